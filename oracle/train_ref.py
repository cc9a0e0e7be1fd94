"""Oracle: one training step of LrgNet (test infrastructure, see oracle/__init__.py).

Restates the training side of /root/reference/learn_region_grow_util.py: the losses (:165-186), ``AdamOptimizer(1e-3)``
(:187-189) on the graph of :106-162, and the batch assembly of /root/reference/train_region_grow.py:156-183.  The
gradients are derived on the graph AS THE REFERENCE WRITES IT -- the pooled feature tiled to every row and concatenated
in front of conv[1] (:128-135), a 1088-wide first head layer -- not on the hoisted form the GPU uses, so the two
derivations check each other.

Pinned by: the loss value against the reference's own graph (the ``loss`` of tests/golden/lrgnet_*.npz, produced by
``LrgNet.__init__`` under the NumPy stand-in), and the gradients against central differences of that loss in float64
(tests/test_oracle_units.py).  Semantics that matter:
  * add_loss = mean over all B*Nn slots of the sparse softmax cross entropy (:174);
  * remove_loss = mean over the POSITIVE slots + mean over the NEGATIVE slots of the batch, an empty class contributing 0
    (:166-172: tf.cond on NaN);
  * tf.reduce_max's gradient is shared equally among tied rows (duplicated rows of a padded set tie by construction);
  * Adam as TensorFlow 1 applies it: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t); m, v updated; var -= lr_t * m / (sqrt(v) + eps)
    with eps = 1e-8 OUTSIDE the square root and t counted from 1.
"""
import numpy as np

from .lrgnet_ref import CONV_CHANNELS, CONV2_CHANNELS, _lite


def _softmax_ce(logits, labels):
    """Per-slot sparse softmax cross entropy and d(ce)/d(logits)."""
    m = logits.max(axis=-1, keepdims=True)
    e = np.exp(logits - m)
    p = e / e.sum(axis=-1, keepdims=True)
    ce = -np.log(np.take_along_axis(p, labels[..., None], axis=-1)[..., 0])
    g = p.copy()
    np.put_along_axis(g, labels[..., None], np.take_along_axis(g, labels[..., None], axis=-1) - 1.0, axis=-1)
    return ce, g


def loss_and_grads(w, inlier, neighbor, add_mask, rmv_mask, lite=0, dtype=np.float64):
    """-> (loss, grads dict keyed like w, scalars dict(add_loss, remove_loss, add_acc, remove_acc))."""
    lite = _lite(lite)
    cc, c2 = CONV_CHANNELS[lite], CONV2_CHANNELS[lite]
    W = {k: np.asarray(v, dtype=dtype) for k, v in w.items()}
    xi, xn = np.asarray(inlier, dtype=dtype), np.asarray(neighbor, dtype=dtype)
    am, rm = np.asarray(add_mask).astype(np.int64), np.asarray(rmv_mask).astype(np.int64)
    B, Ni = xi.shape[:2]
    Nn = xn.shape[1]

    def branch(x, pre):
        acts, h = [x], x
        for i in range(len(cc)):
            h = np.maximum(h @ W[pre + 'kernel%d' % i][0] + W[pre + 'bias%d' % i], 0)
            acts.append(h)
        return acts
    ai, an = branch(xi, 'lrg_'), branch(xn, 'lrg_neighbor_')
    pooled = np.concatenate([ai[-1].max(axis=1), an[-1].max(axis=1)], axis=1)                     # :122-124
    P = pooled.shape[1]
    cat = {'lrg_add_': np.concatenate([np.broadcast_to(pooled[:, None, :], (B, Nn, P)), an[2]], axis=2),      # :132-135
           'lrg_remove_': np.concatenate([np.broadcast_to(pooled[:, None, :], (B, Ni, P)), ai[2]], axis=2)}   # :128-131

    def head(x, pre):
        acts, h = [x], x
        for i in range(len(c2)):
            h = np.maximum(h @ W[pre + 'kernel%d' % i][0] + W[pre + 'bias%d' % i], 0)
            acts.append(h)
        i = len(c2)
        return acts, h @ W[pre + 'kernel%d' % i][0] + W[pre + 'bias%d' % i]
    ha, add = head(cat['lrg_add_'], 'lrg_add_')
    hr, rmv = head(cat['lrg_remove_'], 'lrg_remove_')

    # ---- losses (:165-186) ----
    ce_a, g_a = _softmax_ce(add, am)
    add_loss = ce_a.mean()
    d_add = g_a / ce_a.size
    ce_r, g_r = _softmax_ce(rmv, rm)
    pos, neg = rm.astype(bool), ~rm.astype(bool)
    remove_loss = (ce_r[pos].mean() if pos.any() else 0.0) + (ce_r[neg].mean() if neg.any() else 0.0)
    wgt = np.zeros(ce_r.shape, dtype=dtype)
    if pos.any():
        wgt[pos] = 1.0 / pos.sum()
    if neg.any():
        wgt[neg] = 1.0 / neg.sum()
    d_rmv = g_r * wgt[..., None]
    loss = add_loss + remove_loss

    G = {}

    def head_back(acts, dlog, pre):
        """-> d(cat input) [B,N,P+C1]."""
        i = len(c2)
        G[pre + 'kernel%d' % i] = np.einsum('bnk,bnc->kc', acts[-1], dlog)[None]
        G[pre + 'bias%d' % i] = dlog.sum(axis=(0, 1))
        d = dlog @ W[pre + 'kernel%d' % i][0].T
        for i in range(len(c2) - 1, -1, -1):
            d = d * (acts[i + 1] > 0)
            G[pre + 'kernel%d' % i] = np.einsum('bnk,bnc->kc', acts[i], d)[None]
            G[pre + 'bias%d' % i] = d.sum(axis=(0, 1))
            d = d @ W[pre + 'kernel%d' % i][0].T
        return d
    dcat_a = head_back(ha, d_add, 'lrg_add_')
    dcat_r = head_back(hr, d_rmv, 'lrg_remove_')
    dpooled = dcat_a[:, :, :P].sum(axis=1) + dcat_r[:, :, :P].sum(axis=1)          # the tile's gradient: sum over the rows
    dc1 = {'lrg_neighbor_': dcat_a[:, :, P:], 'lrg_': dcat_r[:, :, P:]}              # conv[1] feeds the head of its own side

    def branch_back(acts, dpool, pre):
        top = acts[-1]
        tie = top == top.max(axis=1, keepdims=True)                                # tf.reduce_max: equal shares among ties
        d = tie * (dpool[:, None, :] / tie.sum(axis=1, keepdims=True))
        for i in range(len(cc) - 1, -1, -1):
            if i == 1:
                d = d + dc1[pre]
            d = d * (acts[i + 1] > 0)
            G[pre + 'kernel%d' % i] = np.einsum('bnk,bnc->kc', acts[i], d)[None]
            G[pre + 'bias%d' % i] = d.sum(axis=(0, 1))
            if i > 0:
                d = d @ W[pre + 'kernel%d' % i][0].T
    C = cc[-1]
    branch_back(ai, dpooled[:, :C], 'lrg_')
    branch_back(an, dpooled[:, C:], 'lrg_neighbor_')
    scalars = dict(add_loss=float(add_loss), remove_loss=float(remove_loss),
                   add_acc=float((add.argmax(-1) == am).mean()), remove_acc=float((rmv.argmax(-1) == rm).mean()))
    return float(loss), G, scalars


class Adam:
    """tf.compat.v1.train.AdamOptimizer(learning_rate) (learn_region_grow_util.py:188), defaults b1=0.9, b2=0.999, eps=1e-8."""

    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, w, grads):
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        out = {}
        for k in w:
            g = np.asarray(grads[k], dtype=np.float64)
            m = self.m.get(k, np.zeros_like(g))
            v = self.v.get(k, np.zeros_like(g))
            m = m + (g - m) * (1.0 - self.b1)
            v = v + (g * g - v) * (1.0 - self.b2)
            self.m[k], self.v[k] = m, v
            out[k] = (np.asarray(w[k], dtype=np.float64) - lr_t * m / (np.sqrt(v) + self.eps)).astype(np.float32)
        return out


def assemble_batch(points, remove, neighbor_points, add, order, rs, batch_size=100, n_inlier=512, n_neighbor=512):
    """train_region_grow.py:156-175: pad / subsample every staged tuple of the batch to the network's point counts with the
    legacy generator, in the reference's call order (inlier choice, then neighbour choice, per tuple)."""
    F = points[0].shape[1]
    xi = np.zeros((batch_size, n_inlier, F), dtype=np.float32)
    xn = np.zeros((batch_size, n_neighbor, F), dtype=np.float32)
    ia = np.zeros((batch_size, n_neighbor), dtype=np.int32)
    ir = np.zeros((batch_size, n_inlier), dtype=np.int32)
    for i in range(batch_size):
        k = order[i]
        for arr, flags, out, fout, npts in ((points[k], remove[k], xi, ir, n_inlier), (neighbor_points[k], add[k], xn, ia, n_neighbor)):
            N = len(arr)
            subset = rs.choice(N, npts, replace=False) if N >= npts else list(range(N)) + list(rs.choice(N, npts - N, replace=True))
            out[i] = arr[subset]
            fout[i] = np.asarray(flags)[subset]
    return xi, xn, ia, ir
