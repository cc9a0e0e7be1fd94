"""Oracle: LrgNet forward pass in NumPy (test infrastructure, see oracle/__init__.py).

Follows /root/reference/learn_region_grow_util.py:
  channel tables            :77-85
  inlier branch             :106-111   relu(conv1d(k=1) + bias), 5 (or 2 / 3) layers
  neighbour branch          :114-119
  max-pool + concat         :122-125   pooled = [max_rows(inlier) | max_rows(neighbour)]
  tile + concat             :128-135   [pooled (2*C_last) first, then conv[1] (64)]
  add head (on neighbours)  :138-149   last layer has no ReLU
  remove head (on inliers)  :151-162
  logged scalars            :165-186   loss, add_acc, remove_acc

``tf.nn.conv1d`` with a ``[1,Cin,Cout]`` filter, stride 1, VALID is a per-point
matrix product, restated as ``x @ W[0]``.  Weights are a dict keyed by the
checkpoint variable names (``lrg_kernel0`` ... ``lrg_remove_bias2``), kernels in the
TF shape ``[1,Cin,Cout]``.
"""
import numpy as np

CONV_CHANNELS = {0: [64, 64, 64, 128, 512], 1: [64, 64], 2: [64, 64, 256]}
CONV2_CHANNELS = {0: [256, 128], 1: [64], 2: [64, 64]}


def _lite(lite):
    return 0 if lite is None else int(lite)


def weight_shapes(feature_size=13, lite=0):
    """name -> shape for every trainable variable (learn_region_grow_util.py:107-159)."""
    lite = _lite(lite)
    cc, c2 = CONV_CHANNELS[lite], CONV2_CHANNELS[lite]
    shapes = {}
    for pre in ('lrg_', 'lrg_neighbor_'):
        for i, c in enumerate(cc):
            shapes['%skernel%d' % (pre, i)] = (1, feature_size if i == 0 else cc[i - 1], c)
            shapes['%sbias%d' % (pre, i)] = (c,)
    for pre in ('lrg_add_', 'lrg_remove_'):
        for i, c in enumerate(c2):
            shapes['%skernel%d' % (pre, i)] = (1, cc[-1] * 2 + cc[1] if i == 0 else c2[i - 1], c)
            shapes['%sbias%d' % (pre, i)] = (c,)
        shapes['%skernel%d' % (pre, len(c2))] = (1, c2[-1], 2)
        shapes['%sbias%d' % (pre, len(c2))] = (2,)
    return shapes


def forward(w, inlier, neighbor, lite=0, dtype=np.float32, return_acts=False):
    """inlier [B,Ni,F], neighbor [B,Nn,F] -> add [B,Nn,2], rmv [B,Ni,2].

    Computed exactly as the reference graph does (un-hoisted 1088-wide head)."""
    lite = _lite(lite)
    cc, c2 = CONV_CHANNELS[lite], CONV2_CHANNELS[lite]
    W = {k: np.asarray(v, dtype=dtype) for k, v in w.items()}
    acts = {}

    def branch(x, pre):
        convs = []
        h = np.asarray(x, dtype=dtype)
        for i in range(len(cc)):
            h = np.maximum(h @ W[pre + 'kernel%d' % i][0] + W[pre + 'bias%d' % i], 0)
            convs.append(h)
        return convs

    conv = branch(inlier, 'lrg_')                       # :106-111
    nconv = branch(neighbor, 'lrg_neighbor_')           # :114-119
    pool = conv[-1].max(axis=1)                         # :122
    npool = nconv[-1].max(axis=1)                       # :123
    pooled = np.concatenate([pool, npool], axis=1)      # :124  inlier pool first
    B, Ni = conv[1].shape[:2]
    Nn = nconv[1].shape[1]
    inl_cat = np.concatenate([np.broadcast_to(pooled[:, None, :], (B, Ni, pooled.shape[1])), conv[1]], axis=2)   # :128-131
    nbr_cat = np.concatenate([np.broadcast_to(pooled[:, None, :], (B, Nn, pooled.shape[1])), nconv[1]], axis=2)  # :132-135

    def head(x, pre):
        hs = []
        h = x
        for i in range(len(c2)):
            h = np.maximum(h @ W[pre + 'kernel%d' % i][0] + W[pre + 'bias%d' % i], 0)
            hs.append(h)
        i = len(c2)
        out = h @ W[pre + 'kernel%d' % i][0] + W[pre + 'bias%d' % i]   # no ReLU :145-149
        return out, hs

    add, add_h = head(nbr_cat, 'lrg_add_')        # add head on the neighbour side :138-149
    rmv, rmv_h = head(inl_cat, 'lrg_remove_')     # remove head on the inlier side :151-162
    if return_acts:
        acts = dict(conv=conv, neighbor_conv=nconv, pooled=pooled, add_hidden=add_h, remove_hidden=rmv_h)
        return add, rmv, acts
    return add, rmv


def _sparse_ce(logits, labels):
    m = logits.max(axis=-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(logits - m).sum(axis=-1))
    picked = np.take_along_axis(logits, labels[..., None].astype(np.int64), axis=-1)[..., 0]
    return lse - picked


def logged_scalars(add, rmv, add_mask, rmv_mask):
    """loss, add_acc, remove_acc as fetched at test_region_grow.py:257
    (learn_region_grow_util.py:165-186)."""
    add = np.asarray(add, dtype=np.float32)
    rmv = np.asarray(rmv, dtype=np.float32)
    add_mask = np.asarray(add_mask).astype(np.int64)
    rmv_mask = np.asarray(rmv_mask).astype(np.int64)
    add_loss = np.float32(_sparse_ce(add, add_mask).mean())                       # :174
    add_acc = np.float32((add.argmax(axis=-1) == add_mask).astype(np.float32).mean())   # :175
    ce = _sparse_ce(rmv, rmv_mask)
    pos = ce[rmv_mask.astype(bool)]
    neg = ce[(1 - rmv_mask).astype(bool)]
    pos_loss = np.float32(pos.mean()) if pos.size else np.float32(0.0)             # :168,:170 (nan -> 0)
    neg_loss = np.float32(neg.mean()) if neg.size else np.float32(0.0)
    rmv_acc = np.float32((rmv.argmax(axis=-1) == rmv_mask).astype(np.float32).mean())  # :180
    return np.float32(add_loss + pos_loss + neg_loss), add_acc, rmv_acc


def confidence(logits):
    """scipy.special.softmax(logits, axis=-1)[:, 1] in float32 (test_region_grow.py:262-263)."""
    logits = np.asarray(logits, dtype=np.float32)
    m = logits.max(axis=-1, keepdims=True)
    e = np.exp(logits - m)
    return (e / e.sum(axis=-1, keepdims=True))[..., 1]
