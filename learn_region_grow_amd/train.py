"""LrgNetTrainer -- the training side of the reference's ``LrgNet`` object on the GPU.

Reference: the losses and ``AdamOptimizer(1e-3)`` of learn_region_grow_util.py:165-189 and the step
``sess.run([net.train_op, net.loss, net.add_prc, net.add_rcl, net.remove_prc, net.remove_rcl], feed)`` of
train_region_grow.py:175.  The forward pass is the inference path's own (``lrg_forward`` with every activation kept);
the backward pass runs through the C-ABI entry points of csrc/lrg_train.hip -- ``lrg_gemm_f32`` on the fp32 matrix cores
for dX = dZ W^T (ReLU gradient in the epilogue) and dW = X^T dZ (reduction over the B*512 rows split over workgroups),
``lrg_ce_grad``, ``lrg_pool_backward`` (tf.reduce_max's tie rule), ``lrg_segment_colsum`` and ``lrg_adam_step`` --
sequenced here, as the reference sequences its step from Python.  torch owns the buffers; it computes nothing.

The first head layer is differentiated in the hoisted form the forward uses (:128-141): with S[b] = the sum of dZ0 over
instance b's rows, dW0[:P] = pooled^T S, d(pooled) += S W0[:P]^T, dW0[P:] = conv[1]^T dZ0, d(conv[1]) += dZ0 W0[P:]^T.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .lrgnet import LrgNetHIP, _ptr, _stream_ptr


def variable_order(net):
    """The checkpoint's variable names in a fixed (sorted) order: the layout of the flat parameter buffer."""
    return sorted(net.variable_shapes())


class LrgNetTrainer:
    def __init__(self, batch_size, num_inlier_points=512, num_neighbor_points=512, feature_size=13, lite=0, device='cuda:0',
                 learning_rate=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8):
        if num_inlier_points % 64 or num_neighbor_points % 64:
            raise ValueError('the fused forward tiles 64 rows: point counts must be multiples of 64')
        self.net = LrgNetHIP(batch_size, 1, num_inlier_points, num_neighbor_points, feature_size, lite, device=device, mode='fused',
                             keep_acts=True)
        self.lib = self.net.lib
        self.dev = self.net.device
        self.B, self.Ni, self.Nn, self.F = batch_size, num_inlier_points, num_neighbor_points, feature_size
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.t = 0
        self.flat = None

    # ---- variables: one flat buffer, the network's tensors are views into it ----
    def load_weights(self, weights):
        net = self.net
        net.load_weights(weights)
        names = variable_order(net)
        sizes = [net.weights[k].numel() for k in names]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        self.names, self.offs = names, offs
        self.flat = torch.empty(int(offs[-1]), dtype=torch.float32, device=self.dev)
        self.gflat = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.views, self.gviews = {}, {}
        for k, o, n in zip(names, offs[:-1], sizes):
            shp = net.weights[k].shape
            self.flat[o:o + n].copy_(net.weights[k].reshape(-1))
            self.views[k] = self.flat[o:o + n].view(shp)
            self.gviews[k] = self.gflat[o:o + n].view(shp)
            net.weights[k] = self.views[k]
        w = net._w
        cc, c2 = net.conv_channels, net.conv2_channels
        for i in range(len(cc)):
            w.inlier_w[i] = self.views['lrg_kernel%d' % i].data_ptr()
            w.inlier_b[i] = self.views['lrg_bias%d' % i].data_ptr()
            w.neighbor_w[i] = self.views['lrg_neighbor_kernel%d' % i].data_ptr()
            w.neighbor_b[i] = self.views['lrg_neighbor_bias%d' % i].data_ptr()
        for i in range(len(c2) + 1):
            w.add_w[i] = self.views['lrg_add_kernel%d' % i].data_ptr()
            w.add_b[i] = self.views['lrg_add_bias%d' % i].data_ptr()
            w.rmv_w[i] = self.views['lrg_remove_kernel%d' % i].data_ptr()
            w.rmv_b[i] = self.views['lrg_remove_bias%d' % i].data_ptr()
        net.pack_weights()
        self.t = 0
        Rm = self.B * max(self.Ni, self.Nn)
        wmax = max(list(cc) + list(c2) + [2])
        self._dz = [torch.empty(Rm * wmax, dtype=torch.float32, device=self.dev) for _ in range(2)]
        self._dc1 = {s: torch.empty(self.B * n * cc[1], dtype=torch.float32, device=self.dev) for s, n in (('in', self.Ni), ('nb', self.Nn))}
        self._dlog = {s: torch.empty((self.B * n, 2), dtype=torch.float32, device=self.dev) for s, n in (('add', self.Nn), ('rmv', self.Ni))}
        self._S = torch.empty((self.B, c2[0]), dtype=torch.float32, device=self.dev)
        self._tmp = torch.empty(self.B * wmax, dtype=torch.float32, device=self.dev)
        self._dpooled = torch.empty((self.B, 2 * cc[-1]), dtype=torch.float32, device=self.dev)
        self._stats = torch.zeros((2, 8), dtype=torch.float64, device=self.dev)
        return self

    def weights_numpy(self):
        """name -> array with the checkpoint's TF shapes ([1,Cin,Cout] kernels)."""
        shapes = self.net.variable_shapes()
        return {k: self.views[k].cpu().numpy().reshape(shapes[k]) for k in self.names}

    def checkpoint_numpy(self):
        """name -> array of everything the reference's ``Saver`` stores for this graph: variables, Adam slots, beta powers and the
        global step (checkpoint.lrgnet_checkpoint_tensors), so that ``Saver().restore`` (test_region_grow.py:92-93) finds every key."""
        from . import checkpoint
        shapes = self.net.variable_shapes()
        m = {k: self.m[o:o + self.views[k].numel()].cpu().numpy().reshape(shapes[k]) for k, o in zip(self.names, self.offs[:-1])}
        v = {k: self.v[o:o + self.views[k].numel()].cpu().numpy().reshape(shapes[k]) for k, o in zip(self.names, self.offs[:-1])}
        return checkpoint.lrgnet_checkpoint_tensors(self.weights_numpy(), m, v, self.t, self.b1, self.b2)

    def evaluate(self, inlier, neighbor, add_mask, rmv_mask):
        """Loss, precision and recall of a batch without the backward pass (the validation loop, train_region_grow.py:186-219, runs
        only ``net.loss`` and the four ratios): forward + the loss kernel, no dW / dX products."""
        return self.backward(inlier, neighbor, add_mask, rmv_mask, loss_only=True)

    # ---- kernels ----
    def _gemm(self, M, N, K, A, lda, tA, B, ldb, tB, C, ldc, addend=None, mask=None, split=1):
        _lib.check(self.lib.lrg_gemm_f32(M, N, K, _ptr(A), lda, tA, _ptr(B), ldb, tB, _ptr(C), ldc, _ptr(addend), _ptr(mask), split,
                                         _stream_ptr(self.dev)), 'lrg_gemm_f32')

    def _dW(self, X, dZ, R, K, N, out):
        """out[K,N] = X[R,K]^T dZ[R,N], the reduction over the R rows split over workgroups."""
        out.zero_()
        self._gemm(K, N, R, X, K, 1, dZ, N, 0, out, N, split=max(1, R // 1024))

    def _colsum(self, x, R, N, out, seg=None):
        """out[N] = column sums of x[R,N] (through per-instance partial sums)."""
        seg = seg or (self.Ni if R == self.B * self.Ni else self.Nn)
        nseg = R // seg
        part = self._tmp[:nseg * N]
        st = _stream_ptr(self.dev)
        _lib.check(self.lib.lrg_segment_colsum(_ptr(x), nseg, seg, N, _ptr(part), st), 'lrg_segment_colsum')
        _lib.check(self.lib.lrg_segment_colsum(_ptr(part), 1, nseg, N, _ptr(out), st), 'lrg_segment_colsum')

    # ---- one step ----
    def backward(self, inlier, neighbor, add_mask, rmv_mask, loss_only=False):
        """Forward + losses + gradients into self.gviews; returns the scalars the reference fetches."""
        net, B, Ni, Nn, F = self.net, self.B, self.Ni, self.Nn, self.F
        cc, c2 = net.conv_channels, net.conv2_channels
        nc, nh = len(cc), len(c2)
        dev = self.dev
        with torch.cuda.device(dev):
            xi = torch.from_numpy(np.ascontiguousarray(inlier, dtype=np.float32)).to(dev)
            xn = torch.from_numpy(np.ascontiguousarray(neighbor, dtype=np.float32)).to(dev)
            am_h = np.ascontiguousarray(add_mask, dtype=np.int32)
            rm_h = np.ascontiguousarray(rmv_mask, dtype=np.int32)
            am, rm = torch.from_numpy(am_h).to(dev), torch.from_numpy(rm_h).to(dev)
            add, rmv = net.forward(xi, xn)
            acts = {'in': [xi.view(B * Ni, F)] + [net.intermediate('conv', i, B).view(B * Ni, cc[i]) for i in range(nc)],
                    'nb': [xn.view(B * Nn, F)] + [net.intermediate('neighbor_conv', i, B).view(B * Nn, cc[i]) for i in range(nc)]}
            hid = {'add': [net.intermediate('add_hidden', i, B).view(B * Nn, c2[i]) for i in range(nh)],
                   'rmv': [net.intermediate('remove_hidden', i, B).view(B * Ni, c2[i]) for i in range(nh)]}
            pooled = net.intermediate('pooled', 0, B).view(B, 2 * cc[-1])
            st = _stream_ptr(dev)
            # ---- losses (:165-186) ----
            self._stats.zero_()
            n_pos, n_neg = int(rm_h.sum()), int(rm_h.size - rm_h.sum())
            _lib.check(self.lib.lrg_ce_grad(_ptr(add), _ptr(am), B * Nn, 1.0 / (B * Nn), 1.0 / (B * Nn), _ptr(self._dlog['add']),
                                            _ptr(self._stats[0]), st), 'lrg_ce_grad')
            _lib.check(self.lib.lrg_ce_grad(_ptr(rmv), _ptr(rm), B * Ni, 1.0 / n_pos if n_pos else 0.0, 1.0 / n_neg if n_neg else 0.0,
                                            _ptr(self._dlog['rmv']), _ptr(self._stats[1]), st), 'lrg_ce_grad')
            if loss_only:
                return self._scalars(self._stats.cpu().numpy())
            # ---- heads (:138-162), down to the gradient of their first layer's pre-activation ----
            P, C1 = 2 * cc[-1], cc[1]
            dz0 = {}
            first = True
            for hd, pre, side, R in (('add', 'lrg_add_', 'nb', B * Nn), ('rmv', 'lrg_remove_', 'in', B * Ni)):
                H, dlog = hid[hd], self._dlog[hd]
                Wf = self.views[pre + 'kernel%d' % nh]
                self._dW(H[-1], dlog, R, c2[-1], 2, self.gviews[pre + 'kernel%d' % nh])
                self._colsum(dlog, R, 2, self.gviews[pre + 'bias%d' % nh])
                cur, nxt = self._dz[0], self._dz[1]
                self._gemm(R, c2[-1], 2, dlog, 2, 0, Wf, 2, 1, cur, c2[-1], mask=H[-1])
                for i in range(nh - 1, 0, -1):
                    self._dW(H[i - 1], cur, R, c2[i - 1], c2[i], self.gviews[pre + 'kernel%d' % i])
                    self._colsum(cur, R, c2[i], self.gviews[pre + 'bias%d' % i])
                    self._gemm(R, c2[i - 1], c2[i], cur, c2[i], 0, self.views[pre + 'kernel%d' % i], c2[i], 1, nxt, c2[i - 1], mask=H[i - 1])
                    cur, nxt = nxt, cur
                # first layer, hoisted: x W0 = pooled W0[:P] (once per instance) + conv[1] W0[P:]
                W0, g0 = self.views[pre + 'kernel0'], self.gviews[pre + 'kernel0']
                self._dW(acts[side][2], cur, R, C1, c2[0], g0[P:])
                _lib.check(self.lib.lrg_segment_colsum(_ptr(cur), B, R // B, c2[0], _ptr(self._S), st), 'lrg_segment_colsum')
                self._colsum(self._S, B, c2[0], self.gviews[pre + 'bias0'], seg=B)
                self._gemm(P, c2[0], B, pooled, P, 1, self._S, c2[0], 0, g0[:P], c2[0])
                self._gemm(B, P, c2[0], self._S, c2[0], 0, W0[:P], c2[0], 1, self._dpooled, P, addend=None if first else self._dpooled)
                # gradient reaching conv[1] through this head (its own side's branch)
                self._gemm(R, C1, c2[0], cur, c2[0], 0, W0[P:], c2[0], 1, self._dc1[side], C1)
                first = False
                dz0[hd] = None
            # ---- branches (:106-123) ----
            for side, pre, R, off in (('in', 'lrg_', B * Ni, 0), ('nb', 'lrg_neighbor_', B * Nn, cc[-1])):
                A = acts[side]
                cur, nxt = self._dz[0], self._dz[1]
                _lib.check(self.lib.lrg_pool_backward(_ptr(A[nc]), ctypes.c_void_p(self._dpooled.data_ptr() + 4 * off), B, R // B, cc[-1], P,
                                                      _ptr(cur), st), 'lrg_pool_backward')
                if nc == 2:            # lite 1: conv[1] IS the pooled layer: its two gradients meet before its ReLU
                    self._gemm(R, C1, C1, self._dc1[side], C1, 0, self._eye(C1), C1, 0, nxt, C1, addend=cur, mask=A[2])
                    cur, nxt = nxt, cur
                for i in range(nc - 1, -1, -1):
                    K = F if i == 0 else cc[i - 1]
                    self._dW(A[i], cur, R, K, cc[i], self.gviews[pre + 'kernel%d' % i])
                    self._colsum(cur, R, cc[i], self.gviews[pre + 'bias%d' % i])
                    if i > 0:
                        extra = self._dc1[side] if (i == 2) else None      # A[2] = conv[1] also feeds the head
                        self._gemm(R, K, cc[i], cur, cc[i], 0, self.views[pre + 'kernel%d' % i], cc[i], 1, nxt, K, addend=extra, mask=A[i])
                        cur, nxt = nxt, cur
            s = self._stats.cpu().numpy()
        return self._scalars(s)

    @staticmethod
    def _scalars(s):
        loss = float(s[0, 0] + s[1, 0])
        return dict(loss=loss, add_loss=float(s[0, 0]), remove_loss=float(s[1, 0]),
                    add_acc=s[0, 1] / s[0, 5], remove_acc=s[1, 1] / s[1, 5],
                    add_prc=s[0, 2] / (s[0, 3] + 1), add_rcl=s[0, 2] / (s[0, 4] + 1),               # :176-177
                    remove_prc=s[1, 2] / (s[1, 3] + 1), remove_rcl=s[1, 2] / (s[1, 4] + 1))         # :182-184

    def _eye(self, n):
        if getattr(self, '_eye_n', None) != n:
            self._eye_t = torch.eye(n, dtype=torch.float32, device=self.dev)
            self._eye_n = n
        return self._eye_t

    def grads_numpy(self):
        shapes = self.net.variable_shapes()
        return {k: self.gviews[k].cpu().numpy().reshape(shapes[k]) for k in self.names}

    def train_step(self, inlier, neighbor, add_mask, rmv_mask):
        """``sess.run([net.train_op, net.loss, net.add_prc, net.add_rcl, net.remove_prc, net.remove_rcl], feed)``."""
        sc = self.backward(inlier, neighbor, add_mask, rmv_mask)
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.lrg_adam_step(_ptr(self.flat), _ptr(self.gflat), _ptr(self.m), _ptr(self.v), self.flat.numel(),
                                              ctypes.c_float(lr_t), ctypes.c_float(self.b1), ctypes.c_float(self.b2), ctypes.c_float(self.eps),
                                              _stream_ptr(self.dev)), 'lrg_adam_step')
            self.net.pack_weights()
        return sc['loss'], sc['add_prc'], sc['add_rcl'], sc['remove_prc'], sc['remove_rcl']
