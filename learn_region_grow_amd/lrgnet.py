"""LrgNetHIP -- host-side mirror of the reference's ``LrgNet`` object for inference.

Reference contract: ``LrgNet(batch_size, seq_len, num_inlier_points, num_neighbor_points,
feature_size, lite=0)`` (learn_region_grow_util.py:76); feeds ``inlier_pl [B*S,Ni,F]``,
``neighbor_pl [B*S,Nn,F]``, ``add_mask_pl``, ``remove_mask_pl`` (:100-103); fetches
``add_output [B,Nn,2]``, ``remove_output [B,Ni,2]``, ``loss``, ``add_acc``, ``remove_acc``
(:149,:162,:186,:175,:180).  Variables are named ``lrg_kernel{i}`` ... ``lrg_remove_bias{j}``
with kernel shape ``[1,Cin,Cout]``.

Here the forward pass runs through the C-ABI ``lrg_forward`` (HIP kernels, gfx950); torch only
owns the device buffers.  The batch size is not baked in: buffers grow on demand.
"""
import ctypes

import numpy as np
import torch

from . import _lib

CONV_CHANNELS = {0: [64, 64, 64, 128, 512], 1: [64, 64], 2: [64, 64, 256]}     # learn_region_grow_util.py:77-85
CONV2_CHANNELS = {0: [256, 128], 1: [64], 2: [64, 64]}


def _stream_ptr(device=None):
    """The current stream OF `device` (None: of the current device).  Kernels are launched on the current device, so
    callers that work on another device than the current one wrap their launches in ``torch.cuda.device(device)``."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class LrgNetHIP:
    def __init__(self, batch_size, seq_len, num_inlier_points, num_neighbor_points, feature_size, lite=0,
                 device='cuda:0', fuse_pool=False, mode='fused', keep_acts=False):
        """mode='fused': whole branch / whole head per 64-row tile in one kernel each (3 launches per evaluation);
        mode='streamed': one launch per layer, every activation through HBM (the layer-by-layer formulation);
        mode='streamed-tiles': the same, every layer launch on the fused stacks' tile (LRG_FWD_STREAM_TILES).
        keep_acts: with 'fused', also copy every intermediate to the workspace so intermediate() works."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.LrgHipError('LrgNetHIP needs a GPU (no CPU fallback)')
        self.device = torch.device(device)
        self.lite = 0 if lite is None else int(lite)
        self.batch = batch_size * seq_len
        self.num_inlier_points = num_inlier_points
        self.num_neighbor_points = num_neighbor_points
        self.feature_size = feature_size
        self.conv_channels = CONV_CHANNELS[self.lite]
        self.conv2_channels = CONV2_CHANNELS[self.lite]
        self.fuse_pool = fuse_pool
        assert mode in ('fused', 'streamed', 'streamed-tiles')
        self.mode = mode
        self.forward_flags = (_lib.LRG_FWD_FUSE_POOL if fuse_pool else 0) | (_lib.LRG_FWD_FUSED if mode == 'fused' else 0) | (_lib.LRG_FWD_STREAM_TILES if mode == 'streamed-tiles' else 0) | \
            (_lib.LRG_FWD_KEEP_ACTS if keep_acts else 0)
        self.weights = {}          # name -> device tensor ([Cin,Cout] / [C])
        self._w = None             # LrgWeights (host struct of device pointers)
        self._ws = None
        self._ws_batch = 0
        self.add_output = None
        self.remove_output = None

    # ---- variables -------------------------------------------------------------------------
    def variable_shapes(self):
        cc, c2, F = self.conv_channels, self.conv2_channels, self.feature_size
        shapes = {}
        for pre in ('lrg_', 'lrg_neighbor_'):
            for i, c in enumerate(cc):
                shapes['%skernel%d' % (pre, i)] = (1, F if i == 0 else cc[i - 1], c)
                shapes['%sbias%d' % (pre, i)] = (c,)
        for pre in ('lrg_add_', 'lrg_remove_'):
            for i, c in enumerate(c2):
                shapes['%skernel%d' % (pre, i)] = (1, cc[-1] * 2 + cc[1] if i == 0 else c2[i - 1], c)
                shapes['%sbias%d' % (pre, i)] = (c,)
            shapes['%skernel%d' % (pre, len(c2))] = (1, c2[-1], 2)
            shapes['%sbias%d' % (pre, len(c2))] = (2,)
        return shapes

    def load_weights(self, weights):
        """weights: name -> array with the checkpoint's names and TF shapes (the Saver.restore step,
        test_region_grow.py:92-93)."""
        shapes = self.variable_shapes()
        missing = [k for k in shapes if k not in weights]
        if missing:
            raise KeyError('missing variables: %s' % missing)
        self.weights = {}
        for name, shp in shapes.items():
            a = np.asarray(weights[name], dtype=np.float32)
            if tuple(a.shape) != tuple(shp):
                raise ValueError('%s: expected shape %s, got %s' % (name, shp, a.shape))
            if a.ndim == 3:
                a = a[0]
            self.weights[name] = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        w = _lib.LrgWeights()
        cc, c2 = self.conv_channels, self.conv2_channels
        w.feature_size = self.feature_size
        w.n_conv = len(cc)
        w.n_head = len(c2) + 1
        for i, c in enumerate(cc):
            w.conv_ch[i] = c
            w.inlier_w[i] = self.weights['lrg_kernel%d' % i].data_ptr()
            w.inlier_b[i] = self.weights['lrg_bias%d' % i].data_ptr()
            w.neighbor_w[i] = self.weights['lrg_neighbor_kernel%d' % i].data_ptr()
            w.neighbor_b[i] = self.weights['lrg_neighbor_bias%d' % i].data_ptr()
        for i, c in enumerate(list(c2) + [2]):
            w.head_ch[i] = c
            w.add_w[i] = self.weights['lrg_add_kernel%d' % i].data_ptr()
            w.add_b[i] = self.weights['lrg_add_bias%d' % i].data_ptr()
            w.rmv_w[i] = self.weights['lrg_remove_kernel%d' % i].data_ptr()
            w.rmv_b[i] = self.weights['lrg_remove_bias%d' % i].data_ptr()
        self._w = w
        self.pack_weights()
        return self

    def pack_weights(self):
        """(Re)build the MFMA-operand image of the kernels (lrg_pack_weights) and publish it in the weights struct;
        call again after changing a variable's device tensor in place."""
        w = self._w
        w.packed = None
        nbytes = self.lib.lrg_packed_weights_bytes(ctypes.byref(w))
        if nbytes == 0:
            raise _lib.LrgHipError('lrg_packed_weights_bytes rejected the configuration')
        self._packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.lrg_pack_weights(ctypes.byref(w), _ptr(self._packed), nbytes, _stream_ptr(self.device)),
                       'lrg_pack_weights')
        w.packed = self._packed.data_ptr()

    # ---- forward ---------------------------------------------------------------------------
    def _workspace(self, B):
        if self._ws is None or self._ws_batch < B:
            nbytes = self.lib.lrg_forward_workspace_bytes(ctypes.byref(self._w), B, self.num_inlier_points,
                                                          self.num_neighbor_points)
            if nbytes == 0:
                raise _lib.LrgHipError('lrg_forward_workspace_bytes rejected the configuration')
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_batch = B
        return self._ws

    def forward(self, inlier, neighbor, add_out=None, rmv_out=None, rows_in=None, rows_nb=None):
        """inlier [B,Ni,F], neighbor [B,Nn,F] float32 CUDA tensors -> (add [B,Nn,2], rmv [B,Ni,2]).
        rows_in / rows_nb ([B] int32 CUDA, optional): evaluate only the leading rows of each instance, the rest
        being copies of them (lrg_forward_rows); logits of the other rows are left unwritten."""
        if self._w is None:
            raise _lib.LrgHipError('load_weights() first')
        assert inlier.is_cuda and neighbor.is_cuda and inlier.dtype == torch.float32 and neighbor.dtype == torch.float32
        assert inlier.is_contiguous() and neighbor.is_contiguous()
        B = inlier.shape[0]
        assert tuple(inlier.shape) == (B, self.num_inlier_points, self.feature_size), inlier.shape
        assert tuple(neighbor.shape) == (B, self.num_neighbor_points, self.feature_size), neighbor.shape
        ws = self._workspace(B)
        if add_out is None:
            add_out = torch.empty((B, self.num_neighbor_points, 2), dtype=torch.float32, device=self.device)
        if rmv_out is None:
            rmv_out = torch.empty((B, self.num_inlier_points, 2), dtype=torch.float32, device=self.device)
        flags = self.forward_flags
        with torch.cuda.device(self.device):
            rc = self.lib.lrg_forward_rows(ctypes.byref(self._w), _ptr(inlier), _ptr(neighbor), B, self.num_inlier_points,
                                           self.num_neighbor_points, _ptr(rows_in), _ptr(rows_nb), _ptr(add_out),
                                           _ptr(rmv_out), _ptr(ws), ws.numel(), flags, _stream_ptr(self.device))
        _lib.check(rc, 'lrg_forward_rows')
        self.add_output, self.remove_output = add_out, rmv_out
        return add_out, rmv_out

    def forward_packed(self, x_in, x_nb, row_inst_in, row_inst_nb, nrows, n_inst, center=None):
        """LrgNet on packed rows (lrg_forward_packed): x_in / x_nb [row_cap,F] hold nrows[0] / nrows[1] valid rows, row r of
        a side belonging to instance row_inst_*[r] (rows of an instance contiguous); with center [n_inst,16] the rows are
        uncentred and enter the network as x - center[instance].  Returns (add [row_cap,2] per neighbour
        row, rmv [row_cap,2] per inlier row, pooled [n_inst, 2*C_last]).  The grow loop's formulation; here for tests."""
        cap = x_in.shape[0]
        assert x_nb.shape[0] == cap and cap % _lib.LRG_ROW_TILE == 0 and nrows.dtype == torch.int32 and nrows.numel() >= 2
        nbytes = self.lib.lrg_forward_packed_workspace_bytes(ctypes.byref(self._w), n_inst, cap)
        if nbytes == 0:
            raise _lib.LrgHipError('lrg_forward_packed_workspace_bytes rejected the configuration')
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        add = torch.zeros((cap, 2), dtype=torch.float32, device=self.device)
        rmv = torch.zeros((cap, 2), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.lrg_forward_packed(ctypes.byref(self._w), _ptr(x_in), _ptr(x_nb), _ptr(center), _ptr(row_inst_in), _ptr(row_inst_nb),
                                             _ptr(nrows), None, n_inst, cap, _ptr(add), _ptr(rmv), _ptr(ws), ws.numel(), 0,
                                             _stream_ptr(self.device))
        _lib.check(rc, 'lrg_forward_packed')
        off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(self.lib.lrg_forward_packed_pooled_view(ctypes.byref(self._w), n_inst, cap, ctypes.byref(off), ctypes.byref(cnt)),
                   'lrg_forward_packed_pooled_view')
        pooled = ws.view(torch.float32)[off.value:off.value + cnt.value].view(n_inst, -1)
        return add, rmv, pooled

    def intermediate(self, kind, index, B):
        """View of a workspace intermediate after forward(): kind in conv|neighbor_conv|pooled|add_hidden|remove_hidden."""
        kinds = {'conv': 0, 'neighbor_conv': 1, 'pooled': 2, 'add_hidden': 3, 'remove_hidden': 4}
        off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
        rc = self.lib.lrg_forward_workspace_view(ctypes.byref(self._w), B, self.num_inlier_points,
                                                 self.num_neighbor_points, kinds[kind], index, ctypes.byref(off),
                                                 ctypes.byref(cnt))
        _lib.check(rc, 'lrg_forward_workspace_view')
        return self._ws.view(torch.float32)[off.value:off.value + cnt.value]

    def run(self, inlier_pl, neighbor_pl, add_mask_pl=None, remove_mask_pl=None):
        """NumPy in / NumPy out, the shape of the reference's
        ``sess.run([net.loss, net.add_output, net.add_acc, net.remove_output, net.remove_acc], feed)``
        (test_region_grow.py:257-258)."""
        xi = torch.from_numpy(np.ascontiguousarray(inlier_pl, dtype=np.float32)).to(self.device)
        xn = torch.from_numpy(np.ascontiguousarray(neighbor_pl, dtype=np.float32)).to(self.device)
        add, rmv = self.forward(xi, xn)
        add, rmv = add.cpu().numpy(), rmv.cpu().numpy()
        if add_mask_pl is None or remove_mask_pl is None:
            return None, add, None, rmv, None
        loss, add_acc, rmv_acc = logged_scalars(add, rmv, add_mask_pl, remove_mask_pl)
        return loss, add, add_acc, rmv, rmv_acc


def _sparse_ce(logits, labels):
    m = logits.max(axis=-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(logits - m).sum(axis=-1))
    return lse - np.take_along_axis(logits, labels[..., None], axis=-1)[..., 0]


def logged_scalars(add, rmv, add_mask, rmv_mask):
    """loss / add_acc / remove_acc of learn_region_grow_util.py:165-186 (host side; only the log line
    at test_region_grow.py:217 consumes them)."""
    add = np.asarray(add, np.float32)
    rmv = np.asarray(rmv, np.float32)
    am = np.asarray(add_mask).astype(np.int64)
    rm = np.asarray(rmv_mask).astype(np.int64)
    add_loss = np.float32(_sparse_ce(add, am).mean())
    add_acc = np.float32((add.argmax(-1) == am).mean(dtype=np.float32))
    ce = _sparse_ce(rmv, rm)
    pos, neg = ce[rm.astype(bool)], ce[(1 - rm).astype(bool)]
    pos_loss = np.float32(pos.mean()) if pos.size else np.float32(0)
    neg_loss = np.float32(neg.mean()) if neg.size else np.float32(0)
    rmv_acc = np.float32((rmv.argmax(-1) == rm).mean(dtype=np.float32))
    return np.float32(add_loss + pos_loss + neg_loss), add_acc, rmv_acc
