"""tf_ops/grouping replacements on torch tensors (HIP kernels through the C-ABI).

Same names and argument order as the reference's Python wrappers
(tf_ops/grouping/tf_grouping.py:8,22,33,48): ``query_ball_point(radius, nsample, xyz1, xyz2)``,
``select_top_k(k, dist)``, ``group_point(points, idx)``, ``knn_point(k, xyz1, xyz2)``, plus
``group_point_grad`` (the registered gradient, tf_grouping.py:42-46).
Shape errors raise ``ValueError`` where the reference op raises ``InvalidArgument``
(tf_grouping.cpp:71-84,113-118,149-155).
"""
import ctypes

import torch

from . import _lib
from .lrgnet import _ptr, _stream_ptr


def _chk(t, ndim, dtype, name):
    if not (t.is_cuda and t.dtype == dtype and t.dim() == ndim):
        raise ValueError('%s: expected a %d-D %s CUDA tensor, got %s %s' % (name, ndim, dtype, tuple(t.shape), t.dtype))
    return t.contiguous()


def query_ball_point(radius, nsample, xyz1, xyz2):
    """xyz1 (b,n,3) dataset, xyz2 (b,m,3) queries -> idx (b,m,nsample) int32, pts_cnt (b,m) int32."""
    xyz1 = _chk(xyz1, 3, torch.float32, 'xyz1')
    xyz2 = _chk(xyz2, 3, torch.float32, 'xyz2')
    if xyz1.shape[2] != 3 or xyz2.shape[2] != 3 or xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError('QueryBallPoint expects (batch_size, ndataset, 3) and (batch_size, npoint, 3)')
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    _lib.check(_lib.load().lrg_query_ball_point(b, n, m, ctypes.c_float(radius), nsample, _ptr(xyz1), _ptr(xyz2),
                                                _ptr(idx), _ptr(cnt), _stream_ptr()), 'lrg_query_ball_point')
    return idx, cnt


def query_ball_group(radius, nsample, xyz1, xyz2, points=None, subtract_center=False):
    """query_ball_point + group_point(xyz1, idx) + group_point(points, idx) in one launch -- the chain of sample_and_group (train_pointnet.py:113-121).
    -> idx (b,m,nsample), pts_cnt (b,m), grouped_xyz (b,m,nsample,3) (minus xyz2[:, :, None] when subtract_center, :117), grouped_points (b,m,nsample,c) or None."""
    xyz1 = _chk(xyz1, 3, torch.float32, 'xyz1')
    xyz2 = _chk(xyz2, 3, torch.float32, 'xyz2')
    if xyz1.shape[2] != 3 or xyz2.shape[2] != 3 or xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError('QueryBallPoint expects (batch_size, ndataset, 3) and (batch_size, npoint, 3)')
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    c = 0
    if points is not None:
        points = _chk(points, 3, torch.float32, 'points')
        if points.shape[0] != b or points.shape[1] != n:
            raise ValueError('GroupPoint expects points (batch_size, ndataset, channel)')
        c = points.shape[2]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    gx = torch.empty((b, m, nsample, 3), dtype=torch.float32, device=xyz1.device)
    gp = torch.empty((b, m, nsample, c), dtype=torch.float32, device=xyz1.device) if points is not None else None
    _lib.check(_lib.load().lrg_query_ball_group(b, n, m, c, ctypes.c_float(radius), nsample, _ptr(xyz1), _ptr(xyz2), _ptr(points) if points is not None else None,
                                                _ptr(idx), _ptr(cnt), _ptr(gx), _ptr(gp) if gp is not None else None, 1 if subtract_center else 0, _stream_ptr()),
               'lrg_query_ball_group')
    return idx, cnt, gx, gp


def select_top_k(k, dist):
    """dist (b,m,n) -> (idx (b,m,n) int32, dist_out (b,m,n)); the first k along n are the k smallest."""
    dist = _chk(dist, 3, torch.float32, 'dist')
    b, m, n = dist.shape
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dist.device)
    out = torch.empty_like(dist)
    _lib.check(_lib.load().lrg_selection_sort(b, n, m, k, _ptr(dist), _ptr(outi), _ptr(out), _stream_ptr()),
               'lrg_selection_sort')
    return outi, out


def group_point(points, idx):
    """points (b,n,c), idx (b,m,nsample) int32 -> (b,m,nsample,c)."""
    points = _chk(points, 3, torch.float32, 'points')
    idx = _chk(idx, 3, torch.int32, 'idx')
    if points.shape[0] != idx.shape[0]:
        raise ValueError('GroupPoint expects matching batch sizes')
    b, n, c = points.shape
    m, ns = idx.shape[1:]
    out = torch.empty((b, m, ns, c), dtype=torch.float32, device=points.device)
    _lib.check(_lib.load().lrg_group_point(b, n, c, m, ns, _ptr(points), _ptr(idx), _ptr(out), _stream_ptr()),
               'lrg_group_point')
    return out


def group_point_grad(points, idx, grad_out):
    """Gradient of group_point w.r.t. points: scatter-add of grad_out (b,m,nsample,c) -> (b,n,c)."""
    points = _chk(points, 3, torch.float32, 'points')
    idx = _chk(idx, 3, torch.int32, 'idx')
    grad_out = _chk(grad_out, 4, torch.float32, 'grad_out')
    b, n, c = points.shape
    m, ns = idx.shape[1:]
    gp = torch.zeros((b, n, c), dtype=torch.float32, device=points.device)
    _lib.check(_lib.load().lrg_group_point_grad(b, n, c, m, ns, _ptr(grad_out), _ptr(idx), _ptr(gp), _stream_ptr()),
               'lrg_group_point_grad')
    return gp


def knn_point(k, xyz1, xyz2, fused=None):
    """xyz1 (b,n,c) dataset, xyz2 (b,m,c) queries -> val (b,m,k) squared L2, idx (b,m,k) int32
    (tf_grouping.py:48-73: distance matrix + select_top_k + slice).  fused=None: lrg_knn_topk (distances and the selection in
    registers, no b x m x n matrix) whenever n <= 4096 and k <= 512, else the reference's three steps; fused=False forces those."""
    xyz1 = _chk(xyz1, 3, torch.float32, 'xyz1')
    xyz2 = _chk(xyz2, 3, torch.float32, 'xyz2')
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    if xyz2.shape[0] != b or xyz2.shape[2] != c or not 0 < k <= n:
        raise ValueError('knn_point expects (b,n,c) and (b,m,c) with 0 < k <= n')
    if fused is None:
        fused = n <= 4096 and k <= 512      # (lrg_rowselect_kernel keeps the row in registers and is instantiated for k <= 512)
    if fused:
        val = torch.empty((b, m, k), dtype=torch.float32, device=xyz1.device)
        idx = torch.empty((b, m, k), dtype=torch.int32, device=xyz1.device)
        _lib.check(_lib.load().lrg_knn_topk(b, n, m, c, k, _ptr(xyz1), _ptr(xyz2), _ptr(val), _ptr(idx), _stream_ptr()), 'lrg_knn_topk')
        return val, idx
    dist = torch.empty((b, m, n), dtype=torch.float32, device=xyz1.device)
    _lib.check(_lib.load().lrg_pairwise_sqdist(b, n, m, c, _ptr(xyz1), _ptr(xyz2), _ptr(dist), _stream_ptr()),
               'lrg_pairwise_sqdist')
    outi, out = select_top_k(k, dist)
    return out[:, :, :k].contiguous(), outi[:, :, :k].contiguous()
