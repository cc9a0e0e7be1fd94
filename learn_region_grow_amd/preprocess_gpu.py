"""Room preprocessing P0 on the GPU (C-ABI ``lrg_preprocess``; the reference block is test_region_grow.py:119-173).

``preprocess_room`` has the signature and return dict of ``preprocess.preprocess_room`` (the host NumPy version).

eig='lapack'  the GPU does equalisation, neighbour gathering and the float64 covariances (bit-identical to the reference
              loop); the 3x3 SVDs run through ``numpy.linalg.svd`` on the host exactly as the reference calls it.  Every
              output equals the host version bit for bit.
eig='jacobi'  everything on the GPU (Jacobi eigen-solve in float64); features agree with the reference to float32
              rounding and the seed order up to ties / differences below ~1e-13 in curvature.
eig='exact'   the Jacobi solve on the GPU, VERIFIED: both solvers are backward stable, so their singular values of one matrix differ by a
              few eps |cov| and their vectors by that over the gap to the next singular value.  With a slack of 256 eps (EXACT_SLACK)
              a float32 feature whose rounding is the same at both ends of its interval, and a curvature further than the slack from
              its neighbours in the seed order, are what LAPACK would have given; the points that fail either test (near-degenerate
              neighbourhoods, values next to a float32 rounding boundary, ties, the candidates for the maximum curvature: a few per
              thousand) are redone with ``numpy.linalg.svd`` on the host exactly as the reference calls it.  ``points`` and ``order`` --
              everything the region-grow loop reads -- equal the host version bit for bit; ``curvatures`` (float64, not read by the
              loop) are LAPACK's where redone and within 1e-13 elsewhere.  The all-GPU rate instead of a host decomposition per point.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


EXACT_SLACK = 256.0 * np.finfo(np.float64).eps      # bound used for |Jacobi - LAPACK| in units of the largest singular value (csrc: PREP_EIG_SLACK)


def _lapack(cov_h):
    """The reference's own calls on a stack of covariances (test_region_grow.py:158-161): |V[2]| and S[2] / sum(S)."""
    _, S, V = np.linalg.svd(cov_h)
    return np.fabs(V[:, 2, :]), np.fabs(S[:, 2] / (S[:, 0] + S[:, 1] + S[:, 2]))


def _exact_finish(lib, dev, ws, M, N, pts, obj_o, cls_o, curv, cov, F, st):
    """eig='exact': Jacobi results from the GPU, the points whose float32 features or seed-order position could differ under LAPACK redone
    with LAPACK.  Returns points / obj_id / cls_id / curvatures / order."""
    nflag = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
    _lib.check(lib.lrg_preprocess_unsafe_normals(_ptr(ws), M, N, _ptr(nflag), st), 'lrg_preprocess_unsafe_normals')
    c = curv[:N].cpu().numpy().copy()                    # un-normalised S[2] / sum(S) by Jacobi
    feats = pts[:N].cpu().numpy()
    unsafe_n = nflag[:N].cpu().numpy().astype(bool)
    stats = dict(points=N)
    if not np.isfinite(c).all():                         # degenerate room (a NaN curvature poisons the maximum, :163): every point through LAPACK
        redo = np.ones(N, dtype=bool)
    else:
        redo = unsafe_n | (c >= c.max() - 2.0 * EXACT_SLACK)        # ... and whoever could be the maximum
    exact = np.zeros(N, dtype=bool)
    normals = None

    def redo_points(mask):
        nonlocal normals
        idx = np.nonzero(mask & ~exact)[0]
        if len(idx) == 0:
            return
        cov_h = cov[torch.from_numpy(idx).to(dev)].cpu().numpy().reshape(-1, 3, 3)
        nrm, cc = _lapack(cov_h)
        c[idx] = cc
        if F >= 12:
            feats[idx, 9:12] = nrm.astype(np.float32)
        exact[idx] = True
    redo_points(redo)
    stats['first_pass'] = int(exact.sum())
    cmax = c[exact].max() if exact.any() else c.max()                  # LAPACK's maximum: the true one is among the candidates
    cn = c / cmax                                                       # (:163; the reference's own operation for the exact ones)
    dn = np.where(exact, 0.0, EXACT_SLACK / cmax * (1.0 + 1e-9))
    amb = (cn - dn).astype(np.float32) != (cn + dn).astype(np.float32)
    s = np.argsort(cn)
    # neighbours in the seed order closer than TWICE the sum of their slacks: after the redo a point has moved by at most its slack, and
    # the true value of an untouched neighbour lies within its own -- what is left of the gap keeps every pair's order
    close = np.diff(cn[s]) <= 2.0 * (dn[s][1:] + dn[s][:-1])
    near = np.zeros(N, dtype=bool)
    near[s[1:][close]] = True
    near[s[:-1][close]] = True
    again = (amb | near) & ~exact
    if again.any():
        redo_points(again)
        cn = c / cmax
    stats['lapack_points'] = int(exact.sum())
    if F >= 13:
        feats[:, 12] = cn.astype(np.float32)
    return dict(points=feats, obj_id=obj_o[:N].cpu().numpy(), cls_id=cls_o[:N].cpu().numpy(), curvatures=cn, order=np.argsort(cn), exact_stats=stats)


def preprocess_room(unequalized_points, obj_id, cls_id, resolution=0.1, feature_size=13, eig='jacobi', device='cuda:0',
                    return_device=False):
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.LrgHipError('preprocess_gpu needs a GPU (use learn_region_grow_amd.preprocess on the host)')
    if eig not in ('jacobi', 'lapack', 'exact'):
        raise ValueError(eig)
    dev = torch.device(device)
    raw_np = np.ascontiguousarray(np.asarray(unequalized_points)[:, :6], dtype=np.float32)
    M = len(raw_np)
    if M == 0:
        raise ValueError('empty room')
    with torch.cuda.device(dev):
        raw = torch.from_numpy(raw_np).to(dev)
        obj = torch.from_numpy(np.ascontiguousarray(obj_id, dtype=np.int32)).to(dev)
        cls = torch.from_numpy(np.ascontiguousarray(cls_id, dtype=np.int32)).to(dev)
        ws = torch.empty(lib.lrg_preprocess_workspace_bytes(M), dtype=torch.uint8, device=dev)
        eq = torch.empty(M, dtype=torch.int32, device=dev)
        uneq = torch.empty(M, dtype=torch.int32, device=dev)
        n_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        mode = {'jacobi': 1, 'lapack': 0, 'exact': 2}[eig]
        pts = torch.empty((M, feature_size), dtype=torch.float32, device=dev) if mode else None
        obj_o = torch.empty(M, dtype=torch.int32, device=dev) if mode else None
        cls_o = torch.empty(M, dtype=torch.int32, device=dev) if mode else None
        curv = torch.empty(M, dtype=torch.float64, device=dev) if mode else None
        cov = torch.empty((M, 9), dtype=torch.float64, device=dev) if mode != 1 else None
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.lrg_preprocess(_ptr(raw), 6, _ptr(obj), _ptr(cls), M, ctypes.c_float(resolution), feature_size, mode, _ptr(ws),
                                ws.numel(), _ptr(pts), _ptr(obj_o), _ptr(cls_o), _ptr(curv), _ptr(eq), _ptr(uneq), _ptr(cov),
                                _ptr(n_dev), st)
        _lib.check(rc, 'lrg_preprocess')
        status = ctypes.c_int32(0)
        _lib.check(lib.lrg_preprocess_status(_ptr(ws), M, ctypes.byref(status), st), 'lrg_preprocess_status')
        if status.value:
            raise _lib.LrgHipError('a point lies outside the +-2^20 voxel window at resolution %g' % resolution)
        N = int(n_dev.item())
        equalized_idx = eq[:N].cpu().numpy().astype(np.int64)
        unequalized_idx = uneq.cpu().numpy().astype(np.int64)
        if mode == 2:
            out = _exact_finish(lib, dev, ws, M, N, pts, obj_o, cls_o, curv, cov, feature_size, st)
            out.update(equalized_idx=equalized_idx, unequalized_idx=unequalized_idx)
            if return_device:
                out['points_device'] = torch.from_numpy(out['points']).to(dev)
            return out
        if mode:
            c = curv[:N].cpu().numpy()
            out = dict(points=pts[:N].cpu().numpy(), obj_id=obj_o[:N].cpu().numpy(), cls_id=cls_o[:N].cpu().numpy(), curvatures=c,
                       order=np.argsort(c), equalized_idx=equalized_idx, unequalized_idx=unequalized_idx)
            if return_device:
                out['points_device'] = pts[:N]
            return out
        cov_h = cov[:N].cpu().numpy().reshape(N, 3, 3)
    # ---- host finish, the reference's own calls (:158-172) ----
    points = raw_np[equalized_idx]
    xyz, rgb = points[:, :3], points[:, 3:6]
    room_coordinates = (xyz - xyz.min(axis=0)) / (xyz.max(axis=0) - xyz.min(axis=0))
    _, S, V = np.linalg.svd(cov_h)
    normals = np.fabs(V[:, 2, :])
    c = np.fabs(S[:, 2] / (S[:, 0] + S[:, 1] + S[:, 2]))
    c = c / c.max()
    if feature_size == 6:
        feats = np.hstack((xyz, room_coordinates)).astype(np.float32)
    elif feature_size == 9:
        feats = np.hstack((xyz, room_coordinates, rgb)).astype(np.float32)
    elif feature_size == 12:
        feats = np.hstack((xyz, room_coordinates, rgb, normals)).astype(np.float32)
    else:
        feats = np.hstack((xyz, room_coordinates, rgb, normals, c.reshape(-1, 1))).astype(np.float32)
    return dict(points=feats, obj_id=np.asarray(obj_id)[equalized_idx].astype(np.int32),
                cls_id=np.asarray(cls_id)[equalized_idx].astype(np.int32), curvatures=c, order=np.argsort(c),
                equalized_idx=equalized_idx, unequalized_idx=unequalized_idx)
