"""Room preprocessing P0 on the host (vectorised NumPy).

Computes what /root/reference/test_region_grow.py:119-173 computes -- first-point-per-voxel
equalisation, per-point PCA normals/curvature over the raw points of the 27 surrounding
voxels, and the 13-column feature stack -- without the reference's per-point Python loops.
The float32-outer-product / float64-accumulate arithmetic and the neighbour visiting order
(offsets in itertools.product order, raw points in file order inside a voxel) are kept so
that results match the reference loop to the last bit of the float64 accumulation.

P0 sits upstream of the accelerated grow loop ("next").
"""
import itertools
import numpy as np

VOX_OFF = 1 << 20
_OFFSETS = np.array(list(itertools.product([-1, 0, 1], [-1, 0, 1], [-1, 0, 1])), dtype=np.int64)


def voxel_keys(xyz, resolution):
    v = np.round(np.asarray(xyz) / resolution).astype(np.int64)   # test_region_grow.py:126
    return v, ((v[:, 0] + VOX_OFF) << 42) | ((v[:, 1] + VOX_OFF) << 21) | (v[:, 2] + VOX_OFF)


def equalized_count(xyz, resolution=0.1):
    """Number of points a room keeps after equalisation (one per resolution-sized voxel, test_region_grow.py:125-134) -- the
    size by which rooms are dealt to GPUs -- without the rest of the preprocessing."""
    return int(len(np.unique(voxel_keys(np.asarray(xyz)[:, :3], resolution)[1])))


def preprocess_room(unequalized_points, obj_id, cls_id, resolution=0.1, feature_size=13, chunk=4096, return_cov=False):
    raw = np.asarray(unequalized_points)
    vox, keys = voxel_keys(raw[:, :3], resolution)
    uniq, first_idx, inverse = np.unique(keys, return_index=True, return_inverse=True)
    appear = np.argsort(first_idx, kind='stable')           # voxels in order of first appearance (:127-129)
    rank = np.empty(len(uniq), dtype=np.int64)
    rank[appear] = np.arange(len(uniq))
    equalized_idx = first_idx[appear]
    unequalized_idx = rank[inverse]                          # :130
    points = raw[equalized_idx]
    obj_eq = np.asarray(obj_id)[equalized_idx]
    cls_eq = np.asarray(cls_id)[equalized_idx]
    xyz = points[:, :3]
    rgb = points[:, 3:6]
    room_coordinates = (xyz - xyz.min(axis=0)) / (xyz.max(axis=0) - xyz.min(axis=0))   # :139

    # CSR of raw points by voxel, file order inside a voxel (normal_grid, :131-133)
    order = np.argsort(keys, kind='stable')
    sorted_keys = keys[order]
    starts = np.searchsorted(sorted_keys, uniq, side='left')
    ends = np.searchsorted(sorted_keys, uniq, side='right')

    p = raw[:, :3]
    # numpy.outer(p,p) keeps p's dtype (float32 rooms -> float32 products), then += into float64 (:155)
    prod = (p[:, :, None] * p[:, None, :]).reshape(len(p), 9).astype(np.float64)
    p64 = p.astype(np.float64)

    N = len(points)
    vq = vox[equalized_idx]
    normals = np.zeros((N, 3))
    curv = np.zeros(N)
    covs = np.zeros((N, 3, 3)) if return_cov else None
    for c0 in range(0, N, chunk):
        c1 = min(N, c0 + chunk)
        nb = vq[c0:c1, None, :] + _OFFSETS[None, :, :]                      # [C,27,3] (:147-148)
        nk = ((nb[..., 0] + VOX_OFF) << 42) | ((nb[..., 1] + VOX_OFF) << 21) | (nb[..., 2] + VOX_OFF)
        pos = np.searchsorted(uniq, nk)
        pos_c = np.minimum(pos, len(uniq) - 1)
        hit = uniq[pos_c] == nk
        s = np.where(hit, starts[pos_c], 0)
        e = np.where(hit, ends[pos_c], 0)
        cnt = (e - s)                                                        # [C,27]
        tot = cnt.sum(axis=1)
        # concatenated neighbour list in reference order
        flat_cnt = cnt.reshape(-1)
        seg_start = np.repeat(s.reshape(-1), flat_cnt)
        within = np.arange(flat_cnt.sum()) - np.repeat(np.cumsum(flat_cnt) - flat_cnt, flat_cnt)
        nbr = order[seg_start + within]
        bounds = np.concatenate([[0], np.cumsum(tot)[:-1]])
        # strictly sequential adds per point, vectorised across the chunk (:155-156).  (numpy.add.reduceat is NOT
        # sequential: on a [n,9] operand its ninth column goes through an unrolled pairwise loop and rounds differently
        # once a neighbourhood has more than a few rows.)
        accA = np.zeros((c1 - c0, 9))
        accB = np.zeros((c1 - c0, 3))
        live = np.arange(c1 - c0)
        for t in range(int(tot.max()) if len(tot) else 0):
            live = live[tot[live] > t]
            idx = nbr[bounds[live] + t]
            accA[live] += prod[idx]
            accB[live] += p64[idx]
        accA = accA.reshape(-1, 3, 3)
        n = tot.astype(np.float64)
        cov = accA / n[:, None, None] - (accB[:, :, None] * accB[:, None, :]) / (n ** 2)[:, None, None]   # :157
        if return_cov:
            covs[c0:c1] = cov
        U, S, V = np.linalg.svd(cov)                                          # :158
        normals[c0:c1] = np.fabs(V[:, 2, :])                                  # :159
        curv[c0:c1] = np.fabs(S[:, 2] / (S[:, 0] + S[:, 1] + S[:, 2]))        # :160-161
    curv = curv / curv.max()                                                  # :163
    if feature_size == 6:                                                     # :165-172
        feats = np.hstack((xyz, room_coordinates)).astype(np.float32)
    elif feature_size == 9:
        feats = np.hstack((xyz, room_coordinates, rgb)).astype(np.float32)
    elif feature_size == 12:
        feats = np.hstack((xyz, room_coordinates, rgb, normals)).astype(np.float32)
    else:
        feats = np.hstack((xyz, room_coordinates, rgb, normals, curv.reshape(-1, 1))).astype(np.float32)
    out = dict(points=feats, obj_id=obj_eq.astype(np.int32), cls_id=cls_eq.astype(np.int32), curvatures=curv,
               order=np.argsort(curv),   # :183 -- same call as the reference (default sort kind; tie order is NumPy's)
               equalized_idx=equalized_idx, unequalized_idx=unequalized_idx)
    if return_cov:
        out['cov'] = covs
    return out
