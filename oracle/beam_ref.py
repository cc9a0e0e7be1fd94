"""Oracle: beam-search region growing of one room (test infrastructure, see oracle/__init__.py).

Restates /root/reference/test_beam_search.py:143-290 (line pins inline): per seed a queue Q of at most BEAM_WIDTH masks;
every queue entry spawns SEARCH_WIDTH stochastic grow steps; the children whose mask changed are scored (``--scoring np``:
mask size), the best BEAM_WIDTH become the next queue; the head of the queue is the answer when growth stalls twice or
no child survives.  Randomness comes from an ``oracle.rng_ref`` stream: the legacy stream reproduces the reference's call
order; under the counter stream a child's draws are keyed (seed point, child ordinal = qid * SEARCH_WIDTH + search id,
level), so children can be evaluated in any order or all at once.

The reference builds its index lists as ``range(n) + list(...)`` (:212, :224) -- Python-2 list arithmetic; the restatement
uses the list form those lines meant (the golden run hands the unmodified script a list-returning ``range``).
"""
import numpy as np

from . import lrgnet_ref
from .grow_ref import GrowResult, fill_unlabeled, pack_voxels, voxelize
from .rng_ref import PURPOSE_ADD, PURPOSE_INLIER, PURPOSE_NEIGHBOR, PURPOSE_RMV


def _format_region(room_id, r, class_name):
    return 'room %d target %3d %.4s: step %3d %4d/%4d points IOU %.3f add %.3f rmv %.3f' % (        # :271
        room_id, r['target'], class_name, r['steps'], r['points'], r['gt'], r['iou'], r['add_acc'], r['rmv_acc'])


def beam_room(points, obj_id, order, weights, stream, *, cls_id=None, classes=None, room_id=0, resolution=0.1, lite=0,
              num_inlier=512, num_neighbor=512, cluster_threshold=10, beam_width=3, search_width=3, policy='net',
              net_fn=None, fill=True, hook=None):
    points = np.ascontiguousarray(points, dtype=np.float32)
    obj_id = np.asarray(obj_id)
    N, F = points.shape
    res = GrowResult()
    if net_fn is None:
        def net_fn(xi, xn):
            return lrgnet_ref.forward(weights, xi, xn, lite=lite)
    point_voxels = voxelize(points[:, :3], resolution)            # :143
    room_keys = pack_voxels(point_voxels)
    cluster_label = np.zeros(N, dtype=np.int64)
    cluster_id = 1
    visited = np.zeros(N, dtype=bool)
    inlier_points = np.zeros((1, num_inlier, F), dtype=np.float32)
    neighbor_points = np.zeros((1, num_neighbor, F), dtype=np.float32)
    add_acc = rmv_acc = float('nan')

    for seed_id in np.arange(N)[np.asarray(order)]:               # :154
        if visited[seed_id]:
            continue
        seed_voxel = point_voxels[seed_id]
        target_id = obj_id[seed_id]
        gt_mask = obj_id == target_id
        currentMask = np.zeros(N, dtype=bool)                     # :164-165
        currentMask[seed_id] = True
        seqMinDims, seqMaxDims = seed_voxel.copy(), seed_voxel.copy()
        steps = 0
        stuck = 0
        bestMask = currentMask
        Q = [(0, currentMask)]                                    # :174
        qid = 0
        newQ = []
        level = 0
        while len(Q) > 0:                                         # :179
            currentScore, currentMask = Q[qid]
            minDims = point_voxels[currentMask, :].min(axis=0)    # :186-187
            maxDims = point_voxels[currentMask, :].max(axis=0)
            if qid == 0:                                          # :188-198
                bestMask = currentMask
                if not np.any(minDims < seqMinDims) and not np.any(maxDims > seqMaxDims):
                    if stuck >= 1:
                        break
                    stuck += 1
                else:
                    stuck = 0
                seqMinDims = np.minimum(seqMinDims, minDims)
                seqMaxDims = np.maximum(seqMaxDims, maxDims)
            currentPoints = points[currentMask, :]
            mask = np.logical_and(np.all(point_voxels >= minDims - 1, axis=1), np.all(point_voxels <= maxDims + 1, axis=1))   # :203-207
            mask = np.logical_and(mask, np.logical_not(currentMask))
            mask = np.logical_and(mask, np.logical_not(visited))
            expandPoints = points[mask, :]
            expandClass = obj_id[mask] == target_id
            rejectClass = obj_id[currentMask] != target_id
            if len(expandPoints) > 0:                             # :212
                for search_id in range(search_width):
                    ctx = (int(seed_id), qid * search_width + search_id, level)
                    nc, ne = len(currentPoints), len(expandPoints)
                    subset = stream.sample(nc, num_inlier, PURPOSE_INLIER, ctx)          # :216-219
                    center = np.median(currentPoints, axis=0)                             # :220
                    ep = np.array(expandPoints)                                            # :221-223
                    ep[:, :2] -= center[:2]
                    ep[:, 6:] -= center[6:]
                    inlier_points[0, :, :] = np.array(currentPoints[subset, :])            # :224-226
                    inlier_points[0, :, :2] -= center[:2]
                    inlier_points[0, :, 6:] -= center[6:]
                    input_remove = np.asarray(rejectClass)[subset].astype(np.int32)        # :227
                    subset_n = stream.sample(ne, num_neighbor, PURPOSE_NEIGHBOR, ctx)      # :228-231
                    neighbor_points[0, :, :] = ep[subset_n, :]                             # :232
                    input_add = np.asarray(expandClass)[subset_n].astype(np.int32)         # :233
                    add, rmv = net_fn(inlier_points, neighbor_points)                      # :234-235
                    add = np.asarray(add, dtype=np.float32)
                    rmv = np.asarray(rmv, dtype=np.float32)
                    _, add_acc, rmv_acc = lrgnet_ref.logged_scalars(add, rmv, input_add[None], input_remove[None])
                    res.total_steps += 1
                    add_conf = lrgnet_ref.confidence(add[0])                               # :237-238
                    rmv_conf = lrgnet_ref.confidence(rmv[0])
                    u_add = stream.uniform(len(add_conf), PURPOSE_ADD, ctx)                # :239-240
                    u_rmv = stream.uniform(len(rmv_conf), PURPOSE_RMV, ctx)
                    if policy == 'net':
                        add_mask = u_add < add_conf
                        rmv_mask = u_rmv < rmv_conf
                        for u_, c_ in ((u_add, add_conf), (u_rmv, rmv_conf)):
                            d_ = np.abs(np.asarray(u_, np.float64) - c_)
                            res.min_margin = min(res.min_margin, float(d_.min()))
                            res.min_rel_margin = min(res.min_rel_margin, float((d_ / (c_ * (1.0 - c_) + 1e-6)).min()))
                    elif policy == 'gt':
                        add_mask = input_add.astype(bool)
                        rmv_mask = input_remove.astype(bool)
                    else:
                        raise ValueError(policy)
                    addPoints = neighbor_points[0, :, :][add_mask]                          # :241-243
                    addPoints[:, :2] += center[:2]
                    addVoxels = voxelize(addPoints[:, :3], resolution)
                    rmvPoints = inlier_points[0, :, :][rmv_mask]                            # :252-254
                    rmvPoints[:, :2] += center[:2]
                    rmvVoxels = voxelize(rmvPoints[:, :3], resolution)
                    if hook is not None:
                        hook(dict(seed=int(seed_id), level=level, qid=qid, search_id=search_id, nc=nc, ne=ne))
                    in_add = np.isin(room_keys, pack_voxels(addVoxels)) if len(addVoxels) else np.zeros(N, bool)
                    in_rmv = np.isin(room_keys, pack_voxels(rmvVoxels)) if len(rmvVoxels) else np.zeros(N, bool)
                    newMask = currentMask.copy()                                            # :264-270
                    updated = bool(np.any(in_add & ~newMask))
                    newMask |= in_add
                    newMask[in_rmv] = False
                    steps += 1                                                              # :274
                    if updated:                                                             # :275-280 (--scoring np)
                        newQ.append((int(np.sum(newMask)), newMask))
            if qid < len(Q) - 1:                                  # :282-287
                qid += 1
            else:
                qid = 0
                Q = sorted(newQ, key=lambda x: x[0], reverse=True)[:beam_width]
                Q = [q for q in Q if q[1].any()]      # an emptied mask has no bounding box (:186 would raise); dropped here
                newQ = []
                level += 1
        visited[bestMask] = True                                  # :289
        labeled = bool(np.sum(bestMask) > cluster_threshold)
        rec = dict(seed=int(seed_id), target=int(target_id), steps=int(steps), points=int(np.sum(bestMask)),
                   gt=int(np.sum(gt_mask)),
                   iou=float(1.0 * np.sum(np.logical_and(gt_mask, bestMask)) / np.sum(np.logical_or(gt_mask, bestMask))),
                   add_acc=float(add_acc), rmv_acc=float(rmv_acc), labeled=labeled)
        if labeled:                                               # :290-293
            cluster_label[bestMask] = cluster_id
            cluster_id += 1
            if classes is not None and cls_id is not None:
                cname = classes[cls_id[np.nonzero(obj_id == target_id)[0][0]]]
                res.lines.append(_format_region(room_id, rec, cname))
        res.regions.append(rec)
    res.cluster_label = cluster_label
    res.filled_label = fill_unlabeled(points, cluster_label) if fill else None
    return res
