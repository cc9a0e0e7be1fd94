"""Oracle: the region-grow loop of one room (test infrastructure, see oracle/__init__.py).

Restates /root/reference/test_region_grow.py:175-316 (greedy) and
/root/reference/test_random_restart.py:141-303 (random restart), line pins inline.

Inputs are what the loop sees at test_region_grow.py:175: ``points[N,F]`` float32
(xyz, room-normalised xyz, rgb, normal, curvature), ``obj_id[N]``, ``cls_id[N]`` and
``order`` = ``numpy.argsort(curvatures)`` (:183; supplied by the caller so that tie order
is explicit).  Randomness comes from an ``oracle.rng_ref`` stream.

Two execution flavours compute identical results:
  faithful=True   per-point Python voxel-tuple set loop (:282-287) -- the CPU baseline
  faithful=False  vectorised set membership on packed voxel keys   -- used by tests
"""
import numpy as np

from . import lrgnet_ref
from .rng_ref import PURPOSE_INLIER, PURPOSE_NEIGHBOR, PURPOSE_ADD, PURPOSE_RMV

VOX_OFF = 1 << 20


def voxelize(xyz, resolution):
    """numpy.round(points[:,:3]/resolution).astype(int) (:126,:175,:272,:276): float32
    divide by the float32-rounded resolution, round-half-even."""
    xyz = np.asarray(xyz, dtype=np.float32)
    return np.round(xyz / np.float32(resolution)).astype(np.int64)


def pack_voxels(v):
    v = np.asarray(v, dtype=np.int64) + VOX_OFF
    return (v[..., 0] << 42) | (v[..., 1] << 21) | v[..., 2]


class GrowResult:
    def __init__(self):
        self.cluster_label = None      # before fill-in (:176, :214)
        self.filled_label = None       # after fill-in (:308-316)
        self.regions = []              # dicts: seed, target, steps, points, gt, iou, add_acc, rmv_acc, reason, labeled
        self.total_steps = 0           # LrgNet evaluations
        self.min_margin = np.inf       # min |u - conf| over all Bernoulli draws
        self.min_rel_margin = np.inf   # min |u - conf| / (conf*(1-conf) + 1e-6): how large a RELATIVE error in a logit
                                       # difference it takes to flip the closest Bernoulli draw.  Same logits, other
                                       # exp/divide: errors ~1e-7; another fp32 evaluation of the network: ~1e-5..1e-4
        self.lines = []                # reference-format log lines (:217)
        self.min_score_gap = np.inf    # --scoring ml: smallest relative lead of a winning restart over the runner-up


def _format_region(room_id, r, class_name):
    return 'room %d target %3d %.4s: step %3d %4d/%4d points IOU %.3f add %.3f rmv %.3f %s' % (
        room_id, r['target'], class_name, r['steps'], r['points'], r['gt'], r['iou'], r['add_acc'], r['rmv_acc'], r['reason'])


def grow_room(points, obj_id, order, weights, stream, *, cls_id=None, classes=None, room_id=0,
              resolution=0.1, lite=0, num_inlier=512, num_neighbor=512, cluster_threshold=10,
              policy='net', restarts=0, faithful=False, net_fn=None, hook=None, max_region_steps=None,
              fill=True, scoring='np', max_total_steps=None):
    """Grow all regions of one room.  restarts=0 -> test_region_grow.py; restarts=R>0 ->
    test_random_restart.py with NUM_RESTARTS=R and --scoring np (default) or ml.

    scoring='ml' (test_random_restart.py:171-172,251-271): the restart score is the summed log-likelihood of the add / remove
    masks sampled along the restart.  Upstream initialises the accumulator with 0 (:164) but resets it to a LIST between restarts
    (:194-196), so from the second restart on ``maskLogProb += float`` no longer accumulates a scalar and ``argmax`` (:177)
    sees a ragged list: only the first restart is scored as written.  Restated here as evidently intended -- one scalar per
    restart, reset to 0 -- which is a documented divergence."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    obj_id = np.asarray(obj_id)
    N, F = points.shape
    res = GrowResult()
    if net_fn is None:
        def net_fn(xi, xn):
            return lrgnet_ref.forward(weights, xi, xn, lite=lite)

    point_voxels = voxelize(points[:, :3], resolution)            # :175
    room_keys = pack_voxels(point_voxels)
    cluster_label = np.zeros(N, dtype=np.int64)                   # :176
    cluster_id = 1                                                # :177
    visited = np.zeros(N, dtype=bool)                             # :178
    inlier_points = np.zeros((1, num_inlier, F), dtype=np.float32)
    neighbor_points = np.zeros((1, num_neighbor, F), dtype=np.float32)
    add_acc = rmv_acc = float('nan')   # the reference leaves these undefined before the first sess.run
    R = max(1, restarts)

    for seed_id in np.arange(N)[np.asarray(order)]:               # :186
        if visited[seed_id]:                                      # :187-188
            continue
        if max_total_steps is not None and res.total_steps >= max_total_steps:
            break          # test extension: a PREFIX of the room's regions (whole regions only) -- 100 k-point scenes within a CPU budget
        seed_voxel = point_voxels[seed_id]
        target_id = obj_id[seed_id]
        gt_mask = obj_id == target_id
        steps = 0                                                 # :203 (never reset between restarts)
        restart_score, restart_mask = [], []
        last_reason = None
        for restart in range(R):
            maskLogProb = 0.0                                     # restart :164 (and, as intended, per restart)
            currentMask = np.zeros(N, dtype=bool)                 # :197-198 / restart :188-189
            currentMask[seed_id] = True
            minDims = seed_voxel.copy()
            maxDims = seed_voxel.copy()
            seqMinDims, seqMaxDims = minDims, maxDims
            stuck = 0
            rstep = 0
            while True:                                           # :208
                # neighbour (box) query :221-231
                mask = np.logical_and(np.all(point_voxels >= minDims - 1, axis=1),
                                      np.all(point_voxels <= maxDims + 1, axis=1))
                mask = np.logical_and(mask, np.logical_not(currentMask))
                mask = np.logical_and(mask, np.logical_not(visited))
                currentPoints = points[currentMask, :].copy()
                expandPoints = points[mask, :].copy()
                expandClass = obj_id[mask] == target_id
                rejectClass = obj_id[currentMask] != target_id
                if len(expandPoints) == 0:                        # :233-235
                    reason = 'noneighbor'
                    break
                if max_region_steps is not None and rstep >= max_region_steps:
                    reason = 'maxsteps'                           # build extension, off by default
                    break
                ctx = (int(seed_id), restart, rstep)
                nc, ne = len(currentPoints), len(expandPoints)
                subset = stream.sample(nc, num_inlier, PURPOSE_INLIER, ctx)      # :237-240
                center = np.median(currentPoints, axis=0)                        # :241
                expandPoints[:, :2] -= center[:2]                                # :243
                expandPoints[:, 6:] -= center[6:]                                # :244
                inlier_points[0, :, :] = currentPoints[subset, :]                # :245
                inlier_points[0, :, :2] -= center[:2]                            # :246
                inlier_points[0, :, 6:] -= center[6:]                            # :247
                input_remove = np.asarray(rejectClass)[subset].astype(np.int32)  # :248
                subset_n = stream.sample(ne, num_neighbor, PURPOSE_NEIGHBOR, ctx)  # :249-252
                neighbor_points[0, :, :] = expandPoints[subset_n, :]             # :253
                input_add = np.asarray(expandClass)[subset_n].astype(np.int32)   # :254
                add, rmv = net_fn(inlier_points, neighbor_points)                # :257-258
                add = np.asarray(add, dtype=np.float32)
                rmv = np.asarray(rmv, dtype=np.float32)
                _, add_acc, rmv_acc = lrgnet_ref.logged_scalars(add, rmv, input_add[None], input_remove[None])
                res.total_steps += 1
                add_conf = lrgnet_ref.confidence(add[0])                         # :262
                rmv_conf = lrgnet_ref.confidence(rmv[0])                         # :263
                u_add = stream.uniform(len(add_conf), PURPOSE_ADD, ctx)          # :266
                u_rmv = stream.uniform(len(rmv_conf), PURPOSE_RMV, ctx)          # :267
                if policy == 'net':
                    add_mask = u_add < add_conf
                    rmv_mask = u_rmv < rmv_conf
                    for u_, c_ in ((u_add, add_conf), (u_rmv, rmv_conf)):
                        d_ = np.abs(np.asarray(u_, np.float64) - c_)
                        res.min_margin = min(res.min_margin, float(d_.min()))
                        res.min_rel_margin = min(res.min_rel_margin, float((d_ / (c_ * (1.0 - c_) + 1e-6)).min()))
                elif policy == 'threshold':                                      # :264-265 (commented out upstream)
                    add_mask = add_conf > 0.5
                    rmv_mask = rmv_conf > 0.5
                    for c_ in (add_conf, rmv_conf):                              # distance of a confidence from the cut
                        d_ = np.abs(np.asarray(c_, np.float64) - 0.5)
                        res.min_margin = min(res.min_margin, float(d_.min()))
                        res.min_rel_margin = min(res.min_rel_margin, float((d_ / 0.25).min()))
                elif policy == 'gt':                                             # :268-269 (commented out upstream)
                    add_mask = input_add.astype(bool)
                    rmv_mask = input_remove.astype(bool)
                else:
                    raise ValueError(policy)
                if scoring == 'ml':                                              # restart :251-271
                    with np.errstate(divide='ignore'):
                        for pts_, conf_, mask_ in ((neighbor_points[0], add_conf, add_mask), (inlier_points[0], rmv_conf, rmv_mask)):
                            q = pts_.copy()
                            q[:, :2] += center[:2]                               # :256 / :264 (row by row upstream; same float32 adds)
                            keys_ = pack_voxels(voxelize(q[:, :3], resolution))
                            in_set = np.isin(keys_, keys_[mask_]) if mask_.any() else np.zeros(len(keys_), bool)
                            c32 = np.asarray(conf_, dtype=np.float32)
                            term = np.where(in_set, np.log(c32), np.log(np.float32(1.0) - c32)) / np.float32(num_neighbor)   # :259-261 (both / NUM_NEIGHBOR_POINT)
                            maskLogProb += float(np.sum(term.astype(np.float64)))
                addPoints = neighbor_points[0, :, :][add_mask]                   # :270
                addPoints[:, :2] += center[:2]                                   # :271
                addVoxels = voxelize(addPoints[:, :3], resolution)               # :272
                rmvPoints = inlier_points[0, :, :][rmv_mask]                     # :274
                rmvPoints[:, :2] += center[:2]                                   # :275
                rmvVoxels = voxelize(rmvPoints[:, :3], resolution)               # :276
                if hook is not None:
                    hook(dict(seed=int(seed_id), restart=restart, step=rstep, nc=nc, ne=ne, center=center.copy(),
                              subset_in=np.asarray(subset).copy(), subset_nb=np.asarray(subset_n).copy(),
                              inlier=inlier_points.copy(), neighbor=neighbor_points.copy(),
                              add=add.copy(), rmv=rmv.copy(), add_conf=add_conf.copy(), rmv_conf=rmv_conf.copy(),
                              u_add=np.asarray(u_add).copy(), u_rmv=np.asarray(u_rmv).copy(),
                              add_mask=add_mask.copy(), rmv_mask=rmv_mask.copy(),
                              input_add=input_add.copy(), input_remove=input_remove.copy(),
                              mask_before=currentMask.copy(), visited=visited.copy(),
                              min_dims=minDims.copy(), max_dims=maxDims.copy()))
                updated = False
                if faithful:                                                     # :273,:277,:282-287
                    addSet = set([tuple(p) for p in addVoxels])
                    rmvSet = set([tuple(p) for p in rmvVoxels])
                    for i in range(len(point_voxels)):
                        if not currentMask[i] and tuple(point_voxels[i]) in addSet:
                            currentMask[i] = True
                            updated = True
                        if tuple(point_voxels[i]) in rmvSet:
                            currentMask[i] = False
                else:
                    in_add = np.isin(room_keys, pack_voxels(addVoxels)) if len(addVoxels) else np.zeros(N, bool)
                    in_rmv = np.isin(room_keys, pack_voxels(rmvVoxels)) if len(rmvVoxels) else np.zeros(N, bool)
                    updated = bool(np.any(in_add & ~currentMask))
                    currentMask |= in_add
                    currentMask[in_rmv] = False
                steps += 1                                                       # :288
                rstep += 1
                if updated:                                                      # :291
                    if not currentMask.any():
                        reason = 'empty'     # reference raises on .min() of an empty mask (:292); defined as a stop
                        break
                    minDims = point_voxels[currentMask, :].min(axis=0)           # :292
                    maxDims = point_voxels[currentMask, :].max(axis=0)           # :293
                    if not np.any(minDims < seqMinDims) and not np.any(maxDims > seqMaxDims):  # :294
                        if stuck >= 1:                                           # :295-297
                            reason = 'stuck'
                            break
                        else:
                            stuck += 1                                           # :299
                    else:
                        stuck = 0                                                # :301
                    seqMinDims = np.minimum(seqMinDims, minDims)                 # :302
                    seqMaxDims = np.maximum(seqMaxDims, maxDims)                 # :303
                else:
                    reason = 'noexpand'                                          # :304-306
                    break
            last_reason = reason
            restart_score.append(maskLogProb if scoring == 'ml' else int(np.sum(currentMask)))   # restart :171-174
            restart_mask.append(currentMask)
        if scoring == 'ml' and len(restart_score) > 1:
            # how far the winner is ahead: GPU and NumPy logarithms differ in the last bit, so a near-tie may legitimately flip
            sc = np.sort(np.asarray(restart_score, dtype=np.float64))[::-1]
            if np.isfinite(sc[0]) and sc[0] != sc[1]:
                res.min_score_gap = min(res.min_score_gap, float((sc[0] - sc[1]) / (abs(sc[0]) + 1e-12)) if np.isfinite(sc[1]) else np.inf)
        bestMask = restart_mask[int(np.argmax(restart_score))]    # restart :177 (first max); R=1 -> the mask itself
        visited[bestMask] = True                                  # :212
        labeled = bool(np.sum(bestMask) > cluster_threshold)      # :213
        rec = dict(seed=int(seed_id), target=int(target_id), steps=int(steps), points=int(np.sum(bestMask)),
                   gt=int(np.sum(gt_mask)),
                   iou=float(1.0 * np.sum(np.logical_and(gt_mask, bestMask)) / np.sum(np.logical_or(gt_mask, bestMask))),
                   add_acc=float(add_acc), rmv_acc=float(rmv_acc), reason=last_reason, labeled=labeled,
                   best_restart=int(np.argmax(restart_score)), restart_scores=[float(x) for x in restart_score])
        if labeled:
            cluster_label[bestMask] = cluster_id                  # :214
            cluster_id += 1                                       # :215
            if classes is not None and cls_id is not None:
                cname = classes[cls_id[np.nonzero(obj_id == target_id)[0][0]]]   # :191
                res.lines.append(_format_region(room_id, rec, cname))           # :217
        res.regions.append(rec)

    res.cluster_label = cluster_label
    res.filled_label = fill_unlabeled(points, cluster_label) if fill else None
    return res


def fill_unlabeled(points, cluster_label):
    """1-NN fill-in of unlabeled points in all feature dims (:308-316), first-min ties."""
    points = np.asarray(points, dtype=np.float32)
    nonzero_idx = np.nonzero(cluster_label)[0]
    filled = cluster_label.copy()
    if len(nonzero_idx) == 0:
        return filled   # reference would raise on argmin of an empty array; defined as a no-op
    nonzero_points = points[nonzero_idx, :]
    for i in np.nonzero(cluster_label == 0)[0]:
        d = np.sum((nonzero_points - points[i]) ** 2, axis=1)
        filled[i] = cluster_label[nonzero_idx[np.argmin(d)]]
    return filled
