"""Training tuples for LrgNet from labeled rooms -- the job of the reference's ``stage_data.py``.

Reference: stage_data.py:115-249.  Every ground-truth object of a room is grown from a random seed by an imperfect
teacher: the neighbours of the right object are accepted and the wrong inliers rejected, each decision flipped with a
probability that starts at 0.2-0.4 and decays by 0.01 per step (:139-140,:201-202), until the mask equals the object or
nothing is left to do.  Every step with a non-empty neighbourhood yields one tuple: the current points with their
"should be removed" flags and the neighbouring points with their "should be added" flags (:183-199), at most 1024 points
per side; afterwards x, y and the feature columns 6.. of both sides are centred on the medians of the inlier side
(:242-249).  File layout of ``data/staged_*.h5``: points / count / remove / neighbor_points / neighbor_count / add / steps /
complete (:251-258), read back by train_region_grow.py:71-125.
"""
import numpy as np

from . import h5lite


def stage_room(points, obj_id, rs, resolution=0.1, cluster_threshold=10, max_points=1024, max_steps=500, log=None):
    """points [N,F] (13-feature rows as the loop sees them), obj_id [N]; rs: numpy RandomState.
    Returns dict of lists: points, remove, neighbor_points, add (one entry per tuple), steps (per object), complete (IoU per tuple)."""
    points = np.asarray(points, dtype=np.float32)
    obj_id = np.asarray(obj_id)
    N = len(points)
    point_voxels = np.round(points[:, :3] / resolution).astype(int)
    visited = np.zeros(N, dtype=bool)
    out = dict(points=[], remove=[], neighbor_points=[], add=[], steps=[], complete=[])
    for seed_id in rs.choice(range(N), N, replace=False):                                  # :117
        if visited[seed_id]:
            continue
        target_id = obj_id[seed_id]
        gt_mask = obj_id == target_id
        currentMask = np.zeros(N, dtype=bool)
        currentMask[seed_id] = True
        minDims = point_voxels[seed_id].copy()
        maxDims = point_voxels[seed_id].copy()
        steps, stuck = 0, False
        add_mistake_prob = rs.randint(2, 5) * 0.1                                          # :139
        remove_mistake_prob = rs.randint(2, 5) * 0.1                                       # :140
        iou = 0.0
        while True:
            currentPoints = points[currentMask]
            mask = np.logical_and(np.all(point_voxels >= minDims - 1, axis=1), np.all(point_voxels <= maxDims + 1, axis=1))   # :153-158
            mask &= ~currentMask
            mask &= ~visited
            expandPoints = points[mask]
            expandClass = obj_id[mask] == target_id
            mask_idx = np.nonzero(mask)[0]
            if stuck:                                                                       # :164-170
                expandID = mask_idx[expandClass]
            else:
                mistake = rs.random_sample(len(mask_idx)) < add_mistake_prob
                expandID = mask_idx[np.logical_xor(expandClass, mistake)]
            rejectClass = obj_id[currentMask] != target_id                                  # :173-181
            cur_idx = np.nonzero(currentMask)[0]
            if stuck:
                rejectID = cur_idx[rejectClass]
            else:
                mistake = rs.random_sample(len(cur_idx)) < remove_mistake_prob
                rejectID = cur_idx[np.logical_xor(rejectClass, mistake)]
            if len(expandPoints) > 0:                                                       # :183-202
                if len(currentPoints) <= max_points:
                    out['points'].append(currentPoints.copy())
                    out['remove'].append(rejectClass.astype(np.int32))
                else:
                    subset = rs.choice(len(currentPoints), max_points, replace=False)
                    out['points'].append(currentPoints[subset])
                    rejectClass = rejectClass[subset]
                    out['remove'].append(rejectClass.astype(np.int32))
                if len(expandPoints) <= max_points:
                    out['neighbor_points'].append(expandPoints.copy())
                    out['add'].append(expandClass.astype(np.int32))
                else:
                    subset = rs.choice(len(expandPoints), max_points, replace=False)
                    out['neighbor_points'].append(expandPoints[subset])
                    expandClass = expandClass[subset]
                    out['add'].append(expandClass.astype(np.int32))
                iou = 1.0 * np.sum(currentMask & gt_mask) / np.sum(currentMask | gt_mask)
                out['complete'].append(iou)
                steps += 1
                add_mistake_prob = max(add_mistake_prob - 0.01, 0.0)
                remove_mistake_prob = max(remove_mistake_prob - 0.01, 0.0)
            if np.all(currentMask == gt_mask):                                              # :204-210 completed
                visited[currentMask] = True
                out['steps'].append(steps)
                break
            if steps < max_steps and (np.any(expandClass) or np.any(rejectClass)):          # :212-227 keep growing
                currentMask[expandID] = True
                if len(rejectID) < len(cur_idx):
                    currentMask[rejectID] = False
                nextMin = point_voxels[currentMask].min(axis=0)
                nextMax = point_voxels[currentMask].max(axis=0)
                if not np.any(nextMin < minDims) and not np.any(nextMax > maxDims):
                    stuck = True
                minDims, maxDims = nextMin, nextMax
            else:                                                                           # :228-236 early termination
                if np.sum(currentMask) > cluster_threshold:
                    visited[currentMask] = True
                    out['steps'].append(steps)
                break
        if log is not None:
            log('target %d: %d steps %d/%d' % (target_id, steps, int(currentMask.sum()), int(gt_mask.sum())))
    return out


def center_tuples(staged):
    """stage_data.py:242-249, in place: x, y and the columns 6.. of both sides minus the inlier side's medians."""
    for i in range(len(staged['points'])):
        p = staged['points'][i]
        center = np.median(p[:, :2], axis=0)
        feature_center = np.median(p[:, 6:], axis=0)
        p[:, :2] -= center
        p[:, 6:] -= feature_center
        q = staged['neighbor_points'][i]
        if len(q) > 0:
            q[:, :2] -= center
            q[:, 6:] -= feature_center
    return staged


def merge(parts):
    out = dict(points=[], remove=[], neighbor_points=[], add=[], steps=[], complete=[])
    for p in parts:
        for k in out:
            out[k].extend(p[k])
    return out


def save_staged(filename, staged):
    """The datasets of data/staged_*.h5 (stage_data.py:251-258)."""
    F = staged['points'][0].shape[1] if staged['points'] else 13
    h5lite.write_file(filename, {
        'points': np.vstack(staged['points']).astype(np.float32) if staged['points'] else np.zeros((0, F), np.float32),
        'count': np.array([len(p) for p in staged['points']], dtype=np.int32),
        'neighbor_points': np.vstack(staged['neighbor_points']).astype(np.float32) if staged['neighbor_points'] else np.zeros((0, F), np.float32),
        'neighbor_count': np.array([len(p) for p in staged['neighbor_points']], dtype=np.int32),
        'add': np.concatenate(staged['add']).astype(np.int32) if staged['add'] else np.zeros(0, np.int32),
        'remove': np.concatenate(staged['remove']).astype(np.int32) if staged['remove'] else np.zeros(0, np.int32),
        'steps': np.array(staged['steps'], dtype=np.int32),
        'complete': np.array(staged['complete'], dtype=np.float32)})


def load_staged(filename, feature_size=13):
    """train_region_grow.py:71-125: the per-tuple arrays, tuples with an empty neighbourhood dropped (:128-133)."""
    f = h5lite.File(filename)
    count, ncount = f['count'].read(), f['neighbor_count'].read()
    pts, npts = f['points'].read(), f['neighbor_points'].read()
    rem, add = f['remove'].read(), f['add'].read()
    P = np.split(pts[:, :feature_size], np.cumsum(count)[:-1]) if len(count) else []
    R = np.split(rem, np.cumsum(count)[:-1]) if len(count) else []
    Q = np.split(npts[:, :feature_size], np.cumsum(ncount)[:-1]) if len(ncount) else []
    A = np.split(add, np.cumsum(ncount)[:-1]) if len(ncount) else []
    keep = [i for i in range(len(ncount)) if ncount[i] > 0]
    return dict(points=[P[i] for i in keep], remove=[R[i] for i in keep], neighbor_points=[Q[i] for i in keep], add=[A[i] for i in keep])


def assemble_batch(data, order, rs, batch_size=100, n_inlier=512, n_neighbor=512):
    """train_region_grow.py:156-175: every tuple of the batch padded / subsampled to the network's point counts with the legacy
    generator in the reference's call order (inlier choice, then neighbour choice, per tuple).
    -> inlier [B,Ni,F], neighbor [B,Nn,F] float32, input_add [B,Nn], input_remove [B,Ni] int32."""
    F = data['points'][0].shape[1]
    xi = np.zeros((batch_size, n_inlier, F), dtype=np.float32)
    xn = np.zeros((batch_size, n_neighbor, F), dtype=np.float32)
    ia = np.zeros((batch_size, n_neighbor), dtype=np.int32)
    ir = np.zeros((batch_size, n_inlier), dtype=np.int32)
    for i in range(batch_size):
        k = order[i]
        N = len(data['points'][k])
        subset = rs.choice(N, n_inlier, replace=False) if N >= n_inlier else list(range(N)) + list(rs.choice(N, n_inlier - N, replace=True))
        xi[i] = data['points'][k][subset]
        ir[i] = np.asarray(data['remove'][k])[subset]
        N = len(data['neighbor_points'][k])
        subset = rs.choice(N, n_neighbor, replace=False) if N >= n_neighbor else list(range(N)) + list(rs.choice(N, n_neighbor - N, replace=True))
        xn[i] = data['neighbor_points'][k][subset]
        ia[i] = np.asarray(data['add'][k])[subset]
    return xi, xn, ia, ir
