"""Oracle: room preprocessing P0 (test infrastructure, see oracle/__init__.py).

Restates /root/reference/test_region_grow.py:119-173 with its own loops:
voxel equalisation (first raw point per voxel), per-point PCA over the raw points of the
27 surrounding voxels (float32 outer products accumulated in float64), normal = |V[2]|,
curvature = S[2]/sum(S) normalised by its max, and the 13-column feature stack.
"""
import itertools
import numpy as np


def preprocess_room(unequalized_points, obj_id, cls_id, resolution=0.1, feature_size=13):
    unequalized_points = np.asarray(unequalized_points)
    equalized_idx = []
    unequalized_idx = []
    equalized_map = {}
    normal_grid = {}
    for i in range(len(unequalized_points)):                                        # :125-133
        k = tuple(np.round(unequalized_points[i, :3] / resolution).astype(int))
        if k not in equalized_map:
            equalized_map[k] = len(equalized_idx)
            equalized_idx.append(i)
        unequalized_idx.append(equalized_map[k])
        if k not in normal_grid:
            normal_grid[k] = []
        normal_grid[k].append(i)
    points = unequalized_points[equalized_idx]                                      # :134
    obj_id = np.asarray(obj_id)[equalized_idx]
    cls_id = np.asarray(cls_id)[equalized_idx]
    xyz = points[:, :3]
    rgb = points[:, 3:6]
    room_coordinates = (xyz - xyz.min(axis=0)) / (xyz.max(axis=0) - xyz.min(axis=0))  # :139

    normals = []
    curvatures = []
    for i in range(len(points)):                                                    # :144-161
        k = tuple(np.round(points[i, :3] / resolution).astype(int))
        neighbors = []
        for offset in itertools.product([-1, 0, 1], [-1, 0, 1], [-1, 0, 1]):
            kk = (k[0] + offset[0], k[1] + offset[1], k[2] + offset[2])
            if kk in normal_grid:
                neighbors.extend(normal_grid[kk])
        accA = np.zeros((3, 3))
        accB = np.zeros(3)
        for n in neighbors:
            p = unequalized_points[n, :3]
            accA += np.outer(p, p)
            accB += p
        cov = accA / len(neighbors) - np.outer(accB, accB) / len(neighbors) ** 2
        U, S, V = np.linalg.svd(cov)
        normals.append(np.fabs(V[2]))
        curvature = S[2] / (S[0] + S[1] + S[2])
        curvatures.append(np.fabs(curvature))
    curvatures = np.array(curvatures)
    curvatures = curvatures / curvatures.max()                                      # :163
    normals = np.array(normals)
    if feature_size == 6:                                                           # :165-172
        feats = np.hstack((xyz, room_coordinates)).astype(np.float32)
    elif feature_size == 9:
        feats = np.hstack((xyz, room_coordinates, rgb)).astype(np.float32)
    elif feature_size == 12:
        feats = np.hstack((xyz, room_coordinates, rgb, normals)).astype(np.float32)
    else:
        feats = np.hstack((xyz, room_coordinates, rgb, normals, curvatures.reshape(-1, 1))).astype(np.float32)
    return dict(points=feats, obj_id=obj_id, cls_id=cls_id, curvatures=curvatures,
                equalized_idx=np.asarray(equalized_idx), unequalized_idx=np.asarray(unequalized_idx))
