"""Synthetic workloads of the shapes BASELINE.json names (no datasets are available offline).

``area5_rooms`` -- "S3DIS Area-5 shape": 68 rooms whose 0.1 m-equalised point counts follow the reference's own
Area-5 run (learn_region_grow_amd.synthetic.AREA5_POINTS), box rooms + cuboid furniture,
preprocessed (P0) to the 13-feature layout the loop consumes.  Generation is seeded and cached as .npz.
"""
import os

import numpy as np

from . import preprocess, synthetic


def make_room(target_points, seed, room_id, resolution=0.1):
    raw = synthetic.area5_shaped_room(target_points, seed, resolution=resolution).astype(np.float32)
    p = preprocess.preprocess_room(raw[:, :6], raw[:, 6].astype(int), raw[:, 7].astype(int), resolution=resolution)
    return dict(points=p['points'], obj_id=p['obj_id'], order=p['order'].astype(np.int32), room_id=room_id)


def area5_rooms(n_rooms=68, seed_base=1000, cache_dir=None, targets=None, scale=1.0, resolution=0.1):
    targets = list(targets if targets is not None else synthetic.AREA5_POINTS)
    rooms = []
    cache = None
    if cache_dir:
        os.makedirs(cache_dir, exist_ok=True)
        cache = os.path.join(cache_dir, 'lrg_area5_%d_%d_%g_%g_%d.npz' % (n_rooms, seed_base, scale, resolution,
                                                                            targets[0] if targets else 0))
        if os.path.exists(cache):
            try:
                z = np.load(cache)
                return [dict(points=z['p%d' % i], obj_id=z['o%d' % i], order=z['s%d' % i], room_id=int(z['ids'][i]))
                        for i in range(n_rooms)]
            except Exception:        # a truncated file left behind by a killed run: generate again
                pass
    for i in range(n_rooms):
        t = max(300, int(targets[i % len(targets)] * scale))
        rooms.append(make_room(t, seed_base + i, seed_base + i, resolution=resolution))
    if cache:
        d = {'ids': np.array([r['room_id'] for r in rooms])}
        for i, r in enumerate(rooms):
            d['p%d' % i], d['o%d' % i], d['s%d' % i] = r['points'], r['obj_id'], r['order']
        # several ranks of one node may generate the same set at the same time: write aside, then rename (atomic)
        tmp = '%s.%d.tmp.npz' % (cache, os.getpid())
        np.savez(tmp, **d)
        os.replace(tmp, cache)
    return rooms


def kitti_scenes(n_scenes=8, seed_base=5000, cache_dir=None, points=100000, resolution=0.3):
    """BASELINE.json configs[4] shape: ~100 k points per scene at 0.3 m resolution (README.md:156 of the reference)."""
    return area5_rooms(n_scenes, seed_base=seed_base, cache_dir=cache_dir, targets=[points], resolution=resolution)


def scannet_rooms(n_rooms=39, seed_base=7000, cache_dir=None, resolution=0.1):
    """BASELINE.json configs[2] shape: ScanNet rooms, equalised count ~ N(5.8 k, 3 k) clipped to [1.2 k, 15.5 k]
    (SURVEY.md section 8d, config 3).  The full set is 312 rooms = 8 GPUs x 39; a rank generates its own 39."""
    rs = np.random.RandomState(seed_base)
    targets = np.clip(rs.normal(5800.0, 3000.0, size=n_rooms), 1200, 15500).astype(int).tolist()
    return area5_rooms(n_rooms, seed_base=seed_base, cache_dir=cache_dir, targets=targets, resolution=resolution)
