"""Seeded synthetic rooms and synthetic LrgNet weights (host side, NumPy).

``generate_room`` follows /root/reference/tools/generate_synthetic_rooms.py:41-99 (box room:
floor, ceiling, four walls = 6 instances, S3DIS-derived size/colour statistics :35-39) but
draws from an explicit ``numpy.random.RandomState`` -- the reference script is unseeded.
``area5_shaped_room`` is a build extension: the same box room plus
axis-aligned cuboid "furniture" instances, with the sampling density chosen so that the
0.1 m-equalised point count hits a target taken from the reference's Area-5 logs.

``make_synthetic_weights`` builds a weight dict with the exact variable names/shapes of
models/lrgnet_model5.ckpt.index (learn_region_grow_util.py:107-159); the trained blob is not
distributed with the reference.
"""
import os

import numpy as np

# tools/generate_synthetic_rooms.py:35-39
ROOM_MIN = np.array([1.0619999, 1.0630007, 2.073])
ROOM_MAX = np.array([44.094, 46.835, 7.647])
ROOM_DIMENSIONS = np.array([5.133024, 5.169554, 3.0433161])
ROOM_VARIATION = np.array([4.2353425, 5.5636344, 0.58006])
COLOR_VARIATION = np.array([0.15274304, 0.15051211, 0.15046296])

# Equalised points per room of the reference's S3DIS Area-5 greedy run
# (results/localsearch/localsearch_greedy.txt) -- "Area-5 shape".
AREA5_POINTS = [13220, 5090, 28756, 8866, 8150, 13469, 7494, 9372, 10622, 8618, 6744, 32196, 7951, 7124, 6643, 20396,
                12840, 7880, 10086, 14518, 11064, 2164, 13275, 13919, 7343, 8732, 8882, 10973, 8336, 16980, 8981, 9471,
                9237, 8586, 9795, 7567, 12548, 7657, 26896, 17368, 8057, 8944, 15599, 7869, 5378, 10910, 12501, 9681,
                17247, 9476, 8793, 30973, 8281, 9213, 26883, 21330, 11566, 45063, 8366, 5563, 7768, 8084, 8309, 12391,
                8111, 12652, 8708, 13365]

CONV_CHANNELS = {0: [64, 64, 64, 128, 512], 1: [64, 64], 2: [64, 64, 256]}     # learn_region_grow_util.py:77-85
CONV2_CHANNELS = {0: [256, 128], 1: [64], 2: [64, 64]}


def _apply_noise_and_color(P, rs, xyz_noise):
    # generate_synthetic_rooms.py:45-50
    P[:, :3] += rs.randn(len(P), 3) * xyz_noise
    mean_color = rs.random_sample(3) - 0.5
    P[:, 3:6] = mean_color + rs.randn(len(P), 3) * COLOR_VARIATION * 0.5
    P[:, 3:6] = np.minimum(0.5, P[:, 3:6])
    P[:, 3:6] = np.maximum(-0.5, P[:, 3:6])


def generate_room(width, length, height, rs, density=0.05, xyz_noise=0.01):
    """[M,8] float64: xyz, rgb, instance id (1..6), class id (0).  generate_synthetic_rooms.py:41-99."""
    room = []

    def face(n, fill, obj):
        pcd = np.zeros((n, 8))
        fill(pcd, n)
        pcd[:, 6] = obj
        _apply_noise_and_color(pcd, rs, xyz_noise)
        room.append(pcd)

    N = int(width * length / density ** 2)

    def floor(p, n):
        p[:, 0] = rs.random_sample(n) * width
        p[:, 1] = rs.random_sample(n) * length

    def ceiling(p, n):
        floor(p, n)
        p[:, 2] = height
    face(N, floor, 1)
    face(N, ceiling, 2)
    N = int(width * height / density ** 2)

    def back(p, n):
        p[:, 0] = rs.random_sample(n) * width
        p[:, 2] = rs.random_sample(n) * height

    def front(p, n):
        p[:, 0] = rs.random_sample(n) * width
        p[:, 1] = length
        p[:, 2] = rs.random_sample(n) * height
    face(N, back, 3)
    face(N, front, 4)
    N = int(length * height / density ** 2)

    def left(p, n):
        p[:, 1] = rs.random_sample(n) * length
        p[:, 2] = rs.random_sample(n) * height

    def right(p, n):
        p[:, 0] = width
        p[:, 1] = rs.random_sample(n) * length
        p[:, 2] = rs.random_sample(n) * height
    face(N, left, 5)
    face(N, right, 6)
    return np.vstack(room)


def room_dims(rs):
    # generate_synthetic_rooms.py:104-106
    wlh = ROOM_DIMENSIONS + rs.randn(3) * ROOM_VARIATION
    wlh = np.maximum(ROOM_MIN, wlh)
    wlh = np.minimum(ROOM_MAX, wlh)
    return wlh


def generate_room_points(target_raw_points, seed, wlh=None):
    """Box room whose raw point count is ~target_raw_points (density solved from the face areas)."""
    rs = np.random.RandomState(seed)
    if wlh is None:
        wlh = room_dims(rs)
    w, l, h = wlh
    area = 2 * (w * l + w * h + l * h)
    density = float(np.sqrt(area / target_raw_points))
    return generate_room(w, l, h, rs, density=density)


def _cuboid(rs, lo, size, n_pts, obj, xyz_noise):
    """Points on the 6 faces of an axis-aligned box (build extension: furniture instance)."""
    areas = np.array([size[0] * size[1], size[0] * size[1], size[0] * size[2], size[0] * size[2],
                      size[1] * size[2], size[1] * size[2]])
    counts = rs.multinomial(n_pts, areas / areas.sum())
    parts = []
    for f, c in enumerate(counts):
        p = np.zeros((c, 8))
        uvw = rs.random_sample((c, 3)) * size
        axis = [2, 2, 1, 1, 0, 0][f]
        uvw[:, axis] = 0.0 if f % 2 == 0 else size[axis]
        p[:, :3] = lo + uvw
        parts.append(p)
    pcd = np.vstack(parts)
    pcd[:, 6] = obj
    pcd[:, 7] = 4
    _apply_noise_and_color(pcd, rs, xyz_noise)
    return pcd


def _area5_geometry(rs_seed, w, l, h, sizes, resolution):
    rs = np.random.RandomState(rs_seed)
    raw_per_area = 2.5 / resolution ** 2
    density = float(np.sqrt(1.0 / raw_per_area))
    parts = [generate_room(w, l, h, rs, density=density)]
    for k in range(len(sizes)):
        sz = np.minimum(sizes[k], [0.8 * w, 0.8 * l, 0.8 * h])
        lo = np.array([rs.uniform(0.05, w - sz[0] - 0.05), rs.uniform(0.05, l - sz[1] - 0.05), 0.0])
        a = 2 * (sz[0] * sz[1] + sz[0] * sz[2] + sz[1] * sz[2])
        parts.append(_cuboid(rs, lo, sz, max(8, int(a * raw_per_area)), 7 + k, 0.01))
    return np.vstack(parts)


def area5_shaped_room(target_equalized_points, seed, n_furniture=None, resolution=0.1):
    """Box room + cuboid furniture whose equalised (one point per `resolution` voxel) count is within a few
    percent of the target: the floor plan is rescaled until the voxel count matches (surfaces are sampled at
    ~2.5 raw points per voxel-sized patch so that equalisation fills them)."""
    rs = np.random.RandomState(seed)
    wlh = room_dims(rs)
    w, l = float(min(wlh[0], 15.0)), float(min(wlh[1], 15.0))
    h = float(np.clip(wlh[2], 2.2, 4.0))
    if n_furniture is None:
        n_furniture = int(rs.randint(40, 101))   # ~72 logged regions per Area-5 room
    sizes = rs.uniform(0.3, 1.5, size=(n_furniture, 3))
    g = 1.0
    room = None
    for _ in range(6):
        ww, ll = max(1.2, w * g), max(1.2, l * g)
        fs = sizes * min(1.0, max(0.3, g))
        room = _area5_geometry(seed + 7919, ww, ll, h, fs, resolution)
        vox = np.round(room[:, :3].astype(np.float32) / np.float32(resolution)).astype(np.int64)
        count = len(np.unique(vox, axis=0))
        if abs(count - target_equalized_points) <= 0.03 * target_equalized_points:
            break
        g *= float(np.sqrt(target_equalized_points / count)) ** 1.15
    return room


def make_synthetic_weights(seed=0, feature_size=13, lite=0, gain=2.0, bias_std=0.2, add_bias_shift=0.0,
                           rmv_bias_shift=-3.0):
    """name -> float32 array with the checkpoint's variable names and TF shapes ([1,Cin,Cout]).

    W ~ U(-a,a), a = gain*sqrt(6/(fan_in+fan_out)) (the reference initialiser
    VarianceScaling(1.0,'fan_avg','uniform'), learn_region_grow_util.py:107, scaled towards the
    spread of trained weights); biases ~ N(0,bias_std); drawn in sorted-name order.  The last
    layer's class-1 bias is shifted so that regions keep growing (add) and rarely shed (remove)."""
    lite = 0 if lite is None else int(lite)
    cc, c2 = CONV_CHANNELS[lite], CONV2_CHANNELS[lite]
    shapes = {}
    for pre in ('lrg_', 'lrg_neighbor_'):
        for i, c in enumerate(cc):
            shapes['%skernel%d' % (pre, i)] = (1, feature_size if i == 0 else cc[i - 1], c)
            shapes['%sbias%d' % (pre, i)] = (c,)
    for pre in ('lrg_add_', 'lrg_remove_'):
        for i, c in enumerate(c2):
            shapes['%skernel%d' % (pre, i)] = (1, cc[-1] * 2 + cc[1] if i == 0 else c2[i - 1], c)
            shapes['%sbias%d' % (pre, i)] = (c,)
        shapes['%skernel%d' % (pre, len(c2))] = (1, c2[-1], 2)
        shapes['%sbias%d' % (pre, len(c2))] = (2,)
    rs = np.random.RandomState(seed)
    w = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if 'kernel' in name:
            a = gain * np.sqrt(6.0 / (shp[1] + shp[2]))
            w[name] = rs.uniform(-a, a, size=shp).astype(np.float32)
        else:
            w[name] = (rs.randn(*shp) * bias_std).astype(np.float32)
    last = len(c2)
    w['lrg_add_bias%d' % last][1] += np.float32(add_bias_shift)
    w['lrg_remove_bias%d' % last][1] += np.float32(rmv_bias_shift)
    return w


def make_reference_init_weights(seed=0, feature_size=13, lite=0):
    """The state a fresh LrgNet starts training from (learn_region_grow_util.py:107-108,...): kernels from
    VarianceScaling(1.0, 'fan_avg', 'uniform') -- U(-a, a) with a = sqrt(6 / (fan_in + fan_out)) -- and zero biases."""
    return make_synthetic_weights(seed=seed, feature_size=feature_size, lite=lite, gain=1.0, bias_std=0.0, add_bias_shift=0.0,
                                  rmv_bias_shift=0.0)


TRAINED_WEIGHTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'weights', 'lrgnet_synthetic_area5.npz')


def load_trained_weights():
    """LrgNet (lite 0, 13 features) trained by THIS repository's training path (tools/train_synthetic.py: 32 Area-5-shaped synthetic
    rooms staged twice by learn_region_grow_amd.stage -- 124 k tuples -- and 12 epochs of train_region_grow.py on one MI355X;
    profiles/r02_train_synthetic.log).  The reference's trained checkpoints are not distributed (models/*.data-* are missing
    upstream), and randomly initialised weights give degenerate growth under the reference's Bernoulli policy; with these the
    policy of test_region_grow.py:266-267 segments the synthetic Area-5-shaped rooms into ~100 regions of ~1 500 steps per room."""
    z = np.load(TRAINED_WEIGHTS)
    return {k: z[k] for k in z.files}
