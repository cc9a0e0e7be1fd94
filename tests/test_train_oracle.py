"""CPU: the training oracle (oracle/train_ref.py) against the reference's own loss graph (goldens made by LrgNet.__init__ under
the NumPy stand-in) and against central differences; Adam's known answers; the staging restatement's file round trip."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from learn_region_grow_amd import stage, synthetic, workloads
from oracle import train_ref

WEIGHT_KW = dict(seed=0, gain=2.0, bias_std=0.2, add_bias_shift=0.0, rmv_bias_shift=-3.0)


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'lrgnet_*.npz'))), ids=os.path.basename)
def test_loss_equals_the_reference_graph(path):
    """net.loss, net.add_acc, net.remove_acc (learn_region_grow_util.py:174-186) as the reference's graph evaluates them."""
    g = np.load(path)
    lite = int(g['lite'])
    lite = None if lite < 0 else lite
    w = synthetic.make_synthetic_weights(feature_size=int(g['feature_size']), lite=lite, **WEIGHT_KW)
    loss, _, sc = train_ref.loss_and_grads(w, g['inlier'], g['neighbor'], g['add_mask'], g['rmv_mask'], lite=lite)
    np.testing.assert_allclose(loss, float(g['loss']), rtol=2e-6)
    assert abs(sc['add_acc'] - float(g['add_acc'])) < 1e-6 and abs(sc['remove_acc'] - float(g['remove_acc'])) < 1e-6


@pytest.mark.parametrize('lite', [0, 1, 2])
def test_gradients_match_central_differences(lite):
    w = synthetic.make_synthetic_weights(feature_size=13, lite=lite, **WEIGHT_KW)
    rs = np.random.RandomState(lite)
    xi, xn = rs.randn(2, 16, 13) * 0.5, rs.randn(2, 16, 13) * 0.5
    xi[0, 8:] = xi[0, rs.randint(0, 8, 8)]                  # duplicated rows tie in the max-pool (tf.reduce_max shares the gradient)
    am, rm = rs.randint(0, 2, (2, 16)), rs.randint(0, 2, (2, 16))
    rm[1] = 0                                               # one instance without positives
    _, G, _ = train_ref.loss_and_grads(w, xi, xn, am, rm, lite=lite)
    assert set(G) == set(w) and all(G[k].shape == np.asarray(w[k]).shape for k in w)
    for name in sorted(w):
        a = np.asarray(w[name], np.float64)
        for _ in range(2):
            idx = tuple(rs.randint(0, s) for s in a.shape)
            wp = {k: np.asarray(v, np.float64).copy() for k, v in w.items()}
            wm = {k: np.asarray(v, np.float64).copy() for k, v in w.items()}
            wp[name][idx] += 1e-6
            wm[name][idx] -= 1e-6
            fd = (train_ref.loss_and_grads(wp, xi, xn, am, rm, lite=lite)[0] - train_ref.loss_and_grads(wm, xi, xn, am, rm, lite=lite)[0]) / 2e-6
            assert abs(fd - G[name][idx]) <= 1e-5 * (abs(fd) + abs(G[name][idx])) + 1e-8, (name, idx, fd, G[name][idx])


def test_remove_loss_ignores_an_empty_class():
    """:166-172: the mean over an empty class is NaN in TensorFlow and replaced by 0."""
    w = synthetic.make_synthetic_weights(**WEIGHT_KW)
    rs = np.random.RandomState(3)
    xi, xn = rs.randn(1, 8, 13), rs.randn(1, 8, 13)
    loss, G, sc = train_ref.loss_and_grads(w, xi, xn, rs.randint(0, 2, (1, 8)), np.zeros((1, 8), int))
    assert np.isfinite(loss) and all(np.isfinite(v).all() for v in G.values())


def test_adam_known_answer():
    """TensorFlow-1 Adam: the first step moves every parameter by lr * sign(g) (up to epsilon), whatever the gradient's scale."""
    opt = train_ref.Adam(lr=1e-3)
    w = {'a': np.array([1.0, -2.0, 3.0], np.float32)}
    g = {'a': np.array([10.0, -0.001, 0.0])}
    w1 = opt.step(w, g)
    np.testing.assert_allclose(w1['a'], [1.0 - 1e-3, -2.0 + 1e-3, 3.0], atol=1e-6)   # (epsilon outside the root: 3e-7 at |g| = 1e-3)
    w2 = opt.step(w1, g)                                    # constant gradient: m_hat / sqrt(v_hat) stays 1
    np.testing.assert_allclose(w2['a'], [1.0 - 2e-3, -2.0 + 2e-3, 3.0], atol=2e-6)


def test_staging_round_trip_and_flags(tmp_path):
    """stage_data.py restated: tuples of one room, the staged file's layout read back as train_region_grow.py:71-133 reads it."""
    room = workloads.make_room(1500, 77, 0)
    a = stage.stage_room(room['points'], room['obj_id'], np.random.RandomState(5))
    b = stage.stage_room(room['points'], room['obj_id'], np.random.RandomState(5))
    assert len(a['points']) == len(b['points']) > 50 and all(np.array_equal(x, y) for x, y in zip(a['add'], b['add']))   # seeded
    assert all(len(p) <= 1024 and len(q) <= 1024 and len(q) > 0 for p, q in zip(a['points'], a['neighbor_points']))
    assert all(len(p) == len(r) for p, r in zip(a['points'], a['remove'])) and all(len(q) == len(f) for q, f in zip(a['neighbor_points'], a['add']))
    assert 0.0 < np.mean([f.mean() for f in a['add']]) < 1.0
    stage.center_tuples(a)
    assert all(abs(np.median(p[:, 0])) < 1e-5 and abs(np.median(p[:, 12])) < 1e-5 for p in a['points'])
    path = str(tmp_path / 'staged_synthetic.h5')
    stage.save_staged(path, a)
    ld = stage.load_staged(path)
    assert len(ld['points']) == len(a['points'])
    for k in ('points', 'remove', 'neighbor_points', 'add'):
        assert all(np.array_equal(x, y) for x, y in zip(ld[k], a[k]))
    xi, xn, ia, ir = train_ref.assemble_batch(ld['points'], ld['remove'], ld['neighbor_points'], ld['add'], np.arange(4), np.random.RandomState(0),
                                              batch_size=4, n_inlier=64, n_neighbor=64)
    assert xi.shape == (4, 64, 13) and xn.shape == (4, 64, 13) and ia.shape == (4, 64) and ir.shape == (4, 64)


def test_staging_reproduces_reference_script():
    """stage_data.py executed unmodified (tests/golden/make_golden.py stage: ``--area 5`` on one synthetic room, global legacy generator
    seeded 0, :11) against learn_region_grow_amd.stage on the room that script equalised: the same tuples in the same order -- counts,
    add / remove flags, steps per object, completeness -- and the same centred rows (:242-249), bit for bit (the first ten tuples are
    stored in full, the complete row arrays as SHA-256)."""
    import hashlib
    from conftest import GOLDEN
    from learn_region_grow_amd import preprocess
    g = np.load(os.path.join(GOLDEN, 'stage_room150.npz'))
    raw = g['raw_room']
    p = preprocess.preprocess_room(raw[:, :6], raw[:, 6].astype(int), raw[:, 7].astype(int))      # stage_data.py:58-113 = test_region_grow.py:119-173
    np.testing.assert_array_equal(p['points'], g['room_points'])
    np.testing.assert_array_equal(p['obj_id'], g['room_obj_id'])
    lines = []
    st = stage.stage_room(g['room_points'], g['room_obj_id'], np.random.RandomState(0), log=lines.append)
    stage.center_tuples(st)
    assert len(st['points']) == len(g['staged_count']) == 191
    np.testing.assert_array_equal([len(x) for x in st['points']], g['staged_count'])
    np.testing.assert_array_equal([len(x) for x in st['neighbor_points']], g['staged_neighbor_count'])
    np.testing.assert_array_equal(np.concatenate(st['add']), g['staged_add'])
    np.testing.assert_array_equal(np.concatenate(st['remove']), g['staged_remove'])
    np.testing.assert_array_equal(st['steps'], g['staged_steps'])
    np.testing.assert_array_equal(np.asarray(st['complete'], dtype=np.float32), g['staged_complete'])
    for k in ('points', 'neighbor_points'):
        a = np.ascontiguousarray(np.vstack(st[k]), dtype=np.float32)
        assert tuple(a.shape) == tuple(g['staged_%s_shape' % k])
        head = g['staged_%s_head' % k]
        np.testing.assert_array_equal(a[:len(head)], head)
        assert hashlib.sha256(a.tobytes()).hexdigest() == str(g['staged_%s_sha256' % k])
    # the script's log lines name the objects in the order they were grown, with their step counts and sizes
    assert len(lines) == len(g['log_lines'])
    for mine, ref in zip(lines, [str(x) for x in g['log_lines']]):
        t, rest = mine.split(':')          # 'target 7: 12 steps 143/143'
        assert ('target %d:' % int(t.split()[1])) in ref and rest.strip() in ref, (mine, ref)
