"""CPU: the oracle against golden vectors produced by the REFERENCE's own code
(tests/golden/make_golden.py executes learn_region_grow_util.LrgNet.__init__,
test_region_grow.py and test_random_restart.py unmodified)."""
import glob
import hashlib
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, CLASSES_S3DIS
from learn_region_grow_amd import synthetic, preprocess
from oracle import grow_ref, lrgnet_ref, metrics_ref, preprocess_ref, rng_ref

WEIGHT_KW = dict(seed=0, gain=2.0, bias_std=0.2, add_bias_shift=0.0, rmv_bias_shift=-3.0)


def digest(w):
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'lrgnet_*.npz'))), ids=os.path.basename)
def test_lrgnet_forward_matches_reference_graph(path):
    g = np.load(path)
    lite = int(g['lite'])
    lite = None if lite < 0 else lite
    F = int(g['feature_size'])
    w = synthetic.make_synthetic_weights(feature_size=F, lite=lite, **WEIGHT_KW)
    assert digest(w) == str(g['weights_digest'])
    assert {k: v.shape for k, v in w.items()} == lrgnet_ref.weight_shapes(F, lite)
    add, rmv, acts = lrgnet_ref.forward(w, g['inlier'], g['neighbor'], lite=lite, return_acts=True)
    np.testing.assert_array_equal(add, g['add_output'])
    np.testing.assert_array_equal(rmv, g['remove_output'])
    np.testing.assert_array_equal(acts['pooled'], g['pooled'])
    for i, a in enumerate(acts['conv']):
        np.testing.assert_array_equal(a, g['conv%d' % i])
        np.testing.assert_array_equal(acts['neighbor_conv'][i], g['neighbor_conv%d' % i])
    for i, a in enumerate(acts['add_hidden']):
        np.testing.assert_array_equal(a, g['add_conv%d' % i])
        np.testing.assert_array_equal(acts['remove_hidden'][i], g['remove_conv%d' % i])
    loss, add_acc, rmv_acc = lrgnet_ref.logged_scalars(add, rmv, g['add_mask'], g['rmv_mask'])
    np.testing.assert_allclose(loss, g['loss'], rtol=1e-6)
    assert add_acc == g['add_acc'] and rmv_acc == g['remove_acc']
    # float64 evaluation of the same graph stays within the stated fp32 tolerance
    add64, rmv64 = lrgnet_ref.forward(w, g['inlier'], g['neighbor'], lite=lite, dtype=np.float64)
    scale = max(1.0, float(np.abs(add64).max()))
    assert np.abs(add - add64).max() <= 1e-4 * scale and np.abs(rmv - rmv64).max() <= 1e-4 * scale


def test_confidence_is_scipy_softmax():
    import scipy.special
    x = np.random.RandomState(0).randn(512, 2).astype(np.float32) * 10
    np.testing.assert_array_equal(lrgnet_ref.confidence(x), scipy.special.softmax(x, axis=-1)[:, 1])


@pytest.mark.parametrize('name,restarts', [('greedy_room100', 0), ('greedy_room101', 0), ('restart_room103', 10),
                                           ('greedy_trained_room114', 0), ('restart_trained_room137', 10)])
def test_grow_loop_reproduces_reference_script(name, restarts):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    # (the *_trained_* goldens: the reference scripts under the weights this repository trained -- 51 and 20 labelled regions, all three
    #  stop reasons, 15 of the 20 restart regions won by a restart other than the first)
    w = synthetic.load_trained_weights() if str(g['weight_kw']) == 'trained' else synthetic.make_synthetic_weights(**WEIGHT_KW)
    assert digest(w) == str(g['weights_digest'])
    raw = g['raw_room']
    # preprocessing: oracle loop and the product's vectorised version both equal the reference's output
    for pre in (preprocess_ref.preprocess_room, preprocess.preprocess_room):
        p = pre(raw[:, :6], raw[:, 6].astype(int), raw[:, 7].astype(int))
        np.testing.assert_array_equal(p['points'], g['points'])
        np.testing.assert_array_equal(p['obj_id'], g['obj_id'])
        np.testing.assert_array_equal(np.argsort(p['curvatures']), g['order'])
    r = grow_ref.grow_room(g['points'], g['obj_id'], g['order'], w, rng_ref.LegacyStream(0), cls_id=g['cls_id'],
                           classes=CLASSES_S3DIS, restarts=restarts)
    np.testing.assert_array_equal(r.filled_label, g['filled_label'])
    assert list(r.lines) == [str(x) for x in g['region_lines']]
    m = metrics_ref.room_metrics(g['obj_id'], r.filled_label)
    np.testing.assert_allclose([m['nmi'], m['ami'], m['ars'], m['prc'], m['rcl'], m['iou']], g['metrics'], rtol=1e-9)
    if 'trained' in name:
        lab = [x for x in r.regions if x['labeled']]
        assert {x['reason'] for x in lab} == {'noexpand', 'stuck', 'noneighbor'}
        if restarts:
            assert len(lab) == 20 and sum(1 for x in lab if x['best_restart'] != 0) == 15      # argmax(restart_score), test_random_restart.py:177, matters
        else:
            assert len(lab) == 51 and r.total_steps > 500


def test_beam_search_reproduces_reference_script():
    """test_beam_search.py (BEAM_WIDTH = SEARCH_WIDTH = 3, --scoring np), executed unmodified with a list-returning
    ``range`` in its globals (its ``range(n) + list(...)`` is Python 2): log lines, labels and metrics."""
    from oracle import beam_ref
    g = np.load(os.path.join(GOLDEN, 'beam_room105.npz'))
    w = synthetic.make_synthetic_weights(**WEIGHT_KW)
    assert digest(w) == str(g['weights_digest'])
    r = beam_ref.beam_room(g['points'], g['obj_id'], g['order'], w, rng_ref.LegacyStream(0), cls_id=g['cls_id'],
                           classes=CLASSES_S3DIS)
    assert list(r.lines) == [str(x) for x in g['region_lines']]
    np.testing.assert_array_equal(r.filled_label, g['filled_label'])
    m = metrics_ref.room_metrics(g['obj_id'], r.filled_label)
    np.testing.assert_allclose([m['nmi'], m['ami'], m['ars'], m['prc'], m['rcl'], m['iou']], g['metrics'], rtol=1e-9)
    assert r.total_steps == sum(x['steps'] for x in r.regions)


def test_faithful_and_vectorised_mask_update_agree():
    g = np.load(os.path.join(GOLDEN, 'greedy_room101.npz'))
    w = synthetic.make_synthetic_weights(**WEIGHT_KW)
    a = grow_ref.grow_room(g['points'], g['obj_id'], g['order'], w, rng_ref.LegacyStream(0), faithful=True)
    np.testing.assert_array_equal(a.filled_label, g['filled_label'])
    assert a.total_steps == sum(x['steps'] for x in a.regions)


@pytest.mark.parametrize('policy', ['gt', 'threshold'])
def test_other_policies_run(policy):
    g = np.load(os.path.join(GOLDEN, 'greedy_room100.npz'))
    w = synthetic.make_synthetic_weights(**WEIGHT_KW)
    r = grow_ref.grow_room(g['points'], g['obj_id'], g['order'], w, rng_ref.CounterStream(1, 0), policy=policy,
                           max_region_steps=50)
    assert r.filled_label.min() >= 0 and len(r.regions) > 0
    if policy == 'gt':     # ground-truth masks never mix instances: every labeled region is pure
        for lab in range(1, int(r.cluster_label.max()) + 1):
            assert len(set(g['obj_id'][r.cluster_label == lab].tolist())) == 1


@pytest.mark.parametrize('path', sorted(__import__('glob').glob(os.path.join(GOLDEN, 'lrgnet_*.npz'))), ids=os.path.basename)
def test_reference_goldens_agree_with_torch_conv1d(path):
    """A second opinion on the stand-in's ``tf.nn.conv1d``: the goldens were made by the reference's
    LrgNet.__init__ with conv1d evaluated as ``x @ W[0]``; tests/golden/torch_check.py recomputes every layer with
    torch.nn.functional.conv1d on the CPU -- an independent convolution kernel and an independent reading of the [1, Cin, Cout]
    filter layout -- from the golden's own inputs (make_golden.py runs the same check when it writes a golden)."""
    sys.path.insert(0, GOLDEN)
    import torch_check
    from learn_region_grow_amd import synthetic
    g = np.load(path)
    lite = int(g['lite'])
    w = synthetic.make_synthetic_weights(feature_size=int(g['feature_size']), lite=None if lite < 0 else lite, seed=0, gain=2.0, bias_std=0.2,
                                         add_bias_shift=0.0, rmv_bias_shift=-3.0)
    assert torch_check.check_against_torch(g, w) < 5e-5


def test_bench_cpu_baseline_on_all_cores_counts_the_steps_of_every_room():
    """bench.py's cpu_baseline.all_cores leg: the oracle on several rooms at once, one spawned single-threaded process per room (no GPU involved)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    w = synthetic.make_synthetic_weights(**WEIGHT_KW)
    rooms = []
    for name in ('greedy_room100.npz', 'greedy_room101.npz', 'greedy_room100.npz'):
        g = np.load(os.path.join(GOLDEN, name))
        rooms.append(dict(points=g['points'], obj_id=g['obj_id'], order=g['order']))
    before = os.environ.get('OMP_NUM_THREADS')
    out = bench.cpu_baseline_all_cores(rooms, w, 1.5, 'net', {0: 100, 1: 200, 2: 100})
    assert out['kind'] == 'port' and out['unit'] == 'instance-steps/s' and 1 <= out['cores'] <= 3
    assert out['value'] > 0 and out['rooms_per_sec'] > 0 and abs(out['per_core'] * out['cores'] - out['value']) < 1e-6
    assert os.environ.get('OMP_NUM_THREADS') == before      # (the caller's thread settings come back)
