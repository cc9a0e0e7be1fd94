"""GPU: the command-line driver end to end -- rooms from an HDF5 file, weights from a checkpoint bundle, region growing
through the C-ABI, the reference's metric lines and PLY export -- against the oracle pipeline on the same files."""
import os
import subprocess
import sys

import numpy as np
import pytest

from learn_region_grow_amd import checkpoint, metrics, synthetic
from learn_region_grow_amd import io as lio
from oracle import grow_ref, metrics_ref, preprocess_ref, rng_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('policy', ['gt', 'threshold'])
def test_cli_matches_oracle_pipeline(cuda_device, tmp_path, policy):
    raw = [synthetic.generate_room_points(2000 + 400 * i, 40 + i, wlh=(1.4 + 0.2 * i, 1.2, 1.0)).astype(np.float32)
           for i in range(3)]
    h5 = str(tmp_path / 'rooms.h5')
    lio.saveToH5(h5, raw)
    weights = synthetic.make_synthetic_weights(seed=0)
    prefix = str(tmp_path / 'model' / 'lrgnet.ckpt')
    checkpoint.write_bundle(prefix, weights)
    out_dir = str(tmp_path / 'ply')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'region_grow.py'), '--h5', h5, '--ckpt', prefix, '--policy', policy,
                        '--seed', '3', '--save', out_dir, '--rooms-in-flight', '2'], capture_output=True, text=True,
                       cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert lines[0] == 'Restored from %s' % prefix
    rooms, obj, cls = lio.loadFromH5(h5)
    want_lines, want_metrics = [], []
    for i in range(3):
        p = preprocess_ref.preprocess_room(rooms[i], obj[i], cls[i])
        order = np.argsort(p['curvatures'])
        res = grow_ref.grow_room(p['points'], p['obj_id'], order, weights, rng_ref.CounterStream(3, i), policy=policy)
        if policy == 'threshold' and res.min_rel_margin < 1e-3:
            pytest.skip('a confidence within fp32 noise of the 0.5 cut (margin %.1e): NumPy and GPU logits may disagree' % res.min_margin)
        m = metrics_ref.room_metrics(p['obj_id'], res.filled_label)
        want_metrics.append(m)
        want_lines.append(metrics.room_line('custom', i, m))
        # the saved cloud: raw xyz, colour of the matched cluster of each raw point's voxel representative
        ply = open(os.path.join(out_dir, '%d.ply' % i)).read().splitlines()
        assert ply[2] == 'element vertex %d' % len(rooms[i]) and len(ply) == 10 + len(rooms[i])
        got_rgb = np.array([[int(v) for v in ln.split()[3:6]] for ln in ply[10:]])
        colors = lio.label_colors(int(m['cluster_label2'].max()) + 1)
        np.testing.assert_array_equal(got_rgb, colors[m['cluster_label2']][p['unequalized_idx']])
        assert ply[10].split()[:3] == ['%f' % v for v in rooms[i][0, :3]]
    got_lines = [ln for ln in lines if ln.startswith('Area custom room')]
    assert got_lines == want_lines
    assert lines[-1] == metrics.aggregate_line(want_metrics)


def test_cli_beam_search(cuda_device, tmp_path):
    """--beam: the test_beam_search.py driver end to end; per-room lines equal the oracle beam search on the same files."""
    from oracle import beam_ref
    raw = [synthetic.generate_room_points(1500, 60 + i, wlh=(1.3, 1.1, 1.0)).astype(np.float32) for i in range(2)]
    h5 = str(tmp_path / 'rooms.h5')
    lio.saveToH5(h5, raw)
    weights = synthetic.make_synthetic_weights(seed=0)
    prefix = str(tmp_path / 'lrgnet.ckpt')
    checkpoint.write_bundle(prefix, weights)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'region_grow.py'), '--h5', h5, '--ckpt', prefix, '--policy', 'gt', '--seed', '4',
                        '--beam', '2', '--search-width', '2'], capture_output=True, text=True, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rooms, obj, cls = lio.loadFromH5(h5)
    want = []
    for i in range(2):
        p = preprocess_ref.preprocess_room(rooms[i], obj[i], cls[i])
        res = beam_ref.beam_room(p['points'], p['obj_id'], np.argsort(p['curvatures']), weights, rng_ref.CounterStream(4, i),
                                 policy='gt', beam_width=2, search_width=2)
        want.append(metrics.room_line('custom', i, metrics_ref.room_metrics(p['obj_id'], res.filled_label)))
    assert [ln for ln in r.stdout.splitlines() if ln.startswith('Area custom room')] == want


def _write_inputs(tmp_path, n_rooms, seed0=70, size0=900):
    raw = [synthetic.generate_room_points(size0 + 250 * i, seed0 + i, wlh=(1.2 + 0.15 * i, 1.1, 1.0)).astype(np.float32) for i in range(n_rooms)]
    h5 = str(tmp_path / 'rooms.h5')
    lio.saveToH5(h5, raw)
    weights = synthetic.make_synthetic_weights(seed=0, gain=2.0, bias_std=0.2, add_bias_shift=0.0, rmv_bias_shift=-3.0)
    prefix = str(tmp_path / 'lrgnet.ckpt')
    checkpoint.write_bundle(prefix, weights)
    return h5, prefix, weights


def test_cli_region_lines_and_timing_table(cuda_device, tmp_path):
    """The per-region lines of test_region_grow.py:217 (target, class, steps, points, IOU, add / remove accuracy of the last
    step, stop reason) equal the oracle's on the same GPU network; --timing prints the reference's table (:382-390) and does not
    change a label."""
    from conftest import CLASSES_S3DIS
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    h5, prefix, weights = _write_inputs(tmp_path, 2)
    cmd = [sys.executable, os.path.join(ROOT, 'region_grow.py'), '--h5', h5, '--ckpt', prefix, '--policy', 'gt', '--seed', '6']
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=cuda_device).load_weights(weights)

    def net_fn(xi, xn):
        _, add, _, rmv, _ = net.run(xi, xn)
        return add, rmv
    rooms, obj, cls = lio.loadFromH5(h5)
    want = []
    for i in range(2):
        p = preprocess_ref.preprocess_room(rooms[i], obj[i], cls[i])
        res = grow_ref.grow_room(p['points'], p['obj_id'], np.argsort(p['curvatures']), None, rng_ref.CounterStream(6, i), net_fn=net_fn,
                                 policy='gt', cls_id=p['cls_id'], classes=CLASSES_S3DIS, room_id=i)
        want += res.lines
    got = [ln for ln in r.stdout.splitlines() if ln.startswith('room ')]
    assert len(want) > 4 and got == want
    rt = subprocess.run(cmd + ['--timing'], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert rt.returncode == 0, rt.stderr[-2000:]
    lines = rt.stdout.splitlines()
    assert [ln for ln in lines if ln.startswith(('room ', 'Area '))] == [ln for ln in r.stdout.splitlines() if ln.startswith(('room ', 'Area '))]
    table = lines[-7:]
    assert [t.split()[0] for t in table] == ['feature', 'net', 'neighbor', 'inlier', 'iter_net', 'iter_neighbor', 'iter_inlier']
    shares = [float(t.split()[-1]) for t in table]
    # (one decimal: a per-iteration bucket of ~50 us beside per-room buckets of seconds may print as 0.0 when a room's feature computation was slow)
    assert abs(sum(shares) - 100.0) < 0.5 and all(x >= 0 for x in shares) and sum(1 for x in shares if x > 0) >= 4


def test_cli_two_ranks_equal_one(cuda_device, tmp_path):
    """--gpus 2 under torch.distributed.run (both ranks on the one GPU of this box, collectives over gloo): the rooms are dealt
    by equalised point count, each rank grows its share, labels and lines are gathered -- same output as a single process."""
    import socket
    h5, prefix, _ = _write_inputs(tmp_path, 6, seed0=80, size0=700)
    base = ['--h5', h5, '--ckpt', prefix, '--policy', 'net', '--seed', '9']
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'region_grow.py')] + base, capture_output=True, text=True, cwd=str(tmp_path), timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, LRG_BENCH_ONE_DEVICE='1')
    two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), os.path.join(ROOT, 'region_grow.py')] + base + ['--gpus', '2'],
                         capture_output=True, text=True, cwd=str(tmp_path), timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    pick = lambda out: [ln for ln in out.splitlines() if ln.startswith(('room ', 'Area ', 'NMI:'))]
    assert pick(two.stdout) == pick(one.stdout) and len(pick(one.stdout)) > 10
    assert any(ln.startswith('gathered the labels of 6 rooms') for ln in two.stdout.splitlines())
