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
