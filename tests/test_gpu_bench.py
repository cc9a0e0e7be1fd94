"""GPU: bench.py end to end on a tiny configuration -- the one JSON line the driver parses, with its roofline and
cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('lanes,graph', [(1, 0), (2, 4)])
def test_bench_line(cuda_device, tmp_path, lanes, graph):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '2', '--iters-per-step', '16',
                        '--rooms', '6', '--fixed-rooms', '12', '--cpu-seconds', '3', '--p0-rooms', '1', '--lanes', str(lanes),
                        '--graph', str(graph), '--cache', str(tmp_path / 'cache')],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'rooms_per_sec', 'fixed_work'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 2 and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic'
    assert d['value'] > 0 and d['unit'] == 'instance-steps/s' and 'workload' in d['config'] and d['config']['lanes'] == lanes
    assert d['config']['iterations_per_step'] == 16 and d['config']['timed_iterations'] == 48 and d['config']['hip_graph_iterations'] == graph
    rf = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'hbm_accounting', 'in_loop'):
        assert k in rf, k
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
    assert rf['instances_per_launch'] == 6 and rf['in_loop']['rows_evaluated_fraction'] > 0
    fw = d['fixed_work']
    assert fw['rooms'] == 12 and fw['rooms_per_sec'] > 0 and fw['all_rooms_labeled_after_gather'] and fw['rccl_ranks'] == 1
    assert d['rooms_per_sec'] == fw['rooms_per_sec']
    cb = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample', 'rooms_per_sec'):
        assert k in cb, k
    assert cb['kind'] == 'port' and cb['value'] > 0 and cb['strong']['value'] > 0 and cb['rooms_per_sec'] > 0
    assert d['preprocessing_p0']['gpu_rooms_per_sec'] > 0
