"""GPU: bench.py end to end on a tiny configuration -- the one JSON line the driver parses, with its roofline and
cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('lanes', [1, 2])
def test_bench_line(cuda_device, tmp_path, lanes):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '40', '--warmup', '10', '--rooms', '6',
                        '--cpu-seconds', '2', '--p0-rooms', '1', '--lanes', str(lanes), '--cache', str(tmp_path / 'cache')],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 40 and d['warmup'] == 10 and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic'
    assert d['value'] > 0 and d['unit'] == 'instance-steps/s' and 'workload' in d['config'] and d['config']['lanes'] == lanes
    rf = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['bound'] == 'hbm' and rf['unit'] == 'GB/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
    assert rf['instances_per_launch'] == 6
    cb = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    assert cb['kind'] == 'port' and cb['value'] > 0 and cb['strong']['value'] > 0
    assert d['preprocessing_p0']['gpu_rooms_per_sec'] > 0
