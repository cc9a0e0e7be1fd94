"""GPU: bench.py end to end on tiny configurations -- the one JSON line the driver parses, with its roofline and cpu_baseline
objects, for both formulations of the loop; `--gpus 2` started WITHOUT a torch.distributed environment (the script spawns its
ranks itself); a WORLD_SIZE that contradicts --gpus is refused."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
            'config', 'roofline', 'rooms_per_sec', 'fixed_work')


def run_bench(argv, env=None, timeout=1200):
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)
    return r, [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


@pytest.mark.parametrize('mode,lanes,graph', [('free', 0, 0), ('lockstep', 1, 0), ('lockstep', 2, 4)])
def test_bench_line(cuda_device, tmp_path, mode, lanes, graph):
    r, lines = run_bench(['--gpus', '1', '--steps', '3', '--warmup', '2', '--iters-per-step', '16', '--step-ms', '2', '--rooms', '6', '--fixed-rooms', '12',
                          '--best-slots', '3,6', '--steady-slots', '9' if mode == 'free' else '', '--named-configs', '1' if mode == 'free' else '0', '--cpu-seconds', '3', '--cpu-box-seconds', '2', '--p0-rooms', '1', '--mode', mode, '--lanes', str(lanes), '--graph', str(graph),
                          '--cache', str(tmp_path / 'cache')])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in CONTRACT + ('cpu_baseline',):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 2 and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic'
    assert d['value'] > 0 and d['unit'] == 'instance-steps/s' and 'workload' in d['config']
    if mode == 'free':
        assert 'free-running' in d['config']['formulation'] and d['config']['lanes'] == 1
    else:
        assert d['config']['lanes'] == lanes and d['config']['iterations_per_step'] == 16 and d['config']['hip_graph_iterations'] == graph
    rf = d['roofline']
    for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_flops_in_loop', 'flops_per_launch', 'avg_us',
              'rows_evaluated_fraction', 'dense'):
        assert k in rf, k
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
    assert 0 < rf['rows_evaluated_fraction'] <= 1 and rf['dense']['frac'] > 0
    # north_star's MLP metric, measured live beside the fused figure: the layer-streamed evaluation (csrc/lrg_stream_layer.inl) priced with SURVEY.md 8d's bytes
    ls = rf['dense']['layer_streamed']
    assert 'error' not in ls, ls
    assert ls['algorithmic_bytes'] == 1088 * 10297344 and 0.2 < ls['frac_of_hbm_peak'] < 1.0, ls
    if mode == 'free':
        # the roofline headline can be recomputed from the line alone: algorithmic FLOPs of the timed launches / their time / the peak
        assert abs(rf['algorithmic_flops_in_loop'] / rf['kernel_seconds'] / 1e12 / rf['peak'] - rf['frac']) < 1e-9
        assert rf['launches'] == 3 and 'lrg_grow_async_kernel' in rf['kernel']
    fw = d['fixed_work']
    assert fw['rooms'] == 12 and fw['rooms_per_sec'] > 0 and fw['all_rooms_labeled_after_gather'] and fw['rccl_ranks'] == 1 and fw['given_up'] == 0
    # one rank: the label gather and the count reductions still went through RCCL (a one-rank communicator on the device)
    assert fw['collective_backend'] == 'nccl' and fw['collective_executed'] and fw['collective_error'] is None
    assert d['rooms_per_sec'] == fw['rooms_per_sec']
    fb = d['fixed_work_best']
    assert fb['all_rooms_labeled_after_gather'] and fb['rooms_per_sec'] >= fw['rooms_per_sec'] and set(fb['sweep']) == {'3', '6'}
    cb = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample', 'rooms_per_sec'):
        assert k in cb, k
    assert cb['kind'] == 'port' and cb['value'] > 0 and cb['strong']['value'] > 0 and cb['rooms_per_sec'] > 0
    assert cb['one_room']['value'] > 0 and cb['one_room']['rooms_per_sec'] > 0 and cb['cores'] >= 1 and cb['per_core'] > 0      # (value: many rooms on many cores)
    assert d['preprocessing_p0']['gpu_rooms_per_sec'] > 0
    if mode == 'free':
        assert d['steady_more_rooms_in_flight']['9']['value'] > 0
        # BASELINE configs 3 and 5 in the line: ScanNet-shaped rooms and KITTI-shaped scenes, eight in flight, as runs of the script of their own
        for wl in ('scannet', 'kitti'):
            nc = d['named_configs'][wl]
            assert 'error' not in nc and nc['value'] > 0 and nc['fixed_work_rooms_per_sec'] > 0 and nc['all_rooms_labeled_after_gather'], nc
        # the fixed-work legs of the free-running launches carry a roofline of their own (device counters over the leg's grow time): the best frac is in the line
        assert 0 < fw['roofline']['frac'] < 1 and 0 < fw['roofline']['rows_evaluated_fraction'] <= 1 and set(fb['sweep']['3']) >= {'roofline_frac'}
    # the measured CPU figure is the headline key of the baseline, the extrapolated one beside it
    if 'measured_small_rooms' in cb:
        assert cb['rooms_per_sec'] == cb['measured_small_rooms']['rooms_per_sec_box'] and not cb['rooms_per_sec_extrapolated']


def test_bench_line_with_restarts_counts_rows_per_slot(cuda_device, tmp_path):
    """test_random_restart.py's loop, 4 restarts per seed batched per launch: rows_evaluated_fraction is per SLOT (rooms x restarts), so it stays <= 1
    (round 5 divided by the rooms: 2.65 in the 16-restart record)."""
    r, lines = run_bench(['--gpus', '1', '--steps', '2', '--warmup', '1', '--iters-per-step', '8', '--rooms', '4', '--restarts', '4', '--fixed-rooms', '0',
                          '--best-slots', '', '--steady-slots', '', '--named-configs', '0', '--cpu-seconds', '0', '--p0-rooms', '0', '--one-room-ks', '', '--cache', str(tmp_path / 'cache')])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(lines[0])
    rf = d['roofline']
    assert d['config']['restarts'] == 4 and 0 < rf['rows_evaluated_fraction'] <= 1 and 0 < rf['slots_evaluated'] <= 16


def test_bench_two_ranks_started_by_the_script_itself(cuda_device, tmp_path):
    """`python bench.py --gpus 2` with no torch.distributed environment: the script starts its two ranks (here both on cuda:0,
    collectives over gloo: LRG_BENCH_ONE_DEVICE=1), shards the fixed work over them and gathers every room's labels."""
    common = ['--steps', '2', '--warmup', '1', '--step-ms', '2', '--rooms', '4', '--fixed-rooms', '8', '--best-slots', '', '--steady-slots', '', '--named-configs', '0', '--cpu-seconds', '0',
              '--p0-rooms', '0', '--cache', str(tmp_path / 'cache')]
    # (both ranks share the one device here: lock-step launches -- two free-running launches side by side on one chip each assume that all
    #  their workgroups are resident at once; on a multi-GPU node every rank has a device to itself)
    r2, l2 = run_bench(['--gpus', '2', '--mode', 'lockstep', '--iters-per-step', '16'] + common, env={'LRG_BENCH_ONE_DEVICE': '1'})
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert len(l2) == 1
    d2 = json.loads(l2[0])
    for k in CONTRACT:
        assert k in d2, k
    assert d2['n_gpus'] == 2 and d2['fixed_work']['rccl_ranks'] == 2 and d2['fixed_work']['rooms'] == 8
    assert d2['fixed_work']['all_rooms_labeled_after_gather'] and d2['fixed_work']['collective_backend'] == 'gloo'
    # the same fixed work on one rank: the same labels (checksum over all rooms) and the same number of instance-steps
    r1, l1 = run_bench(['--gpus', '1'] + common)
    assert r1.returncode == 0, r1.stderr[-3000:]
    d1 = json.loads(l1[0])
    assert d1['n_gpus'] == 1 and d1['fixed_work']['instance_steps'] == d2['fixed_work']['instance_steps']
    assert d1['fixed_work']['labels_crc32'] == d2['fixed_work']['labels_crc32']      # the gathered labels of all rooms equal the one-rank run's
    assert abs(d2['fixed_work']['waves_per_rank'] - 1.0) < 1e-9 and abs(d1['fixed_work']['waves_per_rank'] - 2.0) < 1e-9      # 8 jobs, 4 slots, 2 / 1 ranks
    # the default sizing of the leg: the same R for every N, and four waves of rooms per slot and rank at N = 8
    r3, l3 = run_bench(['--gpus', '1', '--steps', '1', '--warmup', '1', '--step-ms', '2', '--rooms', '3', '--best-slots', '', '--steady-slots', '', '--named-configs', '0', '--cpu-seconds', '0',
                        '--p0-rooms', '0', '--cache', str(tmp_path / 'cache')])
    assert r3.returncode == 0, r3.stderr[-3000:]
    d3 = json.loads(l3[0])
    assert d3['fixed_work']['rooms'] == 4 * 8 * 3 and d3['fixed_work']['waves_per_rank_at_8_gpus'] >= 4.0
    assert 0 < d3['fixed_work']['lpt_balance_at_8_gpus']['by_points'] <= 1.0


def test_bench_refuses_a_world_size_that_contradicts_gpus(cuda_device, tmp_path):
    r, lines = run_bench(['--gpus', '4', '--steps', '1', '--warmup', '0', '--named-configs', '0', '--cache', str(tmp_path / 'cache')],
                         env={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'}, timeout=300)
    assert r.returncode != 0 and not lines and '--gpus 4' in (r.stderr + r.stdout)
