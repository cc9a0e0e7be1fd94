#!/usr/bin/env python3
"""bench.py -- region-grow throughput on synthetic S3DIS-Area-5-shaped rooms (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torch.distributed environment the script starts its N ranks itself (torch.distributed.run, 127.0.0.1); a
WORLD_SIZE that contradicts --gpus is an error.  Inputs (13-D room features, weights) are resident in HBM before any clock starts.

  steady leg (``value``, instance-steps/s).  The rooms of the Area-5-shaped set (BASELINE.json configs[1]), `--rooms` = 68 of them
      in flight; every in-flight room takes region-grow steps (box query, medians, sampling, one LrgNet evaluation on its 512 + 512
      point sets, mask update; test_region_grow.py:208-306), a room that finishes gets its 1-NN fill-in and the slot goes on with
      the next room job (the 68 geometries under fresh random-stream keys), so the batch stays full.
        free-running launches (default up to 96 slots): a *step* is ONE launch of lrg_grow_async with a budget of --step-ms
            (25 ms) in which every slot takes as many grow steps as it can, then the fill-ins of the rooms that finished;
        lock-step iterations (--mode lockstep, and above 96 slots): a *step* is --iters-per-step (512) iterations of
            lrg_grow_step_packed over all slots, on the automatic number of lanes.
      W warm-up steps, then EXACTLY K steps timed between barriers.  N > 1: every rank its own set (weak scaling, no collective in
      the loop); the device counters are summed with one all-reduce after the clock stops.
      An instance-step is counted per slot and iteration of the loop, whatever the number of rows its LrgNet evaluation needed: the
      distinct rows of a 512 + 512 set are evaluated, the copies that pad it are not (``rows_evaluated_fraction``), so the unit is a
      PADDED-EQUIVALENT one -- the roofline object prices the rows that were evaluated, not 271.7 MFLOP per instance-step.
  fixed-work leg (``rooms_per_sec``).  R = max(8 per geometry, 4 waves x 8 GPUs x slots per GPU) room jobs (2 176 at 68 slots: the 68
      geometries x 32 random-stream keys) sharded over the N ranks by point count (longest first), pushed through
      `--rooms` slots per GPU from reset to final labels (grow + fill-in), the labels gathered over RCCL -- the only collective of
      the path.  Same R for every N (strong scaling): at N = 8 a rank still has four waves of rooms per slot (``waves_per_rank``), so
      its time is throughput, not the critical path of its longest room.  ``fixed_work_best``: the same R jobs with the number of
      slots per GPU chosen by a short sweep (rooms/s is not a property of 68 slots).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time
import zlib

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

BYTES_PER_INSTANCE_STEP = 10297344      # layer-streamed algorithmic HBM bytes of one LrgNet evaluation
FLOPS_PER_INSTANCE_STEP = 271712256     # hoisted-head FLOPs of one LrgNet evaluation (512 + 512 rows)
FLOPS_PER_BRANCH_ROW = 165504           # 2 * (13*64 + 64*64 + 64*64 + 64*128 + 128*512)
FLOPS_PER_HEAD_ROW = 98816              # 2 * (64*256 + 256*128 + 128*2)
FLOPS_POOLED_GEMM = 1048576             # per instance: two heads x 2 * 1024 * 256
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
FP32_MATRIX_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: fp32-input MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--mode', default='auto', choices=['auto', 'free', 'lockstep'],
                    help='auto: free-running launches (lrg_grow_async) up to 96 greedy slots per GPU, lock-step iterations above')
    ap.add_argument('--step-ms', type=float, default=25.0, help='free-running launches: budget of one launch = one step')
    ap.add_argument('--iters-per-step', type=int, default=512, help='lock-step iterations per (macro-)step')
    ap.add_argument('--fill-cus', type=int, default=0, help='free-running launches: CUs left out of the launches for the fill-ins of finished rooms (0: fill-ins between two launches)')
    ap.add_argument('--steady-slots', default='192', help='the steady leg again with this many rooms in flight, reported as steady_more_rooms_in_flight (empty = skip)')
    ap.add_argument('--best-slots', default='68,136,192,272,320,400,544', help='slot counts of the fixed_work_best sweep (empty = skip)')
    ap.add_argument('--rooms', type=int, default=68, help='rooms in flight per GPU (the Area-5 set has 68)')
    ap.add_argument('--restarts', type=int, default=1)
    ap.add_argument('--workload', default='area5', choices=['area5', 'kitti', 'scannet'],
                    help='area5: 68 Area-5-shaped rooms (configs[1]); kitti: 100 k-point scenes at 0.3 m (configs[4])')
    ap.add_argument('--policy', default='net', choices=['net', 'gt', 'threshold'],
                    help="mask policy: 'net' = the reference's Bernoulli draws against the network's confidence "
                         "(test_region_grow.py:266-267); 'gt' = its commented-out ground-truth masks (:268-269). "
                         "The network is evaluated every step either way")
    ap.add_argument('--weights', default='trained', choices=['trained', 'random'],
                    help="'trained': LrgNet trained by this repository's own training path on synthetic Area-5-shaped rooms "
                         "(learn_region_grow_amd/weights, tools/train_synthetic.py) -- the reference's policy then gives Area-5-like "
                         "dynamics (~100 labeled regions and ~1 500 steps per room); 'random': seeded random weights (growth degenerates "
                         "under 'net': use --policy gt with them)")
    ap.add_argument('--net-mode', default='fused', choices=['fused', 'streamed', 'streamed-tiles'])
    ap.add_argument('--packed', type=int, default=1, help='1: lrg_grow_step_packed (front kernel + packed rows); 0: the nine-launch lrg_grow_step')
    ap.add_argument('--graph', type=int, default=4, help='packed iterations replayed per host call from a HIP graph (0 = plain launches)')
    ap.add_argument('--fill', type=int, default=1, help='1: finished rooms get their 1-NN fill-in before they are recycled')
    ap.add_argument('--lanes', type=int, default=0, help='groups of slots on their own HIP streams; 0 = auto')
    ap.add_argument('--cu-partition', type=int, default=0, help='1: lanes on disjoint sets of compute units')
    ap.add_argument('--fixed-rooms', type=int, default=-1,
                    help='room jobs of the fixed-work leg over ALL ranks (0 = skip; default: max(8 jobs per geometry, 4 waves x 8 GPUs x slots per GPU) -- '
                         'SURVEY.md 8e: the same R for every N, and at N = 8 every rank still pushes four waves of rooms through its slots)')
    ap.add_argument('--cpu-seconds', type=float, default=25.0, help='budget of the CPU-baseline sample (0 = skip)')
    ap.add_argument('--cpu-box-seconds', type=float, default=12.0,
                    help='cpu_baseline.all_cores: the oracle on many rooms at once, one single-threaded process per room, for this long (0 = skip)')
    ap.add_argument('--speculate', type=int, default=-1,
                    help='regions of ONE room grown side by side in the steady and fixed-work legs (RegionGrower(speculate=K): committed in seed order, voided and grown '
                         'again on a conflict, identical labels); -1 = grow.auto_speculate(rooms in flight): 3 up to 16 rooms, 2 up to 32, off above (the 68-room line); '
                         '0 = off')
    ap.add_argument('--one-room-ks', default='1,2,3,4,6', help='one_room_per_gpu: the speculation depths tried on ONE room on the chip (empty = skip)')
    ap.add_argument('--one-rank-collective', type=int, default=1,
                    help='--gpus 1: 1 = a one-rank RCCL process group is brought up and the fixed-work leg\'s label gather goes through its all_gather '
                         '(the nccl branch of learn_region_grow_amd/dist.py executed on the device); 0 = the single-rank short cut')
    ap.add_argument('--cache', default=os.environ.get('LRG_CACHE', '/tmp/lrg_cache'))
    ap.add_argument('--named-configs', type=int, default=1, help='1: the ScanNet-shaped (8 rooms in flight) and KITTI-shaped (8 scenes in flight) configurations as short runs of their own, reported as named_configs')
    ap.add_argument('--p0-rooms', type=int, default=4, help='rooms of the preprocessing (P0) side measurement (0 = skip)')
    return ap.parse_args()


def _hoisted_numpy_net(weights):
    """LrgNet forward with the pooled-feature product hoisted out of the per-point head (one [1,1024]x[1024,256] product
    per instance instead of 512): the arithmetic a tuned CPU implementation would do.  'Strong CPU' leg only."""
    W = {k: np.asarray(v, np.float32) for k, v in weights.items()}

    def branch(x, pre):
        h, convs = x, []
        for i in range(5):
            h = np.maximum(h @ W[pre + 'kernel%d' % i][0] + W[pre + 'bias%d' % i], 0)
            convs.append(h)
        return convs

    def head(pooled, local, pre):
        k0 = W[pre + 'kernel0'][0]
        h = np.maximum(local @ k0[pooled.shape[-1]:] + (pooled @ k0[:pooled.shape[-1]] + W[pre + 'bias0'])[:, None, :], 0)
        h = np.maximum(h @ W[pre + 'kernel1'][0] + W[pre + 'bias1'], 0)
        return h @ W[pre + 'kernel2'][0] + W[pre + 'bias2']

    def net(xi, xn):
        ci, cn = branch(np.asarray(xi, np.float32), 'lrg_'), branch(np.asarray(xn, np.float32), 'lrg_neighbor_')
        pooled = np.concatenate([ci[-1].max(axis=1), cn[-1].max(axis=1)], axis=-1)
        return head(pooled, cn[1], 'lrg_add_'), head(pooled, ci[1], 'lrg_remove_')
    return net


class _Stop(Exception):
    pass


def _cpu_room(room, weights, seconds, policy, faithful, net_fn, key):
    """One room on the CPU oracle (grow, then fill-in) under a time budget: (steps taken, grow seconds, fill seconds or None)."""
    from oracle import grow_ref, rng_ref      # CPU baseline leg only
    t0 = time.time()
    count = [0]

    def hook(d):
        count[0] += 1
        if time.time() - t0 > seconds:
            raise _Stop()
    try:
        res = grow_ref.grow_room(room['points'], room['obj_id'], room['order'], weights, rng_ref.LegacyStream(key),
                                 faithful=faithful, net_fn=net_fn, hook=hook, fill=False, policy=policy)
    except _Stop:
        return count[0], time.time() - t0, None
    t1 = time.time()
    grow_ref.fill_unlabeled(room['points'], res.cluster_label)
    return count[0], t1 - t0, time.time() - t1


def cpu_baseline(rooms, weights, seconds, policy, gpu_room_steps, box_seconds=0.0):
    """The oracle (faithful NumPy restatement of test_region_grow.py:175-316: per-point Python voxel-set loop, un-hoisted
    1088-wide head) on this box's host cores: the median-size room of the set, grown to the end and filled in when the
    budget allows (rooms/s = 1 / that time), else extrapolated from its step rate and the steps the GPU run took for the same
    room.  Beside it a 'strong CPU' step rate: set membership vectorised, head hoisted."""
    order = np.argsort([len(r['points']) for r in rooms])
    mid = int(order[len(order) // 2])
    room = rooms[mid]
    n, t_grow, t_fill = _cpu_room(room, weights, seconds * 0.75, policy, True, None, 0)
    try:                                       # threads the NumPy BLAS actually runs the matrix products on
        import threadpoolctl
        blas_threads = max([i['num_threads'] for i in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count()
    out = dict(value=n / t_grow, unit='instance-steps/s', cores=blas_threads, kind='port')
    if t_fill is not None:
        out['rooms_per_sec'] = 1.0 / (t_grow + t_fill)
        how = 'grown to the end (%d steps, %.1f s) and filled in (%.1f s): rooms/s = 1 / %.1f s' % (n, t_grow, t_fill, t_grow + t_fill)
    else:
        steps_room = gpu_room_steps.get(mid) or 900
        out['rooms_per_sec'] = (n / t_grow) / steps_room
        out['rooms_per_sec_extrapolated'] = True
        how = ('%d steps in %.1f s, stopped by the budget; rooms/s extrapolated as step rate / %d steps (what the GPU run took for '
               'this room), fill-in not included' % (n, t_grow, steps_room))
    out['sample'] = ('the median-size room of the set (%d points), oracle.grow_ref (faithful=True, policy=%s), %s; BLAS threads = '
                     'cores, Python loops single-threaded' % (len(room['points']), policy, how))
    n2, t2, _ = _cpu_room(room, weights, seconds * 0.25, policy, False, _hoisted_numpy_net(weights), 0)
    out['strong'] = dict(value=n2 / t2, unit='instance-steps/s',
                         sample='%d grow steps of the same room, vectorised voxel-set membership + hoisted head (NumPy/BLAS), %.1f s' % (n2, t2))
    if box_seconds > 0:
        # the figure quoted as the baseline: the whole box at work on many rooms; the single room (BLAS threads = cores, one Python thread) stays beside it
        box = cpu_baseline_all_cores(rooms, weights, box_seconds, policy, gpu_room_steps)
        box['one_room'] = {k: v for k, v in out.items() if k != 'strong'}
        box['strong'] = out['strong']
        return box
    return out


def _cpu_pool_worker(job):
    room, weights, seconds, policy, key = job
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(1)
    except Exception:
        pass
    n, t_grow, t_fill = _cpu_room(room, weights, seconds, policy, True, None, key)
    return n, t_grow, t_fill


def cpu_baseline_all_cores(rooms, weights, seconds, policy, gpu_room_steps):
    """The same oracle on MANY rooms at once, one single-threaded process per room (the rooms are independent, test_region_grow.py:110-183: what a
    user of the reference with a many-core host would do): rooms spread evenly over the set's size order, each grown for `seconds` (or to its end).
    value = all steps / the longest worker's time."""
    import multiprocessing as mp
    cores = max(1, min(len(rooms), (os.cpu_count() or 2) - 2, 64))
    order = np.argsort([len(r['points']) for r in rooms])
    # ... and beside them the SMALLEST rooms of the set grown to the end and filled in (up to four windows each), so that rooms/s is also MEASURED on
    # whole rooms, not only extrapolated from a step rate: eight rooms where the box has 32+ cores
    n_small = min(8, cores // 4, len(rooms))
    workers = cores - n_small
    picks = [int(order[(2 * i + 1) * len(order) // (2 * workers)]) for i in range(workers)]
    small = [int(i) for i in order[:n_small]]
    keep = ('points', 'obj_id', 'order')
    jobs = [({k: rooms[i][k] for k in keep}, weights, seconds, policy, 0) for i in picks]
    jobs += [({k: rooms[i][k] for k in keep}, weights, 4.0 * seconds, policy, 0) for i in small]
    saved = {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS')}
    os.environ.update({k: '1' for k in saved})                    # (the children are spawned: they read these when NumPy loads)
    t0 = time.time()
    try:
        with mp.get_context('spawn').Pool(workers + n_small) as pool:
            res_all = pool.map(_cpu_pool_worker, jobs, chunksize=1)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    wall = time.time() - t0
    res, res_small = res_all[:workers], res_all[workers:]
    steps = sum(r[0] for r in res)
    busy = max(r[1] + (r[2] or 0.0) for r in res)
    finished = sum(1 for r in res if r[2] is not None)
    known = [gpu_room_steps[i] for i in picks if gpu_room_steps.get(i)]
    out = dict(value=steps / busy, unit='instance-steps/s', cores=workers, cores_busy=workers + n_small, kind='port', rooms_finished=finished, wall_seconds=wall,
               per_core=steps / busy / workers)
    if known:
        # (extrapolated: the step rate of the sample over the mean steps the GPU run took for these rooms -- kept beside the measured figure below)
        out['rooms_per_sec_extrapolated_set_mean'] = (steps / busy) / float(np.mean(known))
        out['rooms_per_sec'] = out['rooms_per_sec_extrapolated_set_mean']
        out['rooms_per_sec_extrapolated'] = True
    done_small = [(i, r) for i, r in zip(small, res_small) if r[2] is not None]
    if done_small:
        # measured, on whole rooms: every one of these was grown to its end and filled in by one core; rooms/s of the box = cores / mean seconds per room.
        # (The smallest rooms of the set: fewer steps, and cheaper ones -- the reference's mask update is a Python loop over all N points per step -- than
        #  the set's mean room; an upper bound for the set, next to the extrapolated figure above.)
        secs = [r[1] + r[2] for _, r in done_small]
        out['measured_small_rooms'] = dict(rooms_finished=len(done_small), rooms_started=n_small, points=[len(rooms[i]['points']) for i, _ in done_small],
                                           steps=[r[0] for _, r in done_small], seconds_each=secs, cores_used=len(done_small),
                                           rooms_per_sec_per_core=len(secs) / float(np.sum(secs)), rooms_per_sec_box=(workers + n_small) * len(secs) / float(np.sum(secs)),
                                           what='the %d smallest rooms of the set, each grown to its end and filled in by one single-threaded process: '
                                                'rooms/s per core = rooms / the sum of their seconds; x the %d cores in use = the box' % (n_small, workers + n_small))
        out['rooms_finished'] = finished + len(done_small)
        # the headline key is the MEASURED figure (whole rooms, grown to the end and filled in); the extrapolated one for the set's mean room stays beside it
        out['rooms_per_sec'] = out['measured_small_rooms']['rooms_per_sec_box']
        out['rooms_per_sec_extrapolated'] = False
        out['rooms_per_sec_what'] = 'measured on the %d smallest rooms of the set (an upper bound for the set); rooms_per_sec_extrapolated_set_mean is the step rate over the mean steps per room' % len(done_small)
    out['sample'] = ('%d rooms (evenly over the set\'s sizes: %d .. %d points), one single-threaded process each, oracle.grow_ref (faithful=True, policy=%s) for '
                     '%.0f s or to the room\'s end: %d steps, longest worker %.1f s, %d rooms grown to the end; rooms/s = step rate / the mean steps the GPU run '
                     'took for these rooms' % (workers, min(len(rooms[i]['points']) for i in picks), max(len(rooms[i]['points']) for i in picks), policy,
                                               seconds, steps, busy, finished))
    return out


def p0_rates(n_rooms, dev):
    """Preprocessing P0 (test_region_grow.py:119-173) is upstream of the timed loop and reported separately (SURVEY.md
    8d): rooms/s from raw points in host memory to the 13-feature room, GPU (all on device / LAPACK finish) and host."""
    import torch
    from learn_region_grow_amd import preprocess, preprocess_gpu, synthetic
    targets = [synthetic.AREA5_POINTS[(7 * i) % len(synthetic.AREA5_POINTS)] for i in range(n_rooms)]
    raws = []
    for i, t in enumerate(targets):
        r = synthetic.area5_shaped_room(t, 9000 + i).astype(np.float32)
        raws.append((r[:, :6], r[:, 6].astype(int), r[:, 7].astype(int)))
    preprocess_gpu.preprocess_room(*raws[0], device=dev)
    out = {}
    for name, fn in (('gpu', lambda raw: preprocess_gpu.preprocess_room(*raw, device=dev)),
                     ('gpu_exact', lambda raw: preprocess_gpu.preprocess_room(*raw, eig='exact', device=dev)),
                     ('gpu_lapack_finish', lambda raw: preprocess_gpu.preprocess_room(*raw, eig='lapack', device=dev)),
                     ('host_numpy', lambda raw: preprocess.preprocess_room(*raw))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for raw in raws:
            fn(raw)
        torch.cuda.synchronize()
        out[name + '_rooms_per_sec'] = n_rooms / (time.perf_counter() - t0)
    out['sample'] = '%d Area-5-shaped rooms, %s raw points, host memory in / host memory out' % (n_rooms, [len(r[0]) for r in raws])
    return out


def one_room_corner(net, rooms, picks, ks, grow_kw, dev, step_us):
    """BASELINE configs 3 and 5 as they are named: ONE room (or scene) on the chip.  A room is a chain of dependent steps (test_region_grow.py:186-188: the next
    seed is the next point the regions so far have left unvisited), so one slot per room leaves a 256-CU part on a single ~70 us chain; with speculation
    (RegionGrower(speculate=K)) the regions of the next K seeds grow side by side and are committed in order.  Per room and K: seconds from reset to final labels
    (grow + fill-in), committed steps/s, what was thrown away, and whether the labels equal the K = 1 run's (they must)."""
    import torch
    from learn_region_grow_amd.grow import RegionGrower
    out = {}
    kw = {k: v for k, v in grow_kw.items() if k != 'graph_iterations'}
    for name, idx in picks:
        room = dict(rooms[idx], room_id=424242 + idx)
        per_k, ref = {}, None
        for K in ks:
          try:      # (a depth that fails is recorded as such; the others still count)
              st = torch.cuda.Stream(device=dev)
              with torch.cuda.stream(st):
                  gr = RegionGrower(net, rooms_in_flight=1, seed=0, free_run=True, free_run_budget_us=int(step_us), speculate=K if K > 1 else 0, **kw)
                  gr.load_rooms([room])
                  torch.cuda.synchronize()
                  best = None
                  for rep in range(2):          # (the second pass: module and allocator warm)
                      gr.reset_room(0)
                      torch.cuda.synchronize()
                      t0 = time.perf_counter()
                      gr.grow_loaded(fill=True)
                      torch.cuda.synchronize()
                      dt = time.perf_counter() - t0
                      best = dt if best is None else min(best, dt)
                  lab = gr.d_filled[:gr.room_n[0]].cpu().numpy()
                  rr = gr._read_rooms()[0]
                  rl = gr.d_rlog.cpu().numpy().reshape(-1, 8)[:rr.n_regions]
                  work = gr.a_work.cpu().numpy()
              committed = int(rl[:, 1].sum())
              if ref is None:
                  ref = lab
              per_k[str(K)] = {'seconds_per_room': best, 'committed_steps': committed, 'committed_steps_per_sec': committed / best, 'regions': int(rr.n_regions),
                               'evaluations_executed_two_passes': int(work[0]), 'regions_voided_two_passes': int(work[4]), 'evaluations_voided_two_passes': int(work[5]),
                               'labels_equal_k1': bool(np.array_equal(lab, ref))}
              gr._release_graph()
              del gr
              torch.cuda.empty_cache()
          except Exception as e:      # noqa: BLE001
            per_k[str(K)] = {'error': repr(e)[:300]}
            torch.cuda.synchronize()
        good = {k: v for k, v in per_k.items() if 'error' not in v}
        if str(ks[0]) not in good:
            out[name] = {'points': int(len(room['points'])), 'by_speculation_depth': per_k}
            continue
        k1 = good[str(ks[0])]['seconds_per_room']
        bestk = min(good, key=lambda k: good[k]['seconds_per_room'])
        out[name] = {'points': int(len(room['points'])), 'by_speculation_depth': per_k, 'best_depth': int(bestk), 'speedup_over_one_chain': k1 / good[bestk]['seconds_per_room'],
                     'all_labels_equal': all(v['labels_equal_k1'] for v in good.values()) and len(good) == len(per_k)}
    out['what'] = ('ONE room in flight on the whole GPU, free-running launches; depth K = regions of that room grown side by side (1 = the single dependent chain), '
                   'reset -> grow -> fill-in, best of two passes')
    return out


def _respawn_under_torchrun(args):
    """`python bench.py --gpus N` without a torch.distributed environment: start the N ranks (one per GPU) and become their launcher."""
    # (--standalone: the launcher finds a free port itself; one picked and closed here could be taken before the ranks start)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def _lib_auto_slots():
    from learn_region_grow_amd import _lib
    return _lib.LRG_FREE_RUN_AUTO_SLOTS


def kernel_sources_sha16():
    """Digest of the kernel sources (csrc/*.hip, *.inl, *.h and the C-ABI header): what a committed counter file was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, 'learn_region_grow_amd', 'csrc', '*.*')) + [os.path.join(REPO, 'include', 'lrg_hip.h')]):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


class _Leg:
    """One grower configuration over a list of room jobs: free-running launches on one stream, or lock-step lanes."""

    def __init__(self, net, jobs, slots, mode, args, grow_kw, seed, dev, step_budget_us):
        import torch
        from learn_region_grow_amd.grow import LanedRegionGrower, RegionGrower
        self.torch, self.dev, self.jobs, self.fill = torch, dev, jobs, bool(args.fill)
        greedy = args.restarts == 1
        small = max(len(j['points']) for j in jobs) <= 32 * 4096 if jobs else True      # (_lib.LRG_FREE_RUN_AUTO_POINTS)
        self.free = greedy and (mode == 'free' or (mode == 'auto' and slots <= _lib_auto_slots() and small)) and bool(grow_kw.get('packed'))
        self.slots = slots
        if self.free:
            # (fill-ins beside the next launch: --fill-cus CUs are left out of the launches, grow.fill_streams)
            from learn_region_grow_amd.grow import fill_streams
            fill_cus = int(os.environ.get('LRG_FREE_RUN_FILL_CUS', args.fill_cus))
            self.stream = fill_streams(dev, fill_cus)[0] if fill_cus > 0 else torch.cuda.Stream(device=dev)
            with torch.cuda.stream(self.stream):
                self.gr = RegionGrower(net, rooms_in_flight=slots, seed=seed, free_run=True, free_run_budget_us=int(step_budget_us),
                                       free_run_fill_cus=fill_cus, speculate=int(getattr(args, 'speculate', 0)),
                                       **{k: v for k, v in grow_kw.items() if k != 'graph_iterations'})
                self.gr.load_rooms(jobs)
            self.growers = [self.gr]
            self.lanes = 1
        else:
            self.lg = LanedRegionGrower(net, rooms_in_flight=slots, lanes=args.lanes if args.lanes > 0 else None, cu_partition=args.cu_partition > 0,
                                        seed=seed, free_run=False, **grow_kw)
            self.lg.load_rooms(jobs)
            self.growers = [g for g in self.lg.growers if g.n_rooms]
            self.lanes = len(self.lg.growers)
        torch.cuda.synchronize()

    def stats(self):
        return sum(g.d_stats[:4].cpu().numpy().astype(np.float64) for g in self.growers)

    def work(self):
        """(evaluations, distinct inlier rows, distinct neighbour rows, 32-row tiles per stack) of the free-running launches so far."""
        return self.gr.a_work.cpu().numpy().astype(np.float64) if self.free else None

    def grow_all(self):
        """Every job from reset to final labels (grow + fill-in) on the device."""
        if self.free:
            with self.torch.cuda.stream(self.stream):
                self.gr.grow_loaded(fill=self.fill)
        else:
            self.lg.grow_loaded(fill=self.fill)
        self.torch.cuda.synchronize()

    def labels(self):
        """-> (flat int32 labels of this leg's rooms in grower order, job index of each room, its length)."""
        torch = self.torch
        flat = torch.cat([g.d_filled[int(g.room_off[k]):int(g.room_off[k]) + g.room_n[k]] for g in self.growers for k in range(g.n_rooms)]) \
            if self.jobs else torch.zeros(0, dtype=torch.int32, device=self.dev)
        if self.free:
            order = list(range(self.gr.n_rooms))
        else:
            order = [i for g in self.growers for i in g.room_index]
        lens = [n for g in self.growers for n in g.room_n]
        return flat, order, lens

    def room_steps(self):
        """job index -> instance-steps of its pass (region log)."""
        out = {}
        k0 = 0
        for g in self.growers:
            rl = g.d_rlog.cpu().numpy()
            rr = g._read_rooms()
            idx = list(range(g.n_rooms)) if self.free else g.room_index
            for k in range(g.n_rooms):
                o = int(g.room_off[k])
                out[idx[k]] = int(rl[o:o + rr[k].n_regions, 1].sum())
        return out

    def close(self):
        for g in self.growers:
            g._release_graph()


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        _respawn_under_torchrun(args)
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner through C stdio when a communicator is
    # created, flushed at exit -- after the line): everything but the line goes to stderr; the line is written to the real stdout at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from learn_region_grow_amd import _lib, synthetic, workloads, dist as lrg_dist
    from learn_region_grow_amd.lrgnet import LrgNetHIP, _ptr

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or without a torch.distributed environment)'
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # LRG_BENCH_ONE_DEVICE=1 (testing on a 1-GPU box): every rank uses cuda:0 and the collectives go over gloo
    one_dev = os.environ.get('LRG_BENCH_ONE_DEVICE') == '1'
    if one_dev and world > 1 and args.mode == 'auto':
        args.mode = 'lockstep'      # (ranks sharing one chip: no two free-running launches side by side)
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    coll_dev = dev
    backend = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if one_dev:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            coll_dev = torch.device('cpu')
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        backend = dist.get_backend()
    force_coll = False
    collective_error = None
    if world == 1 and args.one_rank_collective and not dist.is_initialized():
        # One GPU: the exchange of the N > 1 path (learn_region_grow_amd/dist.py: all_gather of the label buffers, all_reduce of the counts) still
        # runs, over a one-rank RCCL communicator -- so that the code an 8-GPU run takes has executed on the device in every record of this
        # script.  Nothing depends on it: if RCCL cannot be brought up the single-rank short cut is taken and the line says why.
        try:
            import socket
            sck = socket.socket()
            sck.bind(('127.0.0.1', 0))
            port = sck.getsockname()[1]
            sck.close()
            dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=dev)
            backend = dist.get_backend()
            force_coll = True
        except Exception as e:      # noqa: BLE001 -- reported in the line
            collective_error = '%s: %s' % (type(e).__name__, e)
            backend = None

    weights = synthetic.load_trained_weights() if args.weights == 'trained' else synthetic.make_synthetic_weights(seed=0)
    resolution = 0.1
    if args.workload == 'kitti':
        resolution = 0.3
        base = workloads.kitti_scenes(min(args.rooms, 8), seed_base=5000, cache_dir=args.cache)
    elif args.workload == 'scannet':
        base = workloads.scannet_rooms(min(args.rooms, 39), seed_base=7000, cache_dir=args.cache)
    else:
        base = workloads.area5_rooms(min(args.rooms, 68), seed_base=1000, cache_dir=args.cache)
    slots = min(args.rooms, 8) if args.workload == 'kitti' else args.rooms
    if args.speculate < 0:
        from learn_region_grow_amd.grow import auto_speculate
        args.speculate = auto_speculate(slots) if (args.restarts == 1 and args.mode != 'lockstep') else 0
    net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev, mode=args.net_mode).load_weights(weights)
    packed = bool(args.packed) and args.net_mode == 'fused' and max(len(r['points']) for r in base) <= (
        _lib.LRG_PACKED_MAX_POINTS if args.packed > 1 else _lib.LRG_PACKED_AUTO_POINTS)      # --packed 2 forces it up to 131072 points
    graph = args.graph if packed else 0
    if graph and args.iters_per_step % graph:
        raise SystemExit('--iters-per-step must be a multiple of --graph')
    grow_kw = dict(restarts=args.restarts, rng='counter', policy=args.policy, resolution=resolution, packed=packed, graph_iterations=graph)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def jobs_of(n, key0):
        """n room jobs: the geometries of the set, job j under the random-stream key key0 + j."""
        return [dict(base[j % len(base)], room_id=key0 + j) for j in range(n)]

    # ------------------------------------------------------------------------------------------------------------------
    # steady leg
    # ------------------------------------------------------------------------------------------------------------------
    step_us = args.step_ms * 1e3
    free_steady = args.restarts == 1 and packed and (args.mode == 'free' or (args.mode == 'auto' and slots <= _lib.LRG_FREE_RUN_AUTO_SLOTS and
                                                                         max(len(r['points']) for r in base) <= _lib.LRG_FREE_RUN_AUTO_POINTS))
    ev_pairs = []
    if free_steady:
        # enough jobs for the warm-up and the timed steps at twice the rate seen so far (~650 rooms/s per GPU), at least two per slot
        n_jobs = max(2 * slots, int((args.warmup + args.steps) * args.step_ms * 1e-3 * 1500) + slots)
        leg = _Leg(net, jobs_of(n_jobs, 1000000 * (rank + 1)), slots, 'free', args, grow_kw, rank, dev, step_us)
        gr = leg.gr
        gr.room_order_mode = 'loaded'      # (the set's geometries in turn, as in every round so far: the timed window sees the same rooms)
        with torch.cuda.stream(leg.stream):
            gr.free_run_begin()

        def iterate(steps, timed):
            with torch.cuda.stream(leg.stream):
                for _ in range(steps):
                    if timed:      # HIP events round the launch, on the stream it is launched on
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(leg.stream)
                        gr.enqueue_free_run()
                        e1.record(leg.stream)
                        ev_pairs.append((e0, e1))
                        gr.poll_done()
                        if args.fill:
                            gr.fill_many([r_ for r_, f_ in zip(gr.done_rooms, gr.done_filled) if not f_])      # (none with the in-launch fill-in)
                        gr.rooms_finished += len(gr.done_rooms)
                        gr.done_rooms = []
                    else:
                        gr.free_run_step(fill=bool(args.fill))
        step_what = 'one free-running launch of lrg_grow_async with a budget of %.0f ms (every in-flight room takes as many region-grow steps ' \
                    'as it can), then the fill-ins of the rooms that finished' % args.step_ms
        iterations = None
    else:
        from learn_region_grow_amd.grow import RegionGrower, auto_lanes, lane_streams
        rooms = base if rank == 0 else ({'kitti': workloads.kitti_scenes, 'scannet': workloads.scannet_rooms}.get(args.workload, workloads.area5_rooms)(
            len(base), seed_base={'kitti': 5000, 'scannet': 7000}.get(args.workload, 1000) + 100 * rank, cache_dir=args.cache))
        rooms = [rooms[i % len(rooms)] if i < len(rooms) else dict(rooms[i % len(rooms)], room_id=50000 + i) for i in range(slots)]
        n_lanes = max(1, min(args.lanes, len(rooms))) if args.lanes > 0 else auto_lanes(len(rooms) * args.restarts)
        by_size = sorted(range(len(rooms)), key=lambda i: -len(rooms[i]['points']))
        parts = [[i for i in by_size[k::n_lanes]] for k in range(n_lanes)]
        lane_streams_ = lane_streams(dev, n_lanes, args.cu_partition > 0)
        growers = []
        for k in range(n_lanes):
            with torch.cuda.stream(lane_streams_[k]):
                g_ = RegionGrower(net, rooms_in_flight=len(parts[k]), seed=rank, free_run=False, **grow_kw)
                g_.load_rooms([rooms[i] for i in parts[k]])
                for g in range(g_.n_groups):
                    g_.bind(g, g)
                growers.append(g_)
        torch.cuda.synchronize()

        class _L:
            pass
        leg = _L()
        leg.growers, leg.free, leg.lanes, leg.slots = growers, False, n_lanes, slots
        leg.stats = lambda: sum(g_.d_stats[:4].cpu().numpy().astype(np.float64) for g_ in growers)
        leg.close = lambda: [g_._release_graph() for g_ in growers]

        def iterate(steps, timed):
            per_call = graph if graph else 1
            for _ in range(0, steps * args.iters_per_step, per_call):
                for lane, g_ in enumerate(growers):
                    with torch.cuda.stream(lane_streams_[lane]):
                        g_.enqueue()
                        gs = g_.poll_done()               # finished rooms get their fill-in (:308-316) and restart at once
                        if args.fill:
                            g_.fill_many([g_.group_room[g] for g in gs])
                        for g in gs:
                            r = g_.group_room[g]
                            g_.reset_room(r)
                            g_.bind(g, r)
        step_what = '%d lock-step iterations of lrg_grow_step_packed (every in-flight room takes one region-grow step per iteration)' % args.iters_per_step
        iterations = args.steps * args.iters_per_step

    iterate(args.warmup, False)
    barrier()
    dbg0 = leg.gr.free_run_ticks() if leg.free else None      # (debug builds only: stage-by-stage ticks of the timed launches -> stderr)
    s0 = leg.stats()
    w0 = leg.work() if leg.free else None
    t0 = time.perf_counter()
    iterate(args.steps, True)
    barrier()
    t1 = time.perf_counter()
    s1 = leg.stats()
    w1 = leg.work() if leg.free else None
    if dbg0 is not None and rank == 0:
        print(json.dumps(leg.gr.free_run_breakdown(since=dbg0)), file=sys.stderr)
    elapsed = lrg_dist.allreduce_max(t1 - t0, device=coll_dev)
    # (speculation: the device's step counter also holds the steps of regions that were voided and grown again -- `value` counts the steps regions KEEP)
    voided_steps = float(w1[6] - w0[6]) if (leg.free and args.speculate > 1) else 0.0
    inst_steps, rooms_cycled, seeds = lrg_dist.allreduce_sum([float(s1[2] - s0[2]) - voided_steps, float(s1[1] - s0[1]), float(s1[0] - s0[0])], device=coll_dev)
    if int(s1[3]):
        raise SystemExit('bench.py: lrg_grow_async gave up on a hand-over (%d): the numbers would be invalid' % int(s1[3]))

    # ------------------------------------------------------------------------------------------------------------------
    # roofline of the kernel the timed loop runs.  Free-running: lrg_grow_async_kernel IS the loop -- its launches were timed with HIP
    # events on their stream, its algorithmic FLOPs follow from the rows its tile teams evaluated (device counters: distinct rows of
    # the 512 + 512 sets, pooled products).  Lock-step: the packed branch stack (the longest of the five launches), timed here with
    # HIP events on the rows of the lanes' last iteration.
    # ------------------------------------------------------------------------------------------------------------------
    S = slots
    roof = {'bound': 'mfma', 'peak': FP32_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s'}
    if leg.free:
        dw = w1 - w0
        launch_ms = [a.elapsed_time(b) for a, b in ev_pairs]
        flops = (dw[1] + dw[2]) * (FLOPS_PER_BRANCH_ROW + FLOPS_PER_HEAD_ROW) + dw[0] * FLOPS_POOLED_GEMM
        t_kernel = sum(launch_ms) * 1e-3
        ach = flops / t_kernel / 1e12
        roof.update({'kernel': 'lrg_grow_async_kernel: the whole timed loop in one launch per step (front workgroups + tile teams: branch stacks, '
                               'pooled products, head stacks on fp32 MFMA, v_mfma_f32_32x32x2_f32)',
                     'achieved': ach, 'frac': ach / FP32_MATRIX_PEAK_TFLOPS,
                     'algorithmic_flops_in_loop': flops, 'flops_per_launch': flops / max(len(launch_ms), 1), 'launches': len(launch_ms),
                     'avg_us': 1e3 * float(np.mean(launch_ms)), 'kernel_seconds': t_kernel,
                     'flops_definition': 'distinct rows evaluated x (165 504 branch + 98 816 head FLOP per row) + evaluations x 1 048 576 (pooled products); '
                                         'copies that pad a set to 512 rows and the copies that pad a slot\'s rows to whole 32-row tiles are NOT counted',
                     'evaluations': dw[0], 'rows_evaluated': [dw[1], dw[2]], 'rows_evaluated_fraction': (dw[1] + dw[2]) / max(dw[0] * 1024.0, 1.0),
                     'tiles_run_per_stack': dw[3], 'rows_in_tiles_fraction': (dw[1] + dw[2]) / max(dw[3] * 32.0, 1.0),
                     'head_tiles_run': dw[7], 'head_rows_in_tiles_fraction': (dw[1] + dw[2]) / max(dw[7] * 32.0, 1.0),
                     'tiles_note': 'tiles_run_per_stack / rows_in_tiles_fraction: the BRANCH stack (with shared tail tiles the rows beyond a slot\'s last full tile share '
                                   'tiles: fewer of them); the head stack keeps a tile of the slot\'s own for its tail: head_tiles_run / head_rows_in_tiles_fraction',
                     'speculation': ({'depth': args.speculate, 'regions_voided': dw[4], 'evaluations_voided': dw[5], 'steps_voided': dw[6],
                                      'note': 'algorithmic_flops_in_loop counts every evaluation executed, voided ones included (work done, not work kept); '
                                              '`value` counts kept steps only'} if args.speculate > 1 else None),
                     'reproduce': 'rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps %d --warmup %d  (profiles/r06_bench_kernel_stats.csv)'
                                  % (args.steps, args.warmup)})
    else:
        reps = 20
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rows = np.zeros(2)
        act = 0
        loop_ms = branch_ms = 0.0
        for g_ in leg.growers:
            if not g_.packed:
                continue
            sr = g_.p_slot_rows.cpu().numpy()
            rows += sr[:, :2].sum(axis=0)
            act += int((sr[:, 0] > 0).sum())
            nr = g_.p_counters[2:4].clone()
            pb = g_.packed_buffers

            def fwd():
                _lib.check(g_.lib.lrg_forward_packed(ctypes.byref(net._w), pb.x_in, pb.x_nb,
                                                     g_.lib.lrg_packed_rows_center(ctypes.byref(g_.params), ctypes.byref(pb)),
                                                     pb.row_slot_in, pb.row_slot_nb, _ptr(nr), None,
                                                     g_.S, pb.row_cap, pb.add_logits, pb.rmv_logits, pb.workspace, pb.workspace_bytes, 0,
                                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'lrg_forward_packed')
            fwd()
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(reps):
                fwd()
            ev1.record()
            torch.cuda.synchronize()
            loop_ms += ev0.elapsed_time(ev1) / reps
        flops = rows.sum() * (FLOPS_PER_BRANCH_ROW + FLOPS_PER_HEAD_ROW) + act * FLOPS_POOLED_GEMM
        ach = flops / (loop_ms * 1e-3) / 1e12 if loop_ms else 0.0
        roof.update({'kernel': 'lrg_forward_packed as the loop issues it (branch stacks + pooled GEMM + head stacks on the packed distinct rows of the '
                               'lanes\' last iteration), HIP events on the launch stream',
                     'achieved': ach, 'frac': ach / FP32_MATRIX_PEAK_TFLOPS, 'algorithmic_flops_in_loop': flops,
                     'flops_per_launch': flops, 'avg_us': 1e3 * loop_ms, 'packed_rows': [int(rows[0]), int(rows[1])],
                     'rows_evaluated_fraction': float(rows.sum()) / (max(act, 1) * 1024.0), 'slots_evaluated': int(act),
                     'flops_definition': 'distinct rows evaluated x (165 504 + 98 816 FLOP per row) + active slots x 1 048 576'})
    # the dense evaluation (all 512 + 512 rows of S instances): the formulation SURVEY.md 8d prices, a side launch outside the timed loop
    reps = 10
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rs = np.random.RandomState(0)
    d_inl = torch.from_numpy((rs.randn(S, 512, 13) * 0.5).astype(np.float32)).to(dev)
    d_nbr = torch.from_numpy((rs.randn(S, 512, 13) * 0.5).astype(np.float32)).to(dev)
    net.forward(d_inl, d_nbr)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        net.forward(d_inl, d_nbr)
    ev1.record()
    torch.cuda.synchronize()
    fwd_ms = ev0.elapsed_time(ev1) / reps
    dflops = S * FLOPS_PER_INSTANCE_STEP
    roof['dense'] = {'note': 'side launch, NOT in the timed loop: lrg_forward on all 512 + 512 rows of %d instances' % S,
                     'achieved': dflops / (fwd_ms * 1e-3) / 1e12, 'frac': dflops / (fwd_ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
                     'ms_per_launch': fwd_ms, 'algorithmic_flops': dflops,
                     'hbm_accounting': {'note': 'SURVEY.md 8d layer-streamed accounting (what an unfused implementation would stream); the fused kernels '
                                                'keep activations in LDS, so this is NOT traffic', 'bytes_per_instance': BYTES_PER_INSTANCE_STEP,
                                        'GBps': S * BYTES_PER_INSTANCE_STEP / (fwd_ms * 1e-3) / 1e9,
                                        'frac_of_hbm_peak': S * BYTES_PER_INSTANCE_STEP / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    # north_star's MLP metric: the SAME evaluation run layer by layer, every activation through HBM (LRG_FWD_STREAM_TILES: csrc/lrg_stream_layer.inl), a side launch
    # at 1 088 instances, priced with the algorithmic bytes of SURVEY.md 8d (the counters' figure: profiles/r05_traffic_streamed_tiles.json)
    try:
        from learn_region_grow_amd.lrgnet import LrgNetHIP
        Sb = 1088
        snet = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev, mode='streamed-tiles').load_weights(weights)
        s_inl = torch.from_numpy((rs.randn(Sb, 512, 13) * 0.5).astype(np.float32)).to(dev)
        s_nbr = torch.from_numpy((rs.randn(Sb, 512, 13) * 0.5).astype(np.float32)).to(dev)
        snet.forward(s_inl, s_nbr)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            snet.forward(s_inl, s_nbr)
        ev1.record()
        torch.cuda.synchronize()
        s_ms = ev0.elapsed_time(ev1) / reps
        roof['dense']['layer_streamed'] = {'note': 'side launch, NOT in the timed loop: one launch per layer, activations through HBM (mode streamed-tiles), %d instances' % Sb,
                                           'ms_per_evaluation': s_ms, 'algorithmic_bytes': Sb * BYTES_PER_INSTANCE_STEP,
                                           'GBps': Sb * BYTES_PER_INSTANCE_STEP / (s_ms * 1e-3) / 1e9,
                                           'frac_of_hbm_peak': Sb * BYTES_PER_INSTANCE_STEP / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           'counters': 'profiles/r05_traffic_streamed_tiles.json (FETCH_SIZE + WRITE_SIZE: 12.16 GB in 3.41 ms under the profiler = 0.446)'}
        del snet, s_inl, s_nbr
    except Exception as e:       # (a side figure: never the reason the line is missing)
        roof['dense']['layer_streamed'] = {'error': repr(e)}
    # HBM traffic / matrix-pipe occupancy of the loop's kernel: PMC passes of their own (tools/pmc_free_run.sh), quoted from the committed
    # file only when it was measured on this ABI and formulation
    roof['traffic'] = None
    tpath = os.path.join(REPO, 'profiles', 'r06_pmc_free_run.json' if leg.free else 'r02_traffic_loop.json')
    for older in ('r05_pmc_free_run.json', 'r04_pmc_free_run.json', 'r03_pmc_free_run.json'):
        if leg.free and not os.path.exists(tpath):
            tpath = os.path.join(REPO, 'profiles', older)
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if leg.free and tj.get('abi') == _lib.load().lrg_abi_version():
            per_launch = tj.get('hbm_bytes_per_launch')
            roof['traffic'] = per_launch
            roof['traffic_quoted_from'] = {'file': 'profiles/' + os.path.basename(tpath), 'measured_at_commit': tj.get('commit'), 'abi': tj.get('abi'),
                                           'launch_ms': tj.get('launch_ms'), 'hbm_GBps': tj.get('hbm_GBps'), 'frac_of_hbm_peak': tj.get('frac_of_hbm_peak'),
                                           'mfma_util_chipwide': tj.get('mfma_util_chipwide'),
                                           'formulation_measured': tj.get('formulation'), 'kernel_sources_sha16_then': tj.get('kernel_sources_sha16'), 'kernel_sources_sha16_now': kernel_sources_sha16(),
                                           'stale': tj.get('kernel_sources_sha16') != kernel_sources_sha16(),
                                           'note': 'rocprofv3 --pmc passes of their own over the same command; bytes per launch of %.0f ms; stale = the kernel '
                                                   'sources (csrc/*, include/lrg_hip.h) changed since the counters were read' % args.step_ms}
        elif not leg.free:
            roof['traffic'] = tj.get('hbm_bytes_per_iteration')
            roof['traffic_quoted_from'] = {'file': 'profiles/' + os.path.basename(tpath), 'note': 'round-2 measurement of the five lock-step launches, per iteration'}
    leg.close()
    del leg
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------------------------------------------------------
    # fixed-work leg(s): R room jobs over all ranks, reset -> final labels (grow + fill-in) -> RCCL gather
    # ------------------------------------------------------------------------------------------------------------------
    room_steps = {}
    coll_warm = []

    def fixed_work(R, n_slots, measure_fill=False):
        jobs = jobs_of(R, 100000)
        sizes = [len(j['points']) for j in jobs]
        mine = lrg_dist.queue_order(lrg_dist.shard_rooms_lpt(sizes, world)[rank], sizes, n_slots)      # passes over the sizes, the smallest rooms last (dist.queue_order)
        my_jobs = [jobs[j] for j in mine]
        fl = _Leg(net, my_jobs, min(n_slots, max(1, len(my_jobs))), args.mode, args, grow_kw, 0, dev, step_us)
        if not coll_warm:
            # the communicator's first all_gather / all_reduce set up its connections and load its kernels (0.1 s on one rank): once, ahead of the timed region, as
            # the steady leg has its warm-up steps
            coll_warm.append(1)
            lrg_dist.gather_flat_labels([rank], [256 * 1024], torch.ones(256 * 1024, dtype=torch.int32, device=dev), world, device=coll_dev, force_collective=force_coll)
            lrg_dist.allreduce_max(0.0, device=coll_dev, force_collective=force_coll)
            lrg_dist.allreduce_sum([0.0, 0.0], device=coll_dev, force_collective=force_coll)
        barrier()
        tf0 = time.perf_counter()
        fl.grow_all()
        tf_grow = time.perf_counter() - tf0
        flat, order, lens = fl.labels()
        ids = [mine[i] for i in order]
        tg0 = time.perf_counter()
        gathered = lrg_dist.gather_flat_labels(ids, lens, flat, R, device=coll_dev, force_collective=force_coll)
        barrier()
        tf1 = time.perf_counter()
        el = lrg_dist.allreduce_max(tf1 - tf0, device=coll_dev, force_collective=force_coll)
        st = fl.stats()
        wk = fl.work()
        kept = float(st[2]) - (float(wk[6]) if (wk is not None and args.speculate > 1) else 0.0)      # (speculation: the steps of voided regions are not the rooms')
        f_steps, f_rooms = lrg_dist.allreduce_sum([kept, float(len(my_jobs))], device=coll_dev, force_collective=force_coll)
        ok = all(gathered[j] is not None and len(gathered[j]) == sizes[j] and int(gathered[j].min()) > 0 for j in range(R))
        crc = 0
        for j in range(R):                                  # one checksum over every room's final labels, in job order
            crc = zlib.crc32(np.ascontiguousarray(gathered[j], dtype=np.int32).tobytes(), crc) if gathered[j] is not None else crc
        for k, v in fl.room_steps().items():
            room_steps.setdefault(mine[k] % len(base), v)
        leg_roof = None
        if wk is not None:
            # the launches' algorithmic FLOPs by the device counters (as for the headline: distinct rows, pooled products) over the WHOLE leg's wall clock -- bind,
            # launches, fill-ins, the start and the tail where slots run dry included: a lower bound of the kernel's own fraction
            fl_flops = lrg_dist.allreduce_sum([float((wk[1] + wk[2]) * (FLOPS_PER_BRANCH_ROW + FLOPS_PER_HEAD_ROW) + wk[0] * FLOPS_POOLED_GEMM)], device=coll_dev,
                                               force_collective=force_coll)[0]
            leg_roof = {'bound': 'mfma', 'peak': FP32_MATRIX_PEAK_TFLOPS * world, 'unit': 'TFLOP/s', 'achieved': fl_flops / tf_grow / 1e12,
                        'frac': fl_flops / tf_grow / 1e12 / (FP32_MATRIX_PEAK_TFLOPS * world), 'algorithmic_flops': fl_flops, 'seconds': tf_grow,
                        'evaluations': float(wk[0]), 'rows_evaluated_fraction': float(wk[1] + wk[2]) / max(float(wk[0]) * 1024.0, 1.0),
                        'rows_in_tiles_fraction': float(wk[1] + wk[2]) / max(float(wk[3]) * 32.0, 1.0),
                        'note': 'device counters of the launches (rank 0 x ranks) over the grow time of the leg (launches + binding + fill-ins), not over kernel time'}
        out = {'rooms': int(f_rooms), 'seconds': el, 'rooms_per_sec': f_rooms / el, 'instance_steps': f_steps, 'instance_steps_per_sec': f_steps / el,
               'roofline': leg_roof,
               'scaling': 'strong', 'slots_per_gpu': fl.slots, 'waves_per_rank': R / float(world) / max(fl.slots, 1),
               'waves_per_rank_at_8_gpus': R / 8.0 / max(n_slots, 1), 'formulation': 'free-running launches' if fl.free else 'lock-step iterations',
               'lanes': fl.lanes, 'grow_seconds_rank0': tf_grow, 'gather_seconds_rank0': tf1 - tg0, 'rccl_ranks': world,
               'collective_backend': backend, 'collective_executed': bool(world > 1 or force_coll), 'collective_error': collective_error,
               'all_rooms_labeled_after_gather': bool(ok), 'labels_crc32': int(crc), 'given_up': int(st[3]),
               'what': '%d room jobs = the %d geometries x %d random-stream keys, LPT-sharded by point count over %d rank(s), queued in passes over the sizes with the smallest rooms last (dist.queue_order), reset -> grow -> '
                       '1-NN fill-in -> all_gather of the labels' % (R, len(base), (R + len(base) - 1) // len(base), world)}
        if measure_fill and fl.free and rank == 0:
            # P14 (the 1-NN fill-in, test_region_grow.py:308-316; "k-NN at scale" of configs[4]) priced by itself: the fill-ins of this leg's rooms
            # once more (labels are final: the same result), HIP events on the leg's stream; work = 3 x F x U x L FLOP per room (a subtract, a
            # multiply and an add per feature and (unlabeled, labeled) pair) against the fp32 vector peak
            g_ = fl.gr
            lab = g_.d_label.cpu().numpy()
            ul = [(int((lab[int(g_.room_off[k]):int(g_.room_off[k]) + g_.room_n[k]] == 0).sum()), int((lab[int(g_.room_off[k]):int(g_.room_off[k]) + g_.room_n[k]] > 0).sum()))
                  for k in range(min(g_.n_rooms, 64))]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(fl.stream):
                g_.fill_many(list(range(len(ul))))
                e0.record(fl.stream)
                for _ in range(5):
                    g_.fill_many(list(range(len(ul))))
                e1.record(fl.stream)
            torch.cuda.synchronize()
            sec = e0.elapsed_time(e1) * 1e-3 / 5
            fl_flops = 3.0 * 13 * sum(u * l for u, l in ul)
            out['p14_fill'] = {'rooms': len(ul), 'unlabeled_mean': float(np.mean([u for u, _ in ul])), 'labeled_mean': float(np.mean([l for _, l in ul])),
                               'algorithmic_flops': fl_flops, 'seconds': sec, 'achieved': fl_flops / sec / 1e12, 'unit': 'TFLOP/s', 'peak': FP32_MATRIX_PEAK_TFLOPS,
                               'frac': fl_flops / sec / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 'bound': 'fp32 vector (the pairwise float32 order of the reference is kept: no MFMA)',
                               'what': 'lrg_nn1_fill_batch over %d finished rooms of this leg, 3 x 13 x U x L FLOP per room' % len(ul)}
        fl.close()
        del fl
        torch.cuda.empty_cache()
        return out

    fixed = best = None
    if args.fixed_rooms < 0:
        args.fixed_rooms = max(8 * len(base), 4 * 8 * slots)      # R >= 8 GPUs x slots per GPU x 4 waves, the same R for every N
    if args.fixed_rooms > 0 and args.restarts == 1:
        fixed = fixed_work(args.fixed_rooms, slots, measure_fill=True)
        sweep = [int(x) for x in args.best_slots.split(',') if x.strip()] if args.workload != 'kitti' else []
        tried = {slots: fixed}
        for n_slots in sweep:
            if n_slots not in tried and n_slots * world <= args.fixed_rooms:
                tried[n_slots] = fixed_work(args.fixed_rooms, n_slots)
        if len(tried) > 1:
            pick = max(tried, key=lambda k: tried[k]['rooms_per_sec'] if tried[k]['all_rooms_labeled_after_gather'] else -1.0)
            best = dict(tried[pick])
            best['sweep'] = {str(k): {'rooms_per_sec': tried[k]['rooms_per_sec'], 'instance_steps_per_sec': tried[k]['instance_steps_per_sec'],
                                      'formulation': tried[k]['formulation'], 'lanes': tried[k]['lanes'],
                                      'roofline_frac': (tried[k]['roofline'] or {}).get('frac')} for k in sorted(tried)}

    if rank == 0:
        out = {
            'metric': 'region-grow steps/sec (rooms/sec alongside), %s shape' % {'area5': 'S3DIS Area-5', 'scannet': 'ScanNet', 'kitti': 'KITTI'}[args.workload],
            'value': inst_steps / elapsed,
            'unit': 'instance-steps/s',
            'value_note': 'an instance-step = one slot taking one region-grow step, whatever the number of distinct rows its LrgNet evaluation had '
                          '(a padded-equivalent unit: see roofline.rows_evaluated_fraction)',
            'rooms_per_sec': fixed['rooms_per_sec'] if fixed else rooms_cycled / elapsed,
            'rooms_per_sec_steady': rooms_cycled / elapsed,
            'regions_per_sec': seeds / elapsed,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'us_per_instance_step_per_slot': 1e6 * elapsed * S * world / max(inst_steps, 1.0),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('Semantic-KITTI-shaped synthetic scenes (~100 k points at 0.3 m), %d in flight' % S if args.workload == 'kitti' else
                                    'ScanNet-shaped synthetic rooms (%d-room set per GPU, %d in flight, recycled under fresh random-stream keys)' % (len(base), S)
                                    if args.workload == 'scannet' else
                                    'S3DIS Area-5-shaped synthetic rooms (68-room set per GPU, %d in flight, recycled under fresh random-stream keys)' % S) +
                                   (', greedy test_region_grow.py loop' if args.restarts == 1 else
                                    ', test_random_restart.py loop with %d restarts per seed batched per launch' % args.restarts),
                       'step': step_what, 'formulation': 'free-running launches (lrg_grow_async)' if free_steady else 'lock-step iterations (lrg_grow_step_packed)' if packed else 'lrg_grow_step',
                       'rooms_in_flight_per_gpu': S, 'slots_per_gpu': S * args.restarts * (args.speculate if (free_steady and args.speculate > 1) else 1),
                       'speculation_depth': args.speculate if (free_steady and args.speculate > 1) else 0, 'lanes': 1 if free_steady else n_lanes, 'policy': args.policy,
                       'compute_units_left_to_the_fill_ins': (int(os.environ.get('LRG_FREE_RUN_FILL_CUS', args.fill_cus)) if free_steady else 0),
                       'restarts': args.restarts, 'points': '512 inlier + 512 neighbour x 13 features', 'rng': 'counter (Philox) stream',
                       'weights': ('trained on synthetic Area-5-shaped rooms by train_region_grow.py (learn_region_grow_amd/weights)'
                                   if args.weights == 'trained' else 'random, seed 0'), 'net_mode': args.net_mode,
                       'hip_graph_iterations': 0 if free_steady else graph},
            'roofline': roof,
        }
        # P2 (bbox + box query, test_region_grow.py:221-235) priced the way SURVEY.md 8a does -- a pass over the room, ~14 bytes per point and
        # step -- against the HBM peak: what the reference's formulation would stream at this step rate (the front answers most boxes from the
        # room's voxel grid, O(box) instead of O(room): this is an algorithmic figure, not traffic)
        mean_n = float(np.mean([len(r['points']) for r in base]))
        out['p2_box_query'] = {'points_per_room_mean': mean_n, 'algorithmic_bytes_per_instance_step': 14.0 * mean_n,
                               'GBps': 14.0 * mean_n * out['value'] / 1e9, 'frac_of_hbm_peak': 14.0 * mean_n * out['value'] / 1e9 / 8000.0}
        if iterations:
            out['ms_per_iteration'] = 1e3 * elapsed / iterations
            out['config']['iterations_per_step'] = args.iters_per_step
            out['config']['active_fraction'] = inst_steps / (iterations * S * args.restarts * world)
        if fixed:
            if 'p14_fill' in fixed:
                out['p14_fill'] = fixed.pop('p14_fill')
            out['fixed_work'] = fixed
            # what the static LPT sharding (dist.shard_rooms_lpt, by point count) leaves of an 8-rank run of these R jobs: mean / max of the
            # ranks' loads, by points (what the sharding sees) and by the instance-steps the rooms actually took in this run (what time follows)
            jobs8 = jobs_of(args.fixed_rooms, 100000)
            sizes8 = [len(j['points']) for j in jobs8]
            steps8 = [float(room_steps.get(j % len(base), 0)) for j in range(len(jobs8))]
            sh8 = lrg_dist.shard_rooms_lpt(sizes8, 8)
            lp, ls = [sum(sizes8[j] for j in s_) for s_ in sh8], [sum(steps8[j] for j in s_) for s_ in sh8]
            fixed['lpt_balance_at_8_gpus'] = {'by_points': float(np.mean(lp) / max(max(lp), 1)), 'by_instance_steps': float(np.mean(ls) / max(max(ls), 1.0)),
                                              'note': 'mean / max of the eight ranks\' loads under the sharding by point count: the efficiency an 8-GPU run of this leg can '
                                                      'reach at best (no 1 -> 8 curve has been measured by the builder: one GPU per box)'}
        if best:
            out['fixed_work_best'] = best
        if world == 1 and args.steady_slots and args.workload == 'area5' and args.restarts == 1:
            # the steady leg again with more rooms of the same set in flight (a run of this script of its own; `value` stays the
            # 68-room configuration the metric is quoted on)
            sweep = {}
            for sl in [int(x) for x in args.steady_slots.split(',') if x]:
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--gpus', '1', '--rooms', str(sl), '--steps', str(max(4, args.steps // 2)),
                                        '--warmup', str(args.warmup), '--step-ms', str(args.step_ms), '--fixed-rooms', '0', '--best-slots', '', '--steady-slots', '',
                                        '--cpu-seconds', '0', '--p0-rooms', '0', '--policy', args.policy, '--weights', args.weights, '--cache', args.cache],
                                       capture_output=True, text=True, timeout=600)
                    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
                    sweep[str(sl)] = {'value': d['value'], 'unit': d['unit'], 'formulation': d['config']['formulation'],
                                      'us_per_instance_step_per_slot': d.get('us_per_instance_step_per_slot')}
                except Exception as e:      # (informational leg: never fails the line)
                    sweep[str(sl)] = {'error': repr(e)[:200]}
            out['steady_more_rooms_in_flight'] = sweep
        if world == 1 and args.named_configs and args.workload == 'area5' and args.restarts == 1:
            # BASELINE configs 3 and 5 in the driver's own line (round-5 review: "the driver's line carries no KITTI or ScanNet figure"): eight ScanNet-shaped rooms and
            # eight Semantic-KITTI-shaped scenes in flight on this GPU, each a short run of this script of its own (steady leg + the rooms / scenes from reset to labels)
            named = {}
            for wl, fixed_n in (('scannet', 39), ('kitti', 8)):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--gpus', '1', '--workload', wl, '--rooms', '8', '--steps', str(max(4, args.steps // 2)),
                                        '--warmup', str(args.warmup), '--step-ms', str(args.step_ms), '--fixed-rooms', str(fixed_n), '--best-slots', '', '--steady-slots', '',
                                        '--one-room-ks', '', '--named-configs', '0', '--cpu-seconds', '0', '--p0-rooms', '0', '--policy', args.policy, '--weights', args.weights,
                                        '--cache', args.cache], capture_output=True, text=True, timeout=600)
                    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
                    fwk = d.get('fixed_work') or {}
                    named[wl] = {'workload': d['config']['workload'], 'value': d['value'], 'unit': d['unit'], 'formulation': d['config']['formulation'],
                                 'speculation_depth': d['config'].get('speculation_depth'), 'roofline_frac': d['roofline']['frac'],
                                 'fixed_work_rooms_per_sec': fwk.get('rooms_per_sec'), 'fixed_work_rooms': fwk.get('rooms'), 'labels_crc32': fwk.get('labels_crc32'),
                                 'all_rooms_labeled_after_gather': fwk.get('all_rooms_labeled_after_gather')}
                except Exception as e:      # (informational legs: never fail the line)
                    named[wl] = {'error': repr(e)[:200]}
            out['named_configs'] = named
        if world == 1 and args.one_room_ks and args.restarts == 1 and packed:
            ks = [int(x) for x in args.one_room_ks.split(',') if x.strip()]
            by_size = sorted(range(len(base)), key=lambda i: len(base[i]['points']))
            picks = [('scene_0', 0)] if args.workload == 'kitti' else [('median_room', by_size[len(by_size) // 2]), ('largest_room', by_size[-1])]
            try:
                out['one_room_per_gpu'] = one_room_corner(net, base, picks, ks, grow_kw, dev, step_us)
            except Exception as e:      # noqa: BLE001 -- a side measurement must not take the line down
                out['one_room_per_gpu'] = {'error': repr(e)[:300]}
        if world == 1 and args.cpu_seconds > 0:
            out['cpu_baseline'] = cpu_baseline(base, weights, args.cpu_seconds, args.policy, room_steps, args.cpu_box_seconds)
        if world == 1 and args.p0_rooms > 0:
            out['preprocessing_p0'] = p0_rates(args.p0_rooms, dev)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
