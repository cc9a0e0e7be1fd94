#!/usr/bin/env python3
"""bench.py -- region-grow throughput on synthetic S3DIS-Area-5-shaped rooms (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Two legs, both with inputs (13-D room features, weights) resident in HBM before the clock starts:

  steady leg (``value``, instance-steps/s).  A *step* is a macro-step of ``--iters-per-step`` (512) lock-step iterations of
      the batched grow loop over the 68-room Area-5-shaped set (BASELINE.json configs[1]) with all rooms in flight: in every
      iteration each in-flight room takes one region-grow step (box query, median, sampling, one LrgNet evaluation on its
      512+512-point sets, mask update).  A room that finishes gets its 1-NN fill-in and restarts at once, so the batch
      stays full.  W warm-up steps roll the rooms into their steady phase, then EXACTLY K steps are timed (20 steps =
      10 240 iterations, about a second).  For N > 1 every rank runs its own 68-room set (weak scaling; no collective
      inside the loop) and the device step counters are summed with one all-reduce after the clock stops.
  fixed-work leg (``rooms_per_sec``).  R = 544 room jobs (the 68 geometries x 8 random-stream keys) are sharded over the N
      ranks by point count (longest first), pushed through 68 slots per GPU from reset to final labels (grow + fill-in),
      and the per-room labels are gathered over RCCL -- the only collective of the path.  Same R for every N (strong scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

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
    ap.add_argument('--iters-per-step', type=int, default=512, help='lock-step iterations per (macro-)step')
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
    ap.add_argument('--net-mode', default='fused', choices=['fused', 'streamed'])
    ap.add_argument('--packed', type=int, default=1, help='1: lrg_grow_step_packed (front kernel + packed rows); 0: the nine-launch lrg_grow_step')
    ap.add_argument('--graph', type=int, default=4, help='packed iterations replayed per host call from a HIP graph (0 = plain launches)')
    ap.add_argument('--fill', type=int, default=1, help='1: finished rooms get their 1-NN fill-in before they are recycled')
    ap.add_argument('--lanes', type=int, default=0, help='groups of slots on their own HIP streams; 0 = auto')
    ap.add_argument('--cu-partition', type=int, default=0, help='1: lanes on disjoint sets of compute units')
    ap.add_argument('--fixed-rooms', type=int, default=-1,
                    help='room jobs of the fixed-work leg over ALL ranks (0 = skip; default: 8 jobs per geometry = 544 for the Area-5 set)')
    ap.add_argument('--cpu-seconds', type=float, default=25.0, help='budget of the CPU-baseline sample (0 = skip)')
    ap.add_argument('--cache', default=os.environ.get('LRG_CACHE', '/tmp/lrg_cache'))
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


def cpu_baseline(rooms, weights, seconds, policy, gpu_room_steps):
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


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from learn_region_grow_amd import _lib, synthetic, workloads, dist as lrg_dist
    from learn_region_grow_amd.lrgnet import LrgNetHIP, _ptr
    from learn_region_grow_amd.grow import LanedRegionGrower, RegionGrower, auto_lanes, lane_streams

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # LRG_BENCH_ONE_DEVICE=1 (testing on a 1-GPU box): every rank uses cuda:0 and the collectives go over gloo
    one_dev = os.environ.get('LRG_BENCH_ONE_DEVICE') == '1'
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

    weights = synthetic.load_trained_weights() if args.weights == 'trained' else synthetic.make_synthetic_weights(seed=0)
    resolution = 0.1
    if args.workload == 'kitti':
        resolution = 0.3
        rooms = workloads.kitti_scenes(min(args.rooms, 8), seed_base=5000 + 100 * rank, cache_dir=args.cache)
        base = workloads.kitti_scenes(min(args.rooms, 8), seed_base=5000, cache_dir=args.cache) if rank else rooms
    elif args.workload == 'scannet':
        rooms = workloads.scannet_rooms(min(args.rooms, 39), seed_base=7000 + 100 * rank, cache_dir=args.cache)
        base = workloads.scannet_rooms(min(args.rooms, 39), seed_base=7000, cache_dir=args.cache) if rank else rooms
    else:
        rooms = workloads.area5_rooms(args.rooms, seed_base=1000 + 100 * rank, cache_dir=args.cache)
        base = workloads.area5_rooms(args.rooms, seed_base=1000, cache_dir=args.cache) if rank else rooms
    net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev, mode=args.net_mode).load_weights(weights)
    packed = bool(args.packed) and args.net_mode == 'fused' and max(len(r['points']) for r in rooms) <= (
        _lib.LRG_PACKED_MAX_POINTS if args.packed > 1 else _lib.LRG_PACKED_AUTO_POINTS)      # --packed 2 forces it up to 131072 points
    graph = args.graph if packed else 0
    if graph and args.iters_per_step % graph:
        raise SystemExit('--iters-per-step must be a multiple of --graph')
    grow_kw = dict(restarts=args.restarts, rng='counter', policy=args.policy, resolution=resolution, packed=packed,
                   graph_iterations=graph)

    # ------------------------------------------------------------------------------------------------------------------
    # steady leg: the rooms in flight dealt over `lanes` growers, each on its own stream (largest rooms first, round the lanes)
    # ------------------------------------------------------------------------------------------------------------------
    n_lanes = max(1, min(args.lanes, len(rooms))) if args.lanes > 0 else auto_lanes(len(rooms) * args.restarts)
    by_size = sorted(range(len(rooms)), key=lambda i: -len(rooms[i]['points']))
    parts = [[i for i in by_size[k::n_lanes]] for k in range(n_lanes)]
    cu_part = args.cu_partition > 0
    lane_streams_ = lane_streams(dev, n_lanes, cu_part)
    growers = []
    for k in range(n_lanes):
        with torch.cuda.stream(lane_streams_[k]):
            g_ = RegionGrower(net, rooms_in_flight=len(parts[k]), seed=rank, **grow_kw)
            g_.load_rooms([rooms[i] for i in parts[k]])
            for g in range(g_.n_groups):
                g_.bind(g, g)
            growers.append(g_)
    torch.cuda.synchronize()
    room_steps = {}          # room index -> instance-steps of its last completed pass (from the device's region log)

    def iterate(iters):
        per_call = graph if graph else 1
        for _ in range(0, iters, per_call):
            for lane, g_ in enumerate(growers):
                with torch.cuda.stream(lane_streams_[lane]):
                    g_.enqueue()
                    for g in g_.poll_done():          # finished rooms get their fill-in (:308-316) and restart at once
                        r = g_.group_room[g]
                        if args.fill:
                            g_.fill(r)
                        g_.reset_room(r)
                        g_.bind(g, r)

    def read_stats():
        return sum(g_.d_stats[:3].cpu().numpy().astype(np.float64) for g_ in growers)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    iterate(args.warmup * args.iters_per_step)
    barrier()
    s0 = read_stats()
    t0 = time.perf_counter()
    iterate(args.steps * args.iters_per_step)
    barrier()
    t1 = time.perf_counter()
    s1 = read_stats()
    elapsed = lrg_dist.allreduce_max(t1 - t0, device=coll_dev)
    inst_steps, rooms_cycled, seeds = lrg_dist.allreduce_sum([float(s1[2] - s0[2]), float(s1[1] - s0[1]), float(s1[0] - s0[0])],
                                                             device=coll_dev)
    iterations = args.steps * args.iters_per_step

    # ------------------------------------------------------------------------------------------------------------------
    # roofline of the LrgNet evaluation (the dominant kernels), HIP events on the launch stream
    # ------------------------------------------------------------------------------------------------------------------
    S = sum(g_.S for g_ in growers)
    reps = 20
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rs = np.random.RandomState(0)
    d_inl = torch.from_numpy((rs.randn(S, 512, 13) * 0.5).astype(np.float32)).to(dev)
    d_nbr = torch.from_numpy((rs.randn(S, 512, 13) * 0.5).astype(np.float32)).to(dev)
    # (a) dense: all 512 + 512 rows of every instance in flight (the formulation SURVEY.md 8d prices)
    net.forward(d_inl, d_nbr)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        net.forward(d_inl, d_nbr)
    ev1.record()
    torch.cuda.synchronize()
    fwd_ms = ev0.elapsed_time(ev1) / reps
    flops = S * FLOPS_PER_INSTANCE_STEP
    tflops = flops / (fwd_ms * 1e-3) / 1e12
    hbm_accounting = S * BYTES_PER_INSTANCE_STEP / (fwd_ms * 1e-3) / 1e9
    # (b) as the loop issues it: the packed distinct rows of the lanes' last iteration
    in_loop = None
    if packed:
        rows = np.zeros(2)
        act = 0
        loop_ms = 0.0
        for g_ in growers:
            sr = g_.p_slot_rows.cpu().numpy()
            rows += sr[:, :2].sum(axis=0)
            act += int((sr[:, 0] > 0).sum())
            # rows as the loop's last evaluation saw them: the hand-over copy of the allocation counters (a slot's rows are
            # allocated in multiples of 8, the padding being copies of its last row)
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
        loop_flops = rows[0] * (FLOPS_PER_BRANCH_ROW + FLOPS_PER_HEAD_ROW) + rows[1] * (FLOPS_PER_BRANCH_ROW + FLOPS_PER_HEAD_ROW) + \
            act * FLOPS_POOLED_GEMM
        in_loop = {'rows_evaluated_fraction': float(rows.sum()) / (S * 1024.0), 'packed_rows': [int(rows[0]), int(rows[1])],
                   'ms_per_evaluation_all_lanes': loop_ms, 'tflops': loop_flops / (loop_ms * 1e-3) / 1e12,
                   'frac_of_fp32_matrix_peak': loop_flops / (loop_ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
                   'useful_tflops_over_the_steady_leg': inst_steps / world / elapsed * loop_flops / max(act, 1) / 1e12}
    traffic = None
    tpath = os.path.join(REPO, 'profiles', 'r02_traffic_%s.json' % args.net_mode)
    if not os.path.exists(tpath):
        tpath = os.path.join(REPO, 'profiles', 'r01_traffic_%s.json' % args.net_mode)
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))['hbm_bytes_per_forward'] * S / 68.0
    sq = None
    spath = os.path.join(REPO, 'profiles', 'r02_pmc_sq_loop.json')
    if os.path.exists(spath):
        sq = json.load(open(spath))
        if in_loop is not None:      # chip-wide matrix-pipe utilisation over the loop's kernels, from the committed SQ counter pass
            in_loop['mfma_util_chipwide'] = sq.get('mfma_util_chipwide')
            in_loop['mfma_util_chipwide_source'] = 'profiles/r02_pmc_sq_loop.json (tools/pmc_sq_loop.sh; round 1 by the same definition: %.3f)' % \
                sq.get('round_1_same_definition', {}).get('mfma_util_chipwide', float('nan'))

    lpath = os.path.join(REPO, 'profiles', 'r02_traffic_loop.json')
    if os.path.exists(lpath) and in_loop is not None:      # HBM bytes of the loop's five launches, from the committed PMC passes
        lt = json.load(open(lpath))
        in_loop['hbm_bytes_per_iteration'] = lt.get('hbm_bytes_per_iteration')
        in_loop['hbm_bytes_source'] = 'profiles/r02_traffic_loop.json (tools/pmc_traffic_loop.sh: FETCH_SIZE / WRITE_SIZE passes, 68 rooms in flight)'

    # ------------------------------------------------------------------------------------------------------------------
    # fixed-work leg: R room jobs over all ranks, reset -> final labels (grow + fill-in) -> RCCL gather
    # ------------------------------------------------------------------------------------------------------------------
    fixed = None
    if args.fixed_rooms < 0:
        args.fixed_rooms = 8 * len(base)
    if args.fixed_rooms > 0 and args.restarts == 1:
        for g_ in growers:
            g_._release_graph()
        del growers
        torch.cuda.empty_cache()
        R = args.fixed_rooms
        jobs = [(j % len(base), j) for j in range(R)]                   # (geometry, random-stream key): 8 keys per geometry at R = 544
        sizes = [len(base[b]['points']) for b, _ in jobs]
        mine = lrg_dist.shard_rooms_lpt(sizes, world)[rank]
        job_rooms = [dict(base[jobs[j][0]], room_id=100000 + jobs[j][1]) for j in mine]
        slots = min(args.rooms, max(1, len(job_rooms)))
        lg = LanedRegionGrower(net, rooms_in_flight=slots, lanes=args.lanes if args.lanes > 0 else None, cu_partition=cu_part, seed=0, **grow_kw)
        lg.load_rooms(job_rooms)
        barrier()
        tf0 = time.perf_counter()
        lg.grow_loaded(fill=bool(args.fill))
        tf_grow = time.perf_counter() - tf0
        # the final gather: every rank ends up with every room's labels (one table + one flat int32 buffer, all_gather)
        # (rooms start at multiples of 16 points in a lane's arena: take each room's own span)
        flat = torch.cat([gr.d_filled[int(gr.room_off[k]):int(gr.room_off[k]) + gr.room_n[k]] for gr in lg.growers for k in range(gr.n_rooms)]) \
            if job_rooms else torch.zeros(0, dtype=torch.int32, device=dev)
        ids = [mine[i] for gr in lg.growers if gr.n_rooms for i in gr.room_index]
        lens = [n for gr in lg.growers if gr.n_rooms for n in gr.room_n]
        tg0 = time.perf_counter()
        gathered = lrg_dist.gather_flat_labels(ids, lens, flat, R, device=coll_dev)
        barrier()
        tf1 = time.perf_counter()
        fixed_elapsed = lrg_dist.allreduce_max(tf1 - tf0, device=coll_dev)
        st = sum(gr.d_stats[:3].cpu().numpy().astype(np.float64) for gr in lg.growers if gr.n_rooms)
        f_steps, f_rooms = lrg_dist.allreduce_sum([float(st[2]), float(len(job_rooms))], device=coll_dev)
        ok = all(gathered[j] is not None and len(gathered[j]) == sizes[j] and int(gathered[j].min()) > 0 for j in range(R))
        for res_lane in [gr for gr in lg.growers if gr.n_rooms]:
            rl = res_lane.d_rlog.cpu().numpy()
            rr = res_lane._read_rooms()
            for k in range(res_lane.n_rooms):
                o = int(res_lane.room_off[k])
                room_steps.setdefault(jobs[mine[res_lane.room_index[k]]][0], int(rl[o:o + rr[k].n_regions, 1].sum()))
        fixed = {'rooms': int(f_rooms), 'seconds': fixed_elapsed, 'rooms_per_sec': f_rooms / fixed_elapsed,
                 'instance_steps': f_steps, 'instance_steps_per_sec': f_steps / fixed_elapsed, 'scaling': 'strong',
                 'slots_per_gpu': slots, 'lanes': len(lg.growers), 'grow_seconds_rank0': tf_grow, 'gather_seconds_rank0': tf1 - tg0,
                 'rccl_ranks': world, 'collective_backend': backend, 'all_rooms_labeled_after_gather': bool(ok),
                 'what': '%d room jobs = the %d geometries x %d random-stream keys, LPT-sharded by point count over %d rank(s), '
                         'reset -> grow -> 1-NN fill-in -> all_gather of the labels' % (R, len(base), (R + len(base) - 1) // len(base), world)}

    if rank == 0:
        out = {
            'metric': 'region-grow steps/sec (rooms/sec alongside), %s shape' % {'area5': 'S3DIS Area-5', 'scannet': 'ScanNet',
                                                                                 'kitti': 'KITTI'}[args.workload],
            'value': inst_steps / elapsed,
            'unit': 'instance-steps/s',
            'rooms_per_sec': fixed['rooms_per_sec'] if fixed else rooms_cycled / elapsed,
            'rooms_per_sec_steady_cycling': rooms_cycled / elapsed,
            'regions_per_sec': seeds / elapsed,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'ms_per_iteration': 1e3 * elapsed / iterations,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('Semantic-KITTI-shaped synthetic scenes (~100 k points at 0.3 m), %d in flight' % len(rooms)
                                    if args.workload == 'kitti' else
                                    'ScanNet-shaped synthetic rooms (%d-room set per GPU, all in flight, cycled)' % len(rooms)
                                    if args.workload == 'scannet' else
                                    'S3DIS Area-5-shaped synthetic rooms (68-room set per GPU, all in flight, cycled)') +
                                   (', greedy test_region_grow.py loop' if args.restarts == 1 else
                                    ', test_random_restart.py loop with %d restarts per seed batched per launch' % args.restarts),
                       'step': '%d lock-step iterations (every in-flight room takes one region-grow step per iteration)' % args.iters_per_step,
                       'iterations_per_step': args.iters_per_step, 'timed_iterations': iterations,
                       'rooms_in_flight_per_gpu': len(rooms), 'slots_per_gpu': S, 'lanes': n_lanes, 'policy': args.policy,
                       'restarts': args.restarts, 'points': '512 inlier + 512 neighbour x 13 features',
                       'rng': 'counter (Philox) stream',
                       'weights': ('trained on synthetic Area-5-shaped rooms by train_region_grow.py (learn_region_grow_amd/weights)'
                                   if args.weights == 'trained' else 'random, seed 0'), 'net_mode': args.net_mode,
                       'iteration': 'lrg_grow_step_packed (front kernel + branch / GEMM / head on packed rows)' if packed else 'lrg_grow_step',
                       'hip_graph_iterations': graph,
                       'active_fraction': inst_steps / (iterations * S * world)},
            'roofline': {'bound': 'mfma', 'kernel': 'lrg_forward: fused branch stacks + pooled GEMM + head stacks of one dense LrgNet evaluation '
                                                   'batch (v_mfma_f32_32x32x2_f32, exact fp32)',
                         'achieved': tflops, 'peak': FP32_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tflops / FP32_MATRIX_PEAK_TFLOPS,
                         'traffic': traffic,
                         'traffic_source': 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of their own, S=68; tools/pmc_run.sh)' % os.path.basename(tpath),
                         'algorithmic_flops': flops, 'flops_per_instance': FLOPS_PER_INSTANCE_STEP,
                         'ms_per_launch': fwd_ms, 'instances_per_launch': S,
                         'hbm_accounting': {'note': 'SURVEY.md 8d layer-streamed accounting (what an unfused implementation would stream); the '
                                                    'fused kernels keep activations in LDS, so this is NOT traffic',
                                            'algorithmic_bytes': S * BYTES_PER_INSTANCE_STEP, 'bytes_per_instance': BYTES_PER_INSTANCE_STEP,
                                            'GBps': hbm_accounting, 'frac_of_hbm_peak': hbm_accounting / HBM_PEAK_GBS},
                         'traffic_ratio': (traffic / (S * BYTES_PER_INSTANCE_STEP)) if traffic else None,
                         'note': 'dense launch: all 512+512 rows of every instance evaluated',
                         'in_loop': in_loop, 'in_loop_sq_counters': sq},
        }
        if fixed:
            out['fixed_work'] = fixed
        if world == 1 and args.cpu_seconds > 0:
            out['cpu_baseline'] = cpu_baseline(base, weights, args.cpu_seconds, args.policy, room_steps)
        if world == 1 and args.p0_rooms > 0:
            out['preprocessing_p0'] = p0_rates(args.p0_rooms, dev)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
