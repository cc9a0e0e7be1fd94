#!/usr/bin/env python3
"""bench.py -- region-grow throughput on synthetic S3DIS-Area-5-shaped rooms (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one lock-step iteration of the batched grow loop (lrg_grow_step) over all rooms in flight: every
in-flight room takes one region-grow step (one LrgNet evaluation on 512+512 points + neighbour query, median,
sampling, mask update).  The workload is the 68-room Area-5-shaped set (BASELINE.json configs[1]) with all rooms
in flight; a room that finishes is replaced at once (the set is cycled) so the batch stays full.  Inputs
(13-D room features, weights) are resident in HBM before the timed region.
For N > 1 every rank runs its own 68-room set (weak scaling, no collective inside the loop) and the final
label gather goes over RCCL.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

BYTES_PER_INSTANCE_STEP = 10297344      # layer-streamed algorithmic HBM bytes of one LrgNet evaluation
FLOPS_PER_INSTANCE_STEP = 271712256     # hoisted-head FLOPs of one LrgNet evaluation
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
FP32_MATRIX_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: fp32-input MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1500)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--rooms', type=int, default=68, help='rooms in flight per GPU (the Area-5 set has 68)')
    ap.add_argument('--restarts', type=int, default=1)
    ap.add_argument('--workload', default='area5', choices=['area5', 'kitti', 'scannet'],
                    help='area5: 68 Area-5-shaped rooms (configs[1]); kitti: 100 k-point scenes at 0.3 m (configs[4])')
    ap.add_argument('--policy', default='gt', choices=['net', 'gt', 'threshold'],
                    help="mask policy: 'net' = the reference's Bernoulli draws against the network's confidence "
                         "(test_region_grow.py:266-267); 'gt' = its commented-out ground-truth masks (:268-269). "
                         "The network is evaluated every step either way; with synthetic weights only 'gt' gives "
                         "Area-5-like region dynamics (regions per room, steps per region)")
    ap.add_argument('--fuse-pool', type=int, default=0)
    ap.add_argument('--net-mode', default='fused', choices=['fused', 'streamed'])
    ap.add_argument('--advance-rounds', type=int, default=1)
    ap.add_argument('--fill', type=int, default=1, help='1: finished rooms get their 1-NN fill-in before they are recycled (rooms/sec as defined)')
    ap.add_argument('--lanes', type=int, default=0, help='half-batches on their own HIP streams; 0 = auto (2 from 64 rooms in flight, else 1)')
    ap.add_argument('--cpu-seconds', type=float, default=20.0, help='budget of the CPU-baseline sample (0 = skip)')
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


def _cpu_sample(rooms, weights, seconds, policy, faithful, net_fn):
    from oracle import grow_ref, rng_ref      # CPU baseline leg only
    order = np.argsort([len(r['points']) for r in rooms])
    t0 = time.time()
    count = [0]

    class Stop(Exception):
        pass

    def hook(d):
        count[0] += 1
        if time.time() - t0 > seconds:
            raise Stop()
    sizes = []
    try:
        for k in range(len(rooms)):        # median-size room first, then outwards
            room = rooms[int(order[(len(order) // 2 + (k + 1) // 2 * (1 if k % 2 else -1)) % len(order)])]
            sizes.append(len(room['points']))
            grow_ref.grow_room(room['points'], room['obj_id'], room['order'], weights, rng_ref.LegacyStream(k),
                               faithful=faithful, net_fn=net_fn, hook=hook, fill=False, policy=policy)
    except Stop:
        pass
    dt = time.time() - t0
    return count[0], sizes, dt


def cpu_baseline(rooms, weights, seconds, policy):
    """The oracle (faithful NumPy restatement of test_region_grow.py:175-316, per-point Python voxel-set loop,
    un-hoisted head) on this box's host cores, for a bounded sample: steps of the median-size room.  Beside it a
    'strong CPU' figure: the same loop with the set membership vectorised and the head hoisted."""
    n, sizes, dt = _cpu_sample(rooms, weights, seconds * 0.7, policy, True, None)
    try:                                       # threads the NumPy BLAS actually runs the matrix products on
        import threadpoolctl
        blas_threads = max([i['num_threads'] for i in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count()
    out = dict(value=n / dt, unit='instance-steps/s', cores=blas_threads, kind='port',
               sample='%d grow steps over %d Area-5-shaped room(s) of %s points, oracle.grow_ref (faithful=True, policy=%s), '
                      '%.1f s; BLAS threads = cores, Python loops single-threaded' % (n, len(sizes), sizes, policy, dt))
    n2, sizes2, dt2 = _cpu_sample(rooms, weights, seconds * 0.3, policy, False, _hoisted_numpy_net(weights))
    out['strong'] = dict(value=n2 / dt2, unit='instance-steps/s',
                         sample='%d grow steps, vectorised voxel-set membership + hoisted head (NumPy/BLAS), %.1f s' % (n2, dt2))
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
    from learn_region_grow_amd import synthetic, workloads, dist as lrg_dist
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    from learn_region_grow_amd.grow import RegionGrower, auto_lanes

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
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if one_dev:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            coll_dev = torch.device('cpu')
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    weights = synthetic.make_synthetic_weights(seed=0)
    resolution = 0.1
    if args.workload == 'kitti':
        resolution = 0.3
        rooms = workloads.kitti_scenes(min(args.rooms, 8), seed_base=5000 + 100 * rank, cache_dir=args.cache)
    elif args.workload == 'scannet':
        rooms = workloads.scannet_rooms(min(args.rooms, 39), seed_base=7000 + 100 * rank, cache_dir=args.cache)
    else:
        rooms = workloads.area5_rooms(args.rooms, seed_base=1000 + 100 * rank, cache_dir=args.cache)
    net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev, fuse_pool=bool(args.fuse_pool), mode=args.net_mode).load_weights(weights)
    # the rooms in flight dealt over `lanes` growers, each on its own stream (largest rooms first, round the lanes)
    n_lanes = max(1, min(args.lanes, len(rooms))) if args.lanes > 0 else auto_lanes(len(rooms) * args.restarts)
    by_size = sorted(range(len(rooms)), key=lambda i: -len(rooms[i]['points']))
    parts = [[rooms[i] for i in by_size[k::n_lanes]] for k in range(n_lanes)]
    lane_streams = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)] if n_lanes > 1 else [torch.cuda.current_stream(dev)]
    growers = []
    for k in range(n_lanes):
        with torch.cuda.stream(lane_streams[k]):
            g_ = RegionGrower(net, rooms_in_flight=len(parts[k]), restarts=args.restarts, rng='counter', seed=rank,
                              policy=args.policy, advance_rounds=args.advance_rounds, resolution=resolution)
            g_.load_rooms(parts[k])
            for g in range(g_.n_groups):
                g_.bind(g, g)
            growers.append(g_)
    gr = growers[0]
    torch.cuda.synchronize()

    def iterate(k):
        for _ in range(k):
            for lane, g_ in enumerate(growers):
                with torch.cuda.stream(lane_streams[lane]):
                    g_.enqueue_iteration()
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

    iterate(args.warmup)
    barrier()
    s0 = read_stats()
    t0 = time.perf_counter()
    iterate(args.steps)
    barrier()
    t1 = time.perf_counter()
    s1 = read_stats()
    elapsed = lrg_dist.allreduce_max(t1 - t0, device=coll_dev)
    inst_steps, rooms_done, seeds = lrg_dist.allreduce_sum([float(s1[2] - s0[2]), float(s1[1] - s0[1]), float(s1[0] - s0[0])],
                                                           device=coll_dev)

    # ---- roofline of the LrgNet evaluation (the dominant kernels), HIP events on the launch stream ----
    # (one dense batch of all the instances in flight on this GPU, the stacked inputs of the lanes' last iteration)
    S = sum(g_.S for g_ in growers)
    b_inl, b_nbr = torch.cat([g_.b_inl for g_ in growers]), torch.cat([g_.b_nbr for g_ in growers])
    b_add, b_rmv = torch.cat([g_.b_add for g_ in growers]), torch.cat([g_.b_rmv for g_ in growers])
    b_rows_in, b_rows_nb = torch.cat([g_.b_rows_in for g_ in growers]), torch.cat([g_.b_rows_nb for g_ in growers])
    reps = 20
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    net.forward(b_inl, b_nbr, b_add, b_rmv)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        net.forward(b_inl, b_nbr, b_add, b_rmv)
    ev1.record()
    torch.cuda.synchronize()
    fwd_ms = ev0.elapsed_time(ev1) / reps
    achieved = S * BYTES_PER_INSTANCE_STEP / (fwd_ms * 1e-3) / 1e9
    tflops = S * FLOPS_PER_INSTANCE_STEP / (fwd_ms * 1e-3) / 1e12
    # the same evaluation as the grow loop issues it: only the distinct leading rows of each padded set
    rows_frac, fwd_rows_ms = 1.0, fwd_ms
    if gr.skip_duplicate_rows:
        rows_frac = float((b_rows_in.float().mean() + b_rows_nb.float().mean()).item()) / 1024.0
        ev0.record()
        for _ in range(reps):
            net.forward(b_inl, b_nbr, b_add, b_rmv, rows_in=b_rows_in, rows_nb=b_rows_nb)
        ev1.record()
        torch.cuda.synchronize()
        fwd_rows_ms = ev0.elapsed_time(ev1) / reps

    # HBM traffic of one dense evaluation from rocprofv3 FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py,
    # gfx950 corrections applied there); collected offline at S=68 and committed under profiles/
    traffic = None
    tpath = os.path.join(REPO, 'profiles', 'r01_traffic_%s.json' % args.net_mode)
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))['hbm_bytes_per_forward'] * S / 68.0

    # ---- final label gather over RCCL (the only collective of the path) ----
    if world > 1:
        labs = [gr.d_label[int(gr.room_off[r]):int(gr.room_off[r]) + gr.room_n[r]].cpu().numpy() for r in range(2)]
        lrg_dist.gather_room_labels([rank * 2, rank * 2 + 1], labs, 2 * world, device=coll_dev)

    if rank == 0:
        out = {
            'metric': 'region-grow steps/sec (rooms/sec alongside), %s shape' % {'area5': 'S3DIS Area-5', 'scannet': 'ScanNet',
                                                                                 'kitti': 'KITTI'}[args.workload],
            'value': inst_steps / elapsed,
            'unit': 'instance-steps/s',
            'rooms_per_sec': rooms_done / elapsed,
            'regions_per_sec': seeds / elapsed,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('Semantic-KITTI-shaped synthetic scenes (~100 k points at 0.3 m), %d in flight' % len(rooms)
                                    if args.workload == 'kitti' else
                                    'S3DIS Area-5-shaped synthetic rooms (68-room set per GPU, all in flight, cycled)') +
                                   (', greedy test_region_grow.py loop' if args.restarts == 1 else
                                    ', test_random_restart.py loop with %d restarts per seed batched per launch' % args.restarts),
                       'rooms_in_flight_per_gpu': len(rooms), 'slots_per_gpu': S, 'lanes': n_lanes, 'policy': args.policy,
                       'restarts': args.restarts, 'points': '512 inlier + 512 neighbour x 13 features',
                       'rng': 'counter (Philox) stream', 'weights': 'synthetic, seed 0', 'net_mode': args.net_mode,
                       'active_fraction': inst_steps / (args.steps * S * world)},
            'roofline': {'bound': 'hbm', 'kernel': 'lrg_forward (all launches of one LrgNet evaluation batch)',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': 'profiles/r01_traffic_%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, S=68)' % args.net_mode,
                         'algorithmic_bytes': S * BYTES_PER_INSTANCE_STEP, 'ms_per_launch': fwd_ms, 'instances_per_launch': S,
                         'bytes_per_instance': BYTES_PER_INSTANCE_STEP,
                         'fp32_matrix_tflops': tflops, 'fp32_matrix_frac': tflops / FP32_MATRIX_PEAK_TFLOPS,
                         'note': 'dense launch: all 512+512 rows of every instance evaluated',
                         'in_loop': {'rows_evaluated_fraction': rows_frac, 'ms_per_launch': fwd_rows_ms,
                                     'padded_equivalent_GBps': S * BYTES_PER_INSTANCE_STEP / (fwd_rows_ms * 1e-3) / 1e9}},
        }
        if world == 1 and args.cpu_seconds > 0:
            out['cpu_baseline'] = cpu_baseline(rooms, weights, args.cpu_seconds, args.policy)
        if world == 1 and args.p0_rooms > 0:
            out['preprocessing_p0'] = p0_rates(args.p0_rooms, dev)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
