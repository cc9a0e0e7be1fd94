#!/usr/bin/env python3
"""Run the benchmark configuration (68 Area-5-shaped rooms in flight, two lanes, HIP-graph replays) several times in one process and
compare the outcomes: rooms are independent and the random stream is keyed by (seed, room), so every repetition must give the
same regions and labels.  usage: determinism_check.py [reps] [policy] [lanes] [graph_iterations] [--restarts R] [--in-flight B] [--workload W] [--hog 1]"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import synthetic, workloads  # noqa: E402
from learn_region_grow_amd.grow import LanedRegionGrower  # noqa: E402
from learn_region_grow_amd.lrgnet import LrgNetHIP  # noqa: E402

import argparse
ap = argparse.ArgumentParser()
ap.add_argument('reps', type=int, nargs='?', default=8)
ap.add_argument('policy', nargs='?', default='gt')
ap.add_argument('lanes', type=int, nargs='?', default=2)
ap.add_argument('graph', type=int, nargs='?', default=4)
ap.add_argument('--restarts', type=int, default=1, help='random restarts per seed, batched (test_random_restart.py)')
ap.add_argument('--in-flight', type=int, default=68, help='rooms in flight (fewer than 68: slots are re-bound to the rooms that wait)')
ap.add_argument('--workload', default='area5', choices=['area5', 'scannet', 'kitti'])
ap.add_argument('--hog', type=int, default=0, help='1: a third stream keeps the chip busy with dense LrgNet evaluations meanwhile')
args = ap.parse_args()
reps, policy, lanes, graph = args.reps, args.policy, args.lanes, args.graph
dev = torch.device('cuda:0')
w = synthetic.make_synthetic_weights(seed=0, gain=2.0, bias_std=0.2, add_bias_shift=0.0, rmv_bias_shift=-3.0) if policy == 'gt' \
    else synthetic.load_trained_weights()
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(w)
kw = {}
if args.workload == 'kitti':
    rooms = workloads.kitti_scenes(8, seed_base=5000, cache_dir='/tmp/lrg_cache')
    kw['resolution'] = 0.3
elif args.workload == 'scannet':
    rooms = workloads.scannet_rooms(39, seed_base=7000, cache_dir='/tmp/lrg_cache')
else:
    rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir='/tmp/lrg_cache')
if args.restarts > 1:
    kw['restarts'] = args.restarts
in_flight = min(args.in_flight, len(rooms))
stop = None
if args.hog:
    import threading
    rs = np.random.RandomState(0)
    xi = torch.from_numpy((rs.randn(68, 512, 13) * 0.5).astype(np.float32)).to(dev)
    xn = torch.from_numpy((rs.randn(68, 512, 13) * 0.5).astype(np.float32)).to(dev)
    hog_net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(w)
    hog_stream = torch.cuda.Stream(device=dev)
    stop = threading.Event()

    def hog():
        with torch.cuda.stream(hog_stream):
            while not stop.is_set():
                for _ in range(8):
                    hog_net.forward(xi, xn)
                hog_stream.synchronize()
    th = threading.Thread(target=hog)
    th.start()
outcomes = [dict() for _ in rooms]
try:
    for rep in range(reps):
        lg = LanedRegionGrower(net, rooms_in_flight=in_flight, lanes=lanes, rng='counter', seed=0, policy=policy, graph_iterations=graph, **kw)
        got = lg.run(rooms)
        for i, res in enumerate(got):
            h = hashlib.sha1(res.filled_label.tobytes() + res.cluster_label.tobytes() +
                             repr([(r['seed'], r['steps'], r['points'], r['reason'], r['labeled']) for r in res.regions]).encode()).hexdigest()[:12]
            outcomes[i].setdefault(h, []).append(rep)
        del lg
finally:
    if stop is not None:
        stop.set()
        th.join()
bad = [(i, len(rooms[i]['points']), {h: v for h, v in o.items()}) for i, o in enumerate(outcomes) if len(o) > 1]
print('%s, policy %s, %d in flight, %d lanes, graph %d, restarts %d%s, %d repetitions: %d of %d rooms with more than one outcome' % (
    args.workload, policy, in_flight, lanes, graph, args.restarts, ', busy chip' if args.hog else '', reps, len(bad), len(rooms)))
for i, n, o in bad[:12]:
    print('  room %d (%d points): %s' % (i, n, o))
