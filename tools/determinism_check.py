#!/usr/bin/env python3
"""Run the benchmark configuration (68 Area-5-shaped rooms in flight, two lanes, HIP-graph replays) several times in one process and
compare the outcomes: rooms are independent and the random stream is keyed by (seed, room), so every repetition must give the
same regions and labels.  usage: determinism_check.py [reps] [policy] [lanes] [graph_iterations]"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import synthetic, workloads  # noqa: E402
from learn_region_grow_amd.grow import LanedRegionGrower  # noqa: E402
from learn_region_grow_amd.lrgnet import LrgNetHIP  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
policy = sys.argv[2] if len(sys.argv) > 2 else 'gt'
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
graph = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device('cuda:0')
w = synthetic.make_synthetic_weights(seed=0, gain=2.0, bias_std=0.2, add_bias_shift=0.0, rmv_bias_shift=-3.0) if policy == 'gt' \
    else synthetic.load_trained_weights()
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(w)
rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir='/tmp/lrg_cache')
outcomes = [dict() for _ in rooms]
for rep in range(reps):
    lg = LanedRegionGrower(net, rooms_in_flight=68, lanes=lanes, rng='counter', seed=0, policy=policy, graph_iterations=graph)
    got = lg.run(rooms)
    for i, res in enumerate(got):
        h = hashlib.sha1(res.filled_label.tobytes() + res.cluster_label.tobytes() +
                         repr([(r['seed'], r['steps'], r['points'], r['reason'], r['labeled']) for r in res.regions]).encode()).hexdigest()[:12]
        outcomes[i].setdefault(h, []).append(rep)
    del lg
bad = [(i, len(rooms[i]['points']), {h: v for h, v in o.items()}) for i, o in enumerate(outcomes) if len(o) > 1]
print('policy %s, %d lanes, graph %d, %d repetitions: %d of %d rooms with more than one outcome' % (policy, lanes, graph, reps, len(bad), len(rooms)))
for i, n, o in bad[:12]:
    print('  room %d (%d points): %s' % (i, n, o))
