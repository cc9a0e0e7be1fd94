#!/usr/bin/env python3
"""Experiment: the 68-room set as two half-batches on two HIP streams (loop kernels of one half overlap the network
evaluation of the other) against one batch on one stream."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import synthetic, workloads
from learn_region_grow_amd.lrgnet import LrgNetHIP
from learn_region_grow_amd.grow import RegionGrower

dev = torch.device('cuda:0')
rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir='/tmp/lrg_cache')
weights = synthetic.make_synthetic_weights(seed=0)
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
order = np.argsort([-len(r['points']) for r in rooms])
parts = [[rooms[i] for i in order[k::nsplit]] for k in range(nsplit)]
streams = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
growers = []
for k in range(nsplit):
    with torch.cuda.stream(streams[k]):
        net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(weights)
        gr = RegionGrower(net, rooms_in_flight=len(parts[k]), rng='counter', seed=0, policy='gt')
        gr.load_rooms(parts[k])
        for g in range(gr.n_groups):
            gr.bind(g, g)
        growers.append(gr)
torch.cuda.synchronize()


host = [0.0, 0.0]


def iterate(n):
    for _ in range(n):
        for k, gr in enumerate(growers):
            with torch.cuda.stream(streams[k]):
                h0 = time.perf_counter()
                gr.enqueue_iteration()
                h1 = time.perf_counter()
                done = gr.poll_done()
                host[0] += h1 - h0
                host[1] += time.perf_counter() - h1
                for g in done:
                    r = gr.group_room[g]
                    gr.reset_room(r)
                    gr.bind(g, r)


iterate(100)
torch.cuda.synchronize()
s0 = sum(int(g.d_stats[2].item()) for g in growers)
t0 = time.perf_counter()
iterate(1000)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
s1 = sum(int(g.d_stats[2].item()) for g in growers)
print('%d stream(s): %.0f instance-steps/s, %.3f ms per round of iterations; host: enqueue %.1f us, poll %.1f us per lane-iteration' % (nsplit, (s1 - s0) / dt, dt, host[0] / (1100 * nsplit) * 1e6, host[1] / (1100 * nsplit) * 1e6))
