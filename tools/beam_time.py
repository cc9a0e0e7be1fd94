#!/usr/bin/env python3
"""Throughput of the beam-search driver on Area-5-shaped rooms (grow steps = LrgNet evaluations per second)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import synthetic, workloads
from learn_region_grow_amd.lrgnet import LrgNetHIP
from learn_region_grow_amd.beam import BeamSearchGrower
dev = torch.device('cuda:0')
rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir='/tmp/lrg_cache')
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sel = sorted(rooms, key=lambda r: len(r['points']))[:NR]
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(synthetic.make_synthetic_weights(seed=0))
gr = BeamSearchGrower(net, rooms_in_flight=len(sel), beam_width=3, search_width=3, seed=0, policy='gt')
t0 = time.perf_counter()
res = gr.run(sel)
dt = time.perf_counter() - t0
steps = sum(r['steps'] for x in res for r in x.regions)
print('%d rooms (%d points), beam 3 x search 3, policy gt: %.2f s, %d grow steps -> %.0f steps/s, %.2f rooms/s' % (
    len(sel), sum(len(r['points']) for r in sel), dt, steps, steps / dt, len(sel) / dt))
