#!/usr/bin/env python3
"""Unlabeled fraction before the fill-in and time of the fill kernels on Area-5-shaped rooms."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import synthetic, workloads
from learn_region_grow_amd.lrgnet import LrgNetHIP
from learn_region_grow_amd.grow import RegionGrower
dev = torch.device('cuda:0')
rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir='/tmp/lrg_cache')
sel = [rooms[i] for i in (0, 5, 11, 23, 40, 57)]
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(synthetic.load_trained_weights())
gr = RegionGrower(net, rooms_in_flight=len(sel), rng='counter', seed=0, policy='net', free_run_fill_cus=0)
res = gr.run(sel, fill=False)
for r, x in enumerate(res):
    n = len(x.cluster_label); u = int((x.cluster_label == 0).sum())
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr.fill(r); torch.cuda.synchronize()
    e0.record(); gr.fill(r); e1.record(); torch.cuda.synchronize()
    print('room %d: n %6d unlabeled %6d (%.1f%%)  regions %4d  fill %.1f us' % (r, n, u, 100.0 * u / n, len(x.regions), 1e3 * e0.elapsed_time(e1)))
