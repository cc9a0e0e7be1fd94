#!/usr/bin/env python3
"""Distribution of the distinct-row counts the grow loop hands to lrg_forward_rows, and the per-kernel time of one such call."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import synthetic, workloads
from learn_region_grow_amd.lrgnet import LrgNetHIP
from learn_region_grow_amd.grow import RegionGrower
dev = torch.device('cuda:0')
rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir='/tmp/lrg_cache')
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(synthetic.make_synthetic_weights(seed=0))
gr = RegionGrower(net, rooms_in_flight=68, rng='counter', seed=0, policy='gt')
gr.load_rooms(rooms)
for g in range(gr.n_groups):
    gr.bind(g, g)
tiles = []
for it in range(400):
    gr.enqueue_iteration()
    for g in gr.poll_done():
        r = gr.group_room[g]; gr.reset_room(r); gr.bind(g, r)
    if it >= 100 and it % 10 == 0:
        torch.cuda.synchronize()
        ri, rn = gr.b_rows_in.cpu().numpy(), gr.b_rows_nb.cpu().numpy()
        tiles.append((np.ceil(ri / 32).sum(), np.ceil(rn / 32).sum(), ri.sum(), rn.sum(), (ri == 512).sum(), (rn == 512).sum(), (ri == 0).sum()))
t = np.array(tiles)
print('per iteration (mean over %d samples): live 32-row tiles inlier %.0f neighbour %.0f (of 1088 each); rows %.0f / %.0f; slots with 512 rows: %.1f / %.1f; idle slots %.1f'
      % (len(t), t[:, 0].mean(), t[:, 1].mean(), t[:, 2].mean(), t[:, 3].mean(), t[:, 4].mean(), t[:, 5].mean(), t[:, 6].mean()))
print('last sample rows_in:', np.sort(ri).tolist())
print('last sample rows_nb:', np.sort(rn).tolist())
