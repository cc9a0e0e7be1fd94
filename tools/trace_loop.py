#!/usr/bin/env python3
"""Cycle stamps of the fused kernels as launched by the grow loop (library built with -DLRG_TRACE=<CAP0>)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import synthetic, workloads, _lib
from learn_region_grow_amd.lrgnet import LrgNetHIP
from learn_region_grow_amd.grow import RegionGrower
dev = torch.device('cuda:0')
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 68
rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir='/tmp/lrg_cache')[:NR]
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(synthetic.load_trained_weights())
gr = RegionGrower(net, rooms_in_flight=NR, rng='counter', seed=0, policy='net')
gr.load_rooms(rooms)
for g in range(gr.n_groups):
    gr.bind(g, g)
lib = _lib.load()
for it in range(300):
    gr.enqueue_iteration()
    for g in gr.poll_done():
        r = gr.group_room[g]; gr.reset_room(r); gr.bind(g, r)
torch.cuda.synchronize()
tr = torch.zeros(2 * 2048 * 32, dtype=torch.int64, device=dev)
lib.lrg_set_trace.argtypes = [ctypes.c_void_p]
lib.lrg_set_trace(ctypes.c_void_p(tr.data_ptr()))
gr.enqueue_iteration()
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(2, 2048, 32)
for y in range(2):
    a = t[y]
    a = a[a[:, 20] > 0]
    if not len(a):
        continue
    t0 = a[:, 0].min()
    life = a[:, 20] - a[:, 0]
    print('prob %d: %d live workgroups; start spread %d cycles (p50 %d, p90 %d); lifetime p10 %d p50 %d p90 %d max %d; last end %d' % (
        y, len(a), a[:, 0].max() - t0, np.median(a[:, 0] - t0), np.percentile(a[:, 0] - t0, 90), np.percentile(life, 10), np.median(life),
        np.percentile(life, 90), life.max(), a[:, 20].max() - t0))
    d = np.diff(a[:, :13], axis=1)
    names = ['stage'] + [x for l in range(5) for x in ('L%d setup' % l, 'L%d run' % l)]
    print('   median phase cycles:', dict(zip(names, np.median(d, axis=0).astype(int).tolist())))
