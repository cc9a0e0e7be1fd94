#!/usr/bin/env python3
"""Phase cycle stamps of lrg_prepare_kernel during a short bench-like run (library built with -DLRG_TRACE=1)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import synthetic, workloads, _lib
from learn_region_grow_amd.lrgnet import LrgNetHIP
from learn_region_grow_amd.grow import RegionGrower
dev = torch.device('cuda:0')
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(synthetic.make_synthetic_weights(seed=0))
rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir='/tmp/lrg_cache')
gr = RegionGrower(net, rooms_in_flight=68, rng='counter', policy='gt')
gr.load_rooms(rooms)
for g in range(68): gr.bind(g, g)
lib = _lib.load()
tr = torch.zeros(68 * 16, dtype=torch.int64, device=dev)
lib.lrg_set_trace2.argtypes = [ctypes.c_void_p]
for it in range(120):
    gr.enqueue_iteration()
torch.cuda.synchronize()
lib.lrg_set_trace2(ctypes.c_void_p(tr.data_ptr()))
acc = []
for it in range(10):
    gr.enqueue_iteration(); torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(68, 16).copy()
    d = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 8]], axis=1)
    act = t[:, 3] > t[:, 0]
    acc.append(d[act])
a = np.concatenate(acc)
print('active samples', len(a))
print('cycles: slot/room load -> median -> gather   (median, p90, max) per phase')
for i, nm in enumerate(['load', 'median', 'sample+gather']):
    print('  %-14s %8d %8d %8d' % (nm, np.median(a[:, i]), np.percentile(a[:, i], 90), a[:, i].max()))
big = a[a[:, 3] > 1024]
print('nc: median %d max %d; slots with nc>1024: %d' % (np.median(a[:, 3]), a[:, 3].max(), len(big)))
for lo, hi in ((0, 256), (256, 1024), (1024, 4096), (4096, 10**9)):
    m = (a[:, 3] > lo) & (a[:, 3] <= hi)
    if m.any():
        print('  nc in (%d,%d]: n=%d  median-phase cycles p50 %d max %d | gather p50 %d | load p50 %d' % (lo, hi, m.sum(), np.median(a[m, 1]), a[m, 1].max(), np.median(a[m, 2]), np.median(a[m, 0])))
