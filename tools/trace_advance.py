#!/usr/bin/env python3
"""Phase cycle stamps of lrg_advance_kernel in the loop (library built with -DLRG_TRACE=1)."""
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
for it in range(150):
    gr.enqueue_iteration()
    for g in gr.poll_done():
        r = gr.group_room[g]; gr.reset_room(r); gr.bind(g, r)
torch.cuda.synchronize()
lib.lrg_set_trace2(ctypes.c_void_p(tr.data_ptr()))
names = ['scan', 'decide', '(sync)', 'bank', 'commit', 'seed+probe', 'reset']
full, short = [], []
n_of = np.array([len(r['points']) for r in rooms])
for it in range(30):
    tr.zero_()
    gr.enqueue_iteration(); torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(68, 16)[:, 9:16].copy()
    for g in range(68):
        if t[g, 6] > 0:
            full.append(np.concatenate([np.diff(t[g]), [t[g, 6] - t[g, 0], n_of[gr.group_room[g]]]]))
        elif t[g, 1] > 0:
            short.append([t[g, 1] - t[g, 0], 0, n_of[gr.group_room[g]]])
    for g in gr.poll_done():
        r = gr.group_room[g]; gr.reset_room(r); gr.bind(g, r)
full, short = np.array(full), np.array(short)
print('slots that went on growing: %d samples; scan p50 %d p90 %d max %d cycles' % (len(short), np.median(short[:, 0]), np.percentile(short[:, 0], 90), short[:, 0].max()))
print('slots that stopped and reseeded: %d samples' % len(full))
for i, nm in enumerate(names[:6]):
    print('  %-12s p50 %7d  p90 %7d  max %7d' % (nm, np.median(full[:, i]), np.percentile(full[:, i], 90), full[:, i].max()))
print('  %-12s p50 %7d  p90 %7d  max %7d' % ('total', np.median(full[:, 6]), np.percentile(full[:, 6], 90), full[:, 6].max()))
big = full[full[:, 7] > 20000]
if len(big):
    print('  rooms > 20k points (%d): ' % len(big) + ', '.join('%s %d' % (nm, np.median(big[:, i])) for i, nm in enumerate(names[:6])) + ', total %d' % np.median(big[:, 6]))
