#!/usr/bin/env python3
"""Phase cycle stamps of lrg_front_kernel in the loop (library built with -DLRG_TRACE=1)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import synthetic, workloads, _lib
from learn_region_grow_amd.lrgnet import LrgNetHIP
from learn_region_grow_amd.grow import RegionGrower
dev = torch.device('cuda:0')
KITTI = os.environ.get('LRG_TRACE_WORKLOAD') == 'kitti'      # 8 KITTI-shaped scenes, ground-truth masks, packed iteration forced
if KITTI:
    net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(synthetic.make_synthetic_weights(seed=0))
    rooms = workloads.kitti_scenes(8, seed_base=5000, cache_dir='/tmp/lrg_cache')
    gr = RegionGrower(net, rooms_in_flight=8, rng='counter', policy='gt', resolution=0.3, packed=True)
else:
    net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(synthetic.load_trained_weights())
    rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir='/tmp/lrg_cache')
    gr = RegionGrower(net, rooms_in_flight=68, rng='counter', policy='net')
gr.load_rooms(rooms)
NS = len(rooms)
for g in range(NS): gr.bind(g, g)
lib = _lib.load()
tr = torch.zeros(NS * 16, dtype=torch.int64, device=dev)
lib.lrg_set_trace2.argtypes = [ctypes.c_void_p]
for it in range(1500 if KITTI else 4000):
    gr.enqueue_iteration()
    for g in gr.poll_done():
        r = gr.group_room[g]; gr.reset_room(r); gr.bind(g, r)
torch.cuda.synchronize()
lib.lrg_set_trace2(ctypes.c_void_p(tr.data_ptr()))
names = ['update', 'advance', 'query', 'sample', 'gather', 'tags', 'end']
rows = []
n_of = np.array([len(r['points']) for r in rooms])
worst = []
for it in range(60):
    tr.zero_()
    gr.enqueue_iteration(); torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(NS, 16)
    for g in range(NS):
        if t[g, 7] > 0:
            d = np.diff(t[g, :8])
            rows.append(np.concatenate([d, [t[g, 7] - t[g, 0], n_of[gr.group_room[g]], t[g, 8], t[g, 14] > 0]]))
    live = t[:, 7] > 0
    if live.any():
        worst.append((t[live, 7].max() - t[live, 0].min(), (t[live, 7] - t[live, 0]).max()))
    for g in gr.poll_done():
        r = gr.group_room[g]; gr.reset_room(r); gr.bind(g, r)
rows = np.array(rows, dtype=np.float64)
print('%d slot-iterations; cycles (p50 / p90 / max)' % len(rows))
for i, nm in enumerate(names + ['total']):
    print('  %-8s %8d %8d %8d' % (nm, np.median(rows[:, i]), np.percentile(rows[:, i], 90), rows[:, i].max()))
# the slowest slot of each launch: which phase makes it slow
slow = []
per_it = 68
k = 0
w = np.array(worst)
print('per launch: span first-start..last-end p50 %d, slowest workgroup p50 %d cycles' % (np.median(w[:, 0]), np.median(w[:, 1])))
for lo, hi in ((0, 1024), (1024, 4096), (4096, 1 << 30)):
    sel = rows[(rows[:, 9] > lo) & (rows[:, 9] <= hi)]
    if len(sel):
        print('  nc in (%d, %d]: %d samples; ' % (lo, hi, len(sel)) + ', '.join('%s %d' % (nm, np.median(sel[:, i])) for i, nm in enumerate(names)) + ', total %d' % np.median(sel[:, 7]))
for lo, hi in ((0, 10000), (10000, 20000), (20000, 1 << 30)):
    sel = rows[(rows[:, 8] > lo) & (rows[:, 8] <= hi)]
    if len(sel):
        print('  room points in (%d, %d]: %d samples; ' % (lo, hi, len(sel)) + ', '.join('%s %d' % (nm, np.median(sel[:, i])) for i, nm in enumerate(names)) + ', total %d' % np.median(sel[:, 7]))
sel = rows[rows[:, 10] > 0]
if len(sel):
    print('  slots that committed + reseeded: %d samples; ' % len(sel) + ', '.join('%s %d' % (nm, np.median(sel[:, i])) for i, nm in enumerate(names)) + ', total %d' % np.median(sel[:, 7]))

# slowest decile of slot-iterations
thr = np.percentile(rows[:, 7], 90)
sel = rows[rows[:, 7] >= thr]
print('  slowest decile (total >= %d): ' % thr + ', '.join('%s %d' % (nm, np.median(sel[:, i])) for i, nm in enumerate(names)) + '; nc p50 %d, room points p50 %d, committed %.0f %%' % (np.median(sel[:, 9]), np.median(sel[:, 8]), 100 * (sel[:, 10] > 0).mean()))
