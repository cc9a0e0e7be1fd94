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
for it in range(int(os.environ.get("LRG_TRACE_WARM", "300"))):
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
# which compute unit each workgroup ran on (HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]) and how many shared it
live = t[(t[:, :, 20] > 0) & (t[:, :, 0] > 0)]
hw = live[:, 22]
cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 7) * 64 + ((hw >> 12) & 1) * 32 + ((hw >> 8) & 0xf)
life_all = live[:, 20] - live[:, 0]
wall = (live[:, 24] - live[:, 23]) / 100.0     # us (wall_clock64 ticks at 100 MHz)
print('cycle counter: %.2f GHz (median over workgroups); workgroup lifetime p50 %.1f us max %.1f us; first start to last end %.1f us' % (
    np.median(life_all / wall) / 1e3, np.median(wall), wall.max(), (live[:, 24].max() - live[:, 23].min()) / 100.0))
vals, inv, cnt = np.unique(cu, return_inverse=True, return_counts=True)
print('%d live workgroups on %d distinct CUs (XCCs used: %s)' % (len(live), len(vals), sorted(set(((hw >> 32) & 0xf).tolist()))))
for c in sorted(set(cnt.tolist())):
    sel = cnt[inv] == c
    print('   workgroups on a CU holding %d of them: %4d, lifetime p50 %d p90 %d max %d' % (c, sel.sum(), np.median(life_all[sel]), np.percentile(life_all[sel], 90), life_all[sel].max()))
for y in range(2):
    a = t[y]
    a = a[(a[:, 20] > 0) & (a[:, 0] > 0)]
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
    # passes of the stamped layer (LRG_TRACE_LAYER): after the MFMAs / after the epilogue, relative to the layer's start
    tl = int(os.environ.get('LRG_TRACE_LAYER', '4'))
    base = a[:, 2 + 2 * tl]
    ps = a[:, 12:20]
    ok = (ps > 0).all(axis=0)
    rel = np.median(ps - base[:, None], axis=0).astype(int)
    print('   layer %d pass stamps (cycles after the layer began; mfma done / epilogue done):' % tl, [int(r) if o else None for r, o in zip(rel, ok)])
    nr = a[:, 21]
    for k in sorted(set(nr.tolist()))[:12]:
        sel = nr == k
        print('   tiles with %2d runs: %4d, lifetime p50 %d max %d' % (k, sel.sum(), np.median(life[sel]), life[sel].max()))
