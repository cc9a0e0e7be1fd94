#!/usr/bin/env python3
"""Speculation (RegionGrower(speculate=K)) on the small test rooms outside pytest: stderr of the runtime stays visible, buffer addresses are printed.
usage: spec_debug.py [K] [rooms in flight] [steps per launch] [n rooms]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from learn_region_grow_amd import synthetic
from learn_region_grow_amd.lrgnet import LrgNetHIP
from learn_region_grow_amd.grow import RegionGrower
from test_gpu_grow import WEIGHT_KW, small_room

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 64
nr = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device('cuda:0')
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev).load_weights(synthetic.make_synthetic_weights(**WEIGHT_KW))
rooms = ([small_room(77, 350, room_id=15)] + [small_room(400 + i, 600 + 200 * i, room_id=10 + i) for i in range(3)])[:nr]
kw = dict(rooms_in_flight=F, rng='counter', seed=123, policy='net')
want = RegionGrower(net, free_run=True, **kw).run(rooms)
print('sequential:', [(len(w.regions), w.total_steps) for w in want], flush=True)
gr = RegionGrower(net, speculate=K, free_run_steps=steps, **kw)
gr.load_rooms(rooms)
for name in ('d_slots', 'd_rooms', 'd_cur', 'd_curidx', 'd_visited', 'd_label', 'd_rlog', 'p_xin', 'a_queue', 'a_sync', 'a_work', 'd_stats'):
    t = getattr(gr, name, None)
    if t is not None:
        print('%-10s %x .. %x' % (name, t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()), flush=True)
if os.environ.get('SPEC_EVENTS'):
    gr.a_work = torch.zeros(8 + 5 + 4 * 65536, dtype=torch.int64, device=dev)
    gr.async_buffers.work = gr.a_work.data_ptr()
gr.grow_loaded()
got = gr.collect()
if os.environ.get('SPEC_EVENTS'):
    w = gr.a_work.cpu().numpy()
    n = int(w[8])
    names = {1: 'PICK', 2: 'UPDATE', 3: 'COMMIT', 4: 'DISCARD', 5: 'LG_ADD', 6: 'LG_RMV', 7: 'GATHER'}
    want_seed = int(os.environ['SPEC_EVENTS'])
    for a_, b_ in zip(got[0].regions, want[0].regions):
        if (a_['seed'], a_['steps'], a_['points'], a_['reason']) != (b_['seed'], b_['steps'], b_['points'], b_['reason']):
            want_seed = a_['seed']
            break
    else:
        want_seed = -1 if len(got[0].regions) == len(want[0].regions) else want_seed
    print('events around seed', want_seed)
    ev = w[9:9 + 4 * min(n, 65536)].reshape(-1, 4)
    idx = [i for i in range(len(ev)) if int(ev[i][1]) == want_seed]
    lo, hi = (max(0, idx[0] - 12), min(len(ev), idx[-1] + 6)) if idx else (0, 40)
    for i in range(lo, hi):
        t, sl = int(ev[i][0]) >> 32, int(ev[i][0]) & 0xFFFFFFFF
        v0, v1, v2 = int(ev[i][1]), int(ev[i][2]), int(ev[i][3])
        if t == 1: d = 'seed %d pos %d cands %d first %d' % (v0, v1, v2 >> 16, v2 & 0xFFFF)
        elif t == 2: d = 'seed %d cnt %d nadd %d st %d nc0 %d ne0 %d' % (v0, v1 >> 16, (v1 >> 4) & 0xFFF, v1 & 15, v2 >> 16, v2 & 0xFFFF)
        elif t == 3: d = 'seed %d count %d reason %d voids %x' % (v0, v1 >> 8, v1 & 255, v2)
        elif t == 7: d = 'seed %d src_in %d src_nb %d centre %08x %08x' % (v0, (v1 >> 32) & 0xFFFFFFFF, v1 & 0xFFFFFFFF, (v2 >> 32) & 0xFFFFFFFF, v2 & 0xFFFFFFFF)
        elif t in (5, 6): d = 'seed %d logits %08x %08x take %d row %d srow %d' % (v0, (v1 >> 32) & 0xFFFFFFFF, v1 & 0xFFFFFFFF, v2 >> 40, (v2 >> 20) & 0xFFFFF, v2 & 0xFFFFF)
        else: d = 'seed %d status %d nc %d' % (v0, v1, v2)
        print('  ev %4d slot %d %-7s %s' % (i, sl, names.get(t, t), d))
print('speculative:', [(len(g.regions), g.total_steps) for g in got], 'work', gr.a_work.cpu().numpy()[:8].tolist(), flush=True)
for g, w in zip(got, want):
    print('same labels', np.array_equal(g.filled_label, w.filled_label), 'same cluster labels', np.array_equal(g.cluster_label, w.cluster_label),
          'regions', len(g.regions), len(w.regions), flush=True)
    for i, (a, b) in enumerate(zip(g.regions, w.regions)):
        if (a['seed'], a['steps'], a['points'], a['reason']) != (b['seed'], b['steps'], b['points'], b['reason']):
            print('first difference at region', i)
            for k in range(max(0, i - 2), min(len(g.regions), i + 4)):
                key = lambda r: (r['seed'], r['steps'], r['points'], r['reason'])
                print('   ', k, 'spec', key(g.regions[k]), ' seq', key(w.regions[k]) if k < len(w.regions) else None)
            break

if os.environ.get('SPEC_TRACE'):
    # single-step launches: after every launch the masks are compared with the lists (a mask bit outside the slot's list is a stale bit)
    import ctypes
    from learn_region_grow_amd._lib import LrgSlot
    tsteps = int(os.environ['SPEC_TRACE'])
    gr = RegionGrower(net, speculate=K, free_run_steps=tsteps, **kw)
    gr.load_rooms(rooms)
    gr.free_run_begin()
    names = {0: 'IDLE', 1: 'ACTIVE', 2: 'NONEIGH', 3: 'NOEXP', 4: 'STUCK', 5: 'EMPTY', 6: 'MAXST', 7: 'DONE', 8: 'WAIT', 9: 'PEND'}
    for it in range(400):
        gr.enqueue_free_run(tsteps, 0)
        torch.cuda.synchronize()
        raw = gr.d_slots.cpu().numpy().tobytes()
        sl = (LrgSlot * gr.S).from_buffer_copy(raw)
        cur = gr.d_cur.cpu().numpy()
        ci = gr.d_curidx.cpu().numpy()
        vis = gr.d_visited.cpu().numpy()
        line = []
        bad = False
        for s in range(gr.S):
            S = sl[s]
            members = set(np.nonzero(cur[s])[0].tolist())
            lst = set(ci[s][:max(S.nc, 0)].tolist()) if S.status in (1, 2, 3, 4, 5, 6, 9) else set()
            stale = members - lst
            line.append('%d:%s seed %d pos %s nc %d step %d cnt %d fl %d |mask| %d%s' % (s, names.get(S.status, S.status), S.seed, S.spec_pos if S.spec_pos < 2**31 - 1 else '-', S.nc, S.step, S.count,
                                                                                       S.spec_flags, len(members), (' STALE %s' % sorted(stale)) if stale else ''))
            bad = bad or bool(stale)
        nreg = int(gr._read_rooms()[0].n_regions)
        rl = gr.d_rlog.cpu().numpy().reshape(-1, 8)[:nreg]
        print('launch %3d  regions %d visited %d | %s | last regions %s' % (it, nreg, int(vis.sum()), ' | '.join(line), [tuple(int(v) for v in r[:4]) for r in rl[-3:]]), flush=True)
        if bad:
            break
        if all(sl[s].status in (0, 7) for s in range(gr.S)):
            break
