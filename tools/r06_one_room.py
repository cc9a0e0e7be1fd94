"""Round 6: ONE room on the chip (bench.py's one_room_per_gpu leg) by launch form, stream and budget: python tools/r06_one_room.py"""
import sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learn_region_grow_amd import synthetic, workloads
from learn_region_grow_amd.lrgnet import LrgNetHIP
from learn_region_grow_amd.grow import RegionGrower
dev = torch.device('cuda:0')
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev, mode='fused').load_weights(synthetic.load_trained_weights())
rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir='/tmp/lrg_cache')
which = int(sys.argv[1]) if len(sys.argv) > 1 else 34
room = dict(sorted(rooms, key=lambda r: len(r['points']))[which], room_id=424242 + which)
print('room of', len(room['points']), 'points')
for waves, k, budget, own_stream in ((1, 0, 25000, True), (-1, 0, 25000, True), (1, 3, 25000, True), (-1, 3, 25000, True), (1, 2, 25000, True), (1, 3, 5000, True)):
    try:
        st = torch.cuda.Stream(device=dev) if own_stream else torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            gr = RegionGrower(net, rooms_in_flight=1, seed=0, rng='counter', policy='net', packed=True, free_run=True, free_run_budget_us=budget, speculate=k, free_run_waves=waves)
            gr.load_rooms([room])
            torch.cuda.synchronize()
            best = None
            for rep in range(2):
                gr.reset_room(0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                gr.grow_loaded(fill=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            steps = int(gr.d_stats[2].item())
        print('waves', waves, 'K', k, 'budget', budget, 'own stream', own_stream, 'seconds per room %.4f' % best, flush=True)
    except Exception as e:
        print('waves', waves, 'K', k, 'budget', budget, 'own stream', own_stream, 'FAILED', repr(e)[:200], flush=True)
