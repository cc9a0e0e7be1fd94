#!/usr/bin/env python3
"""Does a kernel on a second stream run beside a free-running launch that was given fewer CUs than the device has?
Prints when (ms after the launch was enqueued) a small elementwise kernel and a fill-in on the side stream complete, against the launch itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learn_region_grow_amd import synthetic, workloads
from learn_region_grow_amd.lrgnet import LrgNetHIP
from learn_region_grow_amd.grow import RegionGrower

dev = torch.device('cuda:0')
rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir='/tmp/lrg_cache')
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev, mode='fused').load_weights(synthetic.load_trained_weights())
main = torch.cuda.Stream(device=dev)
for fill_cus in [int(x) for x in os.environ.get("FILL_CUS_LIST", "12,64").split(",")]:
    with torch.cuda.stream(main):
        gr = RegionGrower(net, rooms_in_flight=68, seed=0, free_run=True, free_run_budget_us=20000, free_run_fill_cus=fill_cus)
        gr.load_rooms(rooms)
        for g in range(68):
            gr.bind(g, g)
        gr.enqueue_free_run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    x = torch.zeros(1 << 20, device=dev)
    for what in ('elementwise', 'fill'):
        with torch.cuda.stream(main):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(main)
            gr.enqueue_free_run()
            e1.record(main)
        time.sleep(0.002)
        with torch.cuda.stream(side):
            s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
            s0.record(side)
            if what == 'elementwise':
                x.add_(1.0)
            else:
                gr._in_fill_stream = True
                gr.fill(3)
                gr._in_fill_stream = False
            s1.record(side)
        t0 = time.perf_counter()
        s1.synchronize(); t_side = time.perf_counter() - t0
        e1.synchronize(); t_main = time.perf_counter() - t0
        print('fill_cus=%d %s: side stream done %.2f ms after it was enqueued (its kernels took %.3f ms), launch done after %.2f ms (took %.2f ms)'
              % (fill_cus, what, t_side * 1e3, s0.elapsed_time(s1), t_main * 1e3, e0.elapsed_time(e1)))
