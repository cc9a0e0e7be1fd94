#!/usr/bin/env python3
"""Fill-ins beside a free-running launch on CU-masked streams: the launch on a stream confined to all CUs but K, a fill-in on a stream
confined to those K.  Prints how long the fill-in's kernels take beside the launch and whether the launch ends on time.
    FILL_MASKS="8:stride,8:first,16:stride" python tools/r03_masked_streams.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learn_region_grow_amd import _lib, synthetic, workloads
from learn_region_grow_amd.lrgnet import LrgNetHIP
from learn_region_grow_amd.grow import RegionGrower

dev = torch.device('cuda:0')
lib = _lib.load()
ncu = torch.cuda.get_device_properties(dev).multi_processor_count
words = (ncu + 31) // 32


def masked(cus):
    m = (ctypes.c_uint32 * words)()
    for b in cus:
        m[b // 32] |= 1 << (b % 32)
    h = ctypes.c_void_p()
    _lib.check(lib.lrg_stream_create_cu_mask(m, words, ctypes.byref(h)), 'cu mask')
    return torch.cuda.ExternalStream(h.value, device=dev)


rooms = workloads.area5_rooms(68, seed_base=1000, cache_dir='/tmp/lrg_cache')
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev, mode='fused').load_weights(synthetic.load_trained_weights())
for spec in os.environ.get('FILL_MASKS', '8:stride,8:first').split(','):
    k, how = spec.split(':')
    k = int(k)
    fill_cus = list(range(0, ncu, ncu // k))[:k] if how == 'stride' else list(range(k))
    rest = [c for c in range(ncu) if c not in fill_cus]
    main, side = masked(rest), masked(fill_cus)
    os.environ['LRG_FREE_RUN_CUS'] = str(len(rest))
    with torch.cuda.stream(main):
        gr = RegionGrower(net, rooms_in_flight=68, seed=0, free_run=True, free_run_budget_us=20000)
        gr.load_rooms(rooms)
        for g in range(68):
            gr.bind(g, g)
        gr.enqueue_free_run()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        gr.fill(3)
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(main); gr.enqueue_free_run(); e1.record(main)
    time.sleep(0.002)
    with torch.cuda.stream(side):
        s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
        s0.record(side)
        for _ in range(4):
            gr.fill(3)
        s1.record(side)
    t0 = time.perf_counter()
    s1.synchronize(); t_side = time.perf_counter() - t0
    e1.synchronize(); t_main = time.perf_counter() - t0
    st = gr.d_stats.cpu().numpy()
    print('%s (%d CUs for fill-ins): 4 fill-ins done %.2f ms after enqueue (kernels %.3f ms), launch done after %.2f ms (took %.2f ms), given up %d'
          % (spec, k, t_side * 1e3, s0.elapsed_time(s1), t_main * 1e3, e0.elapsed_time(e1), int(st[3])), flush=True)
