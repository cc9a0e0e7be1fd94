#!/usr/bin/env python3
"""Time room preprocessing P0: host NumPy vs GPU (jacobi / lapack finish) on Area-5-shaped rooms."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import synthetic, preprocess, preprocess_gpu
for target in (9300, 20000, 45000):
    r = synthetic.area5_shaped_room(target, 1234).astype(np.float32)
    raw = (r[:, :6], r[:, 6].astype(int), r[:, 7].astype(int))
    preprocess_gpu.preprocess_room(*raw)
    torch.cuda.synchronize()
    out = {}
    for name, fn in (('host', lambda: preprocess.preprocess_room(*raw)), ('gpu jacobi', lambda: preprocess_gpu.preprocess_room(*raw)),
                     ('gpu lapack', lambda: preprocess_gpu.preprocess_room(*raw, eig='lapack'))):
        t = time.perf_counter(); n = 0
        while time.perf_counter() - t < 1.0:
            fn(); n += 1
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t) / n
    print('raw %7d -> equalised ~%5d: ' % (len(r), target) + ', '.join('%s %.1f ms' % (k, v * 1e3) for k, v in out.items()))
