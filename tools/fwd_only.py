#!/usr/bin/env python3
"""Run only the LrgNet evaluation (lrg_forward) on random inputs -- a small target for rocprofv3 counter passes.
usage: fwd_only.py [B] [mode] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import synthetic  # noqa: E402
from learn_region_grow_amd.lrgnet import LrgNetHIP  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 68
mode = sys.argv[2] if len(sys.argv) > 2 else 'fused'
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device('cuda:0')
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev, mode=mode).load_weights(synthetic.make_synthetic_weights(seed=0))
rs = np.random.RandomState(0)
xi = torch.from_numpy((rs.randn(B, 512, 13) * 0.5).astype(np.float32)).to(dev)
xn = torch.from_numpy((rs.randn(B, 512, 13) * 0.5).astype(np.float32)).to(dev)
net.forward(xi, xn)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    net.forward(xi, xn)
e1.record()
torch.cuda.synchronize()
print('B=%d mode=%s: %.1f us per forward' % (B, mode, 1e3 * e0.elapsed_time(e1) / reps))
