#!/usr/bin/env python3
"""Per-phase cycle stamps of the fused kernels (library built with -DLRG_TRACE=1)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import synthetic, _lib
from learn_region_grow_amd.lrgnet import LrgNetHIP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 68
dev = torch.device('cuda:0')
net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev, mode='fused').load_weights(synthetic.make_synthetic_weights(seed=0))
lib = _lib.load()
xi = torch.randn(B, 512, 13, device=dev); xn = torch.randn(B, 512, 13, device=dev)
net.forward(xi, xn); torch.cuda.synchronize()
tr = torch.zeros(2 * 2048 * 32, dtype=torch.int64, device=dev)
lib.lrg_set_trace.argtypes = [ctypes.c_void_p]
lib.lrg_set_trace(ctypes.c_void_p(tr.data_ptr()))
net.forward(xi, xn); torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(2, 2048, 32)
nb = min(2048, B * 512 // (64 if os.environ.get('LRG_TRACE_KERNEL', '2') == '2' else 32))
names = ['stage'] + [x for l in range(5) for x in ('L%d setup' % l, 'L%d run' % l)]
for y in range(2):
    a = t[y, :nb]
    a = a[a[:, 20] > 0]
    d = np.diff(a[:, :13], axis=1)
    print('prob', y, 'workgroups', len(a), 'median cycles per phase:')
    print('  ', dict(zip(names, np.median(d, axis=0).astype(int).tolist())))
    print('   total', int(np.median(a[:, 20] - a[:, 0])), ' last-layer-end -> end', int(np.median(a[:, 20] - a[:, 2 + 2 * (5 if False else 0) + 0])) if False else int(np.median(a[:, 20] - a[:, 0])))
    span = a[:, 20].max() - a[:, 0].min()
    print('   span of traced workgroups (cycles):', int(span), ' sum of WG lifetimes / span =', float((a[:, 20] - a[:, 0]).sum() / span))
