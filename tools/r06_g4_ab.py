"""Round 6: G4 (group_point_grad, tf_grouping_g.cu:61-78) on the harness shape with rotating buffers (860 MB: nothing of a launch's input is left in the Infinity Cache),
as tools/grouping_bench.py times it; the kernel variant by the environment (LRG_GPG_OWN=0: channel slices; LRG_GPG_OWN_NT: points per workgroup of the owned-rows kernel)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learn_region_grow_amd import _lib, grouping          # noqa: E402
from learn_region_grow_amd.lrgnet import _ptr, _stream_ptr   # noqa: E402

HBM = 8e12
dev = torch.device('cuda:0')
rs = np.random.RandomState(0)
b, n, m, ns, c, radius = 32, 512, 128, 64, 64, 0.1
xyz1 = torch.from_numpy(rs.rand(b, n, 3).astype(np.float32)).to(dev)
xyz2 = torch.from_numpy(rs.rand(b, m, 3).astype(np.float32)).to(dev)
idx, _ = grouping.query_ball_point(radius, ns, xyz1, xyz2)
nrot = 12
rot_go = [torch.randn((b, m, ns, c), device=dev) for _ in range(nrot)]
rot_idx = [idx.clone() for _ in range(nrot)]
rot_gp = [torch.zeros((b, n, c), device=dev) for _ in range(nrot)]
lib = _lib.load()
# the result against a float64 scatter-add
lib.lrg_group_point_grad(b, n, c, m, ns, _ptr(rot_go[0]), _ptr(rot_idx[0]), _ptr(rot_gp[0]), _stream_ptr())
want = torch.zeros((b, n, c), dtype=torch.float64, device=dev)
want.scatter_add_(1, idx.long().reshape(b, m * ns, 1).expand(-1, -1, c), rot_go[0].double().reshape(b, m * ns, c))
err = float((rot_gp[0].double() - want).abs().max())
state = {'i': 0}


def fn():
    i = state['i'] = (state['i'] + 1) % nrot
    _lib.check(lib.lrg_group_point_grad(b, n, c, m, ns, _ptr(rot_go[i]), _ptr(rot_idx[i]), _ptr(rot_gp[i]), _stream_ptr()), 'g4')


for _ in range(24):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(240):
    fn()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3 / 240
nbytes = b * m * ns * c * 4 + b * m * ns * 4 + b * n * c * 4
print(json.dumps({'env': {k: v for k, v in os.environ.items() if k.startswith('LRG_GPG')}, 'gpu_us': t * 1e6, 'GBps': nbytes / t / 1e9, 'frac_of_hbm_peak': nbytes / t / HBM,
                  'max_abs_err_vs_float64': err}))
