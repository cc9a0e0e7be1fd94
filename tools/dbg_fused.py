import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from learn_region_grow_amd import synthetic, _lib
from learn_region_grow_amd.lrgnet import LrgNetHIP
hip = ctypes.CDLL('libamdhip64.so')
hip.hipGetErrorName.restype = ctypes.c_char_p
print('err100 =', hip.hipGetErrorName(100), 'err98', hip.hipGetErrorName(98), hip.hipGetErrorName(1), hip.hipGetErrorName(9))
dev = torch.device('cuda:0')
w = synthetic.make_synthetic_weights(seed=0)
for (ni, nn, B, keep) in [(512, 512, 2, False), (64, 128, 2, False), (64, 128, 2, True), (64, 64, 2, True), (128, 128, 2, False)]:
    try:
        net = LrgNetHIP(1, 1, ni, nn, 13, 0, device=dev, mode='fused', keep_acts=keep).load_weights(w)
        xi = torch.randn(B, ni, 13, device=dev); xn = torch.randn(B, nn, 13, device=dev)
        a, r = net.forward(xi, xn); torch.cuda.synchronize()
        print(ni, nn, B, keep, 'ok', float(a.abs().max()))
    except Exception as e:
        print(ni, nn, B, keep, 'FAIL', e)
        try:
            torch.cuda.synchronize()
        except Exception as e2:
            print('  sync:', e2)
