#!/usr/bin/env python3
"""In which order does v_mfma_f32_32x32x2_f32 add its products?  lrg_head_pool_gemv with 8 instances runs the matrix-core GEMM
(lrg_head_gemm_kernel: eight K ranges, per k-group of 8 the MFMAs s = 0..3 take k = 8g + s from lane half 0 and k = 8g + 4 + s from
lane half 1), with 4 instances the vector kernel (k after k).  Compared with float32 chains emulated on the host."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learn_region_grow_amd import _lib
from learn_region_grow_amd.lrgnet import _ptr, _stream_ptr
lib = _lib.load()
dev = torch.device('cuda:0')
rs = np.random.RandomState(0)
P, C = 1024, 256
pooled = np.abs(rs.randn(8, P)).astype(np.float32)
W = (rs.randn(P, C) * 0.1).astype(np.float32)
bias = rs.randn(C).astype(np.float32)
d_p, d_w, d_b = (torch.from_numpy(x).to(dev) for x in (pooled, W, bias))
def run(B):
    out = torch.zeros((B, C), dtype=torch.float32, device=dev)
    _lib.check(lib.lrg_head_pool_gemv(_ptr(d_p), _ptr(d_w), C, _ptr(d_b), _ptr(out), B, P, C, _stream_ptr(dev)), 'gemv')
    torch.cuda.synchronize()
    return out.cpu().numpy()
gemm, gemv = run(8), run(4)
def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
def chain(order_of_group):
    tot = None
    parts = []
    for r in range(8):
        acc = np.zeros((4, C), np.float32)
        for g in range(16):
            for k in order_of_group:
                kk = r * 128 + 8 * g + k
                acc = fma(pooled[:4, kk:kk + 1], W[kk:kk + 1, :], acc)
        parts.append(acc)
    s = parts[0]
    for q in parts[1:]:
        s = (s + q).astype(np.float32)
    return (s + bias).astype(np.float32)
def pairsum():
    parts = []
    for r in range(8):
        acc = np.zeros((4, C), np.float32)
        for g in range(16):
            for s_ in range(4):
                k0, k1 = r * 128 + 8 * g + s_, r * 128 + 8 * g + 4 + s_
                pr = (pooled[:4, k0:k0 + 1].astype(np.float64) * W[k0:k0 + 1, :] + pooled[:4, k1:k1 + 1].astype(np.float64) * W[k1:k1 + 1, :] + acc)
                acc = pr.astype(np.float32)
        parts.append(acc)
    s = parts[0]
    for q in parts[1:]:
        s = (s + q).astype(np.float32)
    return (s + bias).astype(np.float32)
seq = chain(range(8))
inter = chain([0, 4, 1, 5, 2, 6, 3, 7])
ps = pairsum()
def frac(a, b):
    return float((a == b).mean())
print('gemv(B=4) == sequential chain     : %.4f' % frac(gemv, seq))
print('gemm(B=8)[:4] == gemv             : %.4f' % frac(gemm[:4], gemv))
print('gemm == chain k, k+4 interleaved  : %.4f' % frac(gemm[:4], inter))
print('gemm == exact pair sum then round : %.4f' % frac(gemm[:4], ps))
print('gemm == sequential chain          : %.4f' % frac(gemm[:4], seq))
