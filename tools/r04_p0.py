#!/usr/bin/env python3
"""P0 (test_region_grow.py:119-173) rates of the GPU preprocessing by finish: all-GPU Jacobi, verified Jacobi + LAPACK for the uncertain points
(eig='exact': bit-exact features and seed order), LAPACK for every point, host NumPy."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.p0_rates(int(sys.argv[1]) if len(sys.argv) > 1 else 8, torch.device('cuda:0'))))
