#!/usr/bin/env python3
"""Known-traffic kernel for calibrating FETCH_SIZE / WRITE_SIZE: a 256 MiB device-to-device copy (reads 256 MiB,
writes 256 MiB), repeated 5 times."""
import torch
a = torch.empty(256 << 20, dtype=torch.uint8, device='cuda:0').random_(0, 255)
b = torch.empty_like(a)
torch.cuda.synchronize()
for _ in range(5):
    b.copy_(a)
torch.cuda.synchronize()
