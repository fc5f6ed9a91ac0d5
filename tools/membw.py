#!/usr/bin/env python3
"""Calibrate streaming bandwidth of the box with torch fill/copy (development tool)."""
import torch
dev = torch.device("cuda", 0)
n = 166_000_000  # 664 MB of fp32
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
us = t(lambda: a.fill_(1.0)); print(f"fill  {n*4/1e6:.0f} MB: {us:.1f} us -> {n*4/us/1e6:.2f} TB/s written")
us = t(lambda: b.copy_(a)); print(f"copy  {n*4/1e6:.0f} MB: {us:.1f} us -> {2*n*4/us/1e6:.2f} TB/s (r+w)")
us = t(lambda: a.sum()); print(f"read  {n*4/1e6:.0f} MB (sum): {us:.1f} us -> {n*4/us/1e6:.2f} TB/s read")
