#!/usr/bin/env python3
"""what the HBM does for a pure write, a pure read and a copy (torch kernels, 2 GiB buffers: far beyond the 256 MB
Infinity Cache), for the roofline discussion of the write-heavy bf16 stages in DESIGN.md"""
import torch
n = 1 << 29  # 2 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
w = t(lambda: a.fill_(1.0)); r = t(lambda: a.sum()); c = t(lambda: b.copy_(a))
gb = n * 4 / 1e12
print(f"write {gb / w:.2f} TB/s   read {gb / r:.2f} TB/s   copy {2 * gb / c:.2f} TB/s (read + write bytes)")
