#!/usr/bin/env python3
"""Micro-benchmark of the batched Hamming matcher (the kernel bench.py prices against the HBM roof).
usage: python tools/hamming_bench.py [B] [M] [N] [mode]   (LVT_AMD_HAMMING_DEBUG=1 prints phase cycle stamps)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import lvt_amd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
W, H = 1241, 376
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1234)
qd = torch.randint(0, 256, (B, M, 32), dtype=torch.uint8, device=dev, generator=g)
td = torch.randint(0, 256, (B, N, 32), dtype=torch.uint8, device=dev, generator=g)
qxy = (torch.rand((B, M, 2), device=dev, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=dev)).contiguous()
txy = torch.floor(torch.rand((B, N, 2), device=dev, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=dev)).contiguous()
tf = torch.zeros((B, N), dtype=torch.uint8, device=dev)
out = torch.zeros((B, M, 4), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
if os.environ.get("WARM"):   # the statistic bench.py reports: 80 warm-up launches (the clocks settle), then 7 timings of 5 back-to-back launches
    for _ in range(8):
        lvt_amd.hamming_match_batched(qd, qxy, td, txy, tf, 625.0, mode, H, W, out, launches=10)
    us = sorted(lvt_amd.hamming_match_batched(qd, qxy, td, txy, tf, 625.0, mode, H, W, out, launches=5) for _ in range(7))
else:
    us = [lvt_amd.hamming_match_batched(qd, qxy, td, txy, tf, 625.0, mode, H, W, out) for _ in range(8)]
    us = sorted(us[1:])
med = us[len(us) // 2]
byts = B * (40.0 * (M + N) + N + 16.0 * M)
print("B=%d M=%d N=%d mode=%d: median %.1f us  min %.1f us  %.0f GB/s  (%.1f%% of 8 TB/s)" % (B, M, N, mode, med, us[0], byts / med / 1e3, byts / med / 1e3 / 80.0))
# streaming reference: a plain device copy of the same byte count
x = torch.empty(int(byts) // 2, dtype=torch.uint8, device=dev)
y = torch.empty_like(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    y.copy_(x)
e0.record()
for _ in range(10):
    y.copy_(x)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 100.0
print("device copy moving the same bytes (read half + write half): %.1f us -> %.0f GB/s" % (t, byts / t / 1e3))
