#!/usr/bin/env python3
"""Timing of the EuRoC rectification kernel (k_rectify) on device-resident images: python tools/rectify_bench.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import lvt_amd

K = [458.654, 0.0, 367.215, 0.0, 457.296, 248.375, 0.0, 0.0, 1.0]
D = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0]
R = [0.999966347530033, -0.001422739138722922, 0.008079580483432283, 0.001365741834644127, 0.9999741760894847, 0.007055629199258132,
     -0.008089410156878961, -0.007044357138835809, 0.9999424675829176]
P = [435.2046959714599, 0, 367.4517211914062, 0, 435.2046959714599, 252.2008514404297, 0, 0, 1]
W, H, pitch = 752, 480, 768
r = lvt_amd.Rectifier(K, D, R, P, W, H)
L = lvt_amd.load_library()
src = torch.randint(0, 256, (H, W), dtype=torch.uint8, device="cuda")
dst = torch.zeros((H, pitch), dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = torch.cuda.current_stream().cuda_stream
for n in (1, 100):
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        L.lvt_amd_rectify_device(r._h, C.c_void_p(src.data_ptr()), W, C.c_void_p(dst.data_ptr()), pitch, C.c_void_p(st))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    byts = W * H * (8 + 1) + W * H  # maps + output + one pass over the source
    print("%d launch(es): %.2f us each  (%.0f GB/s of map + source + output bytes)" % (n, us, byts / us / 1e3))
