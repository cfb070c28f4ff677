#!/usr/bin/env python3
"""usage: python tools/window_probe.py [K] [depth]
Where a K-step timed window of the bench headline goes (host frames, lvt_amd_track_async, `depth` in flight, device idle on both sides):
enqueue and completion time of every frame relative to the window's start.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, lvt_amd
from lvt_amd.synth import make_world
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w = make_world("kitti", seed=0)
H, W = w.H, w.W
n = 60 + K
hf = torch.empty((n, 2, H, W), dtype=torch.uint8).pin_memory()
for i in range(n):
    hf[i].copy_(w.render_stereo_torch(i, device="cuda")[:, :, :W].cpu())
base, fs = hf.data_ptr(), 2 * H * W
vo = lvt_amd.LvtSystem.create(lvt_amd.kitti_params(), 1)
def run(first, cnt, rec=None):
    infl = 0
    t0 = time.perf_counter()
    for i in range(first, first + cnt):
        vo.track_async_ptr(base + i * fs, base + i * fs + H * W, H, W); infl += 1
        if rec is not None: rec.append(("enq", i - first, (time.perf_counter() - t0) * 1e6))
        if infl >= depth:
            vo.wait(); infl -= 1
            if rec is not None: rec.append(("done", None, (time.perf_counter() - t0) * 1e6))
    while infl:
        vo.wait(); infl -= 1
        if rec is not None: rec.append(("done", None, (time.perf_counter() - t0) * 1e6))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6
run(0, 60)
for rep in range(3):
    rec = []
    tot = run(60, K, rec)
    enq = [t for k, _, t in rec if k == "enq"]; done = [t for k, _, t in rec if k == "done"]
    print("window %.0f us = %.0f frames/s | enqueue returns at" % (tot, K / tot * 1e6), " ".join("%.0f" % t for t in enq))
    print("   frames complete at", " ".join("%.0f" % t for t in done), "| steady intervals", " ".join("%.0f" % (b - a) for a, b in zip(done[4:-1], done[5:])))
