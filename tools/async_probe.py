#!/usr/bin/env python3
"""usage: python tools/async_probe.py [first] [count] [depth] [sync]
per-frame completion times of the asynchronous pipeline on device-resident frames (bring-up tool: finds frames that stall)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, lvt_amd
from lvt_amd.synth import make_world
first = int(sys.argv[1]) if len(sys.argv) > 1 else 380
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
w = make_world("kitti", seed=0)
vo = lvt_amd.LvtSystem.create(lvt_amd.kitti_params(), 1)
frames = []
H, W = w.H, w.W
pitch = ((W + 63) // 64) * 64
for i in range(first + count):
    a = torch.zeros((2, H, pitch), dtype=torch.uint8, device="cuda")
    a[:, :, :W] = w.render_stereo_torch(i, device="cuda")  # (bench.py's renderer)
    frames.append((a[0], a[1]))
torch.cuda.synchronize()
done = []
pend = 0
t0 = time.perf_counter()
sync_first = len(sys.argv) > 4 and sys.argv[4] == "sync"   # the frames before `first` one at a time (bench.py's warm-up)
for i, (L, R) in enumerate(frames):
    if sync_first and i < first:
        vo.track_device(L.data_ptr(), R.data_ptr(), H, W, pitch)   # (returns on the pose; the frame's tail finishes behind the call)
        done.append(time.perf_counter())
        continue
    vo.track_device_async(L.data_ptr(), R.data_ptr(), H, W, pitch)
    pend += 1
    if pend >= (1 if (sync_first and i < first) else depth):
        vo.wait(); pend -= 1; done.append(time.perf_counter())
while pend:
    vo.wait(); pend -= 1; done.append(time.perf_counter())
d = np.diff(np.array(done)) * 1e6
for i in range(max(first - 1, 0), len(d)):
    if d[i] > 250: print("frame", i + 1, "completed %.0f us after its predecessor" % d[i], vo.last_error())
print("mean period of frames %d.. : %.1f us" % (first, d[first:].mean()), "counts", vo.counts())
print("stamps", [int(x) for x in vo.debug_stamps()[32:48]])
