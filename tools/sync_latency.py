#!/usr/bin/env python3
"""Latency of the synchronous entry points on a KITTI-shaped synthetic sequence: lvt_track (host buffers, the reference's
C-ABI) against lvt_amd_track_device (device-resident images).  Run on the GPU box:  python tools/sync_latency.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import lvt_amd
from lvt_amd.synth import make_world

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w = make_world("kitti", seed=0)
prm = lvt_amd.kitti_params()
frames = [w.render_stereo(i) for i in range(n)]
pitch = ((w.W + 63) // 64) * 64
dev = torch.zeros((n, 2, w.H, pitch), dtype=torch.uint8, device="cuda")
for i, (L, R) in enumerate(frames):
    dev[i, 0, :, :w.W] = torch.from_numpy(L).cuda(); dev[i, 1, :, :w.W] = torch.from_numpy(R).cuda()
torch.cuda.synchronize()
pinned = [(torch.from_numpy(L).pin_memory(), torch.from_numpy(R).pin_memory()) for L, R in frames]
pinned_np = [(a.numpy(), b.numpy()) for a, b in pinned]
for name in ("host buffers (lvt_track)", "pinned host buffers (lvt_track, read in place)", "device-resident (lvt_amd_track_device)"):
    vo = lvt_amd.LvtSystem.create(prm, 1)
    ts = []
    for i in range(n):
        t0 = time.perf_counter()
        if name.startswith("host"):
            vo.track(frames[i][0], frames[i][1])
        elif name.startswith("pinned"):
            vo.track(pinned_np[i][0], pinned_np[i][1])
        else:
            p = dev[i].data_ptr()
            vo.track_device(p, p + w.H * pitch, w.H, w.W, pitch)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[10:]) * 1e3
    print("%-50s median %.3f ms  mean %.3f ms  -> %.0f frames/s   state %d  %s" % (name, np.median(ts), ts.mean(), 1e3 / ts.mean(), vo.get_state(), vo.last_error()))
