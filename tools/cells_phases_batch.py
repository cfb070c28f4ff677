#!/usr/bin/env python3
"""k_cells phase stamps (cycles) of sequence 0 / cell 0 / left eye inside a lock-step batch: python tools/cells_phases_batch.py [sequences] [depth]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import lvt_amd
from lvt_amd.synth import make_world

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = 24
worlds = [make_world("kitti", seed=100 + s) for s in range(S)]
prm = lvt_amd.kitti_params()
H, W = worlds[0].H, worlds[0].W
pitch = ((W + 63) // 64) * 64
fr = torch.zeros((S, n, 2, H, pitch), dtype=torch.uint8, device="cuda")
for s, w in enumerate(worlds):
    for i in range(n):
        fr[s, i, :, :, :W] = w.render_stereo_torch(i, device="cuda")
torch.cuda.synchronize()
vo = lvt_amd.LvtBatch(prm, S)
vo.debug_stamps = lvt_amd.LvtSystem.debug_stamps.__get__(vo)
phases = ["gather", "(links start)", "union-find", "replay", "survivors", "sort", "rank", "radii", "decision", "emit", "done"]
inflight = 0
rows = []
for i in range(n):
    vo.track_device_async([fr[s, i, 0].data_ptr() for s in range(S)], [fr[s, i, 1].data_ptr() for s in range(S)], H, W, pitch); inflight += 1
    if inflight >= depth:
        vo.wait(); inflight -= 1
        d = vo.debug_stamps()
        rows.append([int(d[k + 1] - d[k]) for k in range(11)] + [int(d[11] - d[0])])
while inflight:
    vo.wait(); inflight -= 1
r = np.median(np.array(rows[8:]), axis=0)
print("batch of %d, depth %d: k_cells cell 0 of sequence 0, median cycles:" % (S, depth), {phases[k]: int(r[k]) for k in range(11)}, "total", int(r[11]))
