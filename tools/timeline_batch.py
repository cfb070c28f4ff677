#!/usr/bin/env python3
"""The pose-to-pose cycle of a LOCK-STEP BATCH in wall-clock time (sequence 0's stamps, Ctl::dbg[32..47]; no profiler attached):
k_pnp(t) end -> early gate -> early lists / resolution -> k_match_map(t+1) -> k_track_mid -> k_pnp(t+1).   python tools/timeline_batch.py [sequences]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import lvt_amd
from lvt_amd.synth import make_world

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 70
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
lp = [[fr[s, i, 0].data_ptr() for s in range(S)] for i in range(n)]
rp = [[fr[s, i, 1].data_ptr() for s in range(S)] for i in range(n)]
tl, inflight = [], 0
for i in range(n):
    vo.track_device_async(lp[i], rp[i], H, W, pitch); inflight += 1
    if inflight >= 2:
        vo.wait(); inflight -= 1
        tl.append(vo.timeline().copy())
while inflight:
    vo.wait(); inflight -= 1
    tl.append(vo.timeline().copy())
tl = np.array(tl[12:], dtype=np.float64) / 100.0
r, nx = tl[:-1], tl[1:]
def med(x): return float(np.nanmedian(x))
def early(j):   # an early-stream stamp of frame k+1: already in record k (later than its pnp end) or still in record k+1
    v = np.where(r[:, j] > r[:, 13], r[:, j], np.where(nx[:, j] < nx[:, 13], nx[:, j], np.nan))
    return np.where(v > r[:, 13] - 2000.0, v, np.nan)
g1, em0, ed0, ed1 = early(1), early(2), early(4), early(5)
print("lock-step batch of %d: frame period (pnp start to pnp start) %.1f us" % (S, med(np.diff(tl[:, 12]))))
print("  pnp(k) end -> early gate end                 %6.1f" % med(g1 - r[:, 13]))
print("  gate end -> early lists / early_map start    %6.1f" % med(em0 - g1))
print("  -> early_mid start                           %6.1f" % med(ed0 - em0))
print("  early_mid start -> end                       %6.1f" % med(ed1 - ed0))
print("  early_mid end -> match_map(k+1) start        %6.1f   (gate_late started %.1f after early_mid end; triangulate(k) ended %.1f after pnp(k) end)"
      % (med(nx[:, 8] - ed1), med(nx[:, 6] - ed1), med(r[:, 14] - r[:, 13])))
print("  match_map start -> track_mid start           %6.1f" % med(nx[:, 10] - nx[:, 8]))
print("  track_mid start -> pnp start                 %6.1f" % med(nx[:, 12] - nx[:, 10]))
print("  pnp start -> pnp end                         %6.1f" % med(nx[:, 13] - nx[:, 12]))
