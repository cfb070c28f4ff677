#!/usr/bin/env python3
"""cycle stamps of k_pnp (one sequence, synchronous calls):  python tools/pnp_phases.py [frames]
dbg[12..17] = cycles in sweeps (+ their reductions), reductions of the speculative sweeps alone, thread 0's solves, its decisions, the whole solve, solve() calls"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, lvt_amd
from lvt_amd.synth import make_world
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
w = make_world("kitti", seed=100)
H, W = w.H, w.W
pitch = ((W + 63) // 64) * 64
fr = torch.zeros((n, 2, H, pitch), dtype=torch.uint8, device="cuda")
for i in range(n):
    fr[i, :, :, :W] = w.render_stereo_torch(i, device="cuda")
torch.cuda.synchronize()
vo = lvt_amd.LvtSystem.create(lvt_amd.kitti_params())
acc = []
for i in range(n):
    vo.track_device(fr[i, 0].data_ptr(), fr[i, 1].data_ptr(), H, W, pitch)
    d = vo.debug_stamps()
    c = vo.counts()
    if i >= 5:
        acc.append([d[12], d[13], d[14], d[15], d[16], d[17], c["n_matches"], c["pnp_trials"]])
a = np.array(acc, dtype=np.float64)
m = a.mean(axis=0)
print("cycles per frame: sweeps+reductions %.0f  (speculative reductions %.0f)  solves %.0f  decisions %.0f  whole %.0f   solve calls %.1f   matches %.0f  trials %.1f" % tuple(m))
