#!/usr/bin/env python3
"""cycle stamps of k_hamming_batched_lists<ROW> (workgroup 0 of sequence 0) in a lock-step batch:  python tools/lists_phases.py [sequences] [frames]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, lvt_amd
from lvt_amd.synth import make_world
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
worlds = [make_world("kitti", seed=100 + s) for s in range(S)]
H, W = worlds[0].H, worlds[0].W
pitch = ((W + 63) // 64) * 64
fr = torch.zeros((S, n, 2, H, pitch), dtype=torch.uint8, device="cuda")
for s, w in enumerate(worlds):
    for i in range(n):
        fr[s, i, :, :, :W] = w.render_stereo_torch(i, device="cuda")
torch.cuda.synchronize()
vo = lvt_amd.LvtBatch(lvt_amd.kitti_params(), S)
acc = []
for i in range(n):
    vo.track_device_async([fr[s, i, 0].data_ptr() for s in range(S)], [fr[s, i, 1].data_ptr() for s in range(S)], H, W, pitch)
    vo.wait()
    d = vo.debug_stamps()[26:32]
    c = vo.counts(0)
    if i >= 5 and d[0] and c["n_row_matches"]:
        acc.append([d[1] - d[0], d[2] - d[1], d[3] - d[2], d[3] - d[0], c["n_left"], c["n_right"]])
a = np.array(acc, dtype=np.float64)
print("frames with row lists:", len(a))
if len(a):
    m = a.mean(axis=0)
    print("cycles: loads + counting sort + scatter %.0f  query words + descriptors %.0f  wavefront-per-query lists %.0f  total %.0f   (n_left %.0f n_right %.0f)" % tuple(m))
