#!/usr/bin/env python3
"""S sequences on one GPU as H lock-step batches of S / H driven from H host threads (each batch has its own three streams):
python tools/two_batches.py [S] [H] [frames in flight]     (GPU_MAX_HW_QUEUES in the environment decides how many hardware queues the streams get)"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import lvt_amd
from lvt_amd.synth import make_world

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Hn = int(sys.argv[2]) if len(sys.argv) > 2 else 2
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = 124
per = S // Hn
worlds = [make_world("kitti", seed=100 + s) for s in range(S)]
prm = lvt_amd.kitti_params()
H, W = worlds[0].H, worlds[0].W
pitch = ((W + 63) // 64) * 64
fr = torch.zeros((S, n, 2, H, pitch), dtype=torch.uint8, device="cuda")
for s, w in enumerate(worlds):
    for i in range(n):
        fr[s, i, :, :, :W] = w.render_stereo_torch(i, device="cuda")
torch.cuda.synchronize()
vos = [lvt_amd.LvtBatch(prm, per) for _ in range(Hn)]
ptrs = [[([fr[h * per + s, i, 0].data_ptr() for s in range(per)], [fr[h * per + s, i, 1].data_ptr() for s in range(per)]) for i in range(n)] for h in range(Hn)]
bad = [0] * Hn
bar = threading.Barrier(Hn + 1)


def work(h):
    vo = vos[h]
    for i in range(4):
        vo.track_device_async(*ptrs[h][i], H, W, pitch); vo.wait()
    bar.wait()
    inflight = 0
    for i in range(4, n):
        vo.track_device_async(*ptrs[h][i], H, W, pitch); inflight += 1
        if inflight >= depth:
            bad[h] += int((vo.wait()[2] != 2).sum()); inflight -= 1
    while inflight:
        bad[h] += int((vo.wait()[2] != 2).sum()); inflight -= 1
    bar.wait()


th = [threading.Thread(target=work, args=(h,)) for h in range(Hn)]
for t in th:
    t.start()
bar.wait()
t0 = time.perf_counter()
bar.wait()
dt = time.perf_counter() - t0
for t in th:
    t.join()
print("%d sequences as %d batches of %d, %d frames in flight, GPU_MAX_HW_QUEUES=%s: %.0f frames/s (not tracking: %d; errors: %s)"
      % (S, Hn, per, depth, os.environ.get("GPU_MAX_HW_QUEUES", "default"), S * (n - 4) / dt, sum(bad), [v.last_error()[:40] for v in vos]))
