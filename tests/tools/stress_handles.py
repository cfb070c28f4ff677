#!/usr/bin/env python3
"""Stress run: H handles on H host threads, asynchronous (4 frames in flight each), the same HBM-resident sequence for every
handle; all handles must return identical poses and no error.  python tests/tools/stress_handles.py [handles] [frames] [rr]
(rr: all handles driven round-robin from ONE thread instead of a thread each)"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

import lvt_amd
from lvt_amd.synth import make_world

H = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
RR = len(sys.argv) > 3 and sys.argv[3] == "rr"
w = make_world("kitti", seed=2)
prm = lvt_amd.kitti_params()
Hh, W = w.H, w.W
pitch = ((W + 63) // 64) * 64
frames = torch.zeros((n, 2, Hh, pitch), dtype=torch.uint8, device="cuda")
for i in range(n):
    frames[i, :, :, :W] = w.render_stereo_torch(i, device="cuda")
torch.cuda.synchronize()
base, fs = frames.data_ptr(), 2 * Hh * pitch
out = [None] * H
errs = []
modes = [None] * H


def work(k):
    try:
        vo = lvt_amd.LvtSystem.create(prm, 1)
        modes[k] = vo.ordering()
        poses, inflight = [], 0
        for i in range(n):
            vo.track_device_async(base + i * fs, base + i * fs + Hh * pitch, Hh, W, pitch); inflight += 1
            if inflight >= 4:
                poses.append(vo.wait()[1].copy()); inflight -= 1
        while inflight:
            poses.append(vo.wait()[1].copy()); inflight -= 1
        out[k] = np.array(poses)
        if vo.last_error() or vo.get_state() != 2:
            errs.append((k, vo.last_error(), vo.get_state()))
    except Exception as e:  # noqa: BLE001
        errs.append((k, repr(e)))


def round_robin():
    vos = [lvt_amd.LvtSystem.create(prm, 1) for _ in range(H)]
    for k in range(H):
        modes[k] = vos[k].ordering()
    poses = [[] for _ in range(H)]
    for i in range(n):
        for k in range(H):
            vos[k].track_device_async(base + i * fs, base + i * fs + Hh * pitch, Hh, W, pitch)
        if i >= 3:
            for k in range(H):
                poses[k].append(vos[k].wait()[1].copy())
    for _ in range(3):
        for k in range(H):
            poses[k].append(vos[k].wait()[1].copy())
    for k in range(H):
        out[k] = np.array(poses[k])
        if vos[k].last_error() or vos[k].get_state() != 2:
            errs.append((k, vos[k].last_error(), vos[k].get_state()))


t0 = time.perf_counter()
if RR:
    round_robin()
else:
    ths = [threading.Thread(target=work, args=(k,)) for k in range(H)]
    [t.start() for t in ths]; [t.join() for t in ths]
dt = time.perf_counter() - t0
same = all(out[k] is not None and np.array_equal(out[0], out[k]) for k in range(H))
print("STRESS %s handles=%d frames=%d ordering=%s identical=%s errors=%s aggregate %.0f frames/s"
      % ("round-robin" if RR else "threads", H, n, ",".join(m[0] for m in modes), same, errs, H * n / dt))
