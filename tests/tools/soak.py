#!/usr/bin/env python3
"""Soak run: N frames of a KITTI-shaped synthetic sequence through the ASYNCHRONOUS pipeline (4 frames in flight), every pose
compared with the CPU oracle's.  python tests/tools/soak.py [frames] [seed] [host]   (run on the GPU box; a third argument "host" feeds the frames from
page-locked HOST memory through lvt_amd_track_async instead of lvt_amd_track_device_async)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

import lvt_amd
from lvt_amd.synth import make_world
from oracle import pyoracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
host_mode = len(sys.argv) > 3 and sys.argv[3] == "host"
w = make_world("kitti", seed=seed)
prm = lvt_amd.kitti_params()
H, W = w.H, w.W
pitch = ((W + 63) // 64) * 64
CH = 500  # frames rendered per chunk (HBM resident while they are tracked)
vo = lvt_amd.LvtSystem.create(prm, 1)
orc = O.Oracle(prm, 1, threads=2)
worst_t = worst_R = 0.0
bad = 0
t_gpu = 0.0
done = 0
for c0 in range(0, n, CH):
    m = min(CH, n - c0)
    frames = torch.zeros((m, 2, H, pitch), dtype=torch.uint8, device="cuda")
    for i in range(m):
        frames[i, :, :, :W] = w.render_stereo_torch(c0 + i, device="cuda")
    torch.cuda.synchronize()
    host_t = frames[:, :, :, :W].contiguous().cpu().pin_memory()
    host = host_t.numpy()
    hbase, himg = host_t.data_ptr(), H * W
    base, fs = frames.data_ptr(), 2 * H * pitch
    poses, inflight = [], 0
    t0 = time.perf_counter()
    for i in range(m):
        if host_mode:
            assert vo.track_async_ptr(hbase + 2 * i * himg, hbase + (2 * i + 1) * himg, H, W) == 0
        else:
            vo.track_device_async(base + i * fs, base + i * fs + H * pitch, H, W, pitch)
        inflight += 1
        if inflight >= 4:
            poses.append(vo.wait()); inflight -= 1
    while inflight:
        poses.append(vo.wait()); inflight -= 1
    t_gpu += time.perf_counter() - t0
    for i in range(m):
        Ro, to = orc.track(host[i, 0], host[i, 1])
        Rh, th = poses[i]
        et = np.linalg.norm(th - to) / max(np.linalg.norm(to), 1.0)
        eR = np.arccos(np.clip((np.trace(Rh.T @ Ro) - 1) / 2, -1, 1))
        worst_t, worst_R = max(worst_t, et), max(worst_R, eR)
        bad += (et > 1e-4) or (eR > 1e-4)
    done += m
    print("frames %d: state hip %d oracle %d, worst e_t %.2e e_R %.2e, frames over tolerance %d, %.0f frames/s, error '%s'"
          % (done, vo.get_state(), orc.status, worst_t, worst_R, bad, done / t_gpu, vo.last_error()), flush=True)
    if vo.get_state() != orc.status:
        bad += 1
        break
c, co = vo.counts(), orc.counts()
same = all(c[k] == co[k] for k in ("map_size", "n_matches", "n_left", "n_right") if k in c and k in co)
print("SOAK frames=%d bad=%d worst_e_t=%.3e worst_e_R=%.3e map=%d final_counts_equal=%s" % (done, bad, worst_t, worst_R, c["map_size"], same))
