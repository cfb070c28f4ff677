#!/usr/bin/env python3
"""Inter-frame critical path in wall-clock time (no profiler attached): the kernels stamp wall_clock64() at their start / end
into Ctl::dbg[32..47]; this prints the medians over the asynchronous steady state.  python tools/timeline.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import lvt_amd
from lvt_amd.synth import make_world

w = make_world("kitti", seed=0)
prm = lvt_amd.kitti_params()
H, W = w.H, w.W
pitch = ((W + 63) // 64) * 64
n = 160
frames = torch.zeros((n, 2, H, pitch), dtype=torch.uint8, device="cuda")
for i in range(n):
    frames[i, :, :, :W] = w.render_stereo_torch(i, device="cuda")
torch.cuda.synchronize()
vo = lvt_amd.LvtSystem.create(prm, 1)
base, fs = frames.data_ptr(), 2 * H * pitch
host = len(sys.argv) > 1 and sys.argv[1] == "host"   # frames in page-locked host memory through lvt_amd_track_async (the bench headline's entry)
if host:
    hf = torch.empty((n, 2, H, W), dtype=torch.uint8).pin_memory()
    hf.copy_(frames[:, :, :, :W])
    hbase, hfs = hf.data_ptr(), 2 * H * W
    print("host frames (page-locked), lvt_amd_track_async")
tl = []
inflight = 0
for i in range(n):
    if host:
        vo.track_async_ptr(hbase + i * hfs, hbase + i * hfs + H * W, H, W)
    else:
        vo.track_device_async(base + i * fs, base + i * fs + H * pitch, H, W, pitch)
    inflight += 1
    if inflight >= 4:
        vo.wait(); inflight -= 1
        tl.append(vo.timeline().copy())
while inflight:
    vo.wait(); inflight -= 1
    tl.append(vo.timeline().copy())
tl = np.array(tl[40:], dtype=np.float64) / 100.0  # us
names = ["gate start", "gate end", "early_map start", "-", "early_mid start", "early_mid end", "gate_late start", "-", "match_map start", "-",
         "track_mid start", "-", "pnp start", "pnp end", "triangulate end", "feat_done"]
# The early-stream kernels of frame k+1 run during the tail of frame k, so each of their stamps is either already in record k
# (copied at the end of k_triangulate(k): then it is later than that record's pnp end) or still in record k+1 (then it is earlier
# than THAT record's pnp end, i.e. not yet overwritten by frame k+2's early kernels).
def med(x): return float(np.nanmedian(x))
r, nx = tl[:-1], tl[1:]
def early(j):
    v = np.where(r[:, j] > r[:, 13], r[:, j], np.where(nx[:, j] < nx[:, 13], nx[:, j], np.nan))
    return np.where(v > r[:, 13] - 1000.0, v, np.nan)  # (a stamp older than 1 ms before pnp end is a stale one)
g0, g1, em0, ed0, ed1 = r[:, 0], early(1), early(2), early(4), early(5)  # (the gate itself starts long before pnp(k) ends)
def pct(x):
    x = x[~np.isnan(x)]
    return "p10 %.1f  p50 %.1f  p90 %.1f" % tuple(np.percentile(x, [10, 50, 90])) if len(x) else "-"
print("early path of frame k+1 after pnp(k) end, us: gate end [%s]  early_mid end [%s]" % (pct(g1 - r[:, 13]), pct(ed1 - r[:, 13])))
print("features of frame k+1 seen by its gate, us after pnp(k) end (negative = the feature stream was ahead): [%s]" % pct(np.where(r[:, 3] > r[:, 13] - 1000.0, r[:, 3], nx[:, 3]) - r[:, 13]))
print("tracking path: triangulate(k) end [%s]  gate_late(k+1) start [%s]  match_map(k+1) start [%s]" % (pct(r[:, 14] - r[:, 13]), pct(nx[:, 6] - r[:, 13]), pct(nx[:, 8] - r[:, 13])))
# end of k_triangulate(k): stamped after the record copy when the kernel delivers it itself (then it is in record k+1), before the
# copy when the next frame's k_gate_late delivers it (then it is in record k)
tend = np.where(r[:, 11] > r[:, 14], r[:, 11], nx[:, 11])
print("k_triangulate(k): epilogue stamp -> kernel end, us: [%s]   kernel end -> gate_late(k+1) start: [%s]" % (pct(tend - r[:, 14]), pct(nx[:, 6] - tend)))
_e, _t, _m = ed1 - r[:, 13], tend - r[:, 13], nx[:, 8] - r[:, 13]
_ok = ~np.isnan(_e)
print("means, us after pnp(k) end: early_mid(k+1) end %.1f, k_triangulate(k) ended %.1f, later of the two %.1f, match_map(k+1) start %.1f; early stream was the later one in %.0f %% of the frames"
      % (np.mean(_e[_ok]), np.mean(_t[_ok]), np.mean(np.maximum(_e, _t)[_ok]), np.mean(_m[_ok]), 100.0 * np.mean((_e > _t)[_ok])))
print("frame period (pnp start to pnp start)            : %6.1f us" % med(np.diff(tl[:, 12])))
print("pnp(k) end -> gate(k+1) end                       : %6.1f us   (gate started %.1f us before pnp(k) end)" % (med(g1 - r[:, 13]), med(r[:, 13] - g0)))
print("gate end -> early_map start                       : %6.1f us" % med(em0 - g1))
print("early_map start -> early_mid start                : %6.1f us" % med(ed0 - em0))
print("early_mid start -> early_mid end                  : %6.1f us" % med(ed1 - ed0))
print("pnp(k) end -> triangulate(k) end                  : %6.1f us" % med(r[:, 14] - r[:, 13]))
print("early_mid(k+1) end -> match_map(k+1) start        : %6.1f us   (gate_late started %.1f us after early_mid end)" % (med(nx[:, 8] - ed1), med(nx[:, 6] - ed1)))
print("triangulate(k) end -> match_map(k+1) start        : %6.1f us" % med(nx[:, 8] - r[:, 14]))
print("match_map start -> track_mid start -> pnp start   : %6.1f + %.1f us" % (med(nx[:, 10] - nx[:, 8]), med(nx[:, 12] - nx[:, 10])))
print("pnp start -> pnp end                              : %6.1f us" % med(nx[:, 13] - nx[:, 12]))
