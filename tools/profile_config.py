#!/usr/bin/env python3
"""Per-kernel HIP-event times of one synthetic configuration: python tools/profile_config.py {kitti|euroc|tum} [frames]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lvt_amd
from lvt_amd.synth import make_world

kind = sys.argv[1] if len(sys.argv) > 1 else "tum"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
w = make_world(kind, seed=0)
prm = {"kitti": lvt_amd.kitti_params, "euroc": lvt_amd.euroc_params, "tum": lvt_amd.tum_params}[kind]()
sensor = 2 if kind == "tum" else 1
vo = lvt_amd.LvtSystem.create(prm, sensor)
frames = [w.render_rgbd(i) if sensor == 2 else w.render_stereo(i) for i in range(n)]
for a, b in frames[:5]:
    vo.track(a, b)
vo.profile_enable(True)
for a, b in frames[5:]:
    vo.track(a, b)
tot = 0.0
for name, ms, calls in vo.profile_read():
    if calls:
        print("%-60s %8.1f us x %d" % (name, 1e3 * ms / calls, calls))
        tot += ms / calls
print("sum of kernels per frame: %.1f us;  counts:" % (1e3 * tot), {k: v for k, v in vo.counts().items() if k in ("features_left", "map_size", "n_matches")})
