#!/usr/bin/env python3
"""Builds and runs tests/tools/stress_handles.cpp (several handles, one C++ host thread each) on rendered KITTI-shaped frames.
   python tests/tools/stress_handles_cpp.py [frames per handle] [handle counts ...]   e.g.  ... 3000 1 2 4 8"""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np

from lvt_amd.synth import make_world

per = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
counts = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
out = os.environ.get("TMPDIR", "/tmp")
exe = os.path.join(out, "stress_handles")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "tools", "stress_handles.cpp"),
                       "-o", exe, "-L" + os.path.join(ROOT, "lvt_amd", "lib"), "-llvt_c", "-Wl,-rpath," + os.path.join(ROOT, "lvt_amd", "lib"), "-lpthread"])
w = make_world("kitti", seed=2)
n = 96
pitch = ((w.W + 63) // 64) * 64
buf = np.zeros((n, 2, w.H, pitch), np.uint8)
for i in range(n):
    L, R = w.render_stereo(i)
    buf[i, 0, :, :w.W], buf[i, 1, :, :w.W] = L, R
path = os.path.join(out, "stress_frames.bin")
buf.tofile(path)
cfg = os.path.join(out, "stress_vo.yaml")
import lvt_amd
lvt_amd.kitti_params().write_yaml(cfg)
for h in counts:
    env = dict(os.environ)
    r = subprocess.run([exe, path, str(n), str(w.H), str(w.W), str(pitch), str(h), str(per), cfg], env=env, capture_output=True, text=True)
    print("rc", r.returncode, r.stdout.strip(), r.stderr.strip()[-600:])
