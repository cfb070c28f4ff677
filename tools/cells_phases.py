#!/usr/bin/env python3
"""usage: python tools/cells_phases.py [kitti|euroc|tum]
clock64 phase stamps of the single-workgroup kernels (Ctl::dbg, bring-up profiling): k_cells of cell 0 / left eye,
k_pnp and the early map resolver.  Run on the GPU box:  python tools/cells_phases.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lvt_amd
from lvt_amd.synth import make_world

kind = sys.argv[1] if len(sys.argv) > 1 else "kitti"
w = make_world(kind, seed=0)
prm = {"kitti": lvt_amd.kitti_params, "euroc": lvt_amd.euroc_params, "tum": lvt_amd.tum_params}[kind]()
sensor = 2 if kind == "tum" else 1
vo = lvt_amd.LvtSystem.create(prm, sensor)
# stamp k .. k+1 of k_cells (k_features.hip STAMP(k)); dbg[1] is taken right after the corner gather
phases = ["gather segments", "(links start)", "links + union-find + component maxima", "replay of tied components", "survivors (split cells: + waiting for the helper strip, merge)",
          "std::sort emulation", "rank", "radii", "decision radius", "emit", "done"]
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    L, R = w.render_rgbd(i) if sensor == 2 else w.render_stereo(i)
    vo.track(L, R)
    d = vo.debug_stamps()
    tl = vo.timeline()  # (slots 39 / 41 of the stamp block: cell 0's counts and introsort levels)
    print("frame", i, "k_cells cell 0 cycles:", {phases[k]: int(d[k + 1] - d[k]) for k in range(11)}, "total", int(d[11] - d[0]), "| raw corners", int(tl[7]) & 0xFFFF, "after NMS", (int(tl[7]) >> 16) & 0xFFFF, "kept", int(tl[7]) >> 32, "introsort levels", int(tl[9]))
    print("   k_pnp cycles: sweeps (incl. reductions)", d[12], "reductions", d[13], "solve", d[14], "decide", d[15], "all", d[16], "solve() calls", d[17])
    print("   early map resolver (k_early_mid), sums over its super-chunks: fixpoint iterations", d[18], "fixpoint cycles", d[19], "kernel cycles", d[23], "| init", d[31],
          "counts+scan", d[29], "pack", d[30], "longest list (last chunk)", d[28], "| super-chunks", d[24], "slow-path queries", d[25], "most iterations in a chunk", d[26],
          "queries walked", d[27])
    print("   k_triangulate of the last triangulation frame, cycles: staged update", d[20], "row resolution", d[21], "triangulation + append + epilogue", d[22])
