import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, lvt_amd
from lvt_amd.synth import make_world
w = make_world("kitti", seed=0); prm = lvt_amd.kitti_params()
vo = lvt_amd.LvtSystem.create(prm, 1)
names = ["start", "compact", "links", "replay", "survivors", "anms-sort", "rank", "radii", "decision", "emit", "done", "end"]
for i in range(4):
    L, R = w.render_stereo(i); vo.track(L, R)
    d = vo.debug_stamps()
    print("frame", i, "n_raw", d[20], "n_kp", d[21], "n_out", d[22], " cycles:", {names[k + 1]: int(d[k + 1] - d[k]) for k in range(11)}, "total", int(d[11] - d[0]))
    print("   pnp cycles: err", d[12], "build", d[13], "solve", d[14], "decide", d[15], "all", d[16], "calls", d[17], "| block_sum: wait", d[24], "write", d[25], "segments", d[26], "final", d[27], "read", 0)
    print("   resolve(map): fixpoint iterations", d[18], "fixpoint cycles", d[19], "kernel cycles", d[23], "| init", d[31], "counts+scan", d[29], "pack", d[30], "max list len", d[28], "iteration ends", d[24], d[25], d[26], d[27])
