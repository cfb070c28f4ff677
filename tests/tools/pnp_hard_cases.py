"""Search (on the CPU oracle) for motion-only-BA inputs that take the branches of g2o's Levenberg-Marquardt a good prior never reaches: rejected
trials, Terminate, NaN steps.  Its hits are the PNP_HARD table of tests/test_gpu_primitives.py."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import lvt_amd
from oracle import pyoracle as O
prm = lvt_amd.kitti_params()

def case(seed, n, off_t, off_deg, outl, noise=0.4, big=25.0):
    rng = np.random.default_rng(seed)
    X = np.column_stack([rng.uniform(-20, 20, n), rng.uniform(-5, 5, n), rng.uniform(6, 60, n)])
    uv = np.column_stack([prm.fx * X[:, 0] / X[:, 2] + prm.cx, prm.fy * X[:, 1] / X[:, 2] + prm.cy])
    uv = np.rint(uv + rng.normal(0, noise, uv.shape)).astype(np.float32)
    k = rng.random(n) < outl
    uv[k] += rng.uniform(-big, big, (int(k.sum()), 2)).astype(np.float32)
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    a = np.deg2rad(off_deg)
    q0 = np.array([np.cos(a/2), *(np.sin(a/2)*ax)])
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    p0 = off_t * d
    return X, uv, q0, p0

for (n, off_t, off_deg, outl, big) in [(300,1.5,10,0.3,25),(300,2.0,10,0.3,60),(200,2.0,15,0.3,200),(100,3,30,0.4,300),(60,5,60,0.5,400),(40,8,120,0.5,400),(30,10,170,0.3,100)]:
    found=[]
    for seed in range(40):
        X, uv, q0, p0 = case(seed, n, off_t, off_deg, outl, big=big)
        q,p,marks,tr = O.pnp(prm, q0, p0, X, uv)
        nan = bool(np.isnan(tr).any())
        found.append((seed, O.pnp.last_trials, O.pnp.last_rejections, O.pnp.last_terminates, nan, O.pnp.last_solve_calls, int(marks.sum())))
    print((n,off_t,off_deg,outl,big))
    print("  rej>0:", [f for f in found if f[2]>0][:6])
    print("  term>0:", [f for f in found if f[3]>0][:6])
    print("  nan:", [f for f in found if f[4]][:6])
