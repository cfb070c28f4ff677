"""print the LM traces (lambda, chi2, chi2 of the trial, rho) of the oracle and of k_pnp side by side for the PNP_HARD cases"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lvt_amd
from oracle import pyoracle as O
import test_gpu_primitives as T
prm = lvt_amd.kitti_params()
np.set_printoptions(precision=17, linewidth=250)
for case in T.PNP_HARD:
    X, uv, q0, p0 = T._pnp_hard_case(prm, *case)
    qo, po, marks, tro = O.pnp(prm, q0, p0, X, uv)
    so = (O.pnp.last_trials, O.pnp.last_rejections, O.pnp.last_terminates)
    qh, ph, inl, calls, trh, sh = lvt_amd.pnp_trace(prm, q0, p0, X, uv)
    print("CASE", case, "oracle", so, "hip", sh, "inliers", int(marks.sum()), inl, "dp", np.abs(ph - po).max(), "dq", np.abs(qh - qo).max())
    for i in range(max(len(tro), len(trh))):
        a = tro[i] if i < len(tro) else None
        b = trh[i] if i < len(trh) else None
        print("  ", i, "O", None if a is None else ["%.17g" % v for v in a])
        print("  ", i, "H", None if b is None else ["%.17g" % v for v in b])
