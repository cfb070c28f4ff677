#!/usr/bin/env python3
"""Stage-by-stage differential run: HIP path (through the C-ABI) vs the CPU oracle on a synthetic sequence.
Prints, per frame, the first stage that diverges.  Used on the GPU box during bring-up:
    python tests/tools/gpu_parity.py --kind kitti --frames 30
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import lvt_amd  # noqa: E402
from lvt_amd.synth import make_world  # noqa: E402
from oracle import pyoracle as O  # noqa: E402


def box9(img):
    a = np.pad(img.astype(np.int64), 4)
    s = np.zeros(img.shape, np.int64)
    for dy in range(9):
        for dx in range(9):
            s += a[dy:dy + img.shape[0], dx:dx + img.shape[1]]
    return s


def expected_score_plane(img, params):
    """per-cell isolated OAST score, thresholded at the lowered threshold (what k_score must emit)"""
    H, W = img.shape
    cs = params.detection_cell_size
    t_low = int(params.agast_threshold * 0.5 + 0.5)
    out = np.zeros((H, W), np.int32)
    for y0 in range(0, H, cs):
        for x0 in range(0, W, cs):
            roi = np.ascontiguousarray(img[y0:y0 + cs, x0:x0 + cs])
            sm = O.agast_score_map(roi).astype(np.int32)
            sm[sm < t_low] = 0
            out[y0:y0 + cs, x0:x0 + cs] = sm
    return out


def cmp(name, a, b, exact=True, tol=0.0):
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape:
        return f"{name}: shape {a.shape} vs {b.shape}"
    if exact:
        bad = np.argwhere(a != b)
        if len(bad):
            i = tuple(bad[0])
            return f"{name}: {len(bad)} mismatches, first at {i}: hip={a[i]} oracle={b[i]}"
    else:
        err = np.abs(a - b)
        if err.size and err.max() > tol:
            i = np.unravel_index(np.argmax(err), err.shape)
            return f"{name}: max abs err {err.max():.3e} at {i}: hip={a[i]} oracle={b[i]}"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="kitti")
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--planes", type=int, default=1)
    ap.add_argument("--verbose", type=int, default=1)
    ap.add_argument("--set", action="append", default=[], help="param override key=value")
    ap.add_argument("--jump", type=int, default=-1, help="skip 40 frames at this index (forces second pass / LOST)")
    a = ap.parse_args()

    world = make_world(a.kind, seed=a.seed, scale=a.scale)
    mk = {"kitti": lvt_amd.kitti_params, "euroc": lvt_amd.euroc_params, "tum": lvt_amd.tum_params}[a.kind]
    kw = dict(width=world.W, height=world.H, fx=world.fx, fy=world.fy, cx=world.cx, cy=world.cy)
    if a.kind != "tum":
        kw["baseline"] = world.baseline
    prm = mk(**kw)
    for kv in a.set:
        k, v = kv.split('=')
        setattr(prm, k, type(getattr(prm, k))(float(v)))
    sensor = 2 if a.kind == "tum" else 1
    orc = O.Oracle(prm, sensor)
    hip = lvt_amd.LvtSystem.create(prm, sensor)
    n_bad = 0
    worst_t = worst_r = 0.0
    t_hip = t_cpu = 0.0
    for fi in range(a.frames):
        i = fi if (a.jump < 0 or fi < a.jump) else fi + 40
        if sensor == 1:
            L, R = world.render_stereo(i)
            t0 = time.time(); Ro, to = orc.track(L, R); t_cpu += time.time() - t0
            t0 = time.time(); Rh, th = hip.track(L, R); t_hip += time.time() - t0
        else:
            L, D = world.render_rgbd(i)
            t0 = time.time(); Ro, to = orc.track_rgbd(L, D); t_cpu += time.time() - t0
            t0 = time.time(); Rh, th = hip.track(L, D); t_hip += time.time() - t0
        msgs = []
        if hip.last_error():
            msgs.append("hip error: " + hip.last_error())
        if a.planes and fi == 0:
            for eye, img in ((0, L),) + (((1, R),) if sensor == 1 else ()):
                sp = hip.plane(eye, 0)[:, :world.W].astype(np.int32)
                m = cmp(f"score plane eye{eye}", sp, expected_score_plane(img, prm))
                if m: msgs.append(m)
                bp = hip.plane(eye, 1)[:, :world.W].astype(np.int64)
                m = cmp(f"boxsum plane eye{eye}", bp, box9(img))
                if m: msgs.append(m)
        co, ch = orc.counts(), hip.counts()
        for k in co:
            if co[k] != ch.get(k):
                msgs.append(f"count {k}: hip={ch.get(k)} oracle={co[k]}")
        if ch.get("overflow"):
            msgs.append(f"overflow mask {ch['overflow']}")
        for eye in (0, 1):
            xo, ro, do = orc.features(eye); xh, rh, dh = hip.features(eye)
            for nm, x, y in (("xy", xh, xo), ("resp", rh, ro), ("desc", dh, do)):
                m = cmp(f"features eye{eye} {nm}", x, y)
                if m: msgs.append(m)
        fo, po = orc.matches(); fh, ph = hip.matches()
        m = cmp("match feat idx", fh, fo)
        if m: msgs.append(m)
        elif len(fo):
            m = cmp("match xyz", ph, po, exact=False, tol=1e-7)
            if m: msgs.append(m)
        m = cmp("row matches", hip.row_matches(), orc.row_matches())
        if m: msgs.append(m)
        mo, mh = orc.map(), hip.map()
        for nm, x, y, ex in (("map xyz", mh[0], mo[0], False), ("map counter", mh[1], mo[1], True), ("map age", mh[2], mo[2], True),
                             ("map desc", mh[3], mo[3], True)):
            m = cmp(nm, x, y, exact=ex, tol=1e-7)
            if m: msgs.append(m)
        so, sh = orc.staged(), hip.staged()
        for nm, x, y, ex in (("staged xyz", sh[0], so[0], False), ("staged counter", sh[1], so[1], True), ("staged desc", sh[2], so[2], True)):
            m = cmp(nm, x, y, exact=ex, tol=1e-7)
            if m: msgs.append(m)
        if orc.status != hip.get_state():
            msgs.append(f"status hip={hip.get_state()} oracle={orc.status}")
        e_t = np.linalg.norm(th - to) / max(np.linalg.norm(to), 1.0)
        cosang = np.clip((np.trace(Rh.T @ Ro) - 1) / 2, -1, 1)
        e_R = float(np.arccos(cosang))
        worst_t, worst_r = max(worst_t, e_t), max(worst_r, e_R)
        if e_t > 1e-4 or e_R > 1e-4:
            msgs.append(f"pose e_t={e_t:.3e} e_R={e_R:.3e} hip t={th} oracle t={to}")
        Rg, tg = world.pose(i)
        gt_err = np.linalg.norm(to - tg)
        tag = "OK " if not msgs else "BAD"
        if msgs:
            n_bad += 1
        if a.verbose or msgs:
            print(f"[{tag}] frame {fi}: N=({ch['n_left']},{ch['n_right']}) map={ch['map_size']} staged={ch['staged_size']} "
                  f"matches={ch['n_matches']} rowm={ch['n_row_matches']} tri={ch['n_triangulated']} pnp_it={ch['pnp_iters']} "
                  f"e_t={e_t:.2e} e_R={e_R:.2e} oracle_vs_gt={gt_err:.4f}")
        for m in msgs[:12]:
            print("      ", m)
    print(f"SUMMARY kind={a.kind} frames={a.frames} bad_frames={n_bad} worst_e_t={worst_t:.3e} worst_e_R={worst_r:.3e} "
          f"hip_ms/frame={1e3 * t_hip / a.frames:.3f} (incl. H2D of host images) oracle_ms/frame={1e3 * t_cpu / a.frames:.3f}")
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
