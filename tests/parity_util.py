"""Shared helpers: build a (world, params) pair for a named config and diff the HIP path against the oracle
stage by stage (bit-exact for integer stages, tolerance for fp64 stages)."""
import numpy as np

import lvt_amd
from lvt_amd.synth import make_world

POSE_TOL = 1e-4      # BASELINE.json: per-frame SE3 within 1e-4 rel of the CPU reference
XYZ_TOL = 1e-7       # map point positions (fp64, different summation order only)
PRED_TOL = 1e-7      # motion-model prediction (fp64: quaternion w x y z, position) -- it inherits the optimised pose's last digits


def make_case(kind="kitti", seed=0, scale=1.0, overrides=None):
    world = make_world(kind, seed=seed, scale=scale)
    mk = {"kitti": lvt_amd.kitti_params, "euroc": lvt_amd.euroc_params, "tum": lvt_amd.tum_params}[kind]
    kw = dict(width=world.W, height=world.H, fx=world.fx, fy=world.fy, cx=world.cx, cy=world.cy)
    if kind != "tum":
        kw["baseline"] = world.baseline
    prm = mk(**kw)
    for k, v in (overrides or {}).items():
        setattr(prm, k, type(getattr(prm, k))(v))
    return world, prm, (2 if kind == "tum" else 1)


class HardWorld:
    """the synthetic sequences are kind to a tracker: integer-pixel warps of one texture, identical photometry in both eyes and all frames.  This
    wrapper makes the SAME geometry harder the way real footage is: per-frame and per-eye gain / bias (auto-exposure), sensor noise that differs
    between the eyes, a 3 x 3 blur on every third frame (motion blur / defocus), and a right image that sits one row off on every fourth frame
    (imperfect rectification: candidates at the edge of the +-2 row band).  More ambiguous ratio tests, more outliers in front of the chi2 gates,
    fewer features per cell -- and the HIP path must still equal the oracle bit for bit."""

    def __init__(self, world, seed=0):
        self.w, self.seed = world, seed
        self.W, self.H = world.W, world.H

    def _degrade(self, img, rng, blur):
        a = img.astype(np.float32)
        if blur:
            p = np.pad(a, 1, mode="edge")
            a = (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:] + 2 * p[1:-1, :-2] + 4 * p[1:-1, 1:-1] + 2 * p[1:-1, 2:] + p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) / 16.0
        a = a * rng.uniform(0.8, 1.2) + rng.uniform(-12, 12) + rng.normal(0.0, 4.0, a.shape)
        return np.clip(np.rint(a), 0, 255).astype(np.uint8)

    def render_stereo(self, i):
        L, R = self.w.render_stereo(i)
        rng = np.random.default_rng(1000003 * self.seed + i)
        blur = (i % 3) == 2
        L2, R2 = self._degrade(L, rng, blur), self._degrade(R, rng, blur)
        if i % 4 == 3:
            R2 = np.vstack([R2[1:], R2[-1:]])     # one row up
        return np.ascontiguousarray(L2), np.ascontiguousarray(R2)

    def pose(self, i):
        return self.w.pose(i)


def pose_errors(Rh, th, Ro, to):
    e_t = np.linalg.norm(th - to) / max(np.linalg.norm(to), 1.0)
    e_R = float(np.arccos(np.clip((np.trace(Rh.T @ Ro) - 1) / 2, -1, 1)))
    return e_t, e_R


def diff_frame(hip, orc):
    """list of human-readable discrepancies between the two systems after the same frame"""
    msgs = []
    if hip.last_error():
        msgs.append("hip error: " + hip.last_error())
    co, ch = orc.counts(), hip.counts()
    for k, v in co.items():
        if ch.get(k) != v:
            msgs.append(f"count {k}: hip={ch.get(k)} oracle={v}")
    if ch.get("overflow"):
        msgs.append(f"overflow mask {ch['overflow']}")

    def eq(name, a, b):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape:
            msgs.append(f"{name}: shape {a.shape} vs {b.shape}")
        elif not np.array_equal(a, b):
            i = tuple(np.argwhere(a != b)[0])
            msgs.append(f"{name}: first mismatch at {i}: hip={a[i]} oracle={b[i]}")

    def close(name, a, b, tol):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape:
            msgs.append(f"{name}: shape {a.shape} vs {b.shape}")
        elif a.size and np.abs(a - b).max() > tol:
            msgs.append(f"{name}: max abs err {np.abs(a - b).max():.3e} > {tol}")

    for eye in (0, 1):
        xo, ro, do = orc.features(eye)
        xh, rh, dh = hip.features(eye)
        eq(f"features[{eye}].xy", xh, xo); eq(f"features[{eye}].resp", rh, ro); eq(f"features[{eye}].desc", dh, do)
    fo, po = orc.matches(); fh, ph = hip.matches()
    eq("find_matches feature idx", fh, fo)
    if fh.shape == fo.shape:
        close("find_matches map xyz", ph, po, XYZ_TOL)
    eq("row_match pairs", hip.row_matches(), orc.row_matches())
    mo, mh = orc.map(), hip.map()
    close("map xyz", mh[0], mo[0], XYZ_TOL); eq("map counter", mh[1], mo[1]); eq("map age", mh[2], mo[2]); eq("map desc", mh[3], mo[3])
    so, sh = orc.staged(), hip.staged()
    close("staged xyz", sh[0], so[0], XYZ_TOL); eq("staged counter", sh[1], so[1]); eq("staged desc", sh[2], so[2])
    # MM row (lvt_motion_model.cpp:42-65): the pose the frame's matching started from
    if co.get("map_size_at_match", 0) > 0:   # (a prediction is only made for a frame that starts in TRACKING: lvt_system.cpp:196-199)
        qo, po = orc.predicted_pose(); qh, ph = hip.predicted_pose()
        close("predicted pose q", qh, qo, PRED_TOL); close("predicted pose p", ph, po, PRED_TOL)
    if orc.status != hip.get_state():
        msgs.append(f"status hip={hip.get_state()} oracle={orc.status}")
    return msgs


def run_sequence(world, prm, sensor, frame_ids, hip=None, orc=None):
    """track the frames through both systems; returns list of (frame, msgs, e_t, e_R)"""
    from oracle import pyoracle as O
    orc = orc or O.Oracle(prm, sensor)
    hip = hip or lvt_amd.LvtSystem.create(prm, sensor)
    out = []
    for i in frame_ids:
        if sensor == 1:
            a, b = world.render_stereo(i)
            Ro, to = orc.track(a, b)
            Rh, th = hip.track(a, b)
        else:
            a, b = world.render_rgbd(i)
            Ro, to = orc.track_rgbd(a, b)
            Rh, th = hip.track(a, b)
        msgs = diff_frame(hip, orc)
        e_t, e_R = pose_errors(Rh, th, Ro, to)
        if e_t > POSE_TOL or e_R > POSE_TOL:
            msgs.append(f"pose e_t={e_t:.3e} e_R={e_R:.3e}")
        out.append((i, msgs, e_t, e_R))
    return out, hip, orc


def sparse_pair(world, seed=0, n=40):
    """a nearly empty stereo pair (n small squares on flat gray): fewer than 50 features, so find_matches' first pass
    must fall short of LVT_N_MATCHES_TH and the doubled-radius second pass runs (lvt_local_map.cpp:173-199)"""
    rng = np.random.default_rng(seed)
    L = np.full((world.H, world.W), 110, np.uint8)
    R = L.copy()
    for _ in range(n):
        x = int(rng.integers(60, world.W - 60)); y = int(rng.integers(40, world.H - 40)); g = int(rng.integers(180, 255))
        L[y:y + 7, x:x + 7] = g
        R[y:y + 7, x - 9:x - 2] = g
    return L, R
