"""ctypes binding of the CPU oracle (oracle/liblvt_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by lvt_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblvt_oracle.so")
N_COUNTS = 32
COUNT_NAMES = ["n_left", "n_right", "map_size", "staged_size", "n_matches", "second_pass", "n_row_matches",
               "n_triangulated", "triangulated", "retry_left", "retry_right", "pnp_iters", "pnp_inliers",
               "map_size_at_match", "n_staged_erased", "n_staged_promoted", "n_culled", "frame", "overflow", "pnp_borderline",
               None,  # (slot 20: HIP path only)
               "pnp_trials", "pnp_rejections", "pnp_terminates"]


def build(force: bool = False):
    src = os.path.join(_HERE, "lvt_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, ip, dp, fp, u8p = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        L.lvto_create.restype = vp
        L.lvto_create.argtypes = [vp, C.c_int]
        L.lvto_destroy.argtypes = [vp]
        L.lvto_reset.argtypes = [vp]
        L.lvto_set_threads.argtypes = [vp, C.c_int]
        L.lvto_track.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
        L.lvto_track_rgbd.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
        L.lvto_track_with_external_corners.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp]
        L.lvto_get_status.argtypes = [vp]
        L.lvto_get_counts.argtypes = [vp, vp]
        L.lvto_get_features.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
        L.lvto_get_matches.argtypes = [vp, vp, vp, C.c_int]
        L.lvto_get_row_matches.argtypes = [vp, vp, C.c_int]
        L.lvto_get_map.argtypes = [vp, vp, vp, vp, vp, C.c_int]
        L.lvto_get_staged.argtypes = [vp, vp, vp, vp, C.c_int]
        L.lvto_get_pose.argtypes = [vp, vp, vp]
        L.lvto_get_predicted_pose.argtypes = [vp, vp, vp]
        L.lvto_agast_score_map.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.lvto_agast_detect.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.lvto_anms.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_float]
        L.lvto_sort_by_response.argtypes = [vp, C.c_int]
        L.lvto_init_undistort_rectify_map.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]
        L.lvto_remap_bilinear.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp]
        L.lvto_detect_grid.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp]
        L.lvto_brief.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp]
        L.lvto_compute_features.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp]
        L.lvto_hamming_top2.argtypes = [vp, vp, C.c_int, vp, vp]
        L.lvto_pnp.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp]
        L.lvto_pnp_last_gate.argtypes = [vp, C.c_int, vp]
        L.lvto_pnp_last_stats.argtypes = [vp]
        L.lvto_triangulate_one.argtypes = [vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, vp]
        L.lvto_motion_predict.argtypes = [vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(img):
    a = np.ascontiguousarray(img, dtype=np.uint8)
    assert a.ndim == 2
    return a


class Oracle:
    """Mirror of lvt_system (reference lvt/src/lvt_system.h:57-70) over the oracle."""

    def __init__(self, params, sensor_type: int = 1, threads: int = 2):
        self.pod = params.to_pod()
        self.h = lib().lvto_create(C.byref(self.pod), sensor_type)
        if not self.h:
            raise RuntimeError("lvto_create failed")
        lib().lvto_set_threads(self.h, threads)

    def close(self):
        if self.h:
            lib().lvto_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        lib().lvto_reset(self.h)

    def track(self, left, right):
        l, r = _u8(left), _u8(right)
        R = np.zeros(9); t = np.zeros(3)
        lib().lvto_track(self.h, _p(l), _p(r), l.shape[0], l.shape[1], _p(R), _p(t))
        return R.reshape(3, 3), t

    def track_rgbd(self, gray, depth):
        g = _u8(gray); d = np.ascontiguousarray(depth, dtype=np.float32)
        R = np.zeros(9); t = np.zeros(3)
        lib().lvto_track_rgbd(self.h, _p(g), _p(d), g.shape[0], g.shape[1], _p(R), _p(t))
        return R.reshape(3, 3), t

    def track_with_external_corners(self, left, right, cl, cr):
        l, r = _u8(left), _u8(right)
        cl = np.ascontiguousarray(cl, dtype=np.float64); cr = np.ascontiguousarray(cr, dtype=np.float64)
        R = np.zeros(9); t = np.zeros(3)
        lib().lvto_track_with_external_corners(self.h, _p(l), _p(r), l.shape[0], l.shape[1], _p(cl), len(cl), _p(cr), len(cr), _p(R), _p(t))
        return R.reshape(3, 3), t

    @property
    def status(self):
        return lib().lvto_get_status(self.h)

    def counts(self):
        a = np.zeros(N_COUNTS, dtype=np.int32)
        lib().lvto_get_counts(self.h, _p(a))
        return {n: int(a[i]) for i, n in enumerate(COUNT_NAMES) if n}

    def features(self, eye=0, cap=16384):
        xy = np.zeros((cap, 2), np.float32); resp = np.zeros(cap, np.float32); desc = np.zeros((cap, 32), np.uint8)
        n = lib().lvto_get_features(self.h, eye, _p(xy), _p(resp), _p(desc), cap)
        return xy[:n].copy(), resp[:n].copy(), desc[:n].copy()

    def matches(self, cap=65536):
        fi = np.zeros(cap, np.int32); xyz = np.zeros((cap, 3), np.float64)
        n = lib().lvto_get_matches(self.h, _p(fi), _p(xyz), cap)
        return fi[:n].copy(), xyz[:n].copy()

    def row_matches(self, cap=16384):
        pr = np.zeros((cap, 2), np.int32)
        n = lib().lvto_get_row_matches(self.h, _p(pr), cap)
        return pr[:n].copy()

    def map(self, cap=262144):
        xyz = np.zeros((cap, 3)); cnt = np.zeros(cap, np.int32); age = np.zeros(cap, np.int32); desc = np.zeros((cap, 32), np.uint8)
        n = lib().lvto_get_map(self.h, _p(xyz), _p(cnt), _p(age), _p(desc), cap)
        return xyz[:n].copy(), cnt[:n].copy(), age[:n].copy(), desc[:n].copy()

    def staged(self, cap=65536):
        xyz = np.zeros((cap, 3)); cnt = np.zeros(cap, np.int32); desc = np.zeros((cap, 32), np.uint8)
        n = lib().lvto_get_staged(self.h, _p(xyz), _p(cnt), _p(desc), cap)
        return xyz[:n].copy(), cnt[:n].copy(), desc[:n].copy()

    def pose(self):
        q = np.zeros(4); p = np.zeros(3)
        lib().lvto_get_pose(self.h, _p(q), _p(p))
        return q, p

    def predicted_pose(self):
        q = np.zeros(4); p = np.zeros(3)
        lib().lvto_get_predicted_pose(self.h, _p(q), _p(p))
        return q, p


# ---- primitives -------------------------------------------------------------------------------
def agast_score_map(img):
    a = _u8(img); out = np.zeros(a.shape, np.int16)
    lib().lvto_agast_score_map(_p(a), a.shape[0], a.shape[1], a.shape[1], _p(out))
    return out


def agast_detect(img, threshold, nonmax=True, cap=200000):
    a = _u8(img); xyr = np.zeros((cap, 3), np.float32)
    n = lib().lvto_agast_detect(_p(a), a.shape[0], a.shape[1], a.shape[1], threshold, int(nonmax), _p(xyr), cap)
    return xyr[:n].copy()


def anms(xyr, num_to_keep, tx=0.0, ty=0.0):
    a = np.ascontiguousarray(xyr, dtype=np.float32).copy()
    n = lib().lvto_anms(_p(a), len(a), num_to_keep, tx, ty)
    return a[:n].copy()


def sort_by_response(xyr):
    a = np.ascontiguousarray(xyr, dtype=np.float32).copy()
    lib().lvto_sort_by_response(_p(a), len(a))
    return a


def init_undistort_rectify_map(K, D, R, P, w, h):
    """cv::initUndistortRectifyMap(K, D[5], R, P[:3,:3], (w, h), CV_32FC1) -> (map1, map2)"""
    K, D, R, P = (np.ascontiguousarray(a, dtype=np.float64).reshape(-1) for a in (K, D, R, P))
    m1 = np.zeros((h, w), np.float32); m2 = np.zeros((h, w), np.float32)
    lib().lvto_init_undistort_rectify_map(_p(K), _p(D), _p(R), _p(P), w, h, _p(m1), _p(m2))
    return m1, m2


def remap_bilinear(img, map1, map2):
    """cv::remap(img 8UC1, map1, map2, INTER_LINEAR, BORDER_CONSTANT 0)"""
    a = _u8(img); m1 = np.ascontiguousarray(map1, np.float32); m2 = np.ascontiguousarray(map2, np.float32)
    out = np.zeros(m1.shape, np.uint8)
    lib().lvto_remap_bilinear(_p(a), a.shape[1], a.shape[0], a.shape[1], _p(m1), _p(m2), m1.shape[1], m1.shape[0], _p(out))
    return out


def detect_grid(img, params, cap=65536):
    a = _u8(img); pod = params.to_pod(); xyr = np.zeros((cap, 3), np.float32); retry = C.c_int(0)
    n = lib().lvto_detect_grid(_p(a), a.shape[0], a.shape[1], C.byref(pod), _p(xyr), cap, C.byref(retry))
    return xyr[:n].copy(), retry.value


def brief(img, xy):
    a = _u8(img); xy = np.ascontiguousarray(xy, dtype=np.float32)
    kept = np.zeros(len(xy), np.int32); desc = np.zeros((max(len(xy), 1), 32), np.uint8)
    n = lib().lvto_brief(_p(a), a.shape[0], a.shape[1], _p(xy), len(xy), _p(kept), _p(desc))
    return kept[:n].copy(), desc[:n].copy()


def compute_features(img, params, cap=16384):
    a = _u8(img); pod = params.to_pod()
    xy = np.zeros((cap, 2), np.float32); resp = np.zeros(cap, np.float32); desc = np.zeros((cap, 32), np.uint8); retry = C.c_int(0)
    n = lib().lvto_compute_features(_p(a), a.shape[0], a.shape[1], C.byref(pod), _p(xy), _p(resp), _p(desc), cap, C.byref(retry))
    return xy[:n].copy(), resp[:n].copy(), desc[:n].copy(), retry.value


def hamming_top2(query, train, mask=None):
    q = np.ascontiguousarray(query, np.uint8); t = np.ascontiguousarray(train, np.uint8)
    m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
    out = np.zeros(4, np.int32)
    lib().lvto_hamming_top2(_p(q), _p(t), len(t), _p(m), _p(out))
    return tuple(int(x) for x in out)


def pnp(params, q_in, p_in, pts, obs, trace_cap=256):
    pod = params.to_pod()
    q_in = np.ascontiguousarray(q_in, np.float64); p_in = np.ascontiguousarray(p_in, np.float64)
    pts = np.ascontiguousarray(pts, np.float64); obs = np.ascontiguousarray(obs, np.float32)
    q = np.zeros(4); p = np.zeros(3); marks = np.zeros(len(pts), np.int32); tr = np.zeros((trace_cap, 4))
    calls = C.c_int(0)
    n = lib().lvto_pnp(C.byref(pod), _p(q_in), _p(p_in), _p(pts), _p(obs), len(pts), _p(q), _p(p), _p(marks), _p(tr), trace_cap, C.byref(calls))
    pnp.last_solve_calls = calls.value   # optimize() iterations (g2o solve() calls) of both passes
    err = np.zeros(2 * len(pts)); mm = C.c_double(0.0)
    pnp.last_borderline = lib().lvto_pnp_last_gate(_p(err), len(pts), C.byref(mm))   # gate decisions within 1e-8 of the threshold
    pnp.last_min_margin = mm.value
    pnp.last_err = err.reshape(-1, 2)     # the edge errors pass 2's gate saw
    st = np.zeros(3, np.int32)
    lib().lvto_pnp_last_stats(_p(st))
    pnp.last_trials, pnp.last_rejections, pnp.last_terminates = int(st[0]), int(st[1]), int(st[2])
    return q, p, marks, tr[:min(n, trace_cap)].copy()


def triangulate_one(params, q, pos, ul, ur):
    pod = params.to_pod(); out = np.zeros(3)
    q = np.ascontiguousarray(q, np.float64); pos = np.ascontiguousarray(pos, np.float64)
    ok = lib().lvto_triangulate_one(C.byref(pod), _p(q), _p(pos), float(ul[0]), float(ul[1]), float(ur[0]), float(ur[1]), _p(out))
    return bool(ok), out


def motion_predict(state, q, p):
    st = np.ascontiguousarray(state, np.float64).copy(); qo = np.zeros(4); po = np.zeros(3)
    lib().lvto_motion_predict(_p(st), _p(np.ascontiguousarray(q, np.float64)), _p(np.ascontiguousarray(p, np.float64)), _p(qo), _p(po))
    return st, qo, po
