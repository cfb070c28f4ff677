#!/usr/bin/env python3
"""Freezes the outputs of the REAL third-party libraries the reference delegates to -- OpenCV AGAST / BRIEF / BFMatcher /
undistortPoints / initUndistortRectifyMap + remap and g2o's motion-only BA -- into tests/golden/upstream_*.json, through the adapters
under oracle/upstream/ (which call exactly the entry points the reference calls).  Those files are what would PIN the oracle
(SURVEY 8c): tests/test_golden.py picks them up when they exist and holds the oracle to them on every box, with or without the
libraries.

This image and the GPU box have neither OpenCV nor g2o (profiles/r02_gpu_box_probe.txt): run here, the script says so and writes
nothing -- parity stays "unpinned".  On a machine that has them:
    make -C oracle upstream        # oracle/_upstream/liblvt_upstream.so (+ liblvt_upstream_g2o.so)
    python tests/golden/make_upstream_golden.py
also recovers the genuine BRIEF test-pair table (opencv_contrib's generated_32.i is not vendored in the reference) by probing the
extractor with impulse images and rewrites include/lvt_brief256_pattern.inc from it.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
CV_LIB = os.path.join(ROOT, "oracle", "_upstream", "liblvt_upstream.so")
G2O_LIB = os.path.join(ROOT, "oracle", "_upstream", "liblvt_upstream_g2o.so")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def brief_pattern(up):
    """BRIEF-32's 256 test pairs, recovered from the extractor itself: test k compares the 9x9 box sums S(p + a_k) < S(p + b_k)
    (SURVEY A.3).  A single bright pixel at offset o from the key point raises exactly the box sums whose centre lies within 4 px of o:
    bit k of the descriptor flips from 0 to 1 iff b_k is within the box around o and a_k is not.  Scanning o over the 57 x 57
    neighbourhood gives, for every k, the 9x9 footprint of b_k (bit set) -- and with a DARK pixel on a bright image the footprint of a_k."""
    R = 28
    size = 2 * R + 48 + 1
    c = size // 2
    kp = np.array([[float(c), float(c)]], np.float32)
    desc = np.zeros((1, 32), np.uint8); kept = np.zeros(1, np.int32)

    def bits(img):
        k = up.lvtu_brief(_p(img), size, size, _p(kp), 1, _p(desc), _p(kept))
        assert k == 1
        return np.unpackbits(desc[0])           # bit k of byte j = test 8 j + k, MSB first

    foot_b = np.zeros((256, 2 * R + 1, 2 * R + 1), bool)
    foot_a = np.zeros_like(foot_b)
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            img = np.zeros((size, size), np.uint8); img[c + dy, c + dx] = 255
            foot_b[:, dy + R, dx + R] = bits(img) == 1            # S(a) < S(b): only b's box saw the bright pixel
            img = np.full((size, size), 255, np.uint8); img[c + dy, c + dx] = 0
            foot_a[:, dy + R, dx + R] = bits(img) == 1            # S(a) < S(b): only a's box saw the dark pixel
    pairs = []
    for k in range(256):
        row = []
        for f in (foot_a[k], foot_b[k]):
            ys, xs = np.nonzero(f)
            if len(ys) == 0:
                raise RuntimeError(f"test {k}: empty footprint (a == b, or the offset lies outside the probed window)")
            row += [int(round(xs.mean())) - R, int(round(ys.mean())) - R]   # centre of the (possibly clipped by the twin's box) footprint
        pairs.append(row)                                                     # ax, ay, bx, by
    return pairs


def main():
    have_cv, have_g2o = os.path.exists(CV_LIB), os.path.exists(G2O_LIB)
    if not have_cv and not have_g2o:
        print("make_upstream_golden: neither oracle/_upstream/liblvt_upstream.so nor liblvt_upstream_g2o.so exists (`make -C oracle upstream` "
              "found no OpenCV / g2o): nothing written, parity stays unpinned")
        return 0
    import lvt_amd
    from test_oracle_primitives import EUROC_L
    rng = np.random.default_rng(2024)
    if have_cv:
        up = C.CDLL(CV_LIB)
        out = {"opencv_version": int(up.lvtu_opencv_version())}
        # AGAST + NMS on three small images (noise, a quantised one with many tied responses, blocks)
        imgs = [rng.integers(0, 256, (96, 128), dtype=np.uint8), (rng.integers(0, 4, (80, 112)) * 80).astype(np.uint8)]
        blk = np.full((96, 128), 90, np.uint8)
        for _ in range(40):
            x, y, g = int(rng.integers(8, 110)), int(rng.integers(8, 80)), int(rng.integers(150, 255))
            blk[y:y + int(rng.integers(4, 12)), x:x + int(rng.integers(4, 12))] = g
        imgs.append(blk)
        out["agast"] = []
        for img in imgs:
            for th in (12, 20, 25):
                xy = np.zeros((65536, 2), np.float32); resp = np.zeros(65536, np.float32)
                n = up.lvtu_agast(_p(img), img.shape[0], img.shape[1], img.shape[1], th, _p(xy), _p(resp), 65536)
                out["agast"].append({"img": img.tolist(), "threshold": th, "xy": xy[:n].astype(int).tolist(), "response": resp[:n].astype(int).tolist()})
        # BRIEF: descriptors of random key points (incl. the border filter) + the recovered test-pair table
        img = rng.integers(0, 256, (160, 200), dtype=np.uint8)
        xy = np.column_stack([rng.integers(0, 200, 300), rng.integers(0, 160, 300)]).astype(np.float32)
        xy[::7] += np.float32(0.5)
        desc = np.zeros((300, 32), np.uint8); kept = np.zeros(300, np.int32)
        k = up.lvtu_brief(_p(img), 160, 200, _p(np.ascontiguousarray(xy)), 300, _p(desc), _p(kept))
        out["brief"] = {"img": img.tolist(), "xy": xy.tolist(), "kept": kept[:k].tolist(), "desc": desc[:k].tolist()}
        pairs = brief_pattern(up)
        out["brief_pattern"] = pairs
        with open(os.path.join(ROOT, "include", "lvt_brief256_pattern.inc"), "w") as f:
            f.write("// BRIEF-32 test pairs (ax, ay, bx, by), recovered from the installed opencv_contrib extractor by tests/golden/make_upstream_golden.py\n")
            f.write("// (OpenCV %d): the GENUINE table, no longer the stand-in of rounds 1-3\n" % out["opencv_version"])
            for a in pairs:
                f.write("{%d, %d, %d, %d},\n" % tuple(a))
        # masked 2-NN with ties
        proto = rng.integers(0, 256, (5, 32), dtype=np.uint8)
        out["knn2"] = []
        for n in (1, 2, 40, 333):
            train = np.ascontiguousarray(proto[rng.integers(0, 5, n)] if n > 2 else rng.integers(0, 256, (n, 32), dtype=np.uint8))
            mask = (rng.uniform(size=n) < 0.6).astype(np.uint8)
            res = np.zeros(4, np.int32)
            up.lvtu_knn2(_p(proto[0].copy()), _p(train), n, _p(mask), _p(res))
            out["knn2"].append({"query": proto[0].tolist(), "train": train.tolist(), "mask": mask.tolist(), "out": res.tolist()})
        # undistortPoints with the TUM1 coefficients, rectification maps of EuRoC cam0 (hashes + a sample: the maps are 2 x 1.4 MB)
        prm = lvt_amd.tum_params()
        K = np.array([prm.fx, 0, prm.cx, 0, prm.fy, prm.cy, 0, 0, 1], np.float64)
        D = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float64)
        pts = np.column_stack([rng.uniform(0, 640, 200), rng.uniform(0, 480, 200)]).astype(np.float32)
        und = np.zeros_like(pts)
        up.lvtu_undistort_points(_p(pts), 200, _p(K), _p(D), _p(und))
        out["undistort_points"] = {"K": K.tolist(), "D": D.tolist(), "src": pts.tolist(), "dst_bits": und.view(np.uint32).tolist()}
        Ke, De, Re, Pe = (np.array(EUROC_L[k], np.float64) for k in ("K", "D", "R", "P"))
        m1 = np.zeros((480, 752), np.float32); m2 = np.zeros((480, 752), np.float32)
        up.lvtu_rectify_map(_p(Ke), _p(De), _p(Re), _p(Pe), 752, 480, _p(m1), _p(m2))
        import hashlib
        out["rectify_map"] = {"sha1": [hashlib.sha1(m1.tobytes()).hexdigest(), hashlib.sha1(m2.tobytes()).hexdigest()],
                              "row_100_bits": [m1[100].view(np.uint32).tolist(), m2[100].view(np.uint32).tolist()]}
        json.dump(out, open(os.path.join(HERE, "upstream_opencv.json"), "w"))
        print("wrote tests/golden/upstream_opencv.json and the genuine include/lvt_brief256_pattern.inc (rebuild the oracle and the HIP library)")
    if have_g2o:
        g2 = C.CDLL(G2O_LIB)
        g2.lvtu_g2o_pnp.argtypes = [C.c_double] * 5 + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4
        prm = lvt_amd.kitti_params()
        cases = []
        for n in (12, 200, 777):
            X = np.column_stack([rng.uniform(-20, 20, n), rng.uniform(-5, 5, n), rng.uniform(6, 60, n)])
            p_true = rng.normal(0, 0.3, 3)
            Xc = X - p_true
            uv = np.column_stack([prm.fx * Xc[:, 0] / Xc[:, 2] + prm.cx, prm.fy * Xc[:, 1] / Xc[:, 2] + prm.cy])
            uv = np.rint(uv + rng.normal(0, 0.4, uv.shape)).astype(np.float32)
            uv[::9] += 25.0
            q0 = np.array([1.0, 0, 0, 0]); p0 = np.zeros(3)
            q = np.zeros(4); p = np.zeros(3); marks = np.zeros(n, np.int32); chi2 = np.zeros(n)
            inl = g2.lvtu_g2o_pnp(prm.fx, prm.fy, prm.cx, prm.cy, prm.baseline, _p(q0), _p(p0), _p(X), _p(uv), n, _p(q), _p(p), _p(marks), _p(chi2))
            cases.append({"X": X.tolist(), "obs": uv.tolist(), "q": q.tolist(), "p": p.tolist(), "inliers": int(inl), "marks": marks.tolist(), "chi2": chi2.tolist()})
        json.dump({"pnp": cases}, open(os.path.join(HERE, "upstream_g2o.json"), "w"))
        print("wrote tests/golden/upstream_g2o.json")
    return 0


if __name__ == "__main__":
    sys.exit(main())
