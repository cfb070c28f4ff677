#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the CPU oracle (the reference ships no golden vectors and cannot be built
here -- SURVEY 8c -- so these fixtures pin the ORACLE against regressions and give the HIP path a fixed target).
Inputs are re-rendered from seeds by lvt_amd.synth (bit-reproducible numpy); their SHA-1s are stored so generator
drift is detected.   python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from parity_util import make_case  # noqa: E402
from oracle import pyoracle as O  # noqa: E402


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def sequence_fixture(kind, seed, scale, n_frames):
    world, prm, sensor = make_case(kind, seed, scale)
    orc = O.Oracle(prm, sensor)
    frames = []
    for i in range(n_frames):
        if sensor == 1:
            a, b = world.render_stereo(i)
            R, t = orc.track(a, b)
        else:
            a, b = world.render_rgbd(i)
            R, t = orc.track_rgbd(a, b)
        xl, rl, dl = orc.features(0)
        xr, rr, dr = orc.features(1)
        fi, xyz = orc.matches()
        frames.append({
            "img_sha1": [sha(a), sha(b)],
            "counts": orc.counts(),
            "left_xy_sha1": sha(xl), "left_desc_sha1": sha(dl), "right_xy_sha1": sha(xr), "right_desc_sha1": sha(dr),
            "left_head": [[float(x), float(y), float(r)] + [int(v) for v in d[:8]] for (x, y), r, d in zip(xl[:6], rl[:6], dl[:6])],
            "match_feat_idx": [int(v) for v in fi],
            "row_pairs_sha1": sha(orc.row_matches()),
            "R": [float(v) for v in R.ravel()], "t": [float(v) for v in t],
        })
    return {"kind": kind, "seed": seed, "scale": scale, "frames": frames}


def primitive_fixture():
    rng = np.random.default_rng(20260928)
    tile = rng.integers(0, 256, (24, 32), dtype=np.uint8)
    sm = O.agast_score_map(tile)
    det = O.agast_detect(tile, 20, True)
    train = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    query = train[17] ^ np.array([1] + [0] * 31, np.uint8)
    mask = (rng.uniform(size=40) < 0.7).astype(np.uint8)
    return {"tile": tile.tolist(), "score_map": sm.tolist(), "detect_t20": det.tolist(), "train": train.tolist(),
            "query": query.tolist(), "mask": mask.tolist(), "top2": list(O.hamming_top2(query, train, mask)),
            "top2_nomask": list(O.hamming_top2(query, train))}


if __name__ == "__main__":
    json.dump(sequence_fixture("kitti", 0, 0.5, 5), open(os.path.join(HERE, "kitti_half_seed0.json"), "w"), indent=0)
    json.dump(sequence_fixture("tum", 0, 0.5, 3), open(os.path.join(HERE, "tum_half_seed0.json"), "w"), indent=0)
    json.dump(primitive_fixture(), open(os.path.join(HERE, "primitives.json"), "w"), indent=0)
    print("golden fixtures written")
