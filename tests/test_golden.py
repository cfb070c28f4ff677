"""Golden fixtures (tests/golden/*.json, produced by tests/golden/make_golden.py from the oracle):
CPU tier -- the oracle still reproduces them; GPU tier -- the HIP path reproduces them through the C-ABI."""
import hashlib
import json
import os

import numpy as np
import pytest

from parity_util import make_case

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def _check_sequence(fx, system_factory, is_oracle):
    world, prm, sensor = make_case(fx["kind"], fx["seed"], fx["scale"])
    sysm = system_factory(prm, sensor)
    for i, fr in enumerate(fx["frames"]):
        a, b = world.render_stereo(i) if sensor == 1 else world.render_rgbd(i)
        assert [sha(a), sha(b)] == fr["img_sha1"], "synthetic generator drifted: regenerate the fixtures"
        if sensor == 1:
            R, t = sysm.track(a, b)
        else:
            R, t = sysm.track_rgbd(a, b) if is_oracle else sysm.track(a, b)
        xl, rl, dl = sysm.features(0)
        xr, rr, dr = sysm.features(1)
        assert sha(xl) == fr["left_xy_sha1"] and sha(dl) == fr["left_desc_sha1"], f"frame {i}: left features"
        assert sha(xr) == fr["right_xy_sha1"] and sha(dr) == fr["right_desc_sha1"], f"frame {i}: right features"
        head = [[float(x), float(y), float(r)] + [int(v) for v in d[:8]] for (x, y), r, d in zip(xl[:6], rl[:6], dl[:6])]
        assert head == fr["left_head"]
        fi, _ = sysm.matches()
        assert [int(v) for v in fi] == fr["match_feat_idx"], f"frame {i}: match indices"
        assert sha(sysm.row_matches()) == fr["row_pairs_sha1"], f"frame {i}: row matches"
        c = sysm.counts()
        for k, v in fr["counts"].items():
            assert c[k] == v, f"frame {i}: count {k}"
        tol = 0 if is_oracle else 1e-9          # fp64 stages: tolerance for the GPU path (different summation order)
        assert np.allclose(np.asarray(R).ravel(), fr["R"], rtol=0, atol=tol) and np.allclose(t, fr["t"], rtol=0, atol=tol)


@pytest.mark.parametrize("name", ["kitti_half_seed0", "tum_half_seed0"])
def test_oracle_reproduces_golden_sequence(oracle_lib, name):
    fx = json.load(open(os.path.join(G, name + ".json")))
    _check_sequence(fx, lambda prm, sensor: oracle_lib.Oracle(prm, sensor), True)


def test_oracle_reproduces_golden_primitives(oracle_lib):
    fx = json.load(open(os.path.join(G, "primitives.json")))
    tile = np.array(fx["tile"], np.uint8)
    assert oracle_lib.agast_score_map(tile).tolist() == fx["score_map"]
    assert oracle_lib.agast_detect(tile, 20, True).tolist() == fx["detect_t20"]
    train = np.array(fx["train"], np.uint8); q = np.array(fx["query"], np.uint8)
    assert list(oracle_lib.hamming_top2(q, train, np.array(fx["mask"], np.uint8))) == fx["top2"]
    assert list(oracle_lib.hamming_top2(q, train)) == fx["top2_nomask"]
    assert fx["top2_nomask"][:2] == [17, 1]           # planted match: one flipped bit


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["kitti_half_seed0", "tum_half_seed0"])
def test_hip_reproduces_golden_sequence(hip_lib, name):
    fx = json.load(open(os.path.join(G, name + ".json")))
    _check_sequence(fx, lambda prm, sensor: hip_lib.LvtSystem.create(prm, sensor), False)


# ---- vectors frozen from the REAL third-party libraries (tests/golden/make_upstream_golden.py; absent until somebody runs that script
#      on a machine that has OpenCV + opencv_contrib / g2o -- then these tests pin the oracle on every box, libraries or not) ------------
def _upstream(name):
    p = os.path.join(G, name)
    if not os.path.exists(p):
        pytest.skip(f"{name} not generated yet (no OpenCV / g2o where the fixtures are made): parity unpinned")
    return json.load(open(p))


def test_oracle_reproduces_upstream_opencv_vectors(oracle_lib):
    d = _upstream("upstream_opencv.json")
    for c in d["agast"]:
        det = oracle_lib.agast_detect(np.array(c["img"], np.uint8), c["threshold"], True)
        assert det[:, :2].astype(int).tolist() == c["xy"] and det[:, 2].astype(int).tolist() == c["response"]
    b = d["brief"]
    kept, desc = oracle_lib.brief(np.array(b["img"], np.uint8), np.array(b["xy"], np.float32))
    assert kept.tolist() == b["kept"] and desc.tolist() == b["desc"]
    for c in d["knn2"]:
        assert list(oracle_lib.hamming_top2(np.array(c["query"], np.uint8), np.array(c["train"], np.uint8), np.array(c["mask"], np.uint8))) == c["out"]


def test_oracle_reproduces_upstream_g2o_vectors(oracle_lib):
    import lvt_amd
    d = _upstream("upstream_g2o.json")
    prm = lvt_amd.kitti_params()
    for c in d["pnp"]:
        q, p, marks, _ = oracle_lib.pnp(prm, np.array([1.0, 0, 0, 0]), np.zeros(3), np.array(c["X"]), np.array(c["obs"], np.float32))
        assert marks.tolist() == c["marks"] and int(marks.sum()) == c["inliers"]
        assert np.allclose(p, c["p"], rtol=0, atol=1e-6) and np.allclose(q, c["q"], atol=1e-8)
