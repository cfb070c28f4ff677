"""CPU tier: the oracle is validated against something external to itself -- the known ground-truth motion of the
synthetic sequences -- and its control flow is exercised (staging, culling, policies, LOST)."""
import numpy as np

from parity_util import make_case, sparse_pair


def test_oracle_tracks_ground_truth(oracle_lib):
    world, prm, sensor = make_case("kitti", 0, 1.0)       # full KITTI shape: depth variety makes the pose well conditioned
    orc = oracle_lib.Oracle(prm, sensor)
    tri_frames = 0
    for i in range(10):
        L, R = world.render_stereo(i)
        Ro, to = orc.track(L, R)
        Rg, tg = world.pose(i)
        assert orc.status == 2
        assert np.linalg.norm(to - tg) < 0.01, f"frame {i}: {to} vs {tg}"
        assert np.arccos(np.clip((np.trace(Ro.T @ Rg) - 1) / 2, -1, 1)) < 1e-3
        c = orc.counts()
        tri_frames += c["triangulated"]
        if i:
            assert c["n_matches"] >= 50 and c["pnp_iters"] > 0
    assert tri_frames >= 2
    assert orc.counts()["staged_size"] + orc.counts()["n_staged_promoted"] + orc.counts()["n_staged_erased"] >= 0
    xyz, cnt, age, desc = orc.map()
    assert len(xyz) == orc.counts()["map_size"] > 100
    assert (cnt < prm.untracked_threshold).all()      # culled points are gone


def test_oracle_rgbd_and_thread_invariance(oracle_lib):
    world, prm, sensor = make_case("tum", 0, 1.0)
    a = oracle_lib.Oracle(prm, sensor, threads=1)
    for i in range(4):
        g, d = world.render_rgbd(i)
        Ro, to = a.track_rgbd(g, d)
        assert np.linalg.norm(to - world.pose(i)[1]) < 0.03
    world, prm, sensor = make_case("kitti", 3, 0.5)
    one, two = oracle_lib.Oracle(prm, 1, threads=1), oracle_lib.Oracle(prm, 1, threads=2)
    for i in range(4):
        L, R = world.render_stereo(i)
        assert np.array_equal(one.track(L, R)[1], two.track(L, R)[1])


def test_oracle_lost_latch(oracle_lib):
    world, prm, _ = make_case("kitti", 1, 0.5, {"min_num_matches_for_tracking": 60})
    orc = oracle_lib.Oracle(prm, 1)
    for i in range(3):
        orc.track(*world.render_stereo(i))
    last = orc.track(*world.render_stereo(3))[1]
    orc.track(*sparse_pair(world))
    assert orc.status == 3 and orc.counts()["second_pass"] == 1
    assert np.array_equal(orc.track(*world.render_stereo(4))[1], last)


def test_oracle_follows_the_ground_truth_on_degraded_footage(oracle_lib):
    """parity_util.HardWorld (exposure changes, per-eye noise, blur, a right image one row off): the oracle keeps TRACKING, stays within centimetres
    of the ground-truth motion, and the chi2 gates really demote edges (the GPU tier runs the same frames through the HIP path)"""
    from parity_util import HardWorld
    world, prm, _ = make_case("kitti", 71, 0.5)
    hw = HardWorld(world, 71)
    orc = oracle_lib.Oracle(prm, 1)
    demoted = 0
    for i in range(30):
        R, t = orc.track(*hw.render_stereo(i))
        assert orc.status == 2 and np.linalg.norm(t - world.pose(i)[1]) < 0.1, i   # (half-size images: half the disparity)
        c = orc.counts()
        demoted += c["n_matches"] - c["pnp_inliers"]
    assert demoted > 300, demoted
