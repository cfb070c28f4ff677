"""-m gpu: the parity tests proper.  Every case drives the HIP path THROUGH THE C-ABI (lvt_track & co in
liblvt_c.so) and compares it with the CPU oracle on the same seeded synthetic input: key points, descriptors,
match indices, row matches, map bookkeeping bit-exact; map positions and per-frame SE3 within tolerance
(POSE_TOL = 1e-4, the tolerance BASELINE.json states)."""
import numpy as np
import pytest

from parity_util import make_case, run_sequence, diff_frame, sparse_pair, HardWorld, POSE_TOL

pytestmark = pytest.mark.gpu

CASES = [
    # name, kind, seed, scale, overrides, frame ids
    ("kitti_half", "kitti", 0, 0.5, {}, list(range(14))),
    ("kitti_full", "kitti", 1, 1.0, {}, list(range(8))),                       # BASELINE.json configs[1] shape
    ("kitti_dense_anms", "kitti", 2, 1.0, {"agast_threshold": 12, "max_keypoints_per_cell": 60}, list(range(5))),
    ("kitti_low_corner_retry", "kitti", 4, 1.0, {"agast_threshold": 150}, list(range(6))),
    ("kitti_jump", "kitti", 3, 1.0, {}, list(range(8)) + list(range(48, 54))),      # 40 frames skipped: culling, staging, many re-triangulations
    # ~70 features per image: the first pass of find_matches stays below 50 matches on EVERY frame, so the doubled-radius
    # second pass (lvt_local_map.cpp:173-199) runs while tracking continues (asserted below)
    ("kitti_sparse_second_pass", "kitti", 3, 1.0, {"max_keypoints_per_cell": 8}, list(range(12))),
    ("kitti_full_long", "kitti", 9, 1.0, {}, list(range(220))),               # staging / promotion / culling at full size over 220 frames
    # BASELINE.json's sequence LENGTHS (kitti_example.cpp:113-138 walks the whole sequence; cfg 2 has 4 541 frames, cfg 3 3 682, cfg 4 573):
    # long enough for every slow drift of the map bookkeeping (culling ages, staged promotion, the 3-deep match window) to recur many times
    ("kitti_full_1000", "kitti", 13, 1.0, {}, list(range(1000))),
    ("euroc_300", "euroc", 2, 1.0, {}, list(range(300))),
    ("tum_rgbd_300", "tum", 2, 1.0, {}, list(range(300))),
    # degraded footage on the same geometry (parity_util.HardWorld: exposure changes per frame and eye, sensor noise, blur, a right image one row off):
    # ~20 % of the matches end as outliers in front of the chi2 gates, ratio tests are borderline far more often
    ("kitti_hard", "kitti", 70, 1.0, {}, list(range(80))),
    ("kitti_hard_half", "kitti", 71, 0.5, {}, list(range(60))),
    ("kitti_always_triangulate", "kitti", 5, 0.5, {"triangulation_policy": 2, "staged_threshold": 0}, list(range(10))),
    ("kitti_map_size_policy", "kitti", 6, 0.5, {"triangulation_policy": 3}, list(range(10))),
    ("euroc", "euroc", 0, 1.0, {}, list(range(10))),                           # configs[2] shape
    ("tum_rgbd", "tum", 0, 1.0, {}, list(range(8))),                           # configs[3] shape (single 640x480 cell)
    ("tum_rgbd_distorted", "tum", 1, 1.0, {"k1": 0.262383, "k2": -0.953104, "p1": -0.005358, "p2": 0.002628, "k3": 1.163314},
     list(range(5))),
]


@pytest.mark.parametrize("name,kind,seed,scale,overrides,frames", CASES, ids=[c[0] for c in CASES])
def test_sequence_parity(hip_lib, oracle_lib, name, kind, seed, scale, overrides, frames):
    world, prm, sensor = make_case(kind, seed, scale, overrides)
    if name.startswith("kitti_hard"):
        world = HardWorld(world, seed)
    res, hip, orc = run_sequence(world, prm, sensor, frames)
    bad = [(i, m) for i, m, _, _ in res if m]
    assert not bad, f"{name}: first divergence at frame {bad[0][0]}: {bad[0][1][:6]}"
    assert max(r[2] for r in res) <= POSE_TOL and max(r[3] for r in res) <= POSE_TOL
    c = hip.counts()
    if name.startswith("kitti_hard"):
        assert hip.get_state() == 2 and c["pnp_inliers"] < 0.9 * c["n_matches"], (c["pnp_inliers"], c["n_matches"])   # the gates really demote edges here
    if name == "kitti_low_corner_retry":
        assert c["retry_left"] == 1, "the <200-corner retry path was not exercised"
    if name == "kitti_sparse_second_pass":
        assert c["second_pass"] == 1 and hip.get_state() == 2, "the second pass of find_matches was not exercised while TRACKING"
    if name == "kitti_full_long":
        assert hip.get_state() == 2 and c["frame"] == 219
    if name in ("kitti_full_1000", "euroc_300", "tum_rgbd_300"):
        assert hip.get_state() == 2 and c["frame"] == len(frames) - 1, (hip.get_state(), c["frame"])
    if name == "tum_rgbd":
        assert c["n_right"] == 0
        # (more map points than one resolver super-chunk holds: the compacted query list and several super-chunks were exercised)
        assert c["map_size_at_match"] > 2048, c["map_size_at_match"]


@pytest.mark.parametrize("name", ["kitti_full", "kitti_dense_anms", "kitti_jump", "euroc", "tum_rgbd"])
def test_binned_list_kernel_on_a_single_handle(hip_lib, oracle_lib, monkeypatch, name):
    """k_hamming_batched_lists (the binned matcher's list-emitting form; lock-step batches use it by default) forced onto a single
    handle: the early map lists and the row lists it builds must lead to the oracle's matches, maps and poses on every shape -- 3x3 and
    5x5 cell windows (TUM: tracking radius 30), dense ANMS output, a 40-frame jump"""
    monkeypatch.setenv("LVT_AMD_BINNED_LISTS", "1")
    case = next(c for c in CASES if c[0] == name)
    world, prm, sensor = make_case(case[1], case[2], case[3], case[4])
    res, hip, orc = run_sequence(world, prm, sensor, case[5][:10])
    bad = [(i, m) for i, m, _, _ in res if m]
    assert not bad, f"{name}: first divergence at frame {bad[0][0]}: {bad[0][1][:6]}"


def test_brief_without_the_box_sum_plane(hip_lib, oracle_lib, monkeypatch):
    """LVT_AMD_BRIEF_FROM_IMAGE=1: k_score writes no box-sum plane, k_brief_img builds the 9 x 9 sums of every key point's 57 x 57 patch in LDS straight
    from the image.  Same pixels, same sums: descriptors -- hence every later stage -- stay bit-identical to the oracle's, on detected corners and on
    external (fractional, border) corners, which take the clipped element-wise path"""
    from oracle import pyoracle as O
    monkeypatch.setenv("LVT_AMD_BRIEF_FROM_IMAGE", "1")
    world, prm, sensor = make_case("kitti", 21, 0.5)
    hip = hip_lib.LvtSystem.create(prm, 1)
    orc = O.Oracle(prm, 1)
    rng = np.random.default_rng(9)
    for i in range(14):
        a, b = world.render_stereo(i)
        if i % 5 == 3:
            xl, _, _, _ = O.compute_features(a, prm)
            xr, _, _, _ = O.compute_features(b, prm)
            cl = xl.astype(np.float64); cr = xr.astype(np.float64)
            cl[::7] += rng.uniform(-0.5, 0.5, size=cl[::7].shape)
            cl = np.vstack([cl, [[3.0, 3.0], [world.W - 28.5, world.H - 28.5], [27.5, 27.5]]])
            orc.track_with_external_corners(a, b, cl, cr)
            hip.track_with_external_corners(a, b, cl, cr)
        else:
            orc.track(a, b)
            hip.track(a, b)
        msgs = diff_frame(hip, orc)
        assert not msgs, f"frame {i}: {msgs[:5]}"
    assert hip.last_error() == ""


@pytest.mark.parametrize("binned", ["0", "1"], ids=["wave_per_query_lists", "binned_list_kernel"])
def test_row_lists_built_twice_at_once(hip_lib, oracle_lib, monkeypatch, binned):
    """k_triangulate's fallback for late row-match lists (the early stream's gate stood down: 5-ms time-out) builds the lists itself --
    possibly WHILE the early stream's kernel, arriving late, writes the same words.  LVT_AMD_TEST_ROW_FALLBACK=1 forces exactly that on every
    triangulation frame (no wait, both producers run): row pairs, maps and poses must stay the oracle's, with either list kernel on the
    early stream."""
    monkeypatch.setenv("LVT_AMD_TEST_ROW_FALLBACK", "1")
    monkeypatch.setenv("LVT_AMD_BINNED_LISTS", binned)
    monkeypatch.setenv("LVT_AMD_ORDERING", "polling")
    world, prm, sensor = make_case("kitti", 15, 1.0, {"triangulation_policy": 2, "staged_threshold": 0})   # triangulate on every frame
    res, hip, orc = run_sequence(world, prm, sensor, range(10))
    bad = [(i, m) for i, m, _, _ in res if m]
    assert not bad, f"first divergence at frame {bad[0][0]}: {bad[0][1][:6]}"
    assert hip.counts()["row_fallback"] == 1 and hip.counts()["n_row_matches"] > 0


def test_second_pass_and_lost_latch(hip_lib, oracle_lib):
    """a scene cut: the doubled-radius pass runs, then tracking is LOST and stays lost (lvt_system.cpp:161-166)"""
    world, prm, sensor = make_case("kitti", 7, 0.5, {"min_num_matches_for_tracking": 60})
    res, hip, orc = run_sequence(world, prm, sensor, range(4))
    assert not [m for _, m, _, _ in res if m]
    a, b = sparse_pair(world)               # nearly empty scene
    Ro, to = orc.track(a, b); Rh, th = hip.track(a, b)
    assert hip.counts()["second_pass"] == orc.counts()["second_pass"] == 1
    assert not diff_frame(hip, orc)
    assert hip.get_state() == orc.status == 3
    last = th.copy()
    for i in range(2):                      # LOST is sticky: returns the last pose, nothing is computed
        a, b = world.render_stereo(5 + i)
        Ro, to = orc.track(a, b); Rh, th = hip.track(a, b)
        assert hip.get_state() == 3 and np.array_equal(th, last) and np.allclose(th, to, atol=1e-9)
    hip.reset(); orc.reset()                # lvt_system::reset
    res, _, _ = run_sequence(world, prm, sensor, range(3), hip=hip, orc=orc)
    assert not [m for _, m, _, _ in res if m]
    assert hip.get_state() == 2


def test_external_corners(hip_lib, oracle_lib):
    """lvt_track_with_external_corners (lvt_c.cpp:91-132): detection skipped, BRIEF at the given (fractional) corners"""
    from oracle import pyoracle as O
    world, prm, sensor = make_case("kitti", 8, 0.5)
    hip = hip_lib.LvtSystem.create(prm, 1)
    orc = O.Oracle(prm, 1)
    rng = np.random.default_rng(5)
    for i in range(5):
        a, b = world.render_stereo(i)
        xl, _, _, _ = O.compute_features(a, prm)
        xr, _, _, _ = O.compute_features(b, prm)
        cl = xl.astype(np.float64); cr = xr.astype(np.float64)
        cl[::7] += rng.uniform(-0.5, 0.5, size=cl[::7].shape)          # fractional corners
        cr[::5] += 0.5                                                   # exact .5 (round-half cases)
        cl = np.vstack([cl, [[3.0, 3.0], [world.W - 28.5, world.H - 28.5], [27.5, 27.5]]])   # border filter cases
        Ro, to = orc.track_with_external_corners(a, b, cl, cr)
        Rh, th = hip.track_with_external_corners(a, b, cl, cr)
        msgs = diff_frame(hip, orc)
        assert not msgs, f"frame {i}: {msgs[:5]}"
        assert np.allclose(th, to, atol=1e-6) and np.allclose(Rh, Ro, atol=1e-6)


@pytest.mark.parametrize("binned", ["0", "1"], ids=["wave_per_query_lists", "binned_list_kernel"])
def test_candidate_lists_longer_than_their_capacity(hip_lib, oracle_lib, monkeypatch, binned):
    """(both list builders: k_early_map / k_candidates and, forced onto the single handle, k_hamming_batched_lists -- whose lists longer than
    128 entries only report their length and leave the query to the resolvers' exact path, k_lists.hip)
    a 28 x 28 block of corners at 1-px spacing (through the external-corner entry point) puts several hundred features inside the
    search window of the map points that project there and 140 into every 5-row band of row_match: candidate lists overflow their
    128 slots, and the resolvers must take their exact wave-wide path for those queries -- in map order, between ordinary super-chunks"""
    from oracle import pyoracle as O
    monkeypatch.setenv("LVT_AMD_BINNED_LISTS", binned)
    world, prm, sensor = make_case("kitti", 11, 0.5)
    hip = hip_lib.LvtSystem.create(prm, 1)
    orc = O.Oracle(prm, 1)
    gx, gy = np.meshgrid(np.arange(28, dtype=np.float64), np.arange(28, dtype=np.float64))
    block = np.stack([gx.ravel() + world.W // 2 - 14, gy.ravel() + world.H // 2 - 14], axis=1)
    for i in range(4):
        a, b = world.render_stereo(i)
        xl, _, _, _ = O.compute_features(a, prm)
        xr, _, _, _ = O.compute_features(b, prm)
        cl = np.vstack([xl.astype(np.float64), block]); cr = np.vstack([xr.astype(np.float64), block + [[-6.0, 0.0]]])
        Ro, to = orc.track_with_external_corners(a, b, cl, cr)
        Rh, th = hip.track_with_external_corners(a, b, cl, cr)
        msgs = diff_frame(hip, orc)
        assert not msgs, f"frame {i}: {msgs[:5]}"
        assert np.allclose(th, to, atol=1e-6) and np.allclose(Rh, Ro, atol=1e-6)
    assert hip.counts()["n_left"] > 784 and hip.get_state() == orc.status
    assert hip.debug_stamps()[25] > 0, "no map point took the resolver's exact path for over-long lists (bring-up counter of k_early_mid)"


@pytest.mark.parametrize("binned", ["0", "1"], ids=["wave_per_query_lists", "binned_list_kernel"])
def test_more_than_2048_features_per_image(hip_lib, oracle_lib, monkeypatch, binned):
    """2 600 external corners per image: more than the 2 048 features the binned list kernel holds in its LDS image -- with
    LVT_AMD_BINNED_LISTS=1 it must stand down for these frames (k_lists.hip: N > 2048) and the wave-per-query kernels behind it build
    the lists; either way matches, maps and poses are the oracle's"""
    from oracle import pyoracle as O
    monkeypatch.setenv("LVT_AMD_BINNED_LISTS", binned)
    world, prm, sensor = make_case("kitti", 14, 1.0)
    hip = hip_lib.LvtSystem.create(prm, 1)
    orc = O.Oracle(prm, 1)
    rng = np.random.default_rng(14)
    for i in range(4):
        a, b = world.render_stereo(i)
        xl, _, _, _ = O.compute_features(a, prm)
        xr, _, _, _ = O.compute_features(b, prm)
        extra = np.stack([rng.integers(40, world.W - 40, 1800), rng.integers(40, world.H - 40, 1800)], axis=1).astype(np.float64)
        cl = np.vstack([xl.astype(np.float64), extra]); cr = np.vstack([xr.astype(np.float64), extra + [[-8.0, 0.0]]])
        Ro, to = orc.track_with_external_corners(a, b, cl, cr)
        Rh, th = hip.track_with_external_corners(a, b, cl, cr)
        msgs = diff_frame(hip, orc)
        assert not msgs, f"frame {i}: {msgs[:5]}"
        assert np.allclose(th, to, atol=1e-6) and np.allclose(Rh, Ro, atol=1e-6)
    assert 2048 < hip.counts()["n_left"] <= 4096 and hip.get_state() == orc.status == 2


def test_more_features_than_capacity_is_reported(hip_lib):
    """4 900 external corners survive the border filter, the feature arrays hold 4 096: the frame is tracked on the first 4 096 and the cut
    is reported through lvt_amd_last_error (the reference has no such limit: include/lvt_c.h, "capacities") -- never silently"""
    world, prm, sensor = make_case("kitti", 12, 1.0)
    hip = hip_lib.LvtSystem.create(prm, 1)
    gx, gy = np.meshgrid(np.arange(70, dtype=np.float64) * 3 + 40, np.arange(70, dtype=np.float64) * 3 + 40)
    grid = np.stack([gx.ravel(), gy.ravel()], axis=1)
    a, b = world.render_stereo(0)
    hip.track_with_external_corners(a, b, grid, grid)
    c = hip.counts()
    assert c["n_left"] == 4096 and c["n_right"] == 4096 and c["overflow"] != 0
    assert "capacity overflow" in hip.last_error()
    hip.track_with_external_corners(a, b, grid[:500], grid[:500])      # the handle keeps working
    assert hip.counts()["n_left"] == 500 and hip.counts()["overflow"] == 0


def test_create_from_yaml_matches_struct_create(hip_lib, oracle_lib, tmp_path):
    """lvt_create(config.yaml) (lvt_c.cpp:33-48) == lvt_system::create(params): same poses, missing keys read as 0"""
    world, prm, sensor = make_case("kitti", 9, 0.5)
    y = tmp_path / "vo.yaml"
    prm.write_yaml(str(y))
    a = hip_lib.LvtSystem.create_from_file(str(y), 1)
    b = hip_lib.LvtSystem.create(prm, 1)
    for i in range(4):
        L, R = world.render_stereo(i)
        Ra, ta = a.track(L, R); Rb, tb = b.track(L, R)
        assert np.array_equal(ta, tb) and np.array_equal(Ra, Rb)
    with pytest.raises(RuntimeError):
        hip_lib.LvtSystem.create_from_file(str(tmp_path / "missing.yaml"), 1)    # NULL on unreadable file
    with pytest.raises(RuntimeError):
        hip_lib.LvtSystem.create_from_file(str(y), 3)                             # NULL on bad sensor type


def test_full_size_properties(hip_lib):
    """BASELINE.json full-size shape, size-independent properties: determinism across handles, identity first pose,
    rotation matrices orthonormal, device-resident entry point == host-buffer entry point."""
    import torch
    world, prm, sensor = make_case("kitti", 10, 1.0)
    frames = [world.render_stereo(i) for i in range(6)]
    h1 = hip_lib.LvtSystem.create(prm, 1)
    h2 = hip_lib.LvtSystem.create(prm, 1)
    pitch = ((world.W + 63) // 64) * 64
    dev = torch.zeros((6, 2, world.H, pitch), dtype=torch.uint8, device="cuda")
    for i, (L, R) in enumerate(frames):
        dev[i, 0, :, :world.W] = torch.from_numpy(L).cuda(); dev[i, 1, :, :world.W] = torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    for i, (L, R) in enumerate(frames):
        R1, t1 = h1.track(L, R)
        p = dev[i].data_ptr()
        R2, t2 = h2.track_device(p, p + world.H * pitch, world.H, world.W, pitch)
        assert np.array_equal(t1, t2) and np.array_equal(R1, R2), "host-buffer and device-resident entry points disagree"
        assert np.allclose(R1 @ R1.T, np.eye(3), atol=1e-12)
        if i == 0:
            assert np.array_equal(R1, np.eye(3)) and np.array_equal(t1, np.zeros(3))
        Rg, tg = world.pose(i)
        assert np.linalg.norm(t1 - tg) < 0.02, "odometry drifted from ground truth"


def test_async_pipeline_equals_synchronous(hip_lib):
    """lvt_amd_track_device_async / lvt_amd_wait (feature stage of frame t+1 overlapped with the tracking chain of
    frame t, several frames in flight) returns exactly the poses of the synchronous entry point, in FIFO order"""
    import torch
    world, prm, sensor = make_case("kitti", 12, 0.5)
    n = 24
    pitch = ((world.W + 63) // 64) * 64
    dev = torch.zeros((n, 2, world.H, pitch), dtype=torch.uint8, device="cuda")
    for i in range(n):
        L, R = world.render_stereo(i)
        dev[i, 0, :, :world.W] = torch.from_numpy(L).cuda(); dev[i, 1, :, :world.W] = torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    a = hip_lib.LvtSystem.create(prm, 1); b = hip_lib.LvtSystem.create(prm, 1)
    ref = []
    for i in range(n):
        p = dev[i].data_ptr()
        ref.append(a.track_device(p, p + world.H * pitch, world.H, world.W, pitch))
    got = []
    inflight = 0
    for i in range(n):
        p = dev[i].data_ptr()
        b.track_device_async(p, p + world.H * pitch, world.H, world.W, pitch)
        inflight += 1
        if inflight >= 5:
            got.append(b.wait()); inflight -= 1
    while inflight:
        got.append(b.wait()); inflight -= 1
    assert len(got) == n
    for i, ((Ra, ta), (Rb, tb)) in enumerate(zip(ref, got)):
        assert np.array_equal(ta, tb) and np.array_equal(Ra, Rb), f"frame {i}"
    assert a.counts() == b.counts()
    assert b.get_state() == 2 and b.last_error() == ""


def _pinned(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()


def test_async_host_path_equals_the_oracle(hip_lib, oracle_lib):
    """lvt_amd_track_async: borrowed HOST images (lvt_c.cpp:64-89), frames enqueued four deep, the pull of frame t + 1 carried by the corner-cell launch of frame t, beside
    the kernels of frame t.  Held to the ORACLE (not to the synchronous HIP path) over 210 full-size frames (1241 x 376 = 8 mod 16 bytes per
    image): every pose within 1e-4, the state after every frame, and the complete frame diff at the end.  Page-locked buffers (read in place)
    and pageable ones (copied into the staging ring) alternate in blocks; a wrong-size frame arrives mid-flight and must be rejected without
    disturbing the frames around it."""
    from oracle import pyoracle as O
    from parity_util import pose_errors
    world, prm, sensor = make_case("kitti", 21, 1.0)
    n, depth = 210, 4
    hip = hip_lib.LvtSystem.create(prm, 1)
    orc = O.Oracle(prm, 1)
    frames = [world.render_stereo(i) for i in range(n)]
    bufs = []
    for i, (L, R) in enumerate(frames):
        pin = (i // 7) % 2 == 0
        bufs.append((_pinned(L), _pinned(R)) if pin else (np.ascontiguousarray(L), np.ascontiguousarray(R)))
    got, inflight, rejected = [], 0, 0
    small = np.zeros((world.H - 8, world.W), np.uint8)
    for i in range(n):
        if i == 77:   # wrong size, with frames in flight: -1, nothing enqueued, an error string; the FIFO goes on undisturbed
            assert hip.track_async(small, small) == -1
            rejected += 1
        assert hip.track_async(bufs[i][0], bufs[i][1]) == 0
        inflight += 1
        if inflight >= depth:
            got.append(hip.wait_status()); inflight -= 1
    while inflight:
        got.append(hip.wait_status()); inflight -= 1
    assert len(got) == n and rejected == 1
    assert "image size" in hip.last_error()
    worst = (0.0, 0.0)
    for i, (L, R) in enumerate(frames):
        Ro, to = orc.track(L, R)
        Rh, th, st = got[i]
        e_t, e_R = pose_errors(Rh, th, Ro, to)
        assert e_t <= POSE_TOL and e_R <= POSE_TOL and st == orc.status, f"frame {i}: e_t {e_t:.2e} e_R {e_R:.2e} state {st} / {orc.status}"
        worst = (max(worst[0], e_t), max(worst[1], e_R))
    msgs = [m for m in diff_frame(hip, orc) if "image size" not in m]
    assert not msgs, msgs[:6]
    hs = hip.host_stats()
    assert hs["async_host_frames"] == n and hs["planes_in_place"] > 0 and hs["planes_staged"] > 0, hs
    # frames that arrived while their predecessor was still held had their images pulled by that frame's corner-cell launch; the others pulled their own
    assert 0 < hs["pulls_carried_by_the_previous_frame"] < n, hs
    assert hip.get_state() == 2


def test_async_host_rgbd_equals_the_oracle(hip_lib, oracle_lib):
    """lvt_amd_track_rgbd_async (gray u8 + depth f32 host buffers, three in flight) against the oracle on a TUM-shaped sequence"""
    from oracle import pyoracle as O
    from parity_util import pose_errors
    world, prm, sensor = make_case("tum", 4, 1.0)
    n = 24
    hip = hip_lib.LvtSystem.create(prm, 2)
    orc = O.Oracle(prm, 2)
    frames = [world.render_rgbd(i) for i in range(n)]
    bufs = [((_pinned(g), _pinned(np.asarray(d, np.float32))) if i % 2 else (np.ascontiguousarray(g), np.ascontiguousarray(d, dtype=np.float32))) for i, (g, d) in enumerate(frames)]
    got, inflight = [], 0
    for i in range(n):
        assert hip.track_async(bufs[i][0], bufs[i][1]) == 0
        inflight += 1
        if inflight >= 3:
            got.append(hip.wait_status()); inflight -= 1
    while inflight:
        got.append(hip.wait_status()); inflight -= 1
    for i, (g, d) in enumerate(frames):
        Ro, to = orc.track_rgbd(g, d)
        e_t, e_R = pose_errors(got[i][0], got[i][1], Ro, to)
        assert e_t <= POSE_TOL and e_R <= POSE_TOL and got[i][2] == orc.status, f"frame {i}"
    msgs = diff_frame(hip, orc)
    assert not msgs, msgs[:6]


def test_async_device_pipeline_equals_the_oracle(hip_lib, oracle_lib):
    """lvt_amd_track_device_async, four frames in flight, 200 frames: every pose and state against the ORACLE's, the full frame diff at the end"""
    import torch
    from oracle import pyoracle as O
    from parity_util import pose_errors
    world, prm, sensor = make_case("kitti", 22, 0.5)
    n, depth = 200, 4
    pitch = ((world.W + 63) // 64) * 64
    dev = torch.zeros((n, 2, world.H, pitch), dtype=torch.uint8, device="cuda")
    frames = [world.render_stereo(i) for i in range(n)]
    for i, (L, R) in enumerate(frames):
        dev[i, 0, :, :world.W] = torch.from_numpy(L).cuda(); dev[i, 1, :, :world.W] = torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    hip = hip_lib.LvtSystem.create(prm, 1)
    orc = O.Oracle(prm, 1)
    got, inflight = [], 0
    for i in range(n):
        p = dev[i].data_ptr()
        hip.track_device_async(p, p + world.H * pitch, world.H, world.W, pitch)
        inflight += 1
        if inflight >= depth:
            got.append(hip.wait_status()); inflight -= 1
    while inflight:
        got.append(hip.wait_status()); inflight -= 1
    for i, (L, R) in enumerate(frames):
        Ro, to = orc.track(L, R)
        e_t, e_R = pose_errors(got[i][0], got[i][1], Ro, to)
        assert e_t <= POSE_TOL and e_R <= POSE_TOL and got[i][2] == orc.status, f"frame {i}: {e_t:.2e} {e_R:.2e}"
    msgs = diff_frame(hip, orc)
    assert not msgs, msgs[:6]


def test_lockstep_batch_equals_the_oracles(hip_lib, oracle_lib):
    """8 sequences x 60 frames through ONE launch chain (lvt_amd_batch_*, three lock-step frames in flight) against EIGHT oracle instances:
    every pose and state, and each sequence's counters after the last frame"""
    import torch
    from oracle import pyoracle as O
    from parity_util import pose_errors
    B, n, depth = 8, 60, 3
    worlds = [make_case("kitti", 40 + s, 0.5)[0] for s in range(B)]
    prm = make_case("kitti", 40, 0.5)[1]
    W, H = worlds[0].W, worlds[0].H
    pitch = ((W + 63) // 64) * 64
    dev = torch.zeros((B, n, 2, H, pitch), dtype=torch.uint8, device="cuda")
    frames = [[worlds[s].render_stereo(i) for i in range(n)] for s in range(B)]
    for s in range(B):
        for i in range(n):
            dev[s, i, 0, :, :W] = torch.from_numpy(frames[s][i][0]).cuda(); dev[s, i, 1, :, :W] = torch.from_numpy(frames[s][i][1]).cuda()
    torch.cuda.synchronize()
    batch = hip_lib.LvtBatch(prm, B)
    got, inflight = [], 0
    for i in range(n):
        batch.track_device_async([dev[s, i, 0].data_ptr() for s in range(B)], [dev[s, i, 1].data_ptr() for s in range(B)], H, W, pitch)
        inflight += 1
        if inflight >= depth:
            got.append(batch.wait()); inflight -= 1
    while inflight:
        got.append(batch.wait()); inflight -= 1
    assert batch.last_error() == ""
    for s in range(B):
        orc = O.Oracle(prm, 1)
        for i in range(n):
            Ro, to = orc.track(*frames[s][i])
            Rb, tb, st = got[i]
            e_t, e_R = pose_errors(Rb[s], tb[s], Ro, to)
            assert e_t <= POSE_TOL and e_R <= POSE_TOL and st[s] == orc.status, f"sequence {s} frame {i}: {e_t:.2e} {e_R:.2e}"
        co, ch = orc.counts(), batch.counts(s)
        bad = {k: (ch.get(k), v) for k, v in co.items() if ch.get(k) != v}
        assert not bad, f"sequence {s}: counters (hip, oracle) {bad}"


def test_lm_rejections_inside_a_tracked_sequence(hip_lib, oracle_lib):
    """Benign scenes never take the hard branch of g2o's Levenberg-Marquardt (no trial of kitti_jump / kitti_half is ever rejected).  This one
    does while TRACKING: after five frames the camera jumps eight frames at a time, then runs backwards -- the motion model's prior is far
    off, trials are REJECTED (rho <= 0: lambda *= ni, pop(), stale edge errors in front of the 5.991 gate) and passes end in Terminate.  The
    trial / rejection / Terminate counters are part of every frame's diff; here they must also be non-zero."""
    world, prm, sensor = make_case("kitti", 0, 0.5)
    frames = list(range(5)) + [13, 21, 29, 37, 45, 44, 43, 42, 41]
    from oracle import pyoracle as O
    orc = O.Oracle(prm, 1); hip = hip_lib.LvtSystem.create(prm, 1)
    rej = term = 0
    for i in frames:
        res, _, _ = run_sequence(world, prm, sensor, [i], hip=hip, orc=orc)
        assert not res[0][1], f"frame {i}: {res[0][1][:6]}"
        c = hip.counts()
        assert c["pnp_trials"] >= c["pnp_iters"] >= 0
        rej += c["pnp_rejections"]; term += c["pnp_terminates"]
    assert hip.get_state() == 2
    assert rej > 0 and term > 0, (rej, term)


def test_pooled_handles_equal_the_oracle(hip_lib, oracle_lib):
    """POOLED handles (lvt_amd_create_pooled: seats of ONE shared lock-step launch chain per device, fed by a submission thread) driven by three host
    threads with three different sequences through the reference's own entry point (lvt_track, host buffers).  Every frame of every handle is held
    to its own ORACLE instance with the full stage-by-stage diff -- key points, descriptors, match indices, row pairs, map, staged, counters,
    predicted pose, pose -- while the other handles' frames share its steps or leave its seat empty (a step a handle sits out must change nothing
    of its state, not even the frame counter)."""
    import threading
    from oracle import pyoracle as O
    from parity_util import pose_errors
    cases = [make_case("kitti", 50 + k, 0.5) for k in range(3)]
    prm = cases[0][1]
    hs = [hip_lib.LvtSystem.create(prm, 1, pooled=True) for _ in range(3)]
    assert all(h.ordering() == "pooled" for h in hs)
    orcs = [O.Oracle(prm, 1) for _ in range(3)]
    frames = [[cases[k][0].render_stereo(i) for i in range(22)] for k in range(3)]
    fails = [[] for _ in range(3)]

    def work(k):
        n = (22, 15, 18)[k]                       # the handles stop at different frames: the later steps run with empty seats
        for i in range(n):
            L, R = frames[k][i]
            Ro, to = orcs[k].track(L, R)
            Rh, th = hs[k].track(L, R)
            msgs = diff_frame(hs[k], orcs[k])
            e_t, e_R = pose_errors(Rh, th, Ro, to)
            if e_t > POSE_TOL or e_R > POSE_TOL:
                msgs.append(f"pose e_t={e_t:.3e} e_R={e_R:.3e}")
            if msgs:
                fails[k].append((i, msgs[:4]))
                return
    ths = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in ths: t.start()
    for t in ths: t.join()
    assert fails == [[], [], []], fails
    assert [h.counts()["frame"] for h in hs] == [21, 14, 17]            # every handle counted ITS frames, not the pool's steps
    st = hs[0].host_stats()
    assert st["collected"] >= 22 and st["event_ordering"] == 2, st       # (out[1]: steps the pool launched; out[7] == 2: pooled)
    for h in hs: h.close()
    # the pool is gone with its last handle; a new one starts from scratch (identity first frame)
    h = hip_lib.LvtSystem.create(prm, 1, pooled=True)
    R0, t0 = h.track(*frames[0][0])
    assert np.array_equal(R0, np.eye(3)) and np.array_equal(t0, np.zeros(3)) and h.get_state() == 2
    h.close()


@pytest.mark.parametrize("how", ["LVT_AMD_POOL=1", "automatic"])
def test_all_five_reference_symbols_on_pooled_handles(hip_lib, oracle_lib, tmp_path, monkeypatch, how):
    """LVT_AMD_POOL=1, or (round 6) no environment variable at all -- two handles created before either has tracked become seats by themselves:
    every handle lvt_create returns is a seat of the device's shared chain -- and the reference's five entry points
    (lvt_c.h:57-62: lvt_create, lvt_track, lvt_track_with_external_corners, lvt_get_status, lvt_destroy) all work on it.  Two handles side by side,
    one alternating lvt_track and lvt_track_with_external_corners (the corner lists ride the frame's step), the other plain lvt_track: every frame
    of both against its own oracle with the full stage diff."""
    import threading
    from oracle import pyoracle as O
    from parity_util import pose_errors
    cases = [make_case("kitti", 70 + k, 0.5) for k in range(2)]
    prm = cases[0][1]
    prm.write_yaml(str(tmp_path / "vo_config.yaml"))
    if how == "LVT_AMD_POOL=1":
        monkeypatch.setenv("LVT_AMD_POOL", "1")
    else:
        monkeypatch.delenv("LVT_AMD_POOL", raising=False)
        monkeypatch.delenv("LVT_AMD_AUTO_POOL", raising=False)
    hs = [hip_lib.LvtSystem.create_from_file(str(tmp_path / "vo_config.yaml"), 1) for _ in range(2)]   # lvt_create
    assert all(h.ordering() == "pooled" for h in hs)
    fprm = hip_lib.LvtParameters.from_file(str(tmp_path / "vo_config.yaml"))
    orcs = [O.Oracle(fprm, 1) for _ in range(2)]
    frames = [[cases[k][0].render_stereo(i) for i in range(12)] for k in range(2)]
    fails = [[], []]
    rng = np.random.default_rng(11)

    def work(k):
        for i in range(12):
            L, R = frames[k][i]
            if k == 0 and i % 2 == 1:   # external corners: the oracle's own detections, some moved to fractional positions
                xl, _, _, _ = O.compute_features(L, fprm)
                xr, _, _, _ = O.compute_features(R, fprm)
                cl = xl.astype(np.float64); cr = xr.astype(np.float64)
                cl[::7] += rng.uniform(-0.5, 0.5, size=cl[::7].shape)
                cr[::5] += 0.5
                Ro, to = orcs[k].track_with_external_corners(L, R, cl, cr)
                Rh, th = hs[k].track_with_external_corners(L, R, cl, cr)     # lvt_track_with_external_corners
            else:
                Ro, to = orcs[k].track(L, R)
                Rh, th = hs[k].track(L, R)                                   # lvt_track
            msgs = diff_frame(hs[k], orcs[k])
            e_t, e_R = pose_errors(Rh, th, Ro, to)
            if e_t > POSE_TOL or e_R > POSE_TOL:
                msgs.append(f"pose e_t={e_t:.3e} e_R={e_R:.3e}")
            if hs[k].get_state() != orcs[k].status:                          # lvt_get_status
                msgs.append(f"status {hs[k].get_state()} != {orcs[k].status}")
            if hs[k].last_error():
                msgs.append(hs[k].last_error())
            if msgs:
                fails[k].append((i, msgs[:4]))
                return
    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ths: t.start()
    for t in ths: t.join()
    assert fails == [[], []], fails
    # too many corners: cut, and SAID so (never silent) -- on a pooled handle as on a solo one
    L, R = frames[0][0]
    many = np.stack(np.meshgrid(np.arange(30, 230, dtype=np.float64), np.arange(30, 130, dtype=np.float64)), -1).reshape(-1, 2)   # 20 000 > 16 384
    hs[0].track_with_external_corners(L, R, many, many[:100])
    err = hs[0].last_error()                    # (the cut is reported at the call, the feature-slot overflow of the 16 384 corners behind it: either way not silent)
    assert "only the first" in err or "capacity" in err, err
    for h in hs: h.close()                                                   # lvt_destroy


def test_automatic_seats_only_for_handles_created_together(hip_lib, monkeypatch, tmp_path):
    """round 6: a lone handle keeps a launch chain of its own (and keeps it when a second handle appears after it has tracked: a tracker's state is not
    moved between chains); handles created before any of them has tracked share the device's pool; LVT_AMD_AUTO_POOL=0 leaves every handle solo; RGB-D
    handles are never seats.  Poses of automatic seats equal a solo handle's, frame for frame."""
    monkeypatch.delenv("LVT_AMD_POOL", raising=False)
    monkeypatch.delenv("LVT_AMD_AUTO_POOL", raising=False)
    world, prm, sensor = make_case("kitti", 41, 0.5)
    frames = [world.render_stereo(i) for i in range(6)]
    cfg = str(tmp_path / "vo_config.yaml")
    prm.write_yaml(cfg)
    mk = lambda: hip_lib.LvtSystem.create_from_file(cfg, 1)     # lvt_create, the reference's call: the one that hands out automatic seats
    a = mk()
    assert a.ordering() != "pooled"
    ref = [a.track(*f) for f in frames]
    b = mk()                                              # `a` has tracked: both stay on chains of their own
    assert a.ordering() != "pooled" and b.ordering() != "pooled"
    a.close(); b.close()
    hs = [mk() for _ in range(3)]                         # created together: the first two convert when the second appears, the third joins
    assert [h.ordering() for h in hs] == ["pooled"] * 3
    x1 = hip_lib.LvtSystem.create(prm, 1); x2 = hip_lib.LvtSystem.create(prm, 1)   # the extension's create calls hand out what was asked for
    assert x1.ordering() != "pooled" and x2.ordering() != "pooled"
    x1.close(); x2.close()
    for k, h in enumerate(hs):
        got = [h.track(*f) for f in frames[:3 + k]]
        for (Rg, tg), (Rr, tr) in zip(got, ref):
            assert np.array_equal(Rg, Rr) and np.array_equal(tg, tr)
        assert h.last_error() == ""
    late = mk()                                           # the pool exists: a later handle takes a seat as well
    assert late.ordering() == "pooled"
    twr, prm_t, _ = make_case("tum", 3, 0.5)
    cfg_t = str(tmp_path / "tum.yaml")
    prm_t.write_yaml(cfg_t)
    d1 = hip_lib.LvtSystem.create_from_file(cfg_t, 2); d2 = hip_lib.LvtSystem.create_from_file(cfg_t, 2)
    assert d1.ordering() != "pooled" and d2.ordering() != "pooled"
    for h in hs + [late, d1, d2]: h.close()
    monkeypatch.setenv("LVT_AMD_AUTO_POOL", "0")
    hs = [mk() for _ in range(2)]
    assert all(h.ordering() != "pooled" for h in hs)
    for h in hs: h.close()


@pytest.mark.parametrize("ordering", ["events", "polling"])
def test_a_seat_without_frames_is_skipped_under_either_ordering(hip_lib, oracle_lib, monkeypatch, ordering):
    """two seats of one pool used ONE AFTER THE OTHER: while the first tracks, the second has no frame in any step (absent), then the roles swap.  Every
    frame of both equals its oracle -- with the polling gates and (round 6: the tracking chain's prologue finds the feature stage's "published without
    features" word itself) with event ordering, the mode a pool falls back to under tools that serialise dispatches"""
    from oracle import pyoracle as O
    monkeypatch.setenv("LVT_AMD_ORDERING", ordering)
    world, prm, sensor = make_case("kitti", 14, 0.5)
    frames = [world.render_stereo(i) for i in range(8)]
    a = hip_lib.LvtSystem.create(prm, 1, pooled=True); b = hip_lib.LvtSystem.create(prm, 1, pooled=True)
    for name, h in (("first", a), ("second", b)):
        orc = O.Oracle(prm, 1)
        for i, (L, R) in enumerate(frames):
            orc.track(L, R); h.track(L, R)
            msgs = diff_frame(h, orc)
            assert not msgs, f"{ordering}: {name} seat, frame {i}: {msgs[:5]}"
        assert h.get_state() == 2 and h.last_error() == ""
    a.close(); b.close()


def test_a_pooled_handle_takes_more_frames_than_its_queue_before_the_first_wait(hip_lib):
    """one thread enqueues SEVEN asynchronous frames on a pooled handle before it collects the first (the queue bound is four frames deposited or in
    flight: results not read yet must not count, or the caller waits for itself); the poses equal a solo handle's, frame for frame"""
    import threading
    import torch
    world, prm, sensor = make_case("kitti", 33, 0.5)
    n = 14
    H, W = world.H, world.W
    pitch = ((W + 63) // 64) * 64
    fr = torch.zeros((n, 2, H, pitch), dtype=torch.uint8, device="cuda")
    for i in range(n):
        fr[i, :, :, :W] = world.render_stereo_torch(i, device="cuda")
    torch.cuda.synchronize()
    solo = hip_lib.LvtSystem.create(prm, 1)
    ref = [solo.track_device(fr[i, 0].data_ptr(), fr[i, 1].data_ptr(), H, W, pitch) for i in range(n)]
    solo.close()
    h = hip_lib.LvtSystem.create(prm, 1, pooled=True)
    got = []

    def work():
        for i in range(7):
            h.track_device_async(fr[i, 0].data_ptr(), fr[i, 1].data_ptr(), H, W, pitch)
        for i in range(7, n):
            got.append(h.wait())
            h.track_device_async(fr[i, 0].data_ptr(), fr[i, 1].data_ptr(), H, W, pitch)
        for _ in range(7):
            got.append(h.wait())
    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(timeout=60)
    assert not t.is_alive(), "the depositor waits for its own lvt_amd_wait (deadlock)"
    assert len(got) == n
    for i in range(n):
        assert np.array_equal(got[i][0], ref[i][0]) and np.array_equal(got[i][1], ref[i][1]), f"frame {i}"
    assert h.last_error() == ""
    h.close()


def test_pooled_async_handles_equal_a_solo_handle(hip_lib):
    """four pooled handles fed the SAME device-resident sequence asynchronously (three frames in flight each) from four threads that start and stop
    at different times: every handle's poses equal a solo handle's, bit for bit"""
    import threading
    import time
    import torch
    world, prm, sensor = make_case("kitti", 60, 0.5)
    n = 48
    pitch = ((world.W + 63) // 64) * 64
    dev = torch.zeros((n, 2, world.H, pitch), dtype=torch.uint8, device="cuda")
    for i in range(n):
        L, R = world.render_stereo(i)
        dev[i, 0, :, :world.W] = torch.from_numpy(L).cuda(); dev[i, 1, :, :world.W] = torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    solo = hip_lib.LvtSystem.create(prm, 1)
    ref = [solo.track_device(dev[i].data_ptr(), dev[i].data_ptr() + world.H * pitch, world.H, world.W, pitch) for i in range(n)]
    solo.close()
    hs = [hip_lib.LvtSystem.create(prm, 1, pooled=True) for _ in range(4)]
    got = [[] for _ in range(4)]

    def work(k):
        time.sleep(0.002 * k)                     # staggered starts: the first steps run with empty seats
        m = n - 7 * k                             # ... and staggered ends
        inflight = 0
        for i in range(m):
            p = dev[i].data_ptr()
            hs[k].track_device_async(p, p + world.H * pitch, world.H, world.W, pitch)
            inflight += 1
            if inflight >= 3:
                got[k].append(hs[k].wait_status()); inflight -= 1
        while inflight:
            got[k].append(hs[k].wait_status()); inflight -= 1
    ths = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ths: t.start()
    for t in ths: t.join()
    for k in range(4):
        assert len(got[k]) == n - 7 * k
        for i, (R, t, st) in enumerate(got[k]):
            assert st == 2 and np.array_equal(t, ref[i][1]) and np.array_equal(R, ref[i][0]), f"handle {k} frame {i}"
        assert hs[k].last_error() == ""
    st = hs[0].host_stats()
    # (out[2] / out[1]: on average clearly more than one seat of a step was taken -- four Python threads feed the pool through the GIL, and the faster the
    #  chain the fewer frames are waiting when a step is folded: 1.6 - 1.9 in rounds 4 - 5, 1.4 with round 6's list kernels)
    assert st["planes_in_place"] > 1.2 * st["collected"], st
    for h in hs: h.close()


@pytest.mark.parametrize("depth", [1, 3], ids=["one_in_flight", "three_in_flight"])
def test_a_gate_time_out_moves_the_handle_to_event_ordering(hip_lib, monkeypatch, depth):
    """the polling gates are a bet on streams that run side by side; when one runs into its time limit (a tool serialising the dispatches,
    streams sharing a hardware queue) the frame is still tracked -- the late kernels do the early stream's share -- and the handle moves to
    event ordering by itself, between two frames, with frames of both kinds in flight.  LVT_AMD_TEST_GATE_TIMEOUT=6 makes the early gate of
    frame 6 report a time-out: every pose must equal the untouched pipeline's, the change-over must be visible (ordering(), last_error())."""
    import torch
    world, prm, sensor = make_case("kitti", 16, 0.5)
    n = 20
    pitch = ((world.W + 63) // 64) * 64
    dev = torch.zeros((n, 2, world.H, pitch), dtype=torch.uint8, device="cuda")
    for i in range(n):
        L, R = world.render_stereo(i)
        dev[i, 0, :, :world.W] = torch.from_numpy(L).cuda(); dev[i, 1, :, :world.W] = torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    monkeypatch.setenv("LVT_AMD_ORDERING", "polling")
    a = hip_lib.LvtSystem.create(prm, 1)
    ref = [a.track_device(dev[i].data_ptr(), dev[i].data_ptr() + world.H * pitch, world.H, world.W, pitch) for i in range(n)]
    monkeypatch.delenv("LVT_AMD_ORDERING")
    a.close()
    monkeypatch.setenv("LVT_AMD_TEST_GATE_TIMEOUT", "6")
    b = hip_lib.LvtSystem.create(prm, 1)
    assert b.ordering() == "polling"
    got, inflight, seen_switch, said = [], 0, None, []
    for i in range(n):
        p = dev[i].data_ptr()
        b.track_device_async(p, p + world.H * pitch, world.H, world.W, pitch); inflight += 1
        if seen_switch is None and b.ordering() == "events":
            seen_switch = i
        if inflight >= depth:
            got.append(b.wait()); inflight -= 1
            said.append(b.last_error())
    while inflight:
        got.append(b.wait()); inflight -= 1
        said.append(b.last_error())
    for i, ((Ra, ta), (Rb, tb)) in enumerate(zip(ref, got)):
        assert np.array_equal(ta, tb) and np.array_equal(Ra, Rb), f"frame {i} (ordering changed at frame {seen_switch})"
    # frame 6 (1-based) is collected after `depth` more frames were enqueued; the frame enqueued next is the first one under event ordering
    assert seen_switch is not None and 6 <= seen_switch <= 6 + depth, seen_switch
    assert b.ordering() == "events" and any("moves to event ordering" in e for e in said), said
    assert b.counts() == a_counts_of(hip_lib, prm, dev, n, world, pitch)
    assert b.get_state() == 2


def a_counts_of(hip_lib, prm, dev, n, world, pitch):
    v = hip_lib.LvtSystem.create(prm, 1)
    for i in range(n):
        v.track_device(dev[i].data_ptr(), dev[i].data_ptr() + world.H * pitch, world.H, world.W, pitch)
    c = v.counts()
    v.close()
    return c


def test_event_ordering_mode_equals_gated_pipeline(hip_lib, monkeypatch):
    """LVT_AMD_ORDERING=events (event barriers only, no early stream, no polling gates: the mode for tools that serialise
    kernel dispatches) tracks exactly like the default three-stream pipeline, several frames in flight"""
    import torch
    world, prm, sensor = make_case("kitti", 14, 0.5)
    n = 20
    pitch = ((world.W + 63) // 64) * 64
    dev = torch.zeros((n, 2, world.H, pitch), dtype=torch.uint8, device="cuda")
    for i in range(n):
        L, R = world.render_stereo(i)
        dev[i, 0, :, :world.W] = torch.from_numpy(L).cuda(); dev[i, 1, :, :world.W] = torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    monkeypatch.setenv("LVT_AMD_AUTO_POOL", "0")      # two chains side by side are the point here (created together they would become seats of one)
    a = hip_lib.LvtSystem.create(prm, 1)
    monkeypatch.setenv("LVT_AMD_ORDERING", "events")  # read by lvt_create
    b = hip_lib.LvtSystem.create(prm, 1)
    monkeypatch.delenv("LVT_AMD_ORDERING")
    out = {}
    for name, vo in (("gated", a), ("events", b)):
        got, inflight = [], 0
        for i in range(n):
            p = dev[i].data_ptr()
            vo.track_device_async(p, p + world.H * pitch, world.H, world.W, pitch)
            inflight += 1
            if inflight >= 4:
                got.append(vo.wait()); inflight -= 1
        while inflight:
            got.append(vo.wait()); inflight -= 1
        out[name] = got
    for i, ((Ra, ta), (Rb, tb)) in enumerate(zip(out["gated"], out["events"])):
        assert np.array_equal(ta, tb) and np.array_equal(Ra, Rb), f"frame {i}"
    assert a.counts() == b.counts()
    assert a.last_error() == "" and b.last_error() == "" and b.get_state() == 2


_FEWER_QUEUES = r"""
import json, os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, sys.argv[1])
import numpy as np, torch, lvt_amd
from parity_util import make_case
world, prm, sensor = make_case("kitti", 16, 0.5)
n = 16
pitch = ((world.W + 63) // 64) * 64
dev = torch.zeros((n, 2, world.H, pitch), dtype=torch.uint8, device="cuda")
for i in range(n):
    L, R = world.render_stereo(i)
    dev[i, 0, :, :world.W] = torch.from_numpy(L).cuda(); dev[i, 1, :, :world.W] = torch.from_numpy(R).cuda()
torch.cuda.synchronize()
vo = lvt_amd.LvtSystem.create(prm, 1)
out, inflight = [], 0
for i in range(n):
    p = dev[i].data_ptr()
    vo.track_device_async(p, p + world.H * pitch, world.H, world.W, pitch); inflight += 1
    if inflight >= 4:
        out.append(vo.wait()); inflight -= 1
while inflight:
    out.append(vo.wait()); inflight -= 1
print("RESULT " + json.dumps({"t": [t.tolist() for _, t in out], "R": [R.tolist() for R, _ in out], "err": vo.last_error(), "state": vo.get_state()}))
"""


@pytest.mark.parametrize("queues", ["1", "2"])
def test_fewer_hardware_queues_than_streams(hip_lib, queues):
    """Every polling gate waits only for work the host enqueued BEFORE it, so multiplexing the three streams onto fewer
    hardware queues (GPU_MAX_HW_QUEUES=1 / 2: in-order execution of interleaved streams) can neither deadlock nor time
    out, and the poses are those of the default configuration"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for name, env_extra in (("default", {}), ("fewer", {"GPU_MAX_HW_QUEUES": queues})):
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", _FEWER_QUEUES, root], env=env, capture_output=True, text=True, timeout=240)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        assert r.returncode == 0 and line, r.stderr[-2000:]
        runs[name] = json.loads(line[-1][7:])
    assert runs["fewer"]["err"] == "" and runs["fewer"]["state"] == 2
    assert runs["fewer"]["t"] == runs["default"]["t"] and runs["fewer"]["R"] == runs["default"]["R"]


@pytest.mark.parametrize("pieces", ["auto", "1", "2", "3"])
def test_lockstep_batch_equals_independent_handles(hip_lib, monkeypatch, pieces):
    """B sequences through ONE launch chain (lvt_amd_batch_*) == B independent handles, pose for pose -- whether the batch's k_score goes out as
    one launch or as several (LVT_AMD_SCORE_PIECES; unset, the handle chooses by itself from its early gate's waits)"""
    import torch
    if pieces != "auto":
        monkeypatch.setenv("LVT_AMD_SCORE_PIECES", pieces)
    B, n = 3, 10
    worlds = [make_case("kitti", 20 + s, 0.5)[0] for s in range(B)]
    prm = make_case("kitti", 20, 0.5)[1]
    W, H = worlds[0].W, worlds[0].H
    pitch = ((W + 63) // 64) * 64
    dev = torch.zeros((B, n, 2, H, pitch), dtype=torch.uint8, device="cuda")
    for s in range(B):
        for i in range(n):
            L, R = worlds[s].render_stereo(i)
            dev[s, i, 0, :, :W] = torch.from_numpy(L).cuda(); dev[s, i, 1, :, :W] = torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    singles = [hip_lib.LvtSystem.create(prm, 1) for _ in range(B)]
    batch = hip_lib.LvtBatch(prm, B)
    for i in range(n):
        lp = [dev[s, i, 0].data_ptr() for s in range(B)]; rp = [dev[s, i, 1].data_ptr() for s in range(B)]
        batch.track_device_async(lp, rp, H, W, pitch)
        Rb, tb, st = batch.wait()
        for s in range(B):
            Rs, ts = singles[s].track_device(lp[s], rp[s], H, W, pitch)
            assert np.array_equal(ts, tb[s]) and np.array_equal(Rs, Rb[s]), f"sequence {s} frame {i}"
            assert st[s] == 2
    for s in range(B):
        assert batch.counts(s) == singles[s].counts()
    assert batch.last_error() == ""


def test_lockstep_batch_survives_a_gate_time_out(hip_lib, monkeypatch):
    """the change-over to event ordering (test_a_gate_time_out_moves_the_handle_to_event_ordering) in a lock-step batch with three frames in flight:
    poses of every sequence equal those of an untouched batch"""
    import torch
    B, n = 3, 14
    worlds = [make_case("kitti", 30 + s, 0.5)[0] for s in range(B)]
    prm = make_case("kitti", 30, 0.5)[1]
    W, H = worlds[0].W, worlds[0].H
    pitch = ((W + 63) // 64) * 64
    dev = torch.zeros((B, n, 2, H, pitch), dtype=torch.uint8, device="cuda")
    for s in range(B):
        for i in range(n):
            L, R = worlds[s].render_stereo(i)
            dev[s, i, 0, :, :W] = torch.from_numpy(L).cuda(); dev[s, i, 1, :, :W] = torch.from_numpy(R).cuda()
    torch.cuda.synchronize()

    def run(batch):
        out, inflight = [], 0
        for i in range(n):
            lp = [dev[s, i, 0].data_ptr() for s in range(B)]; rp = [dev[s, i, 1].data_ptr() for s in range(B)]
            batch.track_device_async(lp, rp, H, W, pitch); inflight += 1
            if inflight >= 3:
                out.append(batch.wait()); inflight -= 1
        while inflight:
            out.append(batch.wait()); inflight -= 1
        return out
    ref = run(hip_lib.LvtBatch(prm, B))
    monkeypatch.setenv("LVT_AMD_TEST_GATE_TIMEOUT", "5")
    b = hip_lib.LvtBatch(prm, B)
    got = run(b)
    for i, ((Ra, ta, sa), (Rb, tb, sb)) in enumerate(zip(ref, got)):
        assert np.array_equal(ta, tb) and np.array_equal(Ra, Rb) and np.array_equal(sa, sb), f"lock-step frame {i}"
    assert hip_lib.load_library().lvt_amd_get_ordering(b._h) == 1


def test_batch_cells_with_the_small_lds_instance(hip_lib, monkeypatch):
    """a batch with more cell workgroups than CUs runs k_cells with room for 4 700 raw corners (two workgroups per CU) and sends a cell beyond that down
    the exact global-memory path.  Forced here on a small batch (LVT_AMD_CELLS_RAW_CAP) over full-size frames whose texture contrast grows from frame to
    frame, so that cells cross 4 700 (and 10 240) raw corners on the way: features, counts and poses must equal those of independent handles, which
    run the full-size instance."""
    import torch
    B, n = 2, 9
    world, prm, _ = make_case("kitti", 40, 1.0)
    W, H = world.W, world.H
    pitch = ((W + 63) // 64) * 64
    rng = np.random.default_rng(5)
    base = [rng.integers(0, 256, (H, W)).astype(np.float32) for _ in range(B)]
    dev = torch.zeros((B, n, 2, H, pitch), dtype=torch.uint8, device="cuda")
    for s in range(B):
        for i in range(n):
            amp = 0.30 + 0.07 * i + 0.02 * s                   # contrast of the noise texture around mid-gray: ~1 400 ... 11 000 raw corners in a 250 x 250 cell
            L = np.clip(128.0 + amp * (base[s] - 128.0), 0, 255).astype(np.uint8)
            R = np.roll(L, -8, axis=1)
            dev[s, i, 0, :, :W] = torch.from_numpy(L).cuda(); dev[s, i, 1, :, :W] = torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    singles = [hip_lib.LvtSystem.create(prm, 1) for _ in range(B)]
    monkeypatch.setenv("LVT_AMD_CELLS_RAW_CAP", "4700")
    batch = hip_lib.LvtBatch(prm, B)
    raw_seen = []
    for i in range(n):
        lp = [dev[s, i, 0].data_ptr() for s in range(B)]; rp = [dev[s, i, 1].data_ptr() for s in range(B)]
        batch.track_device_async(lp, rp, H, W, pitch)
        Rb, tb, st = batch.wait()
        for s in range(B):
            Rs, ts = singles[s].track_device(lp[s], rp[s], H, W, pitch)
            assert np.array_equal(ts, tb[s]) and np.array_equal(Rs, Rb[s]) and st[s] == singles[s].get_state(), f"sequence {s} frame {i}"
            assert batch.counts(s) == singles[s].counts(), f"sequence {s} frame {i}: {batch.counts(s)} != {singles[s].counts()}"
        raw_seen.append(int(singles[0].timeline()[7]) & 0xFFFF)   # raw corners of cell 0 / left image of sequence 0
    assert min(raw_seen) < 4700 and any(4700 < r <= 10240 for r in raw_seen) and max(raw_seen) > 10240, raw_seen
    assert batch.last_error() == ""


def test_odometry_update_follows_the_tracker(hip_lib):
    """lvt_amd_odometry_update = lvt_track + the node's pose handling: compared with the same arithmetic applied to the poses
    of a second, plain handle; a LOST frame (40-frame jump with a tiny search radius) resets the tracker and publishes nothing"""
    from test_odometry import NodeModel, _same_rotation
    world, prm, sensor = make_case("kitti", 21, 0.5)
    a = hip_lib.LvtSystem.create(prm, 1); b = hip_lib.LvtSystem.create(prm, 1)
    od, ref = hip_lib.Odometry(a, None, True), NodeModel(None, True)
    frames = list(range(10)) + list(range(60, 66))
    published = 0
    for k, i in enumerate(frames):
        L, R = world.render_stereo(i)
        got = od.update(L, R, 0.1 * k)
        Rb, tb = b.track(L, R)
        st = b.get_state()
        exp = ref.push(Rb, tb, st, 0.1 * k)
        if st == 3:
            b.reset()
        assert (got is None) == (exp is None), f"frame {i}"
        if got is not None:
            published += 1
            assert np.allclose(got[0][:3], exp[0][:3], atol=1e-9) and _same_rotation(got[0][3:], exp[0][3:])
            assert np.allclose(got[1], exp[1], atol=1e-6)
    assert published >= 10 and a.last_error() == ""


_UNDER_PMC = r"""
import json, os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, sys.argv[1])
import numpy as np, lvt_amd
from parity_util import make_case
world, prm, sensor = make_case("kitti", 16, 0.5)
vo = lvt_amd.LvtSystem.create(prm, 1)
out = []
for i in range(8):
    L, R = world.render_stereo(i)
    Rm, t = vo.track(L, R)
    out.append({"t": t.tolist(), "state": vo.get_state(), "err": vo.last_error(), "ordering": vo.ordering()})
print("RESULT " + json.dumps(out))
"""


def test_serialised_dispatches_are_reported_not_silently_wrong(hip_lib, tmp_path):
    """rocprofv3 --pmc serialises the dispatches of all queues: a polling gate then holds the only dispatch slot.  In the default
    ordering the library must say so (error string) and must never turn a time-out into LOST (sticky; in the reference a matter of
    match counts only): a frame whose features never arrived is SKIPPED -- last pose, state kept.  Until the first reported problem
    every pose is the right one; with LVT_AMD_ORDERING=events the same run is clean."""
    import json, os, shutil, subprocess, sys
    if not shutil.which("rocprofv3"):
        pytest.skip("rocprofv3 not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for name, extra in (("plain", None), ("pmc_default", {}), ("pmc_events", {"LVT_AMD_ORDERING": "events"})):
        env = dict(os.environ); env["TMPDIR"] = str(tmp_path)
        cmd = [sys.executable, "-c", _UNDER_PMC, root]
        if extra is not None:
            env.update(extra)
            cmd = ["rocprofv3", "--pmc", "SQ_WAVES", "-d", str(tmp_path / name), "-o", "x", "--"] + cmd
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
        except subprocess.TimeoutExpired:
            if extra is None:
                raise
            pytest.skip("rocprofv3 --pmc did not finish in time on this box")
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line and extra is not None:  # the profiler itself failed (not what this test is about)
            pytest.skip("rocprofv3 --pmc produced no run: " + (r.stderr[-300:] or r.stdout[-300:]))
        assert line, (name, r.stdout[-1500:], r.stderr[-1500:])
        runs[name] = json.loads(line[-1][7:])
    ref = runs["plain"]
    assert all(f["state"] == 2 and f["err"] == "" for f in ref)
    assert [f["t"] for f in runs["pmc_events"]] == [f["t"] for f in ref] and all(f["err"] == "" for f in runs["pmc_events"])
    saw_report = False
    for f, g in zip(runs["pmc_default"], ref):
        saw_report |= f["err"] != ""
        assert f["state"] != 3, "a time-out must never cost the track"
        if not saw_report:
            assert f["t"] == g["t"]                      # nothing reported so far: the pose is the right pose
    assert saw_report
    # round 3: a gate that times out is a benign event (the late kernels do the early stream's share) and moves the handle to event ordering:
    # unless a 2-s wait SKIPPED a frame before that, every pose of the run is the right one and the run ends ordered by events, without complaints
    d = runs["pmc_default"]
    if not any("SKIPPED" in f["err"] for f in d):
        assert [f["t"] for f in d] == [f["t"] for f in ref]
        assert d[-1]["ordering"] == "events" and d[-1]["err"] == "", d[-1]


def test_lockstep_batch_with_one_sequence_lost(hip_lib):
    """sequences of one batch are independent state machines: one of them sees a nearly empty frame (second pass, then LOST
    and the sticky last pose) while the other keeps tracking -- both equal their stand-alone handles, frame for frame"""
    import torch
    worlds = [make_case("kitti", 30, 0.5)[0], make_case("kitti", 31, 0.5)[0]]
    prm = make_case("kitti", 30, 0.5, {"min_num_matches_for_tracking": 60})[1]
    n = 10
    W, H = worlds[0].W, worlds[0].H
    pitch = ((W + 63) // 64) * 64
    dev = torch.zeros((2, n, 2, H, pitch), dtype=torch.uint8, device="cuda")
    for s in range(2):
        for k in range(n):
            L, R = sparse_pair(worlds[s]) if (s == 1 and k == 5) else worlds[s].render_stereo(k)
            dev[s, k, 0, :, :W] = torch.from_numpy(L).cuda(); dev[s, k, 1, :, :W] = torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    singles = [hip_lib.LvtSystem.create(prm, 1) for _ in range(2)]
    batch = hip_lib.LvtBatch(prm, 2)
    states = []
    for k in range(n):
        lp = [dev[s, k, 0].data_ptr() for s in range(2)]; rp = [dev[s, k, 1].data_ptr() for s in range(2)]
        batch.track_device_async(lp, rp, H, W, pitch)
        Rb, tb, st = batch.wait()
        for s in range(2):
            Rs, ts = singles[s].track_device(lp[s], rp[s], H, W, pitch)
            assert np.array_equal(ts, tb[s]) and np.array_equal(Rs, Rb[s]), f"sequence {s} frame {k}"
            assert st[s] == singles[s].get_state()
        states.append([int(x) for x in st])
    assert states[4] == [2, 2] and states[5] == [2, 3] and states[-1] == [2, 3], states   # LOST is sticky, the other one tracks on
    for s in range(2):
        assert batch.counts(s) == singles[s].counts()
    assert batch.last_error() == ""


@pytest.mark.parametrize("scale", [0.5, 1.0], ids=["half_size", "kitti_1241x376"])
def test_pinned_host_buffers_are_read_in_place(hip_lib, scale):
    """lvt_track with page-locked host images (torch pin_memory = hipHostMalloc) skips the staging copy -- at the headline shape too,
    whose 466 616 bytes per image are not a multiple of the 16-byte vectors the pull kernel loads (it reads the odd head and tail byte
    by byte), and from a base address that is not 16-byte aligned.  Same poses as with ordinary numpy buffers, and the ROUTE is asserted
    (lvt_amd_get_host_stats), not just the result."""
    import torch
    world, prm, sensor = make_case("kitti", 17, scale)
    a = hip_lib.LvtSystem.create(prm, 1); b = hip_lib.LvtSystem.create(prm, 1)
    n = 8
    for i in range(n):
        L, R = world.render_stereo(i)
        pr = torch.from_numpy(R).pin_memory()
        off = (0, 1, 0, 7, 0, 13, 0, 0)[i]   # page-locked views that start 1, 7, 13 bytes behind a 16-byte boundary
        buf = torch.empty(L.size + 16, dtype=torch.uint8).pin_memory()
        buf[off:off + L.size] = torch.from_numpy(L).reshape(-1)
        Lp = buf[off:off + L.size].numpy().reshape(L.shape)
        assert Lp.ctypes.data % 16 == off
        Ra, ta = a.track(L, R)
        Rb, tb = b.track(Lp, pr.numpy())
        assert np.array_equal(ta, tb) and np.array_equal(Ra, Rb), f"frame {i}"
    assert a.counts() == b.counts() and b.last_error() == ""
    if scale == 1.0:
        assert (world.W * world.H) % 16 != 0      # the case that used to fall back to the staging copy
    sa, sb = a.host_stats(), b.host_stats()
    assert sa["planes_in_place"] == 0 and sa["planes_staged"] == 2 * n, sa        # pageable numpy buffers: copied
    assert sb["planes_in_place"] == 2 * n and sb["planes_staged"] == 0, sb        # page-locked: every plane read where it lies


def test_last_error_reports_the_frame_just_tracked(hip_lib):
    """a synchronous call may return on the pose k_pnp hands over, before the frame's full record exists; lvt_amd_last_error must still
    speak for THAT frame when it is called right behind lvt_track (it collects the frame first) -- here: more external corners than the
    feature arrays hold"""
    world, prm, sensor = make_case("kitti", 21, 0.5)
    vo = hip_lib.LvtSystem.create(prm, 1)
    for i in range(3):
        L, R = world.render_stereo(i)
        vo.track(L, R)
        assert vo.last_error() == ""
    L, R = world.render_stereo(3)
    rng = np.random.default_rng(3)
    many = np.stack([rng.uniform(30, world.W - 30, 5000), rng.uniform(30, world.H - 30, 5000)], axis=1)   # 5000 > the 4096 feature slots
    vo.track_with_external_corners(L, R, many, many[:100])
    err = vo.last_error()          # first call after the frame: no counts() / get_state() in between
    assert "capacity" in err or "overflow" in err, err


def test_rgbd_depth_upload_beside_detection(hip_lib):
    """lvt_amd_track_rgbd copies and pulls the depth image only once the detection kernels are enqueued (k_gather is the first to need it):
    pageable buffers (host copy into the staging buffer) and page-locked ones (read in place) must give the poses and maps of each other,
    frame after frame, with a fresh depth image every call"""
    import torch
    world, prm, sensor = make_case("tum", 3, 1.0)
    a = hip_lib.LvtSystem.create(prm, 2); b = hip_lib.LvtSystem.create(prm, 2)
    for i in range(6):
        g, d = world.render_rgbd(i)
        d = np.ascontiguousarray(d, dtype=np.float32)
        gp, dp = torch.from_numpy(g).pin_memory().numpy(), torch.from_numpy(d).pin_memory().numpy()
        Ra, ta = a.track(g, d)
        Rb, tb = b.track(gp, dp)
        assert np.array_equal(ta, tb) and np.array_equal(Ra, Rb), f"frame {i}"
        assert a.counts() == b.counts(), f"frame {i}"
    assert a.get_state() == 2 and a.last_error() == "" and b.last_error() == ""
    assert a.counts()["map_size"] > 1000  # (the depth filter saw real depths: every kept feature became a map point)


def test_featureless_and_saturated_frames(hip_lib, oracle_lib):
    """degenerate inputs through the whole chain, stage by stage against the oracle: a black first frame (no corner, empty map,
    the <200-corner retry on both eyes) followed by a saturated one (no match -> LOST) and one with a single bright pixel (LOST:
    nothing is computed or reported); after a reset: a single-pixel first frame, ordinary frames, a black frame in the middle of
    tracking (LOST again) and an ordinary one behind it"""
    world, prm, sensor = make_case("kitti", 9, 0.5)
    hip = hip_lib.LvtSystem.create(prm, 1)
    orc = oracle_lib.Oracle(prm, 1)
    H, W = world.H, world.W
    black = np.zeros((H, W), np.uint8)
    white = np.full((H, W), 255, np.uint8)
    dot = black.copy(); dot[H // 2, W // 2] = 255

    def run(seq, tag):
        states = []
        for k, (a, b) in enumerate(seq):
            Ro, to = orc.track(a, b); Rh, th = hip.track(a, b)
            msgs = diff_frame(hip, orc)
            assert not msgs, f"{tag} frame {k}: {msgs[:3]}"
            assert hip.get_state() == orc.status, f"{tag} frame {k}"
            assert np.allclose(th, to, atol=1e-9) and np.allclose(Rh, Ro, atol=1e-9), f"{tag} frame {k}"
            states.append(hip.get_state())
        return states

    assert run([(black, black), (white, white), (dot, dot)], "a") == [2, 3, 3]
    hip.reset(); orc.reset()
    st = run([(dot, dot)] + [world.render_stereo(i) for i in range(3)], "b")
    hip.reset(); orc.reset()
    st = run([world.render_stereo(i) for i in range(4)] + [(black, black), world.render_stereo(5)], "c")
    assert st[:4] == [2, 2, 2, 2] and st[4:] == [3, 3]


def test_handles_are_independent_across_host_threads(hip_lib):
    """two handles driven concurrently from two host threads (ctypes releases the GIL inside the library) track exactly like the
    same sequences run one after the other -- the reference's file-scope statics are per instance here (threading)"""
    import threading
    cases = [make_case("kitti", 40, 0.5), make_case("kitti", 41, 0.5)]
    frames = [[w.render_stereo(i) for i in range(14)] for w, _, _ in cases]
    ref = []
    for (w, prm, _), fr in zip(cases, frames):
        vo = hip_lib.LvtSystem.create(prm, 1)
        ref.append([vo.track(L, R) for L, R in fr])
    got = [None, None]
    errs = []

    def work(k):
        try:
            vo = hip_lib.LvtSystem.create(cases[k][1], 1)
            got[k] = [vo.track(L, R) for L, R in frames[k]]
            if vo.last_error():
                errs.append(vo.last_error())
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert not errs, errs
    for k in range(2):
        for i, ((Ra, ta), (Rb, tb)) in enumerate(zip(ref[k], got[k])):
            assert np.array_equal(ta, tb) and np.array_equal(Ra, Rb), f"handle {k} frame {i}"


def test_a_handle_owns_its_device(hip_lib):
    """a handle records the HIP device it was created on and makes it current inside every entry point (one process may
    drive one handle per GPU from any thread).  On a multi-GPU box: two handles on two devices, each driven by a thread whose
    current device is the OTHER one, both equal to a plain run; each is the first live handle of ITS device (polling gates)."""
    import threading
    import torch
    n = torch.cuda.device_count()
    world, prm, sensor = make_case("kitti", 40, 0.5)
    frames = [world.render_stereo(i) for i in range(6)]
    ref = hip_lib.LvtSystem.create(prm, 1)
    assert ref.device() == torch.cuda.current_device()
    want = [ref.track(a, b)[1].copy() for a, b in frames]
    ref.close()
    with pytest.raises(RuntimeError):
        hip_lib.LvtSystem.create(prm, 1, device=n)              # no such device: NULL, no fallback
    devs = [0, 1] if n >= 2 else [0]
    got, errs = {}, {}

    def drive(d):
        if n >= 2:
            torch.cuda.set_device(1 - d)                         # the calling thread's current device is NOT the handle's
        vo = hip_lib.LvtSystem.create(prm, 1, device=d)
        got[d] = [vo.track(a, b)[1].copy() for a, b in frames]
        errs[d] = (vo.device(), vo.ordering(), vo.last_error(), vo.get_state())
        if n >= 2:
            assert torch.cuda.current_device() == 1 - d          # restored after every call
        vo.close()
    ths = [threading.Thread(target=drive, args=(d,)) for d in devs]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for d in devs:
        assert errs[d] == (d, "polling", "", 2), errs[d]
        assert all(np.array_equal(x, y) for x, y in zip(got[d], want))
    if n < 2:
        pytest.skip("one GPU: the two-device half needs a multi-GPU box (the owning-device bookkeeping above ran)")


def test_wrong_image_size_is_refused_and_reported(hip_lib):
    """lvt_track with images of another size: outputs untouched (like the reference on an exception), an error string, and the
    handle keeps working"""
    world, prm, sensor = make_case("kitti", 42, 0.5)
    vo = hip_lib.LvtSystem.create(prm, 1)
    L, R = world.render_stereo(0)
    R0, t0 = vo.track(L, R)
    small = np.ascontiguousarray(L[:-2, :-3])
    R1, t1 = vo.track(small, small)
    assert "size" in vo.last_error()
    R2, t2 = vo.track(*world.render_stereo(1))
    assert vo.get_state() == 2 and np.linalg.norm(t2) > 0


@pytest.mark.parametrize("depth", [1, 2, 3, 7])
def test_async_depths_and_record_delivery(hip_lib, depth):
    """every number of frames in flight the ring allows: the result record of a frame is delivered by its successor's gate, by
    the courier kernel (the last frame of a burst) or by k_triangulate (synchronous calls mixed in) -- always the same poses"""
    import torch
    world, prm, sensor = make_case("kitti", 50, 0.5)
    n = 18
    pitch = ((world.W + 63) // 64) * 64
    dev = torch.zeros((n, 2, world.H, pitch), dtype=torch.uint8, device="cuda")
    for i in range(n):
        L, R = world.render_stereo(i)
        dev[i, 0, :, :world.W] = torch.from_numpy(L).cuda(); dev[i, 1, :, :world.W] = torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    a = hip_lib.LvtSystem.create(prm, 1); b = hip_lib.LvtSystem.create(prm, 1)
    ptr = lambda i: (dev[i].data_ptr(), dev[i].data_ptr() + world.H * pitch)
    ref = [a.track_device(*ptr(i), world.H, world.W, pitch) for i in range(n)]
    got, inflight = [], 0
    for i in range(n):
        if i in (6, 11):                       # a synchronous call in the middle of the asynchronous stream (drains it first)
            while inflight:
                got.append(b.wait()); inflight -= 1
            got.append(b.track_device(*ptr(i), world.H, world.W, pitch))
            continue
        b.track_device_async(*ptr(i), world.H, world.W, pitch); inflight += 1
        if inflight >= depth:
            got.append(b.wait()); inflight -= 1
    while inflight:
        got.append(b.wait()); inflight -= 1
    assert len(got) == n
    for i, ((Ra, ta), (Rb, tb)) in enumerate(zip(ref, got)):
        assert np.array_equal(ta, tb) and np.array_equal(Ra, Rb), f"frame {i}"
    assert a.counts() == b.counts() and b.last_error() == ""


@pytest.mark.parametrize("ordering", ["auto", "polling"])
def test_many_handles_on_one_gpu(hip_lib, monkeypatch, ordering):
    """eight independent handles driven round-robin from one thread, two frames in flight each (several sequences per
    GPU on separate handles): no gate times out, no handle starves another, every pose equals the stand-alone run -- with the
    default choice (the first live handle polls, the ones created beside it order with events) and with polling gates forced
    on all eight (more than four handles: separate one-wave k_gate_late instead of the parked k_match_map workgroups)"""
    import torch
    if ordering == "polling":
        monkeypatch.setenv("LVT_AMD_ORDERING", "polling")  # read by lvt_create
    else:
        monkeypatch.delenv("LVT_AMD_ORDERING", raising=False)
    H = 8
    n = 10
    cases = [make_case("kitti", 60 + k, 0.5) for k in range(H)]
    world0 = cases[0][0]
    pitch = ((world0.W + 63) // 64) * 64
    dev = torch.zeros((H, n, 2, world0.H, pitch), dtype=torch.uint8, device="cuda")
    for k, (w, _, _) in enumerate(cases):
        for i in range(n):
            L, R = w.render_stereo(i)
            dev[k, i, 0, :, :w.W] = torch.from_numpy(L).cuda(); dev[k, i, 1, :, :w.W] = torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    ptr = lambda k, i: (dev[k, i].data_ptr(), dev[k, i].data_ptr() + world0.H * pitch)
    ref = []
    for k in range(H):
        vo = hip_lib.LvtSystem.create(cases[k][1], 1)
        ref.append([vo.track_device(*ptr(k, i), world0.H, world0.W, pitch) for i in range(n)])
        del vo
    vos = [hip_lib.LvtSystem.create(cases[k][1], 1) for k in range(H)]
    modes = [v.ordering() for v in vos]
    assert modes[1:] == (["polling"] * (H - 1) if ordering == "polling" else ["events"] * (H - 1)), modes
    got = [[] for _ in range(H)]
    for i in range(n):
        for k in range(H):
            vos[k].track_device_async(*ptr(k, i), world0.H, world0.W, pitch)
        if i >= 1:
            for k in range(H):
                got[k].append(vos[k].wait())
    for k in range(H):
        got[k].append(vos[k].wait())
    for k in range(H):
        assert vos[k].last_error() == "", (k, vos[k].last_error())
        for i, ((Ra, ta), (Rb, tb)) in enumerate(zip(ref[k], got[k])):
            assert np.array_equal(ta, tb) and np.array_equal(Ra, Rb), f"handle {k} frame {i}"
