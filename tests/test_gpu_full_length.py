"""-m gpu: BASELINE.json's configurations 2 - 4 at their FULL sequence lengths (kitti_example.cpp:113-138 walks the whole sequence:
KITTI 00 has 4 541 stereo pairs, EuRoC MH_01_easy 3 682, TUM fr1/desk 573 RGB-D frames), through the C-ABI against the oracle.

Every frame is held to a LIGHT diff -- all frame counters (key points per eye, matches, PnP inliers / trials, row pairs, map and staged
sizes, triangulation decisions ...), the find_matches feature indices, the row-match pairs, tracking state, pose within POSE_TOL -- and every
CHECK_EVERY-th frame (and the last one) to the full stage-by-stage diff of parity_util.diff_frame (key points, descriptors, map / staged
arrays).  The frames are rendered on the GPU by the torch twin of the numpy renderer (lvt_amd.synth: same math, compared with it below) and
handed to BOTH systems as host arrays, so the 8 796 frames fit the driver's time limit."""
import numpy as np
import pytest

from parity_util import make_case, diff_frame, pose_errors, POSE_TOL

pytestmark = pytest.mark.gpu

FULL = [
    # id, kind, seed, frames (BASELINE.json configs[1..3])
    ("kitti_full_4541", "kitti", 31, 4541),
    ("euroc_3682", "euroc", 32, 3682),
    ("tum_573", "tum", 33, 573),
]
CHECK_EVERY = 400


def light_diff(hip, orc):
    msgs = []
    if hip.last_error():
        msgs.append("hip error: " + hip.last_error())
    co, ch = orc.counts(), hip.counts()
    msgs += [f"count {k}: hip={ch.get(k)} oracle={v}" for k, v in co.items() if ch.get(k) != v]
    if ch.get("overflow"):
        msgs.append(f"overflow mask {ch['overflow']}")
    fo, _ = orc.matches(); fh, _ = hip.matches()
    if not np.array_equal(fh, fo):
        msgs.append("find_matches feature idx differ")
    if not np.array_equal(hip.row_matches(), orc.row_matches()):
        msgs.append("row_match pairs differ")
    if orc.status != hip.get_state():
        msgs.append(f"status hip={hip.get_state()} oracle={orc.status}")
    return msgs


@pytest.mark.parametrize("name,kind,seed,n_frames", FULL, ids=[c[0] for c in FULL])
def test_full_length_config(hip_lib, oracle_lib, name, kind, seed, n_frames):
    import torch
    from oracle import pyoracle as O
    world, prm, sensor = make_case(kind, seed, 1.0)
    orc = O.Oracle(prm, sensor)
    hip = hip_lib.LvtSystem.create(prm, sensor)
    worst_t = worst_R = 0.0
    tracked = 0
    for i in range(n_frames):
        if sensor == 1:
            st = world.render_stereo_torch(i, device="cuda").cpu().numpy()
            a, b = np.ascontiguousarray(st[0]), np.ascontiguousarray(st[1])
            if i == 0:   # the torch renderer IS the numpy renderer
                na, nb = world.render_stereo(0)
                assert np.array_equal(a, na) and np.array_equal(b, nb)
            Ro, to = orc.track(a, b)
        else:
            g, d = world.render_rgbd_torch(i, device="cuda")
            a, b = np.ascontiguousarray(g.cpu().numpy()), np.ascontiguousarray(d.cpu().numpy())
            if i == 0:
                na, nb = world.render_rgbd(0)
                assert np.array_equal(a, na) and np.array_equal(b, nb)
            Ro, to = orc.track_rgbd(a, b)
        Rh, th = hip.track(a, b)
        full = (i % CHECK_EVERY == CHECK_EVERY - 1) or i == n_frames - 1
        msgs = diff_frame(hip, orc) if full else light_diff(hip, orc)
        e_t, e_R = pose_errors(Rh, th, Ro, to)
        assert not msgs, f"{name}: frame {i}: {msgs[:6]}"
        assert e_t <= POSE_TOL and e_R <= POSE_TOL, f"{name}: frame {i}: pose e_t={e_t:.3e} e_R={e_R:.3e}"
        worst_t, worst_R = max(worst_t, e_t), max(worst_R, e_R)
        tracked += hip.get_state() == 2
    c = hip.counts()
    assert c["frame"] == n_frames - 1
    # the synthetic trajectories keep the tracker alive over the whole length (a LOST latch would make the rest of the run vacuous)
    assert hip.get_state() == 2 and tracked >= n_frames - 1, (hip.get_state(), tracked)
    print(f"{name}: {n_frames} frames, worst e_t {worst_t:.3e} e_R {worst_R:.3e}, map size {c['map_size']}")
