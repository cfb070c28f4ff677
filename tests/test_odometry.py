"""Host logic of the odometry accumulator (lvt_amd_odometry_*) against an independent numpy statement of the
reference node's arithmetic (lvt/src/lvt_ros.cpp:86-92, :215-311).  No GPU: push_pose never touches the device."""
import numpy as np
import pytest

import lvt_amd


def _rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])}[axis]


def _quat_xyzw(R):
    # independent route: eigenvector of eigenvalue 1 and the angle from the trace, sign fixed by the skew part
    ang = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
    if ang < 1e-12:
        return np.array([0, 0, 0, 1.0])
    ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(ang))
    return np.concatenate([ax * np.sin(ang / 2), [np.cos(ang / 2)]])


class NodeModel:
    """the ROS node's pose handling with 4x4 homogeneous matrices"""

    def __init__(self, b2s=None, reset_pose=True):
        a = -1.57079632679
        self.fix = _rot("z", a) @ _rot("x", a)
        self.b2s = np.eye(4) if b2s is None else b2s
        self.acc = np.eye(4)
        self.lastR, self.lastp, self.t_last, self.reset_pose = self.fix.copy(), np.zeros(3), None, reset_pose

    def push(self, R, t, status, stamp):
        if self.t_last is not None and self.t_last > stamp:
            return None
        if status == 3:
            if self.reset_pose:
                self.acc, self.lastR, self.lastp = np.eye(4), self.fix.copy(), np.zeros(3)
            return None
        cR, cp = self.fix @ R, self.fix @ t
        d = np.eye(4); d[:3, :3] = cR @ self.lastR.T; d[:3, 3] = cp - self.lastp
        db = self.b2s @ d @ np.linalg.inv(self.b2s)
        self.acc = self.acc @ db
        tw = np.zeros(6)
        if self.t_last is not None and stamp != self.t_last:
            dt = stamp - self.t_last
            tw[:3] = db[:3, 3] / dt
            q = _quat_xyzw(db[:3, :3])
            ang = 2 * np.arccos(np.clip(q[3], -1, 1)); s2 = 1 - q[3] ** 2
            ax = q[:3] / np.sqrt(s2) if s2 >= 10 * np.finfo(float).eps else np.array([1.0, 0, 0])
            tw[3:] = ax * ang / dt
        self.t_last, self.lastR, self.lastp = stamp, cR, cp
        return np.concatenate([self.acc[:3, 3], _quat_xyzw(self.acc[:3, :3])]), tw


def _trajectory(n, seed):
    rng = np.random.default_rng(seed)
    R, t = np.eye(3), np.zeros(3)
    out = []
    for i in range(n):
        R = R @ _rot("y", rng.normal(0, 0.02)) @ _rot("x", rng.normal(0, 0.005)) @ _rot("z", rng.normal(0, 0.005))
        t = t + R @ np.array([rng.normal(0, 0.02), rng.normal(0, 0.01), 0.8 + rng.normal(0, 0.05)])
        out.append((R.copy(), t.copy()))
    return out


def _same_rotation(qa, qb, tol=1e-9):
    return min(np.abs(qa - qb).max(), np.abs(qa + qb).max()) < tol


@pytest.mark.parametrize("with_base", [False, True])
def test_accumulation_matches_the_node_arithmetic(with_base):
    b2s = None
    if with_base:
        b2s = np.eye(4); b2s[:3, :3] = _rot("z", 0.3) @ _rot("y", -0.2); b2s[:3, 3] = [0.4, -0.1, 1.2]
    od = lvt_amd.Odometry(None, None if b2s is None else b2s[:3, :], True)
    ref = NodeModel(b2s, True)
    for i, (R, t) in enumerate(_trajectory(60, 3)):
        got, exp = od.push_pose(R, t, 2, 0.1 * i), ref.push(R, t, 2, 0.1 * i)
        assert got is not None and exp is not None
        assert np.allclose(got[0][:3], exp[0][:3], atol=1e-9)
        assert _same_rotation(got[0][3:], exp[0][3:])
        assert np.allclose(got[1], exp[1], atol=1e-7)
        assert abs(np.linalg.norm(got[0][3:]) - 1) < 1e-12


def test_identity_motion_in_the_camera_frame_is_forward_in_the_odom_frame():
    od = lvt_amd.Odometry()
    p0 = od.push_pose(np.eye(3), [0, 0, 0], 2, 0.0)
    assert np.allclose(p0[0], [0, 0, 0, 0, 0, 0, 1], atol=1e-9) and np.allclose(p0[1], 0)
    p1 = od.push_pose(np.eye(3), [0, 0, 2.0], 2, 0.5)      # 2 m along the camera's optical axis in half a second
    assert np.allclose(p1[0][:3], [2, 0, 0], atol=1e-9)     # = x forward
    assert np.allclose(p1[1][:3], [4, 0, 0], atol=1e-9) and np.allclose(p1[1][3:], 0, atol=1e-9)
    p2 = od.push_pose(np.eye(3), [-1.0, 0, 2.0], 2, 1.0)    # 1 m to the camera's left = +y
    assert np.allclose(p2[0][:3], [2, 1, 0], atol=1e-9)
    p3 = od.push_pose(np.eye(3), [-1.0, -0.5, 2.0], 2, 1.5)  # camera up (-y) = +z
    assert np.allclose(p3[0][:3], [2, 1, 0.5], atol=1e-9)


def test_stale_stamps_and_lost_policy():
    traj = _trajectory(12, 5)
    for reset_pose in (True, False):
        od, ref = lvt_amd.Odometry(None, None, reset_pose), NodeModel(None, reset_pose)
        for i, (R, t) in enumerate(traj):
            stamp = 0.1 * i if i != 5 else 0.1          # frame 5 carries an old stamp: ignored by both
            status = 3 if i == 8 else 2                   # frame 8: LOST -> nothing published, (optional) pose reset
            got, exp = od.push_pose(R, t, status, stamp), ref.push(R, t, status, stamp)
            assert (got is None) == (exp is None), i
            if got is not None:
                assert np.allclose(got[0][:3], exp[0][:3], atol=1e-9) and _same_rotation(got[0][3:], exp[0][3:])
        if reset_pose:  # after the reset the accumulated pose restarted from the identity at frame 9
            assert np.linalg.norm(got[0][:3]) < 10.0


def test_bad_arguments():
    od = lvt_amd.Odometry()
    L = lvt_amd.load_library()
    assert L.lvt_amd_odometry_push_pose(od._h, None, None, 2, 0.0, None, None) == -1
    assert L.lvt_amd_odometry_update(od._h, None, None, 1, 1, 0.0, None, None) == -1   # no tracker attached
