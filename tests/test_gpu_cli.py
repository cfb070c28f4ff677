"""examples/lvt_kitti / lvt_tum / lvt_euroc (the dataset command line harnesses over the C-ABI) on synthetic
dataset-layout directories: PNG / PGM decoding, calibration + config parsing and the trajectory files.  Every trajectory is held
FIRST against the CPU oracle's trajectory on the same decoded frames (per-frame SE3 within 1e-4, the tolerance of BASELINE.json),
and then -- tighter, as a plumbing check -- against the Python binding of the same library."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXE = os.path.join(ROOT, "examples", "lvt_kitti")
POSE_TOL = 1e-4


def _se3_close(R_cli, t_cli, R_orc, t_orc, what):
    from parity_util import pose_errors
    e_t, e_R = pose_errors(np.asarray(R_cli), np.asarray(t_cli), np.asarray(R_orc), np.asarray(t_orc))
    assert e_t <= POSE_TOL and e_R <= POSE_TOL, f"{what}: e_t={e_t:.3e} e_R={e_R:.3e} against the oracle"


def _quat_xyzw_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _png(path, img, color=False, filt=0):
    h, w = img.shape[:2]
    raw = bytearray()
    data = img if not color else np.repeat(img[:, :, None], 3, axis=2)
    bpp = 3 if color else 1
    prev = np.zeros(w * bpp, np.int32)
    for y in range(h):
        cur = data[y].reshape(-1).astype(np.int32)
        if filt == 0:
            out = cur
        elif filt == 1:   # Sub
            out = (cur - np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])) & 255
        elif filt == 2:   # Up
            out = (cur - prev) & 255
        else:             # Paeth
            a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
            c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            out = (cur - pred) & 255
        raw.append(filt if filt < 3 else 4)
        raw += out.astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2 if color else 0, 0, 0, 0)
    comp = zlib.compress(bytes(raw), 6)
    with open(path, "wb") as f:  # two IDAT chunks: the reader must concatenate them
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b""))


@pytest.mark.gpu
def test_kitti_cli_matches_the_oracle_and_the_binding(tmp_path):
    import lvt_amd
    from lvt_amd.synth import make_world
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    world = make_world("kitti", seed=3, scale=0.5)
    prm = lvt_amd.kitti_params(width=world.W, height=world.H, fx=world.fx, fy=world.fy, cx=world.cx, cy=world.cy, baseline=world.baseline)
    seq = tmp_path / "sequences" / "07"
    (seq / "image_0").mkdir(parents=True)
    (seq / "image_1").mkdir(parents=True)
    (tmp_path / "calib").mkdir()
    n = 6
    frames = [world.render_stereo(i) for i in range(n)]
    for i, (L, R) in enumerate(frames):
        _png(str(seq / "image_0" / f"{i:06d}.png"), L, color=(i % 2 == 1), filt=i % 4)   # gray and RGB files, every filter type
        if i < 3:
            _png(str(seq / "image_1" / f"{i:06d}.png"), R, filt=(i + 1) % 4)
        else:
            with open(seq / "image_1" / f"{i:06d}.pgm", "wb") as f:
                f.write(b"P5\n# synthetic\n%d %d\n255\n" % (R.shape[1], R.shape[0]) + R.tobytes())
    with open(tmp_path / "calib" / "07.yml", "w") as f:
        f.write("%%YAML:1.0\n\ncamera_matrix: !!opencv-matrix\n  rows: 3\n  cols: 3\n  dt: d\n  data: [ %.12e, 0, %.12e, 0, %.12e, %.12e, 0, 0, 1 ]\n\nbaseline: %.10e\n"
                % (prm.fx, prm.cx, prm.fy, prm.cy, prm.baseline))
    prm.write_yaml(str(tmp_path / "vo_config.yaml"))
    out = subprocess.run([EXE, str(tmp_path / "sequences"), "7"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    traj = np.loadtxt(tmp_path / "07.txt").reshape(-1, 3, 4)
    assert traj.shape[0] == n

    # the oracle on the same frames, parameters as the harness builds them (YAML floats; calibration narrowed to float, which the
    # parameter record holds as float anyway)
    from oracle import pyoracle as O
    orc = O.Oracle(lvt_amd.LvtParameters.from_file(str(tmp_path / "vo_config.yaml")), 1)
    for i, (L, R) in enumerate(frames):
        Ro, to = orc.track(L, R)
        assert orc.status == 2
        _se3_close(traj[i, :, :3], traj[i, :, 3], Ro, to, f"lvt_kitti frame {i}")

    # the binding, with the parameters exactly as the harness builds them (YAML floats, calibration narrowed to float)
    ref = lvt_amd.LvtSystem.create_from_file(str(tmp_path / "vo_config.yaml"), lvt_amd.eSensor_STEREO)
    for i, (L, R) in enumerate(frames):
        Rm, t = ref.track(L, R)
        assert ref.get_state() == 2
        assert np.allclose(traj[i, :, :3], Rm, atol=2e-9) and np.allclose(traj[i, :, 3], t, atol=2e-9), f"frame {i}"


def _png16(path, img16):
    h, w = img16.shape
    raw = bytearray()
    be = img16.astype(">u2")
    for y in range(h):
        raw.append(0)
        raw += be[y].tobytes()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))


@pytest.mark.gpu
def test_tum_cli_matches_the_oracle_and_the_binding(tmp_path):
    import lvt_amd
    from lvt_amd.synth import make_world
    exe = os.path.join(ROOT, "examples", "lvt_tum")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    world = make_world("tum", seed=2, scale=0.5)
    prm = lvt_amd.tum_params(width=world.W, height=world.H, fx=world.fx, fy=world.fy, cx=world.cx, cy=world.cy)
    ds = tmp_path / "root" / "fr1_synth"
    (ds / "rgb").mkdir(parents=True)
    (ds / "depth").mkdir(parents=True)
    (tmp_path / "assoc").mkdir()
    n = 5
    frames = []
    with open(tmp_path / "assoc" / "fr1_synth.txt", "w") as af:
        for i in range(n):
            gray, depth = world.render_rgbd(i)
            d16 = np.clip(np.rint(depth.astype(np.float64) * 5000.0), 0, 65535).astype(np.uint16)
            _png(str(ds / "rgb" / f"{i:04d}.png"), gray, color=True, filt=(i % 4))
            _png16(str(ds / "depth" / f"{i:04d}.png"), d16)
            af.write(f"{1305031102.175304 + 0.033 * i:.6f} rgb/{i:04d}.png {1305031102.160407 + 0.033 * i:.6f} depth/{i:04d}.png\n")
            frames.append((gray, d16.astype(np.float32) * np.float32(1.0 / 5000.0)))
    prm.write_yaml(str(tmp_path / "config.yaml"))
    out = subprocess.run([exe, str(tmp_path / "root"), str(tmp_path / "assoc"), "fr1_synth", str(tmp_path / "config.yaml")], cwd=tmp_path, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    traj = np.loadtxt(tmp_path / "fr1_synth.txt").reshape(-1, 8)
    assert traj.shape[0] == n

    from oracle import pyoracle as O
    orc = O.Oracle(lvt_amd.LvtParameters.from_file(str(tmp_path / "config.yaml")), 2)
    for i, (gray, depth) in enumerate(frames):
        Ro, to = orc.track_rgbd(gray, depth)
        assert orc.status == 2
        _se3_close(_quat_xyzw_to_R(traj[i, 4:8]), traj[i, 1:4], Ro, to, f"lvt_tum frame {i}")

    import ctypes as C
    L = lvt_amd.load_library()
    pod = lvt_amd.ParamsPOD()
    assert L.lvt_amd_params_from_file(str(tmp_path / "config.yaml").encode(), C.byref(pod)) == 1
    ref = lvt_amd.LvtSystem(L.lvt_amd_create(C.byref(pod), 2), 2)
    for i, (gray, depth) in enumerate(frames):
        ref.track(gray, depth)
        assert ref.get_state() == 2
        q, p = ref.pose()
        assert np.allclose(traj[i, 1:4], p, atol=2e-7) and np.allclose(traj[i, 4:8], [q[1], q[2], q[3], q[0]], atol=2e-7), f"frame {i}"


@pytest.mark.gpu
def test_euroc_cli_matches_the_oracle_and_the_binding(tmp_path):
    """lvt_euroc = PNG read + GPU rectification (both cameras) + track + body-frame TUM trajectory, against the same steps
    done through the Python binding (Rectifier.rectify, LvtSystem.track) and numpy for T_BS / the quaternion"""
    import lvt_amd
    from lvt_amd.synth import make_world
    from test_oracle_primitives import EUROC_L
    exe = os.path.join(ROOT, "examples", "lvt_euroc")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    world = make_world("euroc", seed=1)
    assert (world.W, world.H) == (752, 480)
    ds = tmp_path / "root" / "MH_synth" / "mav0"
    (ds / "cam0" / "data").mkdir(parents=True)
    (ds / "cam1" / "data").mkdir(parents=True)
    (tmp_path / "stamps").mkdir()
    n = 4
    frames = []
    with open(tmp_path / "stamps" / "MH_synth.txt", "w") as f:
        for i in range(n):
            L, R = world.render_stereo(i)
            stamp = 1403636579763555584 + 50000000 * i
            _png(str(ds / "cam0" / "data" / f"{stamp}.png"), L, filt=i % 4)
            _png(str(ds / "cam1" / "data" / f"{stamp}.png"), R, filt=(i + 2) % 4)
            f.write(f"{stamp}\n")
            frames.append((stamp, L, R))
    prm = lvt_amd.euroc_params()
    prm.write_yaml(str(tmp_path / "config.yaml"))
    out = subprocess.run([exe, str(tmp_path / "root"), str(tmp_path / "stamps"), "MH_synth", str(tmp_path / "config.yaml")], cwd=tmp_path, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    traj = np.loadtxt(tmp_path / "MH_synth.txt").reshape(-1, 8)
    assert traj.shape[0] == n

    cam1 = dict(K=[457.587, 0.0, 379.999, 0.0, 456.134, 255.238, 0.0, 0.0, 1.0], D=[-0.28368365, 0.07451284, -0.00010473, -3.555907e-05, 0.0],
                R=[0.9999633526194376, -0.003625811871560086, 0.007755443660172947, 0.003680398547259526, 0.9999684752771629, -0.007035845251224894,
                   -0.007729688520722713, 0.007064130529506649, 0.999945173484644], P=EUROC_L["P"])
    rl = lvt_amd.Rectifier(EUROC_L["K"], EUROC_L["D"], EUROC_L["R"], EUROC_L["P"], 752, 480)
    rr = lvt_amd.Rectifier(cam1["K"], cam1["D"], cam1["R"], cam1["P"], 752, 480)
    import ctypes as C
    Lib = lvt_amd.load_library()
    pod = lvt_amd.ParamsPOD()
    assert Lib.lvt_amd_params_from_file(str(tmp_path / "config.yaml").encode(), C.byref(pod)) == 1
    pod.fx = pod.fy = 435.2046959714599
    pod.cx, pod.cy, pod.baseline = 367.4517211914062, 252.2008514404297, 0.110077842
    pod.img_width, pod.img_height = 752, 480
    ref = lvt_amd.LvtSystem(Lib.lvt_amd_create(C.byref(pod), 1), 1)
    Tbs = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975], [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
                    [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949], [0, 0, 0, 1.0]])
    # the oracle chain: its own rectification maps + bilinear remap, then its tracker, with the parameters the harness derives
    from oracle import pyoracle as O
    oprm = lvt_amd.LvtParameters.from_file(str(tmp_path / "config.yaml"))
    oprm.fx = oprm.fy = float(np.float32(435.2046959714599))
    oprm.cx, oprm.cy, oprm.baseline = float(np.float32(367.4517211914062)), float(np.float32(252.2008514404297)), float(np.float32(0.110077842))
    oprm.img_width, oprm.img_height = 752, 480
    orc = O.Oracle(oprm, 1)
    maps = [O.init_undistort_rectify_map(c["K"], c["D"], c["R"], c["P"], 752, 480) for c in (EUROC_L, cam1)]
    for i, (stamp, Lm, Rm) in enumerate(frames):
        Ro, to = orc.track(O.remap_bilinear(Lm, *maps[0]), O.remap_bilinear(Rm, *maps[1]))
        if orc.status != 2:
            break
        To = np.eye(4); To[:3, :3] = Ro; To[:3, 3] = to
        Bo = Tbs @ To
        _se3_close(_quat_xyzw_to_R(traj[i, 4:8]), traj[i, 1:4], Bo[:3, :3], Bo[:3, 3], f"lvt_euroc frame {i}")
    else:
        i = n
    assert i >= 3, "the synthetic EuRoC sequence lost tracking too early to compare anything"

    lost = False
    for i, (stamp, Lm, Rm) in enumerate(frames):
        if lost:
            assert np.allclose(traj[i, 1:], [0, 0, 0, 0, 0, 0, 1])   # default pose after a LOST, as in the reference
            continue
        Rc, tc = ref.track(rl.rectify(Lm), rr.rectify(Rm))
        T = np.eye(4); T[:3, :3] = Rc; T[:3, 3] = tc
        Bm = Tbs @ T
        m = Bm[:3, :3]
        tr = np.trace(m)
        assert tr > 0
        w = 0.5 * np.sqrt(tr + 1.0); s4 = 0.5 / np.sqrt(tr + 1.0)
        q = [(m[2, 1] - m[1, 2]) * s4, (m[0, 2] - m[2, 0]) * s4, (m[1, 0] - m[0, 1]) * s4, w]
        assert abs(traj[i, 0] - stamp / 1e9) < 1e-5
        assert np.allclose(traj[i, 1:4], Bm[:3, 3], atol=2e-7) and np.allclose(traj[i, 4:8], q, atol=2e-7), f"frame {i}"
        lost = ref.get_state() == 3


FACADE_SRC = r'''
#include "lvt_system.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
// argv: config.yaml frames.bin sensor(1|2).  frames.bin = int32 n, H, W, then n x (left u8 H*W, right u8 H*W | depth f32 H*W)
int main(int argc, char **argv) {
    if (argc < 4) return 2;
    lvt_parameters params;
    if (!params.init_from_file(argv[1])) return 3;
    const int sensor = std::atoi(argv[3]);
    FILE *f = std::fopen(argv[2], "rb");
    int hdr[3];
    if (!f || std::fread(hdr, 4, 3, f) != 3) return 4;
    const int n = hdr[0], H = hdr[1], W = hdr[2];
    lvt_system *vo = lvt_system::create(params, sensor == 2 ? lvt_system::eSensor_RGBD : lvt_system::eSensor_STEREO);
    if (!vo) return 5;
    if ((int)vo->get_sensor_type() != sensor || vo->get_state() != lvt_system::eState_NOT_INITIALIZED) return 6;
    const size_t pad = 7;                                           // rows are NOT tightly packed: the facade has to repack them (cv::Mat::step)
    std::vector<unsigned char> L((size_t)H * (W + pad)), R((size_t)H * (W + pad));
    std::vector<float> D((size_t)H * (W + pad));
    auto show = [&](const char *tag, const lvt_pose &p) {
        const lvt_quaternion q = p.get_orientation_quaternion();
        const lvt_vector3 t = p.get_position();
        const lvt_matrix33 M = p.get_orientation_matrix();
        std::printf("%s %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d\n", tag, q.w(), q.x(), q.y(), q.z(), t.x(), t.y(), t.z(), M(0, 1), M(2, 0), (int)vo->get_state());
    };
    for (int pass = 0; pass < 2; pass++) {
        std::fseek(f, 12, SEEK_SET);
        for (int i = 0; i < (pass == 0 ? n : 2); i++) {
            for (int y = 0; y < H; y++)
                if (std::fread(&L[(size_t)y * (W + pad)], 1, W, f) != (size_t)W) return 7;
            lvt_pose p;
            if (sensor == 2) {
                for (int y = 0; y < H; y++)
                    if (std::fread(&D[(size_t)y * (W + pad)], 4, W, f) != (size_t)W) return 7;
                p = vo->track(lvt_image_view(L.data(), H, W, W + pad), lvt_image_view(D.data(), H, W, sizeof(float) * (W + pad)));
            } else {
                for (int y = 0; y < H; y++)
                    if (std::fread(&R[(size_t)y * (W + pad)], 1, W, f) != (size_t)W) return 7;
                p = vo->track(lvt_image_view(L.data(), H, W, W + pad), lvt_image_view(R.data(), H, W, W + pad));
            }
            show(pass == 0 ? "pose" : "again", p);
            if (pass == 0 && i == 2) {                               // a frame of the wrong size: refused, the state and the pose stay what they were
                const lvt_pose q = vo->track(lvt_image_view(L.data(), H - 1, W, W + pad), lvt_image_view(R.data(), H - 1, W, W + pad));
                show("bad", q);
            }
        }
        if (pass == 0) {
            vo->reset();                                              // lvt_system.cpp:44-68: back to NOT_INITIALIZED, the next frame founds a new map
            if (vo->get_state() != lvt_system::eState_NOT_INITIALIZED) return 8;
        }
    }
    std::printf("quit %d\n", (int)vo->should_quit());
    lvt_system::destroy(vo);
    return 0;
}
'''


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["kitti", "tum"])
def test_lvt_system_facade_tracks_like_the_oracle(tmp_path, kind):
    """include/lvt_system.h ON the GPU box (SURVEY 8b C++ API / 8f row 4): a caller written against the reference's class names -- create, track with
    row-padded image views (stereo u8 pairs, RGB-D gray + f32 depth), get_state, reset, destroy -- gets the ORACLE's poses frame by frame; a frame of
    the wrong size is refused without touching the state; after reset() the first frame founds a new map at the identity, as lvt_system.cpp:44-68 does"""
    import lvt_amd
    from parity_util import make_case
    from oracle import pyoracle as O
    world, prm, sensor = make_case(kind, seed=5, scale=0.5)
    n = 6
    frames = [(world.render_rgbd(i) if sensor == 2 else world.render_stereo(i)) for i in range(n)]
    prm.write_yaml(str(tmp_path / "vo_config.yaml"))
    with open(tmp_path / "frames.bin", "wb") as f:
        f.write(struct.pack("<iii", n, world.H, world.W))
        for a, b in frames:
            f.write(np.ascontiguousarray(a, np.uint8).tobytes())
            f.write(np.ascontiguousarray(b, np.float32 if sensor == 2 else np.uint8).tobytes())
    src = tmp_path / "facade.cpp"
    src.write_text(FACADE_SRC)
    exe = tmp_path / "facade"
    libdir = os.path.dirname(lvt_amd.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Werror", "-DLVT_SYSTEM_NO_OPENCV", "-DLVT_SYSTEM_NO_EIGEN", "-I", os.path.join(ROOT, "include"), "-o", str(exe),
                           str(src), "-L", libdir, "-llvt_c", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe), str(tmp_path / "vo_config.yaml"), str(tmp_path / "frames.bin"), str(sensor)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout[-400:], out.stderr[-400:])
    rows = [ln.split() for ln in out.stdout.strip().splitlines()]
    poses = [r for r in rows if r[0] == "pose"]
    again = [r for r in rows if r[0] == "again"]
    bad = [r for r in rows if r[0] == "bad"]
    assert len(poses) == n and len(again) == 2 and len(bad) == 1 and rows[-1] == ["quit", "0"]

    def unpack(r):
        v = [float(x) for x in r[1:10]]
        w, x, y, z = v[0:4]
        return _quat_xyzw_to_R((x, y, z, w)), np.array(v[4:7]), v[7], v[8], int(r[10])

    def check(orc, rows_, tag):
        for i, r in enumerate(rows_):
            Rm, t, m01, m20, st = unpack(r)
            Ro, to = (orc.track_rgbd if sensor == 2 else orc.track)(*frames[i])
            assert st == orc.status == 2, (tag, i, st, orc.status)
            _se3_close(Rm, t, Ro, to, f"{tag} frame {i}")
            assert abs(Rm[0, 1] - m01) < 1e-12 and abs(Rm[2, 0] - m20) < 1e-12, "get_orientation_matrix and get_orientation_quaternion describe different rotations"
    prm_file = lvt_amd.LvtParameters.from_file(str(tmp_path / "vo_config.yaml"))
    check(O.Oracle(prm_file, sensor), poses, "facade")
    check(O.Oracle(prm_file, sensor), again, "facade after reset()")          # a NEW oracle: reset() founds a new map on the next frame
    assert bad[0][1:] == poses[2][1:], "a frame of the wrong size changed the pose or the state"
