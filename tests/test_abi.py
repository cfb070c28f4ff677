"""CPU tier: the C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol the headers declare."""
import ctypes
import os
import re

import lvt_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in ("lvt_c.h", "lvt_amd_ext.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names += re.findall(r"LVT_API\s+[\w\s\*]+?\b(lvt_\w+)\s*\(", txt)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(hip_lib):
    decl = declared_symbols()
    assert {"lvt_create", "lvt_destroy", "lvt_track", "lvt_track_with_external_corners", "lvt_get_status"} <= set(decl)
    assert sorted(decl) == sorted(lvt_amd.ABI_SYMBOLS)
    lib = ctypes.CDLL(lvt_amd.LIB_PATH)
    for s in decl:
        assert hasattr(lib, s), f"{s} declared in include/ but not exported"


def test_reference_signatures_are_preserved():
    """the five entry points keep the reference's argument lists (lvt/src/lvt_c.h:57-62)"""
    txt = open(os.path.join(ROOT, "include", "lvt_c.h")).read()
    assert re.search(r"lvt_handle\s+lvt_create\(const char \*config_file_name, int sensor_type\)", txt)
    assert re.search(r"void\s+lvt_track\(lvt_handle vo_system, unsigned char \*left_img, unsigned char \*right_img,\s*int n_rows, int n_cols, double R\[3\]\[3\], double t\[3\]\)", txt)
    assert re.search(r"int\s+lvt_get_status\(lvt_handle vo_system\)", txt)
    assert "typedef void *lvt_handle;" in txt


def test_no_cpu_fallback_without_gpu(hip_lib):
    """without a HIP device lvt_create / lvt_amd_create return NULL: the product never routes through the oracle"""
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    with pytest.raises(RuntimeError):
        lvt_amd.LvtSystem.create(lvt_amd.kitti_params(), 1)


def test_product_does_not_reference_the_oracle():
    """the oracle is test infrastructure: only tests/ (incl. tests/tools/), __graft_entry__.smoke() and bench.py's cpu_baseline leg
    touch it -- not the package, not the examples, not the developer tools"""
    for top in ("lvt_amd", "examples", "tools", "include"):
        for base, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".sh", ".inc")):
                    txt = open(os.path.join(base, f), errors="ignore").read()
                    assert "pyoracle" not in txt and "lvt_oracle" not in txt and "liblvt_oracle" not in txt and "from oracle" not in txt, (top, f)


def test_lvt_system_facade_compiles_without_opencv_and_eigen(tmp_path):
    """include/lvt_system.h (the reference's C++ API names over the C-ABI) is self-contained: a caller written
    like the reference's examples compiles with neither OpenCV nor Eigen installed, and links against liblvt_c.so"""
    import subprocess
    src = tmp_path / "caller.cpp"
    src.write_text(r'''
#include "lvt_system.h"
#include <cstdio>
int main(int argc, char **argv) {
    lvt_parameters params;                      // defaults of lvt_parameters.cpp:29-52
    if (argc > 1 && !params.init_from_file(argv[1])) return 2;
    params.fx = params.fy = 718.856f; params.cx = 607.19f; params.cy = 185.2f; params.baseline = 0.537f;
    params.img_width = 1241; params.img_height = 376;
    if (params.tracking_radius != 25 || params.staged_threshold != 2 || params.triangulation_policy != lvt_parameters::etriangulation_policy_decreasing_matches) return 3;
    lvt_system *vo = lvt_system::create(params, lvt_system::eSensor_STEREO);   // NULL without a GPU: no CPU fallback
    if (!vo) { std::printf("no device\n"); return 0; }
    std::vector<unsigned char> l(1241 * 376, 7), r(1241 * 376, 7);
    lvt_pose pose = vo->track(lvt_image_view(l.data(), 376, 1241), lvt_image_view(r.data(), 376, 1241));
    lvt_vector3 p = pose.get_position(); lvt_matrix33 R = pose.get_orientation_matrix(); lvt_quaternion q = pose.get_orientation_quaternion();
    lvt_pose_array poses(1); poses[0] = pose;
    std::printf("%g %g %g %d %d\n", p.x() + p(1) + p.z(), R(0, 0), q.w(), (int)vo->get_state(), (int)vo->should_quit());
    lvt_system::destroy(vo);
    return 0;
}
''')
    exe = tmp_path / "caller"
    libdir = os.path.dirname(lvt_amd.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-DLVT_SYSTEM_NO_OPENCV", "-DLVT_SYSTEM_NO_EIGEN", "-I", os.path.join(ROOT, "include"), "-o", str(exe),
                           str(src), "-L", libdir, "-llvt_c", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
