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
