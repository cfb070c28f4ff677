"""Host-side mirror of `struct lvt_parameters` (reference lvt/src/lvt_parameters.h:29-64).

Same field names, same defaults (lvt_parameters.cpp:29-52) and the same YAML-loading behaviour
(lvt_parameters.cpp:54-93: every field is overwritten, a missing key reads as 0).  The C++ side of the
product has its own reader for lvt_create(); this one serves the Python harness (tests, bench).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, fields, asdict

_FLOAT = ("fx", "fy", "cx", "cy", "baseline", "k1", "k2", "p1", "p2", "k3", "near_plane_distance",
          "far_plane_distance", "triangulation_ratio_test_threshold", "tracking_ratio_test_threshold",
          "descriptor_matching_threshold")
_INT = ("img_width", "img_height", "min_num_matches_for_tracking", "tracking_radius", "detection_cell_size",
        "max_keypoints_per_cell", "agast_threshold", "untracked_threshold", "staged_threshold",
        "triangulation_policy")


class ParamsPOD(C.Structure):
    """Binary layout shared by `lvt_amd_params` (include/lvt_amd_ext.h) and `lvto_params` (oracle)."""
    _fields_ = [
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("baseline", C.c_float),
        ("img_width", C.c_int), ("img_height", C.c_int),
        ("k1", C.c_float), ("k2", C.c_float), ("p1", C.c_float), ("p2", C.c_float), ("k3", C.c_float),
        ("near_plane_distance", C.c_float), ("far_plane_distance", C.c_float),
        ("triangulation_ratio_test_threshold", C.c_float), ("tracking_ratio_test_threshold", C.c_float),
        ("descriptor_matching_threshold", C.c_float),
        ("min_num_matches_for_tracking", C.c_int), ("tracking_radius", C.c_int), ("detection_cell_size", C.c_int),
        ("max_keypoints_per_cell", C.c_int), ("agast_threshold", C.c_int), ("untracked_threshold", C.c_int),
        ("staged_threshold", C.c_int), ("triangulation_policy", C.c_int),
    ]


@dataclass
class LvtParameters:
    fx: float = 0.5
    fy: float = 0.5
    cx: float = 0.5
    cy: float = 0.5
    baseline: float = 0.0
    img_width: int = 0
    img_height: int = 0
    k1: float = 0.0
    k2: float = 0.0
    p1: float = 0.0
    p2: float = 0.0
    k3: float = 0.0
    near_plane_distance: float = 0.1
    far_plane_distance: float = 500.0
    triangulation_ratio_test_threshold: float = 0.60
    tracking_ratio_test_threshold: float = 0.80
    descriptor_matching_threshold: float = 30.0
    min_num_matches_for_tracking: int = 10
    tracking_radius: int = 25
    detection_cell_size: int = 250
    max_keypoints_per_cell: int = 150
    agast_threshold: int = 25
    untracked_threshold: int = 10
    staged_threshold: int = 2
    triangulation_policy: int = 1

    def to_pod(self) -> ParamsPOD:
        p = ParamsPOD()
        for f in fields(self):
            setattr(p, f.name, getattr(self, f.name))
        return p

    @classmethod
    def from_file(cls, path: str) -> "LvtParameters":
        """OpenCV-FileStorage-style `%YAML:1.0` flat key: value file; missing key -> 0."""
        vals = {}
        with open(path, "r") as fh:
            for line in fh:
                line = line.split("#", 1)[0].strip()
                if not line or line.startswith("%") or line.startswith("---") or ":" not in line:
                    continue
                k, v = line.split(":", 1)
                v = v.strip()
                try:
                    vals[k.strip()] = float(v)
                except ValueError:
                    pass
        p = cls()
        for n in _FLOAT:
            setattr(p, n, float(vals.get(n, 0.0)))
        for n in _INT:
            setattr(p, n, int(vals.get(n, 0)))
        return p

    def write_yaml(self, path: str):
        with open(path, "w") as fh:
            fh.write("%YAML:1.0\n\n")
            for k, v in asdict(self).items():
                fh.write(f"{k}: {v!r}\n")
            fh.write("enable_logging: 0\nenable_visualization: 0\n")


def kitti_params(width=1241, height=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157,
                 baseline=0.53716571886) -> LvtParameters:
    """examples/kitti/vo_config.yaml + calib/00.yml of the reference."""
    return LvtParameters(fx=fx, fy=fy, cx=cx, cy=cy, baseline=baseline, img_width=width, img_height=height,
                         near_plane_distance=0.01, far_plane_distance=500.0,
                         triangulation_ratio_test_threshold=0.60, tracking_ratio_test_threshold=0.80,
                         descriptor_matching_threshold=30, min_num_matches_for_tracking=10, tracking_radius=25,
                         agast_threshold=25, detection_cell_size=250, max_keypoints_per_cell=150,
                         untracked_threshold=10, staged_threshold=2, triangulation_policy=1)


def euroc_params(width=752, height=480, fx=435.2, fy=435.2, cx=367.4, cy=252.2, baseline=0.11) -> LvtParameters:
    """examples/euroc/vo_config_euroc.yaml of the reference."""
    return LvtParameters(fx=fx, fy=fy, cx=cx, cy=cy, baseline=baseline, img_width=width, img_height=height,
                         near_plane_distance=0.01, far_plane_distance=500.0,
                         triangulation_ratio_test_threshold=0.60, tracking_ratio_test_threshold=0.70,
                         descriptor_matching_threshold=30, min_num_matches_for_tracking=10, tracking_radius=25,
                         agast_threshold=20, detection_cell_size=250, max_keypoints_per_cell=100,
                         untracked_threshold=10, staged_threshold=0, triangulation_policy=1)


def tum_params(width=640, height=480, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989) -> LvtParameters:
    """examples/tum_rgbd/config_tum1.yaml of the reference (distortion zeroed for the synthetic stand-in)."""
    return LvtParameters(fx=fx, fy=fy, cx=cx, cy=cy, baseline=0.0, img_width=width, img_height=height,
                         near_plane_distance=0.1, far_plane_distance=5.0,
                         triangulation_ratio_test_threshold=0.60, tracking_ratio_test_threshold=0.70,
                         descriptor_matching_threshold=30, min_num_matches_for_tracking=10, tracking_radius=30,
                         agast_threshold=18, detection_cell_size=2000, max_keypoints_per_cell=1000,
                         untracked_threshold=10, staged_threshold=0, triangulation_policy=2)
