"""lvt_amd -- MI355X-native drop-in for the per-frame tracking hot path of SAR-Research-Lab/lvt.

The product is the C-ABI shared library `lvt_amd/lib/liblvt_c.so` (hand-written gfx950 HIP kernels
behind the reference's own `lvt_c` boundary, see include/lvt_c.h).  This module is the thin Python
mirror of the reference's `lvt_system` interface (lvt/src/lvt_system.h:57-70) used by the tests and
bench harness: same operation names, same argument meaning, same "silent failure" semantics.

There is NO CPU fallback: importing works anywhere (so the CPU test tier can check the exported symbols),
but creating a system without the HIP library or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .params import LvtParameters, ParamsPOD, kitti_params, euroc_params, tum_params  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LVT_AMD_LIB") or os.path.join(_HERE, "lib", "liblvt_c.so")  # override: kernel A/B experiments

# every symbol include/lvt_c.h and include/lvt_amd_ext.h declare
ABI_SYMBOLS = [
    "lvt_create", "lvt_destroy", "lvt_track", "lvt_track_with_external_corners", "lvt_get_status",
    "lvt_amd_default_params", "lvt_amd_params_from_file", "lvt_amd_create", "lvt_amd_reset",
    "lvt_amd_track_rgbd", "lvt_amd_track_device", "lvt_amd_track_device_async", "lvt_amd_wait",
    "lvt_amd_set_stream", "lvt_amd_last_error", "lvt_amd_get_counts", "lvt_amd_get_features",
    "lvt_amd_get_matches", "lvt_amd_get_row_matches", "lvt_amd_get_map", "lvt_amd_get_staged",
    "lvt_amd_get_pose", "lvt_amd_get_last_pose", "lvt_amd_get_predicted_pose", "lvt_amd_get_plane", "lvt_amd_pnp", "lvt_amd_pnp_detail",
    "lvt_amd_hamming_match_batched", "lvt_amd_hamming_match_batched_n", "lvt_amd_rectifier_create", "lvt_amd_rectifier_destroy",
    "lvt_amd_rectify_device", "lvt_amd_rectify", "lvt_amd_rectifier_get_maps",
    "lvt_amd_odometry_create", "lvt_amd_odometry_destroy", "lvt_amd_odometry_reset", "lvt_amd_odometry_push_pose", "lvt_amd_odometry_update", "lvt_amd_profile_enable", "lvt_amd_profile_read", "lvt_amd_get_debug", "lvt_amd_get_timeline", "lvt_amd_get_ordering",
    "lvt_amd_batch_create", "lvt_amd_batch_size", "lvt_amd_batch_track_device_async", "lvt_amd_batch_wait",
    "lvt_amd_batch_get_counts", "lvt_amd_create_on_device", "lvt_amd_get_device", "lvt_amd_wait_status", "lvt_amd_get_host_stats",
    "lvt_amd_track_async", "lvt_amd_track_rgbd_async", "lvt_amd_pnp_trace", "lvt_amd_create_pooled", "lvt_amd_wait_pose",
]

N_COUNTS = 32
COUNT_NAMES = ["n_left", "n_right", "map_size", "staged_size", "n_matches", "second_pass", "n_row_matches",
               "n_triangulated", "triangulated", "retry_left", "retry_right", "pnp_iters", "pnp_inliers",
               "map_size_at_match", "n_staged_erased", "n_staged_promoted", "n_culled", "frame", "overflow", "pnp_borderline", "row_fallback", "pnp_trials", "pnp_rejections", "pnp_terminates"]

eState_NOT_INITIALIZED, eState_TRACKING, eState_LOST = 1, 2, 3
eSensor_STEREO, eSensor_RGBD = 1, 2

_lib = None


def load_library():
    """dlopen the HIP library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C lvt_amd/csrc` "
                           "(there is no CPU fallback for the tracking path)")
    # PyTorch wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, same as /opt/rocm's).  If this library
    # were loaded first, a later `import torch` would bring a SECOND HIP runtime into the process and whichever
    # initialises last reports "no ROCm-capable device".  Importing torch first makes the dynamic loader resolve our
    # DT_NEEDED libamdhip64.so.7 to the copy that is already mapped: one runtime per process.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional: plain C / ctypes callers simply use the system runtime
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.lvt_create.restype = vp
    L.lvt_create.argtypes = [C.c_char_p, C.c_int]
    L.lvt_amd_create.restype = vp
    L.lvt_amd_create.argtypes = [vp, C.c_int]
    L.lvt_amd_create_on_device.restype = vp
    L.lvt_amd_create_on_device.argtypes = [vp, C.c_int, C.c_int]
    L.lvt_amd_get_device.argtypes = [vp]
    L.lvt_amd_create_pooled.restype = vp
    L.lvt_amd_create_pooled.argtypes = [vp, C.c_int, C.c_int]
    L.lvt_destroy.argtypes = [vp]
    L.lvt_amd_reset.argtypes = [vp]
    L.lvt_track.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
    L.lvt_track_with_external_corners.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp]
    L.lvt_get_status.argtypes = [vp]
    L.lvt_amd_params_from_file.argtypes = [C.c_char_p, vp]
    L.lvt_amd_default_params.argtypes = [vp]
    L.lvt_amd_track_rgbd.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
    L.lvt_amd_track_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.lvt_amd_track_device_async.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int]
    L.lvt_amd_wait.argtypes = [vp, vp, vp]
    L.lvt_amd_track_async.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    L.lvt_amd_track_async.restype = C.c_int
    L.lvt_amd_track_rgbd_async.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    L.lvt_amd_track_rgbd_async.restype = C.c_int
    L.lvt_amd_wait_status.argtypes = [vp, vp, vp]
    L.lvt_amd_wait_status.restype = C.c_int
    L.lvt_amd_set_stream.argtypes = [vp, vp]
    L.lvt_amd_last_error.restype = C.c_char_p
    L.lvt_amd_last_error.argtypes = [vp]
    L.lvt_amd_get_counts.argtypes = [vp, vp]
    L.lvt_amd_get_features.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
    L.lvt_amd_get_matches.argtypes = [vp, vp, vp, C.c_int]
    L.lvt_amd_get_row_matches.argtypes = [vp, vp, C.c_int]
    L.lvt_amd_get_map.argtypes = [vp, vp, vp, vp, vp, C.c_int]
    L.lvt_amd_get_staged.argtypes = [vp, vp, vp, vp, C.c_int]
    L.lvt_amd_get_pose.argtypes = [vp, vp, vp]
    L.lvt_amd_get_last_pose.argtypes = [vp, vp, vp]
    L.lvt_amd_get_predicted_pose.argtypes = [vp, vp, vp]
    L.lvt_amd_get_plane.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp]
    L.lvt_amd_pnp.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, vp, vp]
    L.lvt_amd_pnp_detail.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.lvt_amd_pnp_trace.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]
    L.lvt_amd_hamming_match_batched.restype = C.c_float
    L.lvt_amd_hamming_match_batched.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, vp, vp]
    L.lvt_amd_get_timeline.argtypes = [vp, vp]
    L.lvt_amd_get_ordering.argtypes = [vp]
    L.lvt_amd_get_ordering.restype = C.c_int
    L.lvt_amd_rectifier_create.restype = vp
    L.lvt_amd_rectifier_create.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int]
    L.lvt_amd_rectifier_destroy.argtypes = [vp]
    L.lvt_amd_rectify_device.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp]
    L.lvt_amd_rectify.argtypes = [vp, vp, vp]
    L.lvt_amd_rectifier_get_maps.argtypes = [vp, vp, vp]
    L.lvt_amd_odometry_create.restype = vp
    L.lvt_amd_odometry_create.argtypes = [vp, vp, C.c_int]
    L.lvt_amd_odometry_destroy.argtypes = [vp]
    L.lvt_amd_odometry_reset.argtypes = [vp]
    L.lvt_amd_odometry_push_pose.restype = C.c_int
    L.lvt_amd_odometry_push_pose.argtypes = [vp, vp, vp, C.c_int, C.c_double, vp, vp]
    L.lvt_amd_odometry_update.restype = C.c_int
    L.lvt_amd_odometry_update.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_double, vp, vp]
    L.lvt_amd_hamming_match_batched_n.restype = C.c_float
    L.lvt_amd_hamming_match_batched_n.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]
    L.lvt_amd_profile_enable.argtypes = [vp, C.c_int]
    L.lvt_amd_batch_create.restype = vp
    L.lvt_amd_batch_create.argtypes = [vp, C.c_int, C.c_int]
    L.lvt_amd_batch_size.argtypes = [vp]
    L.lvt_amd_batch_track_device_async.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int]
    L.lvt_amd_batch_wait.argtypes = [vp, vp, vp, vp]
    L.lvt_amd_batch_get_counts.argtypes = [vp, C.c_int, vp]
    L.lvt_amd_get_debug.argtypes = [vp, vp]
    L.lvt_amd_get_host_stats.argtypes = [vp, vp]
    L.lvt_amd_profile_read.argtypes = [vp, C.c_int, C.c_char_p, C.c_int, vp, vp]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(img):
    a = np.ascontiguousarray(img, dtype=np.uint8)
    assert a.ndim == 2
    return a


class LvtSystem:
    """Mirror of `lvt_system` (reference lvt/src/lvt_system.h:41-109) over the HIP C-ABI."""

    def __init__(self, handle, sensor_type):
        self._h = handle
        self._sensor = sensor_type

    # lvt_system::create(const lvt_parameters&, eSensor)  -- lvt_system.cpp:70-127
    @classmethod
    def create(cls, params: LvtParameters, sensor_type: int = eSensor_STEREO, device: int = -1, pooled: bool = False) -> "LvtSystem":
        """device >= 0: the handle owns that HIP device whatever the calling thread's current device is (lvt_amd_create_on_device);
        pooled: a seat of the device's shared lock-step chain (lvt_amd_create_pooled)"""
        L = load_library()
        pod = params.to_pod()
        if pooled:
            h = L.lvt_amd_create_pooled(C.byref(pod), sensor_type, device)
        else:
            h = L.lvt_amd_create_on_device(C.byref(pod), sensor_type, device) if device >= 0 else L.lvt_amd_create(C.byref(pod), sensor_type)
        if not h:
            raise RuntimeError("lvt_amd_create failed (bad parameters, or no usable HIP device -- no CPU fallback)")
        return cls(h, sensor_type)

    # lvt_create(config_file, sensor)  -- lvt_c.cpp:33-48
    @classmethod
    def create_from_file(cls, config_file: str, sensor_type: int = eSensor_STEREO) -> "LvtSystem":
        L = load_library()
        h = L.lvt_create(config_file.encode(), sensor_type)
        if not h:
            raise RuntimeError("lvt_create returned NULL")
        return cls(h, sensor_type)

    @staticmethod
    def destroy(system: "LvtSystem"):
        system.close()

    def close(self):
        if self._h:
            load_library().lvt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        load_library().lvt_amd_reset(self._h)

    def get_sensor_type(self):
        return self._sensor

    def device(self) -> int:
        return load_library().lvt_amd_get_device(self._h)

    def get_state(self):
        return load_library().lvt_get_status(self._h)

    def last_error(self) -> str:
        return load_library().lvt_amd_last_error(self._h).decode()

    def host_stats(self) -> dict:
        """host-side counters: frames enqueued / collected, host-buffer planes read in place (page-locked caller buffers) / staged"""
        a = (C.c_longlong * 8)()
        load_library().lvt_amd_get_host_stats(self._h, a)
        return {"enqueued": int(a[0]), "collected": int(a[1]), "planes_in_place": int(a[2]), "planes_staged": int(a[3]),
                "async_host_frames": int(a[4]), "pulls_carried_by_the_previous_frame": int(a[5]), "score_pieces": int(a[6]), "event_ordering": int(a[7])}

    # lvt_system::track(img1, img2)  -- lvt_system.cpp:157-207
    def track(self, img1, img2):
        R = np.zeros((3, 3)); t = np.zeros(3)
        a = _u8(img1)
        if self._sensor == eSensor_STEREO:
            b = _u8(img2)
            load_library().lvt_track(self._h, _p(a), _p(b), a.shape[0], a.shape[1], _p(R), _p(t))
        else:
            d = np.ascontiguousarray(img2, dtype=np.float32)
            load_library().lvt_amd_track_rgbd(self._h, _p(a), _p(d), a.shape[0], a.shape[1], _p(R), _p(t))
        return R, t

    # lvt_system::track_with_external_corners  -- lvt_system.cpp:209-250
    def track_with_external_corners(self, left, right, corners_left, corners_right):
        R = np.zeros((3, 3)); t = np.zeros(3)
        a, b = _u8(left), _u8(right)
        cl = np.ascontiguousarray(corners_left, dtype=np.float64).reshape(-1, 2)
        cr = np.ascontiguousarray(corners_right, dtype=np.float64).reshape(-1, 2)
        load_library().lvt_track_with_external_corners(self._h, _p(a), _p(b), a.shape[0], a.shape[1], _p(cl), len(cl),
                                                       _p(cr), len(cr), _p(R), _p(t))
        return R, t

    # zero-copy: images already resident in HBM (torch uint8 CUDA tensors or raw device pointers)
    def track_device(self, d_left: int, d_right: int, rows: int, cols: int, pitch: int):
        R = np.zeros((3, 3)); t = np.zeros(3)
        load_library().lvt_amd_track_device(self._h, C.c_void_p(d_left), C.c_void_p(d_right), rows, cols, pitch, _p(R), _p(t))
        return R, t

    def track_device_async(self, d_left: int, d_right: int, rows: int, cols: int, pitch: int):
        load_library().lvt_amd_track_device_async(self._h, C.c_void_p(d_left), C.c_void_p(d_right), rows, cols, pitch)

    def track_async(self, img1, img2) -> int:
        """lvt_amd_track_async / lvt_amd_track_rgbd_async: HOST images, the call returns once the frame is enqueued (0) or rejected (-1).
        The arrays are used as they are (no copy here): a page-locked one must stay alive and unchanged until the frame is collected."""
        a = img1 if (isinstance(img1, np.ndarray) and img1.dtype == np.uint8 and img1.flags.c_contiguous) else _u8(img1)
        if self._sensor == eSensor_STEREO:
            b = img2 if (isinstance(img2, np.ndarray) and img2.dtype == np.uint8 and img2.flags.c_contiguous) else _u8(img2)
            return load_library().lvt_amd_track_async(self._h, _p(a), _p(b), a.shape[0], a.shape[1])
        d = img2 if (isinstance(img2, np.ndarray) and img2.dtype == np.float32 and img2.flags.c_contiguous) else np.ascontiguousarray(img2, dtype=np.float32)
        return load_library().lvt_amd_track_rgbd_async(self._h, _p(a), _p(d), a.shape[0], a.shape[1])

    def track_async_ptr(self, p_left: int, p_right: int, rows: int, cols: int) -> int:
        """the same on raw host addresses (stereo; bench.py: no per-call numpy work inside the timed region)"""
        return load_library().lvt_amd_track_async(self._h, C.c_void_p(p_left), C.c_void_p(p_right), rows, cols)

    def wait(self):
        R = np.zeros((3, 3)); t = np.zeros(3)
        load_library().lvt_amd_wait(self._h, _p(R), _p(t))
        return R, t

    def wait_status(self):
        """(R, t, state after that frame) of the oldest un-collected frame"""
        R = np.zeros((3, 3)); t = np.zeros(3)
        st = load_library().lvt_amd_wait_status(self._h, _p(R), _p(t))
        return R, t, st

    def set_stream(self, hip_stream: int):
        load_library().lvt_amd_set_stream(self._h, C.c_void_p(hip_stream))

    # ---- introspection of the last frame (for stage-by-stage parity tests) ----
    def counts(self):
        a = np.zeros(N_COUNTS, dtype=np.int32)
        load_library().lvt_amd_get_counts(self._h, _p(a))
        return {n: int(a[i]) for i, n in enumerate(COUNT_NAMES)}

    def features(self, eye=0, cap=16384):
        xy = np.zeros((cap, 2), np.float32); resp = np.zeros(cap, np.float32); desc = np.zeros((cap, 32), np.uint8)
        n = load_library().lvt_amd_get_features(self._h, eye, _p(xy), _p(resp), _p(desc), cap)
        return xy[:n].copy(), resp[:n].copy(), desc[:n].copy()

    def matches(self, cap=65536):
        fi = np.zeros(cap, np.int32); xyz = np.zeros((cap, 3), np.float64)
        n = load_library().lvt_amd_get_matches(self._h, _p(fi), _p(xyz), cap)
        return fi[:n].copy(), xyz[:n].copy()

    def row_matches(self, cap=16384):
        pr = np.zeros((cap, 2), np.int32)
        n = load_library().lvt_amd_get_row_matches(self._h, _p(pr), cap)
        return pr[:n].copy()

    def map(self, cap=65536):
        xyz = np.zeros((cap, 3)); cnt = np.zeros(cap, np.int32); age = np.zeros(cap, np.int32); desc = np.zeros((cap, 32), np.uint8)
        n = load_library().lvt_amd_get_map(self._h, _p(xyz), _p(cnt), _p(age), _p(desc), cap)
        return xyz[:n].copy(), cnt[:n].copy(), age[:n].copy(), desc[:n].copy()

    def staged(self, cap=65536):
        xyz = np.zeros((cap, 3)); cnt = np.zeros(cap, np.int32); desc = np.zeros((cap, 32), np.uint8)
        n = load_library().lvt_amd_get_staged(self._h, _p(xyz), _p(cnt), _p(desc), cap)
        return xyz[:n].copy(), cnt[:n].copy(), desc[:n].copy()

    def pose(self):
        q = np.zeros(4); p = np.zeros(3)
        load_library().lvt_amd_get_pose(self._h, _p(q), _p(p))
        return q, p

    def predicted_pose(self):
        q = np.zeros(4); p = np.zeros(3)
        load_library().lvt_amd_get_predicted_pose(self._h, _p(q), _p(p))
        return q, p

    def profile_enable(self, on: bool = True):
        load_library().lvt_amd_profile_enable(self._h, 1 if on else 0)

    def profile_read(self):
        """[(kernel name, total ms, launches)] accumulated since profile_enable(True)"""
        out = []
        name = C.create_string_buffer(64); ms = C.c_double(0); calls = C.c_long(0)
        for slot in range(64):
            if not load_library().lvt_amd_profile_read(self._h, slot, name, 64, C.byref(ms), C.byref(calls)):
                if slot >= 22:
                    break
                continue
            out.append((name.value.decode(), ms.value, calls.value))
        return out

    def ordering(self):
        """'polling' (gates + early stream), 'events' (barriers only) or 'pooled'; see lvt_amd_get_ordering"""
        return ("polling", "events", "pooled")[load_library().lvt_amd_get_ordering(self._h)]

    def timeline(self):
        a = np.zeros(16, dtype=np.int64)
        load_library().lvt_amd_get_timeline(self._h, _p(a))
        return a

    def debug_stamps(self):
        a = np.zeros(32, dtype=np.int64)
        load_library().lvt_amd_get_debug(self._h, _p(a))
        return a

    def plane(self, eye=0, what=0):
        """what=0: corner score map (u8), what=1: 9x9 box sums (u16); returns (rows, pitch) array"""
        cap = 64 << 20
        buf = np.zeros(cap // 8, dtype=np.uint64)
        pitch = C.c_int(0)
        nbytes = load_library().lvt_amd_get_plane(self._h, eye, what, _p(buf), cap, C.byref(pitch))
        if nbytes < 0:
            raise RuntimeError("lvt_amd_get_plane failed")
        raw = buf.view(np.uint8)[:nbytes]
        a = raw.view(np.uint8 if what == 0 else np.uint16)
        return a.reshape(-1, pitch.value).copy()


class LvtBatch:
    """B independent sequences advanced in lock-step by one launch chain on one GPU (include/lvt_amd_ext.h)."""

    def __init__(self, params: LvtParameters, n_sequences: int, sensor_type: int = eSensor_STEREO):
        L = load_library()
        pod = params.to_pod()
        self._h = L.lvt_amd_batch_create(C.byref(pod), sensor_type, n_sequences)
        if not self._h:
            raise RuntimeError("lvt_amd_batch_create failed (no usable HIP device -- no CPU fallback)")
        self.B = n_sequences

    def close(self):
        if self._h:
            load_library().lvt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def track_device_async(self, d_left, d_right, rows, cols, pitch):
        a = (C.c_void_p * self.B)(*[int(x) for x in d_left]); b = (C.c_void_p * self.B)(*[int(x) for x in d_right])
        load_library().lvt_amd_batch_track_device_async(self._h, a, b, rows, cols, pitch)

    def wait(self):
        R = np.zeros((self.B, 3, 3)); t = np.zeros((self.B, 3)); st = np.zeros(self.B, np.int32)
        load_library().lvt_amd_batch_wait(self._h, _p(R), _p(t), _p(st))
        return R, t, st

    def counts(self, seq):
        a = np.zeros(N_COUNTS, dtype=np.int32)
        load_library().lvt_amd_batch_get_counts(self._h, seq, _p(a))
        return {n: int(a[i]) for i, n in enumerate(COUNT_NAMES)}

    def last_error(self) -> str:
        return load_library().lvt_amd_last_error(self._h).decode()

    def profile_enable(self, on: bool = True):
        load_library().lvt_amd_profile_enable(self._h, 1 if on else 0)

    profile_read = LvtSystem.profile_read
    timeline = LvtSystem.timeline   # (sequence 0's stamps)
    host_stats = LvtSystem.host_stats
    debug_stamps = LvtSystem.debug_stamps


def pnp(params: LvtParameters, q_in, p_in, pts, obs):
    """stage entry: motion-only BA on caller data (reference lvt_pnp_solver.cpp:60-128)"""
    L = load_library()
    pod = params.to_pod()
    q_in = np.ascontiguousarray(q_in, np.float64); p_in = np.ascontiguousarray(p_in, np.float64)
    pts = np.ascontiguousarray(pts, np.float64); obs = np.ascontiguousarray(obs, np.float32)
    q = np.zeros(4); p = np.zeros(3); calls = C.c_int(0)
    inl = L.lvt_amd_pnp(C.byref(pod), _p(q_in), _p(p_in), _p(pts), _p(obs), len(pts), _p(q), _p(p), C.byref(calls))
    if inl < 0:
        raise RuntimeError("lvt_amd_pnp failed")
    return q, p, inl, calls.value


def pnp_detail(params: LvtParameters, q_in, p_in, pts, obs):
    """lvt_amd_pnp with the chi2 gates laid open: (q, p, inliers, solve calls, err (n x 2: what the second gate saw), level (n: 1 = demoted),
    borderline = gate decisions taken within 1e-8 of the threshold)"""
    L = load_library()
    pod = params.to_pod()
    q_in = np.ascontiguousarray(q_in, np.float64); p_in = np.ascontiguousarray(p_in, np.float64)
    pts = np.ascontiguousarray(pts, np.float64); obs = np.ascontiguousarray(obs, np.float32)
    n = len(pts)
    q = np.zeros(4); p = np.zeros(3); calls = C.c_int(0); border = C.c_int(0)
    err = np.zeros((n, 2)); level = np.zeros(n, np.int32)
    inl = L.lvt_amd_pnp_detail(C.byref(pod), _p(q_in), _p(p_in), _p(pts), _p(obs), n, _p(q), _p(p), C.byref(calls), _p(err), _p(level), C.byref(border))
    if inl < 0:
        raise RuntimeError("lvt_amd_pnp_detail failed")
    return q, p, inl, calls.value, err, level, border.value


def pnp_trace(params: LvtParameters, q_in, p_in, pts, obs, trace_cap=256):
    """lvt_amd_pnp_trace: (q, p, inliers, solve calls, trace rows (trials x 4: lambda, chi2 at the estimate, chi2 of the trial, rho),
    (trials, rejections, terminates))"""
    L = load_library()
    pod = params.to_pod()
    q_in = np.ascontiguousarray(q_in, np.float64); p_in = np.ascontiguousarray(p_in, np.float64)
    pts = np.ascontiguousarray(pts, np.float64); obs = np.ascontiguousarray(obs, np.float32)
    q = np.zeros(4); p = np.zeros(3); calls = C.c_int(0)
    tr = np.zeros((trace_cap, 4)); st = np.zeros(3, np.int32)
    inl = L.lvt_amd_pnp_trace(C.byref(pod), _p(q_in), _p(p_in), _p(pts), _p(obs), len(pts), _p(q), _p(p), C.byref(calls), None, None, None,
                              _p(tr), trace_cap, _p(st))
    if inl < 0:
        raise RuntimeError("lvt_amd_pnp_trace failed")
    return q, p, inl, calls.value, tr[:min(int(st[0]), trace_cap)].copy(), tuple(int(x) for x in st)


def hamming_match_batched(q_desc, q_xy, t_desc, t_xy, t_flag, r2: float, mode: int, img_rows: int, img_cols: int, out, stream: int = 0,
                          launches: int = 1):
    """Batched masked 2-NN Hamming matcher on DEVICE tensors (torch): q_desc (B,M,32) u8, q_xy (B,M,2) f32,
    t_desc (B,N,32) u8, t_xy (B,N,2) f32, t_flag (B,N) u8, out (B,M,4) i32.  Returns kernel time in us."""
    L = load_library()
    B, M = q_desc.shape[0], q_desc.shape[1]
    N = t_desc.shape[1]
    us = L.lvt_amd_hamming_match_batched_n(C.c_void_p(q_desc.data_ptr()), C.c_void_p(q_xy.data_ptr()), C.c_void_p(t_desc.data_ptr()),
                                           C.c_void_p(t_xy.data_ptr()), C.c_void_p(t_flag.data_ptr()), B, M, N, float(r2), int(mode),
                                           int(img_rows), int(img_cols), C.c_void_p(out.data_ptr()), C.c_void_p(stream), int(launches))
    if us < 0:
        raise RuntimeError("lvt_amd_hamming_match_batched failed")
    return us  # (0.0 for an empty train set: nothing was timed)


class Rectifier:
    """cv::initUndistortRectifyMap + cv::remap(INTER_LINEAR) on the GPU (the EuRoC example's pre-step)."""

    def __init__(self, K, D, R, P, width: int, height: int):
        L = load_library()
        a = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in (K, D, R, P)]
        self._keep = a
        self.w, self.h = width, height
        self._h = L.lvt_amd_rectifier_create(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), width, height)
        if not self._h:
            raise RuntimeError("lvt_amd_rectifier_create returned NULL")

    def maps(self):
        m1 = np.zeros((self.h, self.w), np.float32); m2 = np.zeros((self.h, self.w), np.float32)
        if load_library().lvt_amd_rectifier_get_maps(self._h, _p(m1), _p(m2)) != 0:
            raise RuntimeError("lvt_amd_rectifier_get_maps failed")
        return m1, m2

    def rectify(self, img):
        a = _u8(img)
        out = np.zeros((self.h, self.w), np.uint8)
        if load_library().lvt_amd_rectify(self._h, _p(a), _p(out)) != 0:
            raise RuntimeError("lvt_amd_rectify failed")
        return out

    def close(self):
        if self._h:
            load_library().lvt_amd_rectifier_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class Odometry:
    """What the reference's ROS node does with every pose (lvt_ros.cpp:215-311), without ROS: x-forward / z-up re-expression,
    base-frame deltas accumulated into base_to_odom, stale frames ignored, tracker (+ optionally pose) reset on LOST."""

    def __init__(self, system=None, base_to_sensor=None, reset_pose_on_lost=True):
        L = load_library()
        self._sys = system
        b = None if base_to_sensor is None else np.ascontiguousarray(base_to_sensor, dtype=np.float64).reshape(12)
        self._h = L.lvt_amd_odometry_create(system._h if system is not None else None, None if b is None else _p(b), 1 if reset_pose_on_lost else 0)
        if not self._h:
            raise RuntimeError("lvt_amd_odometry_create returned NULL")

    def _call(self, fn, *args):
        pose = np.zeros(7); twist = np.zeros(6)
        rc = fn(self._h, *args, _p(pose), _p(twist))
        if rc < 0:
            raise ValueError("bad arguments")
        return (pose, twist) if rc == 1 else None

    def push_pose(self, R, t, status, stamp):
        """returns (pose[x y z qx qy qz qw], twist[6]) or None (LOST -> reset, or a stale time stamp)"""
        Rm = np.ascontiguousarray(R, dtype=np.float64).reshape(3, 3); tv = np.ascontiguousarray(t, dtype=np.float64).reshape(3)
        return self._call(load_library().lvt_amd_odometry_push_pose, _p(Rm), _p(tv), int(status), float(stamp))

    def update(self, left, right, stamp):
        a, b = _u8(left), _u8(right)
        return self._call(load_library().lvt_amd_odometry_update, _p(a), _p(b), a.shape[0], a.shape[1], float(stamp))

    def reset(self):
        load_library().lvt_amd_odometry_reset(self._h)

    def close(self):
        if self._h:
            load_library().lvt_amd_odometry_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
