"""Deterministic synthetic stereo / RGB-D sequences shaped like the reference's datasets.

No KITTI / EuRoC / TUM data exists offline, so every config of BASELINE.json is
run on a stand-in of identical shape: a world of textured fronto-parallel layers at several depths,
viewed by a rectified stereo rig that moves on a smooth bounded trajectory (ground truth known).
Rendering is an exact per-layer homography warp with bilinear sampling, so left/right disparity and
frame-to-frame motion are geometrically consistent and the odometry can be checked against truth.

Pure numpy (bit-reproducible on any machine: only IEEE +,-,*,/ and floor are used per pixel).  A torch
backend with identical math renders on the GPU for the benchmark where hundreds of full-size frames are
needed quickly.  This module is harness-side plumbing (inputs), not part of the tracking hot path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

# KITTI sequence 00 intrinsics: /root/reference/examples/kitti/calib/00.yml:7,9
KITTI = dict(width=1241, height=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, baseline=0.53716571886)
# EuRoC rectified-ish stand-in (752x480): /root/reference/examples/euroc/euroc_example.cpp:109-113 (approximate)
EUROC = dict(width=752, height=480, fx=435.2, fy=435.2, cx=367.4, cy=252.2, baseline=0.11)
# TUM fr1: /root/reference/examples/tum_rgbd/config_tum1.yaml:3-6,11-12 (distortion set to 0 for the stand-in)
TUM1 = dict(width=640, height=480, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, baseline=0.0)


def _rot(yaw: float, pitch: float, roll: float) -> np.ndarray:
    cy_, sy_ = math.cos(yaw), math.sin(yaw)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cr, sr = math.cos(roll), math.sin(roll)
    Ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]], dtype=np.float64)
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]], dtype=np.float64)
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]], dtype=np.float64)
    return Ry @ Rx @ Rz


@dataclass
class _Layer:
    depth: float
    scale: float          # metres per texel
    tex: np.ndarray       # uint8 (th, tw)
    alpha: np.ndarray     # bool  (th, tw); all True for the farthest layer


class SynthWorld:
    """Layered textured world + smooth camera trajectory.  Camera frame: x right, y down, z forward."""

    def __init__(self, width: int, height: int, fx: float, fy: float, cx: float, cy: float, baseline: float,
                 seed: int = 0, depths=(8.0, 15.0, 30.0, 60.0), n_rect: int = 4000, motion_scale: float = 1.0,
                 noise: int = 2):
        self.W, self.H = int(width), int(height)
        self.fx, self.fy, self.cx, self.cy, self.baseline = float(fx), float(fy), float(cx), float(cy), float(baseline)
        self.seed = int(seed)
        self.noise = int(noise)
        self.motion_scale = float(motion_scale)
        self.K = np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1]], dtype=np.float64)
        rng = np.random.Generator(np.random.PCG64([0x4C56545F30, self.seed]))
        self.layers: list[_Layer] = []
        tw, th = self.W + 900, self.H + 700
        far = max(depths)
        for d in sorted(depths, reverse=True):        # far -> near (painter's order)
            tex = self._make_texture(rng, tw, th, n_rect)
            if d == far:
                alpha = np.ones((th, tw), dtype=bool)
            else:
                alpha = np.zeros((th, tw), dtype=bool)
                for _ in range(int(28 * (tw * th) / (2141 * 1076)) + 6):
                    w = int(rng.integers(90, 380)); h = int(rng.integers(60, 260))
                    x = int(rng.integers(0, tw - w)); y = int(rng.integers(0, th - h))
                    alpha[y:y + h, x:x + w] = True
            self.layers.append(_Layer(depth=float(d), scale=float(d) / self.fx, tex=tex, alpha=alpha))
        self._torch_cache = None

    @staticmethod
    def _make_texture(rng, tw, th, n_rect):
        canvas = np.full((th, tw), 110, dtype=np.uint8)
        n = int(n_rect * (tw * th) / (2141 * 1076)) + 50
        xs = rng.integers(0, tw, size=n); ys = rng.integers(0, th, size=n)
        ws = rng.integers(6, 41, size=n); hs = rng.integers(6, 41, size=n)
        gs = rng.integers(0, 256, size=n)
        for x, y, w, h, g in zip(xs, ys, ws, hs, gs):
            canvas[y:y + h, x:x + w] = g
        # 3x3 box blur (integer arithmetic)
        c = canvas.astype(np.int32)
        p = np.pad(c, 1, mode="edge")
        s = sum(p[dy:dy + th, dx:dx + tw] for dy in range(3) for dx in range(3))
        return ((s + 4) // 9).astype(np.uint8)

    # ------------------------------------------------------------------ ground truth trajectory
    def pose(self, i: int):
        """camera-to-world (R, t) of the LEFT camera at frame i; identity at i = 0."""
        m = self.motion_scale
        t = np.array([1.2 * m * math.sin(2 * math.pi * i / 97.0),
                      0.25 * m * math.sin(2 * math.pi * i / 61.0),
                      1.5 * m * math.sin(2 * math.pi * i / 131.0)], dtype=np.float64)
        yaw = math.radians(4.0) * m * math.sin(2 * math.pi * i / 113.0)
        pitch = math.radians(1.5) * m * math.sin(2 * math.pi * i / 79.0)
        roll = math.radians(2.0) * m * math.sin(2 * math.pi * i / 149.0)
        return _rot(yaw, pitch, roll), t

    def _homographies(self, i: int, eye: int):
        R, t = self.pose(i)
        c = t + R @ np.array([self.baseline * eye, 0.0, 0.0])
        out = []
        for L in self.layers:
            th, tw = L.tex.shape
            s = L.scale
            M = np.array([[s, 0, -s * tw / 2.0 - c[0]],
                          [0, s, -s * th / 2.0 - c[1]],
                          [0, 0, L.depth - c[2]]], dtype=np.float64)
            Hm = self.K @ R.T @ M          # (a, b, 1) texture -> image
            out.append((np.linalg.inv(Hm), R, c))
        return out

    # ------------------------------------------------------------------ numpy renderer
    def render(self, i: int, eye: int = 0, want_depth: bool = False):
        H, W = self.H, self.W
        u = np.arange(W, dtype=np.float64)[None, :]
        v = np.arange(H, dtype=np.float64)[:, None]
        img = np.zeros((H, W), dtype=np.float64)
        depth = np.zeros((H, W), dtype=np.float64) if want_depth else None
        for L, (Hi, R, c) in zip(self.layers, self._homographies(i, eye)):
            th, tw = L.tex.shape
            wq = Hi[2, 0] * u + Hi[2, 1] * v + Hi[2, 2]
            a = (Hi[0, 0] * u + Hi[0, 1] * v + Hi[0, 2]) / wq
            b = (Hi[1, 0] * u + Hi[1, 1] * v + Hi[1, 2]) / wq
            a0 = np.floor(a); b0 = np.floor(b)
            ok = (a0 >= 0) & (a0 < tw - 1) & (b0 >= 0) & (b0 < th - 1)
            ai = np.clip(a0, 0, tw - 2).astype(np.int64); bi = np.clip(b0, 0, th - 2).astype(np.int64)
            fa = a - a0; fb = b - b0
            t = L.tex
            v00 = t[bi, ai].astype(np.float64); v01 = t[bi, ai + 1].astype(np.float64)
            v10 = t[bi + 1, ai].astype(np.float64); v11 = t[bi + 1, ai + 1].astype(np.float64)
            val = (v00 * (1 - fa) + v01 * fa) * (1 - fb) + (v10 * (1 - fa) + v11 * fa) * fb
            ar = np.clip(np.floor(a + 0.5), 0, tw - 1).astype(np.int64)
            br = np.clip(np.floor(b + 0.5), 0, th - 1).astype(np.int64)
            ok &= L.alpha[br, ar]
            img = np.where(ok, val, img)
            if want_depth:
                # camera-frame z of the plane point seen at this pixel
                Xw = (a - tw / 2.0) * L.scale - c[0]
                Yw = (b - th / 2.0) * L.scale - c[1]
                Zw = L.depth - c[2]
                zc = R[0, 2] * Xw + R[1, 2] * Yw + R[2, 2] * Zw
                depth = np.where(ok, zc, depth)
        if self.noise > 0:
            rng = np.random.Generator(np.random.PCG64([0x4E4F495345, self.seed, int(i), int(eye)]))
            img = img + rng.integers(-self.noise, self.noise + 1, size=(H, W)).astype(np.float64)
        out = np.clip(np.floor(img + 0.5), 0, 255).astype(np.uint8)
        if want_depth:
            return out, depth.astype(np.float32)
        return out

    def render_stereo(self, i: int):
        return self.render(i, 0), self.render(i, 1)

    def render_rgbd(self, i: int):
        return self.render(i, 0, want_depth=True)

    # ------------------------------------------------------------------ torch renderer (same math)
    def render_stereo_torch(self, i: int, device="cuda"):
        """Returns a (2, H, W) uint8 tensor on `device` (left, right).  Same math as render(); used by
        bench.py to pre-render hundreds of full-size frames directly into HBM."""
        return self._render_torch(i, (0, 1), device, False)

    def render_rgbd_torch(self, i: int, device="cuda"):
        """(gray u8 (H, W), depth f32 (H, W)) tensors on `device`: render_rgbd()'s math (the full-length RGB-D parity test renders 573 frames this way)"""
        g, d = self._render_torch(i, (0,), device, True)
        return g[0], d

    def _render_torch(self, i: int, eyes, device, want_depth):
        import torch
        if self._torch_cache is None or self._torch_cache[0] != str(device):
            lay = [(torch.from_numpy(L.tex).to(device), torch.from_numpy(L.alpha).to(device)) for L in self.layers]
            u = torch.arange(self.W, dtype=torch.float64, device=device)[None, :]
            v = torch.arange(self.H, dtype=torch.float64, device=device)[:, None]
            self._torch_cache = (str(device), lay, u, v)
        _, lay, u, v = self._torch_cache
        outs = []
        depth = None
        for eye in eyes:
            img = torch.zeros((self.H, self.W), dtype=torch.float64, device=device)
            if want_depth:
                depth = torch.zeros((self.H, self.W), dtype=torch.float64, device=device)
            for L, (tex, alpha), (Hi, R, c) in zip(self.layers, lay, self._homographies(i, eye)):
                th, tw = L.tex.shape
                wq = Hi[2, 0] * u + Hi[2, 1] * v + Hi[2, 2]
                a = (Hi[0, 0] * u + Hi[0, 1] * v + Hi[0, 2]) / wq
                b = (Hi[1, 0] * u + Hi[1, 1] * v + Hi[1, 2]) / wq
                a0 = torch.floor(a); b0 = torch.floor(b)
                ok = (a0 >= 0) & (a0 < tw - 1) & (b0 >= 0) & (b0 < th - 1)
                ai = a0.clamp(0, tw - 2).long(); bi = b0.clamp(0, th - 2).long()
                fa = a - a0; fb = b - b0
                v00 = tex[bi, ai].double(); v01 = tex[bi, ai + 1].double()
                v10 = tex[bi + 1, ai].double(); v11 = tex[bi + 1, ai + 1].double()
                val = (v00 * (1 - fa) + v01 * fa) * (1 - fb) + (v10 * (1 - fa) + v11 * fa) * fb
                ar = torch.floor(a + 0.5).clamp(0, tw - 1).long(); br = torch.floor(b + 0.5).clamp(0, th - 1).long()
                ok = ok & alpha[br, ar]
                img = torch.where(ok, val, img)
                if want_depth:
                    Xw = (a - tw / 2.0) * L.scale - c[0]
                    Yw = (b - th / 2.0) * L.scale - c[1]
                    Zw = L.depth - c[2]
                    depth = torch.where(ok, R[0, 2] * Xw + R[1, 2] * Yw + R[2, 2] * Zw, depth)
            if self.noise > 0:
                rng = np.random.Generator(np.random.PCG64([0x4E4F495345, self.seed, int(i), int(eye)]))
                nz = torch.from_numpy(rng.integers(-self.noise, self.noise + 1, size=(self.H, self.W)).astype(np.float64))
                img = img + nz.to(device)
            outs.append(torch.floor(img + 0.5).clamp(0, 255).to(torch.uint8))
        if want_depth:
            return torch.stack(outs, 0), depth.to(torch.float32)
        return torch.stack(outs, 0)


def make_world(kind: str = "kitti", seed: int = 0, scale: float = 1.0, **kw) -> SynthWorld:
    """kind in {kitti, euroc, tum}.  `scale` < 1 shrinks the image (and intrinsics) for fast CPU tests."""
    base = dict({"kitti": KITTI, "euroc": EUROC, "tum": TUM1}[kind])
    if scale != 1.0:
        base["width"] = int(round(base["width"] * scale)); base["height"] = int(round(base["height"] * scale))
        for k in ("fx", "fy", "cx", "cy"):
            base[k] = base[k] * scale
    if kind == "tum":
        kw.setdefault("depths", (1.2, 2.0, 3.0, 4.5))
        kw.setdefault("motion_scale", 0.15)
    if kind == "euroc":
        kw.setdefault("depths", (3.0, 5.0, 8.0, 14.0))
        kw.setdefault("motion_scale", 0.3)
    return SynthWorld(seed=seed, **base, **kw)
