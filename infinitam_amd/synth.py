"""Deterministic synthetic inputs for parity tests and the benchmark.

Scene: a sphere of radius 0.5 m centred at (0, 0, 1.5) m in front of a wall z = 2.5 m (world frame =
camera-0 frame).  Everything is computed in float32 with numpy (no FMA, fixed operation order), and
the fixtures carry SHA-256 digests of the generated frames so that generator drift is caught before
it is mistaken for a kernel bug.  This is host-side input plumbing, not part of the hot path.
"""
from __future__ import annotations

import hashlib

import numpy as np

F = np.float32


def intrinsics_for(w: int, h: int):
    """(fx, fy, cx, cy): the reference default at 640x480 (Objects/ITMIntrinsics.h:45-50), scaled."""
    s = w / 640.0
    return (580.0 * s, 580.0 * s, 320.0 * s, 240.0 * s)


def pose_matrix(t) -> np.ndarray:
    """World->camera M_d for a camera at position t with identity rotation (column-major 16 floats)."""
    m = np.eye(4, dtype=F)
    m[0, 3], m[1, 3], m[2, 3] = F(-F(t[0])), F(-F(t[1])), F(-F(t[2]))
    return np.ascontiguousarray(m.T).reshape(16).copy()  # column-major storage


def pose_matrix_yaw(t, yaw: float) -> np.ndarray:
    """World->camera pose with a rotation about the y axis (exercises Matrix4::inv)."""
    c, s = F(np.cos(F(yaw))), F(np.sin(F(yaw)))
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=F)      # camera->world rotation
    Rt = R.T
    tt = -(Rt @ np.asarray(t, F))
    m = np.eye(4, dtype=F)
    m[:3, :3] = Rt
    m[:3, 3] = tt.astype(F)
    return np.ascontiguousarray(m.T).reshape(16).copy()


def parity_position(k: int, stream: int = 0):
    """Parity trajectory: camera k at (0.01 k, 0, 0), stream g shifted by 0.05 g in y."""
    return (F(0.01) * F(k), F(0.05) * F(stream), F(0.0))


def _tri(k: int) -> int:
    return abs(((k + 25) % 100) - 50) - 25


def bench_position(k: int, stream: int = 0):
    """Benchmark trajectory: bounded triangle waves, exact in fp32 (period 100 frames)."""
    return (F(0.004) * F(_tri(k)), F(F(0.002) * F(_tri(2 * k))) + F(0.05) * F(stream), F(0.0))


def depth_z(w: int, h: int, t, intr=None) -> np.ndarray:
    """Ideal z-depth image (metres, float32) of the sphere+wall scene from camera position t."""
    fx, fy, cx, cy = [F(v) for v in (intr or intrinsics_for(w, h))]
    xs = np.arange(w, dtype=F)[None, :]
    ys = np.arange(h, dtype=F)[:, None]
    dx = (xs - cx) / fx
    dy = (ys - cy) / fy
    ox, oy, oz = F(t[0]), F(t[1]), F(F(t[2]) - F(1.5))
    A = (dx * dx + dy * dy) + F(1.0)
    Bq = F(2.0) * ((ox * dx + oy * dy) + oz)
    Cq = F(F(F(ox * ox) + F(oy * oy)) + F(oz * oz)) - F(0.25)
    disc = Bq * Bq - F(4.0) * A * Cq
    with np.errstate(invalid="ignore"):
        tau = (-Bq - np.sqrt(np.where(disc > 0, disc, F(0.0)).astype(F))) / (F(2.0) * A)
    z = np.where((disc > 0) & (tau > 0), tau, F(2.5)).astype(F)
    return np.ascontiguousarray(np.broadcast_to(z, (h, w)))


def raw_depth_mm(w, h, t, intr=None, noise_seed=None) -> np.ndarray:
    """Raw sensor frame: int16 millimetres (affine calib 1/1000, 0; Objects/ITMDisparityCalib.h:40-45).
    Optional +-2 mm LCG noise (s = s*1664525 + 1013904223)."""
    z = depth_z(w, h, t, intr)
    raw = (z * F(1000.0)).astype(np.int16)
    if noise_seed is not None:
        n = w * h
        s = np.uint32(noise_seed)
        out = np.empty(n, np.int16)
        state = np.empty(n, np.uint32)
        cur = int(s)
        for i in range(n):  # small images only; the benchmark uses noise-free frames
            cur = (cur * 1664525 + 1013904223) & 0xFFFFFFFF
            state[i] = cur
        out = ((state >> 16) % 5).astype(np.int16) - 2
        raw = (raw.reshape(-1) + out).astype(np.int16).reshape(h, w)
    return np.ascontiguousarray(raw)


def depth_from_raw(raw: np.ndarray) -> np.ndarray:
    """Host mirror of convertDepthAffineToFloat for building float inputs directly."""
    r = raw.astype(F)
    out = r * F(0.001) + F(0.0)
    out[(raw <= 0) | (raw > 32000)] = F(-1.0)
    return np.ascontiguousarray(out.astype(F))


def depth_frame(w, h, t, intr=None) -> np.ndarray:
    return depth_from_raw(raw_depth_mm(w, h, t, intr))


def rgb_frame(w, h) -> np.ndarray:
    """Colour test pattern (x & 255, y & 255, (x ^ y) & 255, 255) as uchar4."""
    xs = np.arange(w, dtype=np.int32)[None, :]
    ys = np.arange(h, dtype=np.int32)[:, None]
    img = np.empty((h, w, 4), np.uint8)
    img[..., 0] = xs & 255
    img[..., 1] = ys & 255
    img[..., 2] = (xs ^ ys) & 255
    img[..., 3] = 255
    return img


def sha256(arr: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()
