"""Shared helpers of the test-suite: backend loading (product / oracle / reference shim), scenario
runner and state comparison.

Only test code may load anything from oracle/ (see oracle/itm_oracle.cpp header).
"""
from __future__ import annotations

import os
import subprocess
import sys
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from infinitam_amd import capi, synth  # noqa: E402
from infinitam_amd.capi import (BUF_ALLOCATION_LIST, BUF_EXCESS_LIST, BUF_HASH_ENTRIES,  # noqa: E402
                                BUF_RANGE_IMAGE, BUF_RAYCAST_IMAGE, BUF_RAYCAST_RESULT,
                                BUF_VISIBLE_IDS, BUF_VISIBLE_TYPE, BUF_VOXEL_BLOCKS, INDEX_DENSE,
                                INDEX_HASH, VOXEL_F, VOXEL_F_RGB, VOXEL_S, VOXEL_S_RGB, Backend,
                                DevBuffer, View, default_params)

ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libitm_oracle.so")
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libitm_ref.so")
ORACLE_OMP_LIB = os.path.join(ORACLE_DIR, "libitm_oracle_omp.so")      # timing only (bench.py all-cores baseline)
REF_POOL40000_LIB = os.path.join(ORACLE_DIR, "_ref", "libitm_ref_pool40000.so")   # reference built with SDF_LOCAL_BLOCK_NUM=0x40000
REF_OMP_LIB = os.path.join(ORACLE_DIR, "_ref", "libitm_ref_omp.so")    # timing only
REFERENCE_TREE = "/root/reference/InfiniTAM"
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

_cache = {}


def oracle_backend() -> Backend:
    if "oracle" not in _cache:
        if not os.path.exists(ORACLE_LIB):
            subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
        _cache["oracle"] = Backend(ORACLE_LIB, "itmo_")
    return _cache["oracle"]


def oracle_omp_backend() -> Optional[Backend]:
    """The OpenMP build of the oracle (timing only, never a parity checker); None if it cannot be built."""
    if "oracle_omp" not in _cache:
        if not os.path.exists(ORACLE_OMP_LIB):
            subprocess.run(["make", "-C", ORACLE_DIR, "libitm_oracle_omp.so"], capture_output=True)
        _cache["oracle_omp"] = Backend(ORACLE_OMP_LIB, "itmo_") if os.path.exists(ORACLE_OMP_LIB) else None
    return _cache["oracle_omp"]


def reference_backend() -> Optional[Backend]:
    """The reference's own CPU engines behind the same ABI; None where /root/reference is absent
    and no prebuilt oracle/_ref exists."""
    if "ref" not in _cache:
        if not os.path.exists(REF_LIB) and os.path.isdir(REFERENCE_TREE):
            subprocess.run(["make", "-C", ORACLE_DIR, "ref"], check=True, capture_output=True)
        _cache["ref"] = Backend(REF_LIB, "itmr_") if os.path.exists(REF_LIB) else None
    return _cache["ref"]


def reference_pool40000_backend() -> Optional[Backend]:
    """The reference's CPU engines compiled with the upstream pool size 0x40000 (oracle/Makefile target ref40000)."""
    if "ref40000" not in _cache:
        if not os.path.exists(REF_POOL40000_LIB) and os.path.isdir(REFERENCE_TREE):
            subprocess.run(["make", "-C", ORACLE_DIR, "ref40000"], check=True, capture_output=True)
        _cache["ref40000"] = Backend(REF_POOL40000_LIB, "itmr_") if os.path.exists(REF_POOL40000_LIB) else None
    return _cache["ref40000"]


def hip_backend() -> Backend:
    import infinitam_amd
    alt = os.environ.get("ITM_TEST_LIB")          # development: run the suite against another build of the product library
    if alt:
        if "hip_alt" not in _cache:
            _cache["hip_alt"] = Backend(alt, "itm_")
        return _cache["hip_alt"]
    return infinitam_amd.load()


@dataclass
class Scenario:
    """One deterministic sequence."""
    name: str = "hash_s_5mm"
    w: int = 640
    h: int = 480
    voxelType: int = VOXEL_S
    indexType: int = INDEX_HASH
    voxelSize: float = 0.005
    mu: float = 0.02
    maxW: int = 100
    stopIntegratingAtMaxW: bool = False
    frames: int = 3
    trajectory: str = "parity"      # "parity" | "bench" | "yaw"
    stream: int = 0
    localBlockNum: int = 0
    bucketNum: int = 0
    excessNum: int = 0
    denseSize: tuple = (0, 0, 0)
    denseOffset: Optional[tuple] = None
    maxRenderingBlocks: int = 0
    colour: bool = False
    noise_seed: Optional[int] = None
    origin: tuple = (0.0, 0.0, 0.0)   # metres added to every camera position: the scene and its trajectory far from the world origin (same depth images)
    frame_stride: int = 1             # frame k of the scenario is position k * frame_stride of its trajectory (bench.py's config 5 keeps every 4th pose)
    yaw_rate: float = 0.02            # "yaw": radians per frame

    def params(self):
        return default_params(self.voxelSize, self.mu, self.maxW, 0.35, 3.0, self.stopIntegratingAtMaxW)

    def intr(self):
        return synth.intrinsics_for(self.w, self.h)

    def position(self, k):
        k = k * self.frame_stride
        if self.trajectory == "bench":
            return synth.bench_position(k, self.stream)
        return synth.parity_position(k, self.stream)

    def pose(self, k):
        t = self.position(k)
        if any(self.origin):
            t = tuple(np.float32(a) + np.float32(b) for a, b in zip(t, self.origin))
        if self.trajectory == "yaw":
            return synth.pose_matrix_yaw(t, self.yaw_rate * k)
        return synth.pose_matrix(t)

    def depth(self, k):
        t = self.position(k)
        if self.trajectory == "yaw":
            # depth rendered from the translated camera only; the rotation just changes the pose fed
            # to the engines (any consistent input is fine for parity)
            pass
        if self.noise_seed is not None:
            return synth.depth_from_raw(synth.raw_depth_mm(self.w, self.h, t, self.intr(), self.noise_seed + k))
        return synth.depth_frame(self.w, self.h, t, self.intr())


@dataclass
class RunResult:
    counters: list = field(default_factory=list)       # per frame dict
    hash: Optional[np.ndarray] = None
    voxels: Optional[np.ndarray] = None
    excess: Optional[np.ndarray] = None
    alloc_list: Optional[np.ndarray] = None
    visible_ids: Optional[np.ndarray] = None
    visible_type: Optional[np.ndarray] = None
    range_image: Optional[np.ndarray] = None
    raycast: Optional[np.ndarray] = None
    image: Optional[np.ndarray] = None
    points: Optional[np.ndarray] = None
    normals: Optional[np.ndarray] = None
    per_frame: list = field(default_factory=list)       # optional per-frame snapshots


class Session:
    """A scene + render state + I/O buffers on one backend."""

    def __init__(self, be: Backend, sc: Scenario, deferred_fusion=True):
        self.be, self.sc = be, sc
        self.scene = be.create_scene(sc.voxelType, sc.indexType, sc.params(), bucketNum=sc.bucketNum,
                                     excessNum=sc.excessNum, localBlockNum=sc.localBlockNum,
                                     denseSize=sc.denseSize, denseOffset=sc.denseOffset,
                                     maxRenderingBlocks=sc.maxRenderingBlocks)
        self.scene.reco.ResetScene()
        # the sessions of the suite accept the recording contract (include/itm_hip.h, itm_scene_set_deferred_fusion): with fused=False
        # every call is flushed before the next, with fused="four" the calls are recorded; tests/test_deferred_fusion.py covers a
        # scene that never asked (the default)
        self.enable_deferred_fusion(deferred_fusion)
        self.rs = self.scene.vis.CreateRenderState((sc.w, sc.h))
        P = sc.w * sc.h
        self.points = DevBuffer(be, P * 16, np.float32, (sc.h, sc.w, 4))
        self.normals = DevBuffer(be, P * 16, np.float32, (sc.h, sc.w, 4))
        self.rgb = be.to_backend(synth.rgb_frame(sc.w, sc.h)) if sc.colour else None
        self._depth = None

    def enable_deferred_fusion(self, on=True):
        self.scene.set_deferred_fusion(on)

    def view(self, k) -> View:
        sc = self.sc
        self._depth = self.be.to_backend(sc.depth(k))
        return View(self._depth, sc.w, sc.h, M_d=sc.pose(k), intr_d=sc.intr(), rgb=self.rgb,
                    w_rgb=sc.w, h_rgb=sc.h, intr_rgb=sc.intr())

    def frame(self, k, fused=False) -> View:
        """fused=True: itm_process_frame.  fused=False: the four engine calls, each LAUNCHED before the next is made (itm_flush in
        between: the separate kernels of every call).  fused="four": the four calls back to back, as ITMMainEngine::ProcessFrame
        issues them -- the product records the first three and launches the fused frame at the fourth (pending.hip)."""
        v = self.view(k)
        s, rs = self.scene, self.rs
        if fused is True:
            s.process_frame(v, rs, self.points, self.normals)
        else:
            sep = fused is False
            s.reco.AllocateSceneFromDepth(v, rs)
            if sep: s.flush(rs)
            s.reco.IntegrateIntoScene(v, rs)
            if sep: s.flush(rs)
            s.vis.CreateExpectedDepths(v.M_d, v.intr_d, rs)
            if sep: s.flush(rs)
            s.vis.CreateICPMaps(v, rs, self.points, self.normals)
        return v

    def snapshot(self, with_voxels=True) -> RunResult:
        s, rs = self.scene, self.rs
        r = RunResult()
        r.counters = [s.counters(rs)]
        if s.is_hash:
            r.hash = s.download(BUF_HASH_ENTRIES)
            r.excess = s.download(BUF_EXCESS_LIST)
            r.visible_ids = s.download(BUF_VISIBLE_IDS, rs)
            r.visible_type = s.download(BUF_VISIBLE_TYPE, rs)
        r.alloc_list = s.download(BUF_ALLOCATION_LIST)
        if with_voxels:
            r.voxels = s.download(BUF_VOXEL_BLOCKS)
        r.range_image = s.download(BUF_RANGE_IMAGE, rs)
        r.raycast = s.download(BUF_RAYCAST_RESULT, rs)
        r.image = s.download(BUF_RAYCAST_IMAGE, rs)
        r.points = self.points.numpy()
        r.normals = self.normals.numpy()
        return r

    def close(self):
        self.rs.close()
        self.scene.close()


def run_scenario(be: Backend, sc: Scenario, fused=False, with_voxels=True, per_frame_hook=None) -> RunResult:
    ses = Session(be, sc)
    counters = []
    for k in range(sc.frames):
        ses.frame(k, fused=fused)
        counters.append(ses.scene.counters(ses.rs))
        if per_frame_hook:
            per_frame_hook(k, ses)
    res = ses.snapshot(with_voxels=with_voxels)
    res.counters = counters
    ses.close()
    return res


def assert_fields_equal(a: np.ndarray, b: np.ndarray, what: str):
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    for name in a.dtype.names:
        if not np.array_equal(a[name], b[name]):
            bad = np.nonzero(np.any((a[name] != b[name]).reshape(len(a), -1), axis=1))[0]
            raise AssertionError(f"{what}.{name}: {len(bad)} entries differ, first at {bad[:5]}: "
                                 f"{a[name][bad[:5]]} vs {b[name][bad[:5]]}")


def range_region(img: np.ndarray, w: int, h: int) -> np.ndarray:
    """The part of the range image the raycaster reads: [0, ceil(W/8)) x [0, ceil(H/8))."""
    return img[: (h + 7) // 8, : (w + 7) // 8]


def compare_results(a: RunResult, b: RunResult, sc: Scenario, exact=True, what=""):
    """Bit-exact comparison of two runs (integer state always exact; float maps exact when both
    sides were built without FP contraction, which is the configuration of this repo).
    Stated tolerances for the float maps when exact=False: raycast xyz 1e-3 voxel, points 1e-5 m,
    normals 1e-4, grey +-1."""
    tag = f"[{what or sc.name}] "
    for c in list(a.counters) + list(b.counters):
        assert (c.get("statusFlags", 0) & 2) == 0, tag + "the one-pass visible list gave up waiting for a predecessor (statusFlags bit 1)"
    for ca, cb in zip(a.counters, b.counters):
        for key in ("lastFreeBlockId", "lastFreeExcessListId", "noVisibleEntries", "noRenderingBlocks"):
            if key == "noRenderingBlocks" and (ca[key] == 0 or cb[key] == 0):
                continue  # the reference shim cannot report it
            assert ca[key] == cb[key], f"{tag}counter {key}: {ca[key]} vs {cb[key]} ({ca} vs {cb})"
    if a.hash is not None:
        assert_fields_equal(a.hash, b.hash, tag + "hash")
        assert np.array_equal(a.excess, b.excess), tag + "excess list"
        nv = a.counters[-1]["noVisibleEntries"]
        assert np.array_equal(a.visible_ids[:nv], b.visible_ids[:nv]), tag + "visible ids"
        if not np.array_equal(a.visible_type, b.visible_type):
            d = np.nonzero(a.visible_type != b.visible_type)[0]
            raise AssertionError(f"{tag}visible types: {len(d)} differ, slots {d[:8].tolist()} have {a.visible_type[d[:8]].tolist()} vs {b.visible_type[d[:8]].tolist()}; "
                                 f"entries there {a.hash[d[:4]].tolist()} vs {b.hash[d[:4]].tolist()}")
    assert np.array_equal(a.alloc_list, b.alloc_list), tag + "allocation list"
    if a.voxels is not None and b.voxels is not None:
        assert_fields_equal(a.voxels, b.voxels, tag + "voxels")
    ra, rb = range_region(a.range_image, sc.w, sc.h), range_region(b.range_image, sc.w, sc.h)
    assert np.array_equal(ra, rb), tag + "range image (raycast-visible region)"
    wa, wb = a.raycast[..., 3], b.raycast[..., 3]
    assert np.array_equal(wa, wb), tag + f"raycast hit mask: {(wa != wb).sum()} pixels differ"
    if exact:
        assert np.array_equal(a.raycast, b.raycast), tag + "raycast result"
        assert np.array_equal(a.points, b.points), tag + "ICP points"
        assert np.array_equal(a.normals, b.normals), tag + "ICP normals"
        assert np.array_equal(a.image, b.image), tag + "raycast image"
    else:
        hit = wa > 0
        assert np.abs(a.raycast[hit][:, :3] - b.raycast[hit][:, :3]).max(initial=0) <= 1e-3, tag + "raycast xyz"
        assert np.array_equal(a.points[..., 3], b.points[..., 3]), tag + "ICP validity"
        assert np.abs(a.points - b.points).max() <= 1e-5, tag + "ICP points"
        assert np.abs(a.normals - b.normals).max() <= 1e-4, tag + "ICP normals"
        assert np.abs(a.image.astype(int) - b.image.astype(int)).max() <= 1, tag + "grey image"
