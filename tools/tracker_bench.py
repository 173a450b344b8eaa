#!/usr/bin/env python3
"""Cost of one cost / gradient / Hessian evaluation of the ICP tracker per pyramid level, and of whole TrackCamera calls on
the bench workload (measurement tool).  usage: python tools/tracker_bench.py"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import infinitam_amd as itm
from infinitam_amd import capi, synth
from infinitam_amd.capi import DevBuffer, TrackerConfig, TrackerGH
be = capi.Backend(os.environ["ITM_LIB"], "itm_") if os.environ.get("ITM_LIB") else itm.load()
W, H = 640, 480
intr = synth.intrinsics_for(W, H)
scene = be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.004), localBlockNum=0x40000)
scene.reco.ResetScene()
rs = scene.vis.CreateRenderState((W, H))
pts = DevBuffer(be, W * H * 16); nrm = DevBuffer(be, W * H * 16)
for k in range(10):
    t = synth.bench_position(k)
    d = be.to_backend(synth.depth_frame(W, H, t, intr))
    v = capi.View(d, W, H, M_d=synth.pose_matrix(t), intr_d=intr)
    scene.process_frame(v, rs, pts, nrm)
be.sync()
nxt = be.to_backend(synth.depth_frame(W, H, synth.bench_position(10), intr))
def fp(a):
    a = np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1)); return a, a.ctypes.data_as(C.POINTER(C.c_float))
M = synth.pose_matrix(synth.bench_position(9)).astype(np.float32)
inv = np.linalg.inv(M.reshape(4, 4).T).T.astype(np.float32).reshape(16)
levels = [(nxt, W, H, np.array(intr, np.float32))]
for _ in range(4):
    pd, pw, ph, pi = levels[-1]
    nd = DevBuffer(be, (pw // 2) * (ph // 2) * 4)
    be.check(be.fn["filter_subsample_with_holes"](pd.ptr, pw, ph, nd.ptr, None), "sub")
    levels.append((nd, pw // 2, ph // 2, pi * np.float32(0.5)))
be.sync()
_, si = fp(intr); _, ip = fp(inv); _, sp = fp(M)
out = TrackerGH()
for lev, (ld, lw, lh, li) in enumerate(levels):
    _, vi = fp(li)
    for mode in (3, 1):
        f = lambda: be.check(be.fn["tracker_compute_g_and_h"](ld.ptr, lw, lh, vi, pts.ptr, nrm.ptr, W, H, si, ip, sp, 0.01, mode, C.byref(out), None), "gh")
        for _ in range(5): f()
        t0 = time.perf_counter()
        for _ in range(50): f()
        dt = (time.perf_counter() - t0) / 50
        print(f"level {lev} {lw}x{lh} mode {mode}: {dt*1e6:7.1f} us per evaluation, valid {out.noValidPoints}")
cfg = TrackerConfig.default()
view = capi.View(nxt, W, H, M_d=M, intr_d=intr).struct()
res = (C.c_float * 16)()
g = lambda: be.check(be.fn["track_camera"](C.byref(cfg), C.byref(view), pts.ptr, nrm.ptr, sp, res, None), "track")
for _ in range(5): g()
t0 = time.perf_counter()
for _ in range(50): g()
print(f"track_camera: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per call")
