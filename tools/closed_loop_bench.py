#!/usr/bin/env python3
"""Closed tracking + mapping loop on the bench workload without external poses: raw short depth in HBM ->
itm_update_view (conversion, optional 5-pass bilateral filter) -> itm_track_camera (ICP against the previous
frame's maps) -> itm_process_frame (allocate, integrate, expected depths, ICP maps).  Prints per-stage wall times.
usage: python tools/closed_loop_bench.py [frames]      (measurement tool)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import infinitam_amd as itm  # noqa: E402
from infinitam_amd import capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W, H = 640, 480
be = capi.Backend(os.environ["ITM_LIB"], "itm_") if os.environ.get("ITM_LIB") else itm.load()
intr = synth.intrinsics_for(W, H)
scene = be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.004), localBlockNum=0x40000)
scene.reco.ResetScene()
rs = scene.vis.CreateRenderState((W, H))
pts = capi.DevBuffer(be, W * H * 16); nrm = capi.DevBuffer(be, W * H * 16)
raws = [be.to_backend(np.round(synth.depth_frame(W, H, synth.bench_position(k), intr) * 1000.0).astype(np.int16)) for k in range(N)]
depth = capi.DevBuffer(be, W * H * 4); scratch = capi.DevBuffer(be, W * H * 4)
ip = (C.c_float * 4)(*intr)
cfg = capi.TrackerConfig.default()
# one view structure and one result array for the whole run, the pose copied in place: the stage times below are the library's,
# not those of rebuilding ctypes structures in Python (~25 us per structure)
view = capi.View(depth, W, H, M_d=synth.pose_matrix(synth.bench_position(0)), intr_d=intr).struct()
out = (C.c_float * 16)()
outp = C.cast(out, C.POINTER(C.c_float))
Mview = np.ctypeslib.as_array(view.M_d)          # float32[16] aliasing the structure's pose
Mout = np.ctypeslib.as_array(out)


def track_and_map(k, sync):
    if k > 0:
        be.check(be.fn["track_camera"](C.byref(cfg), C.byref(view), pts.ptr, nrm.ptr, C.cast(view.M_d, C.POINTER(C.c_float)), outp, None), "track")
        Mview[:] = Mout
    if sync:
        be.sync()
    t = time.perf_counter()
    scene.process_frame(view, rs, pts, nrm)
    return t


for bilateral in (0, 1):
    scene.reco.ResetScene()
    Mview[:] = synth.pose_matrix(synth.bench_position(0)).astype(np.float32).reshape(16)
    t = {"update_view": 0.0, "track": 0.0, "map": 0.0}
    for k in range(N):
        t0 = time.perf_counter()
        be.check(be.fn["update_view"](raws[k].ptr, W, H, 1, 0.001, 0.0, ip, bilateral, 0, depth.ptr, scratch.ptr, None, None, None), "update_view")
        be.sync(); t1 = time.perf_counter()
        t2 = track_and_map(k, True)
        be.sync(); t3 = time.perf_counter()
        if k >= 5:
            t["update_view"] += t1 - t0; t["track"] += t2 - t1; t["map"] += t3 - t2
    n = N - 5
    gt = synth.pose_matrix(synth.bench_position(N - 1))
    err = float(np.abs(Mview[12:15] - gt[12:15]).max())
    tot = sum(t.values()) / n
    # the same loop without the per-stage synchronisations (they only exist for the breakdown above): the tracker's
    # own wait is the only host <-> device rendezvous per frame
    scene.reco.ResetScene()
    Mview[:] = synth.pose_matrix(synth.bench_position(0)).astype(np.float32).reshape(16)
    be.sync(); tp0 = None
    for k in range(N):
        if k == 5:
            be.sync(); tp0 = time.perf_counter()
        be.check(be.fn["update_view"](raws[k].ptr, W, H, 1, 0.001, 0.0, ip, bilateral, 0, depth.ptr, scratch.ptr, None, None, None), "update_view")
        track_and_map(k, False)
    be.sync(); pipelined = (time.perf_counter() - tp0) / n
    err2 = float(np.abs(Mview[12:15] - gt[12:15]).max())
    print(json.dumps({"bilateral": bool(bilateral), "frames": N, "pipelined_fps": round(1.0 / pipelined, 1), "pipelined_final_error_m": round(err2, 5), "ms_per_frame": round(tot * 1e3, 4), "fps": round(1.0 / tot, 1),
                      "stage_us": {k: round(v / n * 1e6, 1) for k, v in t.items()}, "final_translation_error_m": round(err, 5)}))
