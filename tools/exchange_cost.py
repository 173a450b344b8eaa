#!/usr/bin/env python3
"""Where the cost of a per-frame exchange goes: the bench frame + itm_exchange_step with parts of the hand-off switched off
(ITM_EXCHANGE_EXPERIMENT, read when the exchange is created), frames/s and host time per frame for each.
usage: python tools/exchange_cost.py [frames]      (measurement tool)"""
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
from infinitam_amd.streams import NativeExchange  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
W, H = 640, 480
be = itm.load()
intr = synth.intrinsics_for(W, H)
scene = be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.004), localBlockNum=0x40000)
scene.reco.ResetScene()
rs = scene.vis.CreateRenderState((W, H))
pts = capi.DevBuffer(be, W * H * 16); nrm = capi.DevBuffer(be, W * H * 16)
frames = [be.to_backend(synth.depth_frame(W, H, synth.bench_position(k), intr)) for k in range(100)]
views = [capi.View(frames[k], W, H, M_d=synth.pose_matrix(synth.bench_position(k)), intr_d=intr).struct() for k in range(100)]
poses = [(C.c_float * 16)(*[float(x) for x in synth.pose_matrix(synth.bench_position(k))]) for k in range(100)]
fn = be.fn["process_frame"]
sh, rh, pp, np_ = C.c_void_p(scene.h), C.c_void_p(rs.h), C.c_void_p(pts.ptr), C.c_void_p(nrm.ptr)

NAMES = {-1: "no exchange", 1: "record copy kernel only", 2: "+ event on the frame stream", 3: "+ side stream waits for it",
         4: "+ collective on the side stream, no release event", 0: "full hand-off", 5: "collective on the FRAME stream, no events"}
for exp in (-1, 1, 2, 3, 4, 0, 5):
    for copy in ("0", "1"):
        if exp in (-1, 1, 2, 3) and copy == "1":
            continue
        os.environ["ITM_EXCHANGE_EXPERIMENT"] = str(max(exp, 0))
        os.environ["ITM_EXCHANGE_DEVICE_COPY"] = copy
        ex = NativeExchange(be, 1, 0, 16384, batch=1) if exp >= 0 else None
        for k in range(40):
            fn(sh, C.byref(views[k % 100]), rh, pp, np_, None)
            if ex: ex.step(rs.h, poses[k % 100], None)
        be.sync()
        t0 = time.perf_counter(); host = 0.0
        for k in range(40, 40 + N):
            fn(sh, C.byref(views[k % 100]), rh, pp, np_, None)
            h0 = time.perf_counter()
            if ex: ex.step(rs.h, poses[k % 100], None)
            host += time.perf_counter() - h0
        submit = time.perf_counter() - t0
        be.sync()
        dt = time.perf_counter() - t0
        print(json.dumps({"experiment": NAMES[exp], "collective": ("device copy" if copy == "1" else "ncclAllGather") if exp in (4, 0, 5) else None,
                          "fps": round(N / dt, 1), "us_per_frame": round(dt / N * 1e6, 1), "host_submit_us_per_frame": round(submit / N * 1e6, 1),
                          "host_us_in_exchange_step": round(host / N * 1e6, 1)}), flush=True)
        if ex: ex.close()
