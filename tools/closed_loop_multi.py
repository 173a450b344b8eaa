#!/usr/bin/env python3
"""k closed tracking + mapping loops on one GPU (k scenes, k HIP streams, k tracker handles, one host thread each): raw short depth
in HBM -> itm_update_view -> itm_tracker_track_camera -> itm_process_frame, no external poses.  The single loop leaves the GPU idle
most of the time (the tracker is a host <-> device ping-pong, the ray cast a few long chains); this measures how much of that k
independent sensors fill.  usage: python tools/closed_loop_multi.py [k=4] [frames=100]      (measurement tool)"""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import infinitam_amd as itm  # noqa: E402
from infinitam_amd import capi, synth  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W, H = 640, 480
be = capi.Backend(os.environ["ITM_LIB"], "itm_") if os.environ.get("ITM_LIB") else itm.load()
intr = synth.intrinsics_for(W, H)
ip = (C.c_float * 4)(*intr)
raws = [be.to_backend(np.round(synth.depth_frame(W, H, synth.bench_position(k), intr) * 1000.0).astype(np.int16)) for k in range(N)]
gt = synth.pose_matrix(synth.bench_position(N - 1))


class Loop:
    def __init__(self):
        self.scene = be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.004), localBlockNum=0x40000)
        self.scene.reco.ResetScene()
        self.rs = self.scene.vis.CreateRenderState((W, H))
        self.pts = capi.DevBuffer(be, W * H * 16); self.nrm = capi.DevBuffer(be, W * H * 16)
        self.depth = capi.DevBuffer(be, W * H * 4); self.scratch = capi.DevBuffer(be, W * H * 4)
        self.stream = C.c_void_p(); be.check(be.fn["stream_create"](C.byref(self.stream)), "stream_create")
        self.tracker = C.c_void_p(); be.check(be.fn["tracker_create"](C.byref(self.tracker)), "tracker_create")
        self.cfg = capi.TrackerConfig.default()
        self.view = capi.View(self.depth, W, H, M_d=synth.pose_matrix(synth.bench_position(0)), intr_d=intr).struct()
        self.out = (C.c_float * 16)()
        self.Mview = np.ctypeslib.as_array(self.view.M_d); self.Mout = np.ctypeslib.as_array(self.out)
        self.err = None

    def run(self, gate):
        try:
            outp = C.cast(self.out, C.POINTER(C.c_float)); mp = C.cast(self.view.M_d, C.POINTER(C.c_float))
            gate.wait()
            for k in range(N):
                be.check(be.fn["update_view"](raws[k].ptr, W, H, 1, 0.001, 0.0, ip, 0, 0, self.depth.ptr, self.scratch.ptr, None, None, self.stream), "update_view")
                if k > 0:
                    be.check(be.fn["tracker_track_camera"](self.tracker, C.byref(self.cfg), C.byref(self.view), self.pts.ptr, self.nrm.ptr, mp, outp, self.stream), "track")
                    self.Mview[:] = self.Mout
                self.scene.process_frame(self.view, self.rs, self.pts, self.nrm, stream=self.stream.value)
            be.sync(self.stream.value)
        except Exception as e:      # noqa: BLE001
            self.err = e


for k_loops in sorted({1, 2, K}):
    loops = [Loop() for _ in range(k_loops)]
    gate = threading.Barrier(k_loops + 1)
    threads = [threading.Thread(target=l.run, args=(gate,)) for l in loops]
    for t in threads:
        t.start()
    be.sync()
    gate.wait()
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    errs = [l.err for l in loops if l.err]
    if errs:
        raise errs[0]
    print(json.dumps({"closed_loops": k_loops, "frames_each": N, "aggregate_fps": round(k_loops * N / dt, 1), "per_loop_fps": round(N / dt, 1),
                      "final_translation_error_m": [round(float(np.abs(l.Mview[12:15] - gt[12:15]).max()), 5) for l in loops]}))
    for l in loops:
        be.check(be.fn["tracker_destroy"](l.tracker), "tracker_destroy")
        l.scene.close()
    del loops
