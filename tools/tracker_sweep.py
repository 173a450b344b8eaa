#!/usr/bin/env python3
"""TrackCamera from many perturbed starting poses, level counts and image sizes: the resident evaluation kernel and the
launch-per-evaluation path (debug key 10) must return identical poses, bit for bit.  usage: python tools/tracker_sweep.py [cases=40]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import itm_testlib as T  # noqa: E402
from infinitam_amd import capi  # noqa: E402
from infinitam_amd.capi import TrackerConfig  # noqa: E402
from itm_testlib import Scenario  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
hip = T.hip_backend()
rng = np.random.default_rng(11)
bad = 0
for size in ((640, 480), (328, 248), (208, 152)):
    sc = Scenario(name="trk_sweep_%dx%d" % size, voxelSize=0.01, frames=3, stream=3, trajectory="yaw", w=size[0], h=size[1])
    ses = T.Session(hip, sc)
    for k in range(sc.frames):
        v = ses.frame(k)
    d = hip.to_backend(sc.depth(sc.frames))
    for i in range(cases):
        M = np.array(v.M_d, np.float32).copy()
        M[12:15] += rng.normal(0, 0.004, 3).astype(np.float32)          # a few millimetres off
        cfg = TrackerConfig.default()
        levels = int(rng.integers(2, 6)) if min(size) >= 240 else int(rng.integers(2, 4))
        cfg.noHierarchyLevels = levels
        regime = [int(rng.integers(1, 4)) for _ in range(levels)]          # rotation only / translation only / both, any mix
        cfg.trackingRegime[:levels] = regime
        view = capi.View(d, sc.w, sc.h, M_d=M, intr_d=sc.intr()).struct()
        sp = np.ascontiguousarray(np.array(v.M_d, np.float32))
        spp = sp.ctypes.data_as(C.POINTER(C.c_float))
        out = []
        for key10 in (0, 1):
            hip.check(hip.fn["debug_set"](10, key10), "debug_set")
            o = (C.c_float * 16)()
            hip.check(hip.fn["track_camera"](C.byref(cfg), C.byref(view), ses.points.ptr, ses.normals.ptr, spp, o, None), "track_camera")
            out.append(np.array(o[:], np.float32))
        hip.check(hip.fn["debug_set"](10, 0), "debug_set")
        if not np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32)):
            bad += 1
            print("size", size, "case", i, "levels", levels, "differs by", float(np.abs(out[0] - out[1]).max()), flush=True)
    ses.close()
print(f"{3 * cases} cases: {bad} differences")
sys.exit(1 if bad else 0)
