#!/usr/bin/env python3
"""Dense integration (BASELINE configs[2], 512^3) frame by frame: integrate-kernel time for the classification modes of debug key 16
and the class counts of the check mode, as the weights grow and saturate.
usage: python tools/dense_modes.py [frames]     (measurement tool)"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import infinitam_amd as itm  # noqa: E402
from infinitam_amd import capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 130
# mode = key 16 value + 10 * key 17 value: 0 strips + classification, 1 strips, 10 / 11 / 12 the launch shape of rounds 1-2, 13 its check mode
MODES = [int(m) for m in os.environ.get("ITM_DENSE_MODES", "11,1,0,13").split(",")]
W, H = 640, 480
be = capi.Backend(os.environ["ITM_LIB"], "itm_") if os.environ.get("ITM_LIB") else itm.load()
for kv in os.environ.get("ITM_DEBUG_KV", "").split(","):         # e.g. ITM_DEBUG_KV=3:1024 -> debug_set(3, 1024)
    if ":" in kv:
        be.check(be.fn["debug_set"](int(kv.split(":")[0]), int(kv.split(":")[1])), "debug_set")
intr = synth.intrinsics_for(W, H)
frames = [be.to_backend(synth.depth_frame(W, H, synth.bench_position(k), intr)) for k in range(100)]
views = [capi.View(frames[k], W, H, M_d=synth.pose_matrix(synth.bench_position(k)), intr_d=intr) for k in range(100)]
REPORT = [int(k) for k in os.environ.get("ITM_REPORT", "0,1,5,20,50,90,99,100,101,110,129,150,199").split(",")]
for mode in MODES:
    be.check(be.fn["debug_set"](16, mode % 10), "debug_set")
    be.check(be.fn["debug_set"](17, int(os.environ.get("ITM_SPLITS", "1")) if mode // 10 else 0), "debug_set")
    scene = be.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, capi.default_params(voxelSize=0.004, stopIntegratingAtMaxW=True))
    scene.reco.ResetScene()
    rs = scene.vis.CreateRenderState((W, H))
    scene.profile_enable(1 << 3)
    out = {}
    c = (C.c_int32 * 4)()
    for k in range(N):
        scene.profile_read(reset=True)
        if mode % 10 == 3:
            be.check(be.fn["debug_dense_classify_check"](c, 1), "check")
        scene.reco.IntegrateIntoScene(views[k % 100], rs)
        be.sync()
        p = scene.profile_read(reset=True)["integrate"]
        if k in REPORT:
            out[k] = round(p["total_ms"] * 1e3, 1)
            if mode % 10 == 3:
                be.check(be.fn["debug_dense_classify_check"](c, 1), "check")
                out[k] = {"us": out[k], "free": c[0], "shadow": c[1], "mixed": c[2], "violations": c[3]}
    print(json.dumps({"lib": os.path.basename(os.environ.get("ITM_LIB", "default")), "kv": os.environ.get("ITM_DEBUG_KV", ""), "mode": mode, "integrate_us_by_frame": out}), flush=True)
    scene.close()
be.check(be.fn["debug_set"](16, 0), "debug_set")
be.check(be.fn["debug_set"](17, 0), "debug_set")
