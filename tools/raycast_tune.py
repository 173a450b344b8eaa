#!/usr/bin/env python3
"""Times single stages of the frame on the bench workload for one build of the library.
usage: raycast_tune.py <libitmhip.so> [frames]   (development tool, not part of the product)"""
import ctypes as C, sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from infinitam_amd import capi, synth
be = capi.Backend(sys.argv[1], "itm_")
if os.environ.get("ITM_NO_DIRECTORY"):
    be.check(be.fn["debug_set"](5, 1), "debug_set")   # A/B: table walk instead of the block directory
W, H = 640, 480
scene = be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.004), localBlockNum=0x40000)
scene.reco.ResetScene()
rs = scene.vis.CreateRenderState((W, H))
intr = synth.intrinsics_for(W, H)
pts = capi.DevBuffer(be, W*H*16); nrm = capi.DevBuffer(be, W*H*16)
views = []
for k in range(30):
    t = synth.bench_position(k)
    d = be.to_backend(synth.depth_frame(W, H, t, intr))
    views.append(capi.View(d, W, H, M_d=synth.pose_matrix(t), intr_d=intr))
for v in views: scene.process_frame(v, rs, pts, nrm)
be.sync()
v = views[-1]
def timeit(name, f, n=50):
    for _ in range(5): f()
    be.sync(); t0 = time.perf_counter()
    for _ in range(n): f()
    be.sync(); dt = (time.perf_counter() - t0) / n
    print(f"{name:28s} {dt*1e6:8.1f} us")
timeit("find_surface (raycast)", lambda: scene.vis.FindSurface(v.M_d, v.intr_d, rs))
timeit("create_expected_depths", lambda: scene.vis.CreateExpectedDepths(v.M_d, v.intr_d, rs))
timeit("create_icp_maps", lambda: scene.vis.CreateICPMaps(v, rs, pts, nrm))
timeit("integrate", lambda: scene.reco.IntegrateIntoScene(v, rs))
timeit("allocate", lambda: scene.reco.AllocateSceneFromDepth(v, rs))
timeit("process_frame", lambda: scene.process_frame(v, rs, pts, nrm))
