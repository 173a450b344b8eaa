#!/usr/bin/env python3
"""Sweeps the persistent-workgroup count of the (fused) hash integration launch on the bench workload."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import infinitam_amd as itm
from infinitam_amd import capi, synth
be = itm.load()
W, H = 640, 480
scene = be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.004), localBlockNum=0x40000)
scene.reco.ResetScene()
rs = scene.vis.CreateRenderState((W, H))
intr = synth.intrinsics_for(W, H)
pts = capi.DevBuffer(be, W*H*16); nrm = capi.DevBuffer(be, W*H*16)
views = []
for k in range(100):
    t = synth.bench_position(k)
    views.append(capi.View(be.to_backend(synth.depth_frame(W, H, t, intr)), W, H, M_d=synth.pose_matrix(t), intr_d=intr))
def run(n=200):
    for k in range(20): scene.process_frame(views[k % 100], rs, pts, nrm)
    be.sync(); t0 = time.perf_counter()
    for k in range(n): scene.process_frame(views[(20 + k) % 100], rs, pts, nrm)
    be.sync(); return (time.perf_counter() - t0) / n * 1e6
for fused in (1, 0):
    be.lib.itm_debug_set(4, 0 if fused else 1)
    for g in (0, 512, 640, 768, 800, 896, 1024, 1280, 1536, 2048):
        be.lib.itm_debug_set(3, g)
        print("fused", fused, "grid", g, "%.1f us/frame" % run())
