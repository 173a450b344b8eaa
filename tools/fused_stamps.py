#!/usr/bin/env python3
"""Who finishes last in the fused integrate + projection launch?  Needs a build with -DITM_EXP_FUSED_STAMPS=1 (measurement tool).
usage: python tools/fused_stamps.py gpurun_variants/lib_stamps.so"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from infinitam_amd import capi, synth
be = capi.Backend(sys.argv[1], "itm_")
W, H = 640, 480
scene = be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.004), localBlockNum=0x40000)
scene.reco.ResetScene()
rs = scene.vis.CreateRenderState((W, H))
intr = synth.intrinsics_for(W, H)
pts = capi.DevBuffer(be, W * H * 16); nrm = capi.DevBuffer(be, W * H * 16)
for k in range(30):
    t = synth.bench_position(k)
    d = be.to_backend(synth.depth_frame(W, H, t, intr))
    v = capi.View(d, W, H, M_d=synth.pose_matrix(t), intr_d=intr)
    scene.process_frame(v, rs, pts, nrm)
be.sync()
n = 2048
st = np.zeros((n, 2), np.uint64)
assert be.lib.itm_debug_read_fused_stamps(st.ctypes.data_as(C.c_void_p), n * 2) == 0
t0 = st[:, 0].min()
s = (st - t0).astype(np.float64) / 100.0     # us (100 MHz clock)
proj, integ = s[:32], s[32:]   # (with dynamic queues the first 32 workgroups project, then integrate as well)
print("projection  workgroups: start %.2f..%.2f us, end %.2f..%.2f us, duration mean %.2f max %.2f" % (proj[:, 0].min(), proj[:, 0].max(), proj[:, 1].min(), proj[:, 1].max(), (proj[:, 1] - proj[:, 0]).mean(), (proj[:, 1] - proj[:, 0]).max()))
print("integration workgroups: start %.2f..%.2f us, end %.2f..%.2f us, duration mean %.2f max %.2f" % (integ[:, 0].min(), integ[:, 0].max(), integ[:, 1].min(), integ[:, 1].max(), (integ[:, 1] - integ[:, 0]).mean(), (integ[:, 1] - integ[:, 0]).max()))
print("integration end percentiles (us):", np.percentile(integ[:, 1], [10, 50, 90, 99, 100]).round(2))
