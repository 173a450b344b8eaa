import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import infinitam_amd as itm
from infinitam_amd import capi, synth
be = itm.load()
W, H = 640, 480
intr = synth.intrinsics_for(W, H)
scene = be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.004), localBlockNum=0x40000)
scene.reco.ResetScene()
rs = scene.vis.CreateRenderState((W, H))
pts = capi.DevBuffer(be, W * H * 16); nrm = capi.DevBuffer(be, W * H * 16)
views = []
for k in range(50):
    t = synth.bench_position(k)
    d = be.to_backend(synth.depth_frame(W, H, t, intr))
    views.append((d, capi.View(d, W, H, M_d=synth.pose_matrix(t), intr_d=intr).struct()))
fn = be.fn["process_frame"]
sh, rh, pp, np_ = C.c_void_p(scene.h), C.c_void_p(rs.h), C.c_void_p(pts.ptr.value if hasattr(pts.ptr,'value') else pts.ptr), C.c_void_p(nrm.ptr.value if hasattr(nrm.ptr,'value') else nrm.ptr)
for k in range(50): fn(sh, C.byref(views[k][1]), rh, pp, np_, None)
be.sync()
t0 = time.perf_counter()
for r in range(4):
    for k in range(50): fn(sh, C.byref(views[k][1]), rh, pp, np_, None)
t1 = time.perf_counter()
be.sync()
t2 = time.perf_counter()
print("host submission %.1f us per frame; with final sync %.1f us per frame" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
