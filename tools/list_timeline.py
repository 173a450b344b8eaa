#!/usr/bin/env python3
"""Per-workgroup timeline of the visible-list launch (with the allocation sweep inside): start, after the sweep, after counting +
publishing, after the look-back, before compaction.  Needs -DITM_EXP_LIST_STAMPS=1:  tools/build_variant.sh ls "-DITM_EXP_LIST_STAMPS=1" alloc
usage: python tools/list_timeline.py gpurun_variants/lib_ls.so   (measurement tool)"""
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
n = 640                       # 576 chunks + the 64 early workgroups that only sweep an excess-region chunk (they leave no stamps)
raw = np.zeros((n, 6), np.uint64)
assert be.lib.itm_debug_read_list_stamps(raw.ctypes.data_as(C.c_void_p), n * 6) == 0
raw = raw[raw[:, 0] > 0]      # the workgroups that are chunks (round 3 builds: rows 0 .. 575, round 4: 64 .. 639)
n = len(raw)
t0 = raw[:, 0].min()
us = (raw[:, :5].astype(np.float64) - float(t0)) / 100.0
pc = lambda a: np.percentile(a, [0, 10, 50, 90, 100]).round(2)
for k, name in enumerate(["start", "after sweep", "counted + published", "after look-back", "before compaction"]):
    print("%-22s" % name, pc(us[:, k]))
print("count phase duration  ", pc(us[:, 2] - us[:, 1]))
print("look-back duration    ", pc(us[:, 3] - us[:, 2]))
ex = np.arange(n) >= n - 64
print("ordered chunks end    ", pc(us[~ex, 4]), " excess-region chunks end", pc(us[ex, 4]))
