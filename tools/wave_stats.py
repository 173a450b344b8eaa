#!/usr/bin/env python3
"""Per-wave cycle traces of the ray-cast kernel (s_memtime stamps inside cast_ray).  Needs a library built with
-DITM_EXP_WAVE_TIMING=1:   tools/build_variant.sh wt "-DITM_EXP_WAVE_TIMING=1"
usage: wave_stats.py gpurun_variants/lib_wt.so  -> gpurun_out/wave_stats.npy, wave_trace.npy; view with
tools/wave_trace_ww.py <wave ids>.   (development tool, not part of the product)"""
import ctypes as C, sys, os
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
for k in range(30):
    t = synth.bench_position(k)
    d = be.to_backend(synth.depth_frame(W, H, t, intr))
    v = capi.View(d, W, H, M_d=synth.pose_matrix(t), intr_d=intr)
    scene.process_frame(v, rs, pts, nrm)
be.sync()
for _ in range(3): scene.vis.FindSurface(v.M_d, v.intr_d, rs)
be.sync()
n = 4800
st = np.zeros((n, 12), np.uint64)
assert be.lib.itm_debug_read_wave_stats(st.ctypes.data_as(C.c_void_p), n * 12) == 0
tr = np.zeros((160, 64, 4), np.uint64)
assert be.lib.itm_debug_read_wave_trace(tr.ctypes.data_as(C.c_void_p), tr.size) == 0
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "wave_stats.npy"), st)
np.save(os.path.join(ROOT, "gpurun_out", "wave_trace.npy"), tr)
tot, outer = st[:, 0].astype(float), st[:, 1].astype(float)
if st[:, 2].any():   # directory march: extra columns
    order = np.argsort(-tot)
    lab = ["cycles", "outer", "inner", "steps", "run_cyc", "mem_cyc", "tri_cyc", "runs"]
    print("slowest 8 waves  ", lab); print(st[order[:8], :8].astype(np.int64))
    print("median 8 waves   "); print(st[order[len(order) // 2: len(order) // 2 + 8], :8].astype(np.int64))
    print("means            ", st[:, :8].astype(float).mean(axis=0).round(0))
print("wave cycles: mean %.0f p50 %.0f p90 %.0f max %.0f; outer iterations: mean %.1f max %.0f" % (
    tot.mean(), *np.percentile(tot, [50, 90]), tot.max(), outer.mean(), outer.max()))
