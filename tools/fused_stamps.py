#!/usr/bin/env python3
"""Who finishes last in the fused integrate + projection launch?  Needs a build with -DITM_EXP_FUSED_STAMPS=1 (measurement tool).
usage: python tools/fused_stamps.py gpurun_variants/lib_stamps.so"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from infinitam_amd import capi, synth
be = capi.Backend(sys.argv[1], "itm_")
G = int(os.environ.get("ITM_STAMP_GRID", "0"))
if G:
    be.check(be.fn["debug_set"](3, G), "debug_set")
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
n = G or 2048
st = np.zeros((n, 2), np.uint64)
assert be.lib.itm_debug_read_fused_stamps(st.ctypes.data_as(C.c_void_p), n * 2) == 0
t0 = st[:, 0].min()
s = (st - t0).astype(np.float64) / 100.0     # us (100 MHz clock)
proj, integ = s[:32], s[32:]   # (with dynamic queues the first 32 workgroups project, then integrate as well)
print("projection  workgroups: start %.2f..%.2f us, end %.2f..%.2f us, duration mean %.2f max %.2f" % (proj[:, 0].min(), proj[:, 0].max(), proj[:, 1].min(), proj[:, 1].max(), (proj[:, 1] - proj[:, 0]).mean(), (proj[:, 1] - proj[:, 0]).max()))
print("integration workgroups: start %.2f..%.2f us, end %.2f..%.2f us, duration mean %.2f max %.2f" % (integ[:, 0].min(), integ[:, 0].max(), integ[:, 1].min(), integ[:, 1].max(), (integ[:, 1] - integ[:, 0]).mean(), (integ[:, 1] - integ[:, 0]).max()))
print("integration end percentiles (us):", np.percentile(integ[:, 1], [10, 50, 90, 99, 100]).round(2))
print("integration start percentiles (us):", np.percentile(integ[:, 0], [10, 50, 90, 99, 100]).round(2))
d = integ[:, 1] - integ[:, 0]
print("integration duration percentiles (us):", np.percentile(d, [10, 50, 90, 99, 100]).round(2))
nv = scene.counters(rs)["noVisibleEntries"]
busy = integ[: (nv + 7) // 8]            # one item (a whole block) per wave: workgroup g holds items 8 g .. 8 g + 7 (+ multiples of the wave count)
idle = integ[(nv + 7) // 8:]
print("visible blocks %d: %d workgroups with work (duration mean %.2f, start mean %.2f, end max %.2f), %d without (duration mean %.2f, end max %.2f)" % (
    nv, len(busy), (busy[:, 1] - busy[:, 0]).mean(), busy[:, 0].mean(), busy[:, 1].max(), len(idle), (idle[:, 1] - idle[:, 0]).mean() if len(idle) else 0, idle[:, 1].max() if len(idle) else 0))
for lo in range(0, len(integ), max(1, len(integ) // 8)):
    seg = integ[lo:lo + max(1, len(integ) // 8)]
    print("  workgroups %4d..%4d: start %.2f..%.2f  end %.2f..%.2f" % (lo + 32, lo + 32 + len(seg) - 1, seg[:, 0].min(), seg[:, 0].max(), seg[:, 1].min(), seg[:, 1].max()))
