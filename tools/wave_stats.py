#!/usr/bin/env python3
"""Per-wave cycle accounting of the ray-cast kernel (needs a library built with -DITM_EXP_WAVE_TIMING=1
-DITM_RAY_WHILE_WHILE=0).  usage: wave_stats.py <lib.so>   (development tool)"""
import ctypes as C, sys, os
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
rc = be.lib.itm_debug_read_wave_stats(st.ctypes.data_as(C.c_void_p), n * 12)
assert rc == 0
st = st.astype(np.float64)
tot, iters, triit, near, tri, lanesIt, lanesTri, maxidx, setup, loop, refine, maxiter = st.T
start = np.zeros_like(tot)
end = start + tot
print("kernel span (cycles, from earliest start to latest end; counters may differ per XCD):", end.max() - start.min())
print("wave total cycles: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f max %.0f" % (tot.mean(), *np.percentile(tot, [50, 90, 99]), tot.max()))
print("start offsets: p50 %.0f p99 %.0f max %.0f" % tuple(np.percentile(start - start.min(), [50, 99, 100])))
order = np.argsort(-(tot+setup))[:16]
print("latest-finishing waves: idx total iters triIters nearCyc triCyc lanes/iter lanes/tri startOff")
for i in order:
    print(int(i), int(tot[i]), int(iters[i]), int(triit[i]), int(near[i]), int(tri[i]), "%.1f" % (lanesIt[i] / max(iters[i], 1)), "%.1f" % (lanesTri[i] / max(triit[i], 1)), "maxIter", int(maxiter[i]), "at", int(maxidx[i]), "setup", int(setup[i]), "refine", int(refine[i]))
long = iters > 30
print("long waves:", long.sum(), " mean total %.0f  near/iter %.0f  tri/triIter %.0f  triIters %.1f iters %.1f other %.0f" % (
    tot[long].mean(), (near[long] / iters[long]).mean(), (tri[long] / np.maximum(triit[long], 1)).mean(), triit[long].mean(), iters[long].mean(),
    (tot[long] - near[long] - tri[long]).mean()))
short = ~long
print("short waves:", short.sum(), " mean total %.0f  near/iter %.0f  tri/triIter %.0f  triIters %.1f iters %.1f other %.0f" % (
    tot[short].mean(), (near[short] / iters[short]).mean(), (tri[short] / np.maximum(triit[short], 1)).mean(), triit[short].mean(), iters[short].mean(),
    (tot[short] - near[short] - tri[short]).mean()))
for nm, m in (("long", long), ("short", short)):
    print(nm, "setup %.0f loop %.0f refine %.0f  loop-(near+tri) %.0f" % (setup[m].mean(), loop[m].mean(), refine[m].mean(), (loop[m] - near[m] - tri[m]).mean()))
# per-iteration-index profile is not recorded; histogram of near cycles per iteration over long waves
np.save(os.path.join(ROOT, "gpurun_out", "wave_stats.npy"), st)

tr = np.zeros((160, 64, 4), np.uint64)
be.lib.itm_debug_read_wave_trace(tr.ctypes.data_as(C.c_void_p), tr.size)
np.save(os.path.join(ROOT, "gpurun_out", "wave_trace.npy"), tr)
sel = [w for w in range(150) if True]
# print the traced waves with the largest totals
traced = [(tot[32 * k + 1], k) for k in range(150)]
traced.sort(reverse=True)
for T, k in traced[:4]:
    wv = 32 * k + 1
    print("wave", wv, "total", int(T), "iters", int(iters[wv]))
    for i in range(int(min(iters[wv], 64))):
        t0, t1, t2, t3 = [int(v) for v in tr[k, i]]
        lanes = t3 >> 48; t3 &= (1 << 48) - 1
        print("   it %2d lanes %2d start %7d near %6d tri %6d rest %6d" % (i, lanes, t0, t1 - t0, t2 - t1, t3 - t2))
