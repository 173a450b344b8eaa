#!/usr/bin/env python3
"""Timeline of the in-frame ray-cast launch: per wave start / prologue end / phase-1 end / phase-2 end on the constant 100 MHz clock.
Needs a build with -DITM_EXP_RAYCAST_STAMPS=1:  tools/build_variant.sh rs "-DITM_EXP_RAYCAST_STAMPS=1"
usage: python tools/raycast_timeline.py gpurun_variants/lib_rs.so   (measurement tool, not part of the product)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from infinitam_amd import capi, synth
be = capi.Backend(sys.argv[1], "itm_")
W, H = 640, 480
scene = be.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, capi.default_params(voxelSize=0.004, stopIntegratingAtMaxW=True)) if len(sys.argv) > 2 and sys.argv[2] == "dense" else be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.004), localBlockNum=0x40000)
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
n = 4800
raw = np.zeros((n, 4), np.uint64)
assert be.lib.itm_debug_read_raycast_stamps(raw.ctypes.data_as(C.c_void_p), n * 4) == 0
served = (raw[:, 3] >> np.uint64(52)).astype(np.int64)
raw[:, 3] &= np.uint64((1 << 52) - 1)
t0 = raw[:, 0].min()
has2 = raw[:, 3] > 0
end = np.where(has2, raw[:, 3], raw[:, 2])
us = lambda a: (a.astype(np.float64) - float(t0)) / 100.0
start, pro, p1, e = us(raw[:, 0]), us(raw[:, 1]), us(raw[:, 2]), us(end)
pc = lambda a: np.percentile(a, [0, 10, 50, 90, 99, 100]).round(2)
print("percentiles                 min / p10 / p50 / p90 / p99 / max (us)")
print("wave start               ", pc(start))
print("prologue duration        ", pc(pro - start))
print("phase 1 duration (wave)  ", pc(p1 - pro))
print("phase 1 end (wave)       ", pc(p1))
idx = np.arange(0, n, 4)
wg_p1_first = np.minimum.reduceat(p1, idx)
wg_p1_end = np.maximum.reduceat(p1, idx)
wg_end = np.maximum.reduceat(e, idx)
wg_parked = served.reshape(-1, 4).sum(axis=1)
two = wg_parked > 0
print("workgroups with parked rays: %d of %d; parked rays: mean %.0f max %d" % (two.sum(), n // 4, wg_parked[two].mean() if two.any() else 0, wg_parked.max()))
print("WG end, no parked rays   ", pc(wg_end[~two]))
if two.any():
    print("WG first wave out of ph.1", pc(wg_p1_first[two]))
    print("WG last wave out of ph.1 ", pc(wg_p1_end[two]))
    print("WG end - last ph.1 end   ", pc((wg_end - wg_p1_end)[two]))
    print("WG end, parked           ", pc(wg_end[two]))
print("kernel span (first start -> last end): %.2f us" % e.max())
order = np.argsort(-wg_end)[:12]
print("slowest workgroups: id, tile x, tile y, end, parked, per-wave phase-1 end, per-wave end, per-wave rays served in phase 2")
for g in order:
    print("  %4d %3d %3d  %6.2f  %3d  %s %s %s" % (g, g % 40, g // 40, wg_end[g], wg_parked[g], p1[4 * g:4 * g + 4].round(1), e[4 * g:4 * g + 4].round(1), served[4 * g:4 * g + 4]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "raycast_timeline.npy"), raw)
