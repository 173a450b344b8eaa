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
C5 = len(sys.argv) > 2 and sys.argv[2] == "c5"          # BASELINE configs[4]: 1280x960, ITMVoxel_f_rgb, 2 mm
W, H = (1280, 960) if C5 else (640, 480)
scene = be.create_scene(capi.VOXEL_F_RGB if C5 else capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.002 if C5 else 0.004), localBlockNum=0x40000)
scene.reco.ResetScene()
rs = scene.vis.CreateRenderState((W, H))
intr = synth.intrinsics_for(W, H)
pts = capi.DevBuffer(be, W * H * 16); nrm = capi.DevBuffer(be, W * H * 16)
FRAMES = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for k in range(FRAMES):
    t = synth.bench_position(k)
    d = be.to_backend(synth.depth_frame(W, H, t, intr))
    v = capi.View(d, W, H, M_d=synth.pose_matrix(t), intr_d=intr, rgb=(be.to_backend(synth.rgb_frame(W, H)) if C5 else None), w_rgb=W, h_rgb=H, intr_rgb=intr)
    scene.process_frame(v, rs, pts, nrm)
be.sync()
n = 640                       # 576 chunks + the 64 early workgroups that only sweep an excess-region chunk (they leave no stamps)
raw = np.zeros((n, 6), np.uint64)
assert be.lib.itm_debug_read_list_stamps(raw.ctypes.data_as(C.c_void_p), n * 6) == 0
early = raw[:64][raw[:64, 5] > 0] if raw[64:, 0].all() and n == 640 else raw[:0]      # round 4: the first 64 workgroups only sweep an excess-region chunk
if len(early):
    e0 = float(raw[raw[:, 0] > 0][:, 0].min())
    print("early workgroups: start", np.percentile((early[:, 0] - e0) / 100.0, [0, 50, 100]).round(2), " end", np.percentile((early[:, 5].astype(np.float64) - e0) / 100.0, [0, 50, 90, 100]).round(2),
          " with requests:", int((early[:, 1] > early[:, 0]).sum()), " stamped at", np.percentile((early[early[:, 1] > early[:, 0]][:, 1].astype(np.float64) - e0) / 100.0, [0, 50, 100]).round(2) if (early[:, 1] > early[:, 0]).any() else "-")
    raw = raw[64:]
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
print("all done (stamp 5)    ", pc((raw[:, 5].astype(np.float64) - float(t0)) / 100.0))
print("ordered chunks end    ", pc(us[~ex, 4]), " excess-region chunks end", pc(us[ex, 4]))
# the slowest chunks, and how many slots of each were visible before / after this frame (the re-test candidates are the
# previously visible slots no pixel asked for)
if len(sys.argv) > 4:
    order = np.argsort(-(us[:, 2] - us[:, 1]))[:6]
    types = scene.download(capi.BUF_VISIBLE_TYPE, rs)
    per = (types.reshape(-1, 2048) != 0).sum(axis=1)
    hashv = scene.download(capi.BUF_HASH_ENTRIES)
    alloc = (hashv["ptr"].reshape(-1, 2048) >= 0).sum(axis=1)
    for c in order:
        print("chunk %4d: count phase %.2f us, published at %.2f, visible slots now %d, allocated slots %d" % (c, us[c, 2] - us[c, 1], us[c, 2], per[c], alloc[c]))
    print("visible slots per chunk: ordered max %d mean %.1f; excess region %s" % (per[:512].max(), per[:512].mean(), per[512:].tolist()))
if len(sys.argv) > 4:
    exi = np.where(ex)[0]
    late = exi[np.argsort(-us[exi, 1])[:8]]
    for c in late:
        print("excess chunk %4d (e = %2d): start %.2f  after worker phase + sweep %.2f  published %.2f  after look-back %.2f  end %.2f" % (c, c - exi[0], us[c, 0], us[c, 1], us[c, 2], us[c, 3], us[c, 4]))
