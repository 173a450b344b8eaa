#!/usr/bin/env python3
"""Census of the ray cast's dependent chains on the bench workload (CPU oracle, test infrastructure: the reference's castRay with per-ray
counters): how many steps of which kind a ray takes, per 16x16-pixel tile the longest chain.  A step is one dependent round trip on
the GPU (two when the value is in the band: single-voxel read, then trilinear read).  usage: python tests/golden/ray_census.py [frames=30]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from infinitam_amd import capi, synth  # noqa: E402

ob = capi.Backend(os.path.join(ROOT, "oracle", "libitm_oracle.so"), "itmo_")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
W, H = 640, 480
scene = ob.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.005), localBlockNum=0x40000)
scene.reco.ResetScene()
rs = scene.vis.CreateRenderState((W, H))
intr = synth.intrinsics_for(W, H)
pts = capi.DevBuffer(ob, W * H * 16); nrm = capi.DevBuffer(ob, W * H * 16)
tr = np.zeros((H, W, 8), np.int32)
for k in range(frames):
    t = synth.bench_position(k)
    v = capi.View(ob.to_backend(synth.depth_frame(W, H, t, intr)), W, H, M_d=synth.pose_matrix(t), intr_d=intr)
    if k == frames - 1:
        ob.lib.itmo_debug_ray_trace(tr.ctypes.data_as(C.c_void_p), W)
    scene.process_frame(v, rs, pts, nrm)
ob.lib.itmo_debug_ray_trace(None, 0)
names = ["steps", "band", "miss", "unitNear", "unitBand", "far(sdf==1)", "longestMissRun", "afterRun"]
flat = tr.reshape(-1, 8)
print("per ray: mean / p50 / p90 / p99 / max")
for i, n in enumerate(names):
    c = flat[:, i]
    print(f"  {n:16s} {c.mean():7.2f} {np.percentile(c, 50):6.0f} {np.percentile(c, 90):6.0f} {np.percentile(c, 99):6.0f} {c.max():6d}")
# dependent round trips as the shipped kernel takes them: a step = 1, a band step = 2; parked runs of misses in look-aheads of 6
rounds = flat[:, 0] + flat[:, 1]
tiles = rounds.reshape(H // 16, 16, W // 16, 16).max(axis=(1, 3))
print("rounds per ray (steps + band steps): mean %.1f p90 %.0f p99 %.0f max %d" % (rounds.mean(), np.percentile(rounds, 90), np.percentile(rounds, 99), rounds.max()))
print("longest chain per 16x16 tile: p50 %.0f p90 %.0f p99 %.0f max %d; tiles %d" % (np.percentile(tiles, 50), np.percentile(tiles, 90), np.percentile(tiles, 99), tiles.max(), tiles.size))
worst = np.argsort(-rounds)[:12]
print("the 12 longest rays:", names)
for w in worst:
    print("  pixel", (int(w % W), int(w // W)), flat[w].tolist())
# what speculation on unit steps could remove: unitNear in groups of 4, unitBand pairs
spec = flat[:, 0] + flat[:, 1] - (flat[:, 3] * 3) // 4 - flat[:, 4]
tspec = spec.reshape(H // 16, 16, W // 16, 16).max(axis=(1, 3))
print("with unit-step speculation (4 single-voxel unit steps per round, band unit step = 1 round): per tile p50 %.0f p90 %.0f p99 %.0f max %d" % (
    np.percentile(tspec, 50), np.percentile(tspec, 90), np.percentile(tspec, 99), tspec.max()))
np.save(os.path.join(ROOT, "gpurun_out", "ray_census.npy"), tr)
