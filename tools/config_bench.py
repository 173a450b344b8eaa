#!/usr/bin/env python3
"""Per-kernel timings (hipEvent timers of the library) for the other BASELINE configs:
  3: dense ITMPlainVoxelArray 512^3, ITMVoxel_s, 4 mm, stopIntegratingAtMaxW (pure HBM-bound integrate)
  5: 1280x960, ITMVoxel_f_rgb, 2 mm, hash pool 0x40000 (large-volume stress)
  2: the bench.py workload, for reference
usage: python tools/config_bench.py [2|3|5] [frames]      (development / measurement tool)
Prints one JSON line per config with frames/s, per-kernel average microseconds and, for the integrate
kernel, algorithmic bytes and achieved GB/s."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import infinitam_amd as itm  # noqa: E402
from infinitam_amd import capi, synth  # noqa: E402


def run(cfg: int, frames: int):
    be = capi.Backend(os.environ['ITM_LIB'], 'itm_') if os.environ.get('ITM_LIB') else itm.load()
    for key in os.environ.get('ITM_DEBUG_KEYS', '').split(','):      # e.g. ITM_DEBUG_KEYS=4,6 -> debug_set(4,1), debug_set(6,1)
        if key.strip():
            be.check(be.fn['debug_set'](int(key), 1), 'debug_set')
    for kv in os.environ.get('ITM_DEBUG_KV', '').split(','):         # e.g. ITM_DEBUG_KV=3:1056 -> debug_set(3, 1056)
        if ':' in kv:
            be.check(be.fn['debug_set'](int(kv.split(':')[0]), int(kv.split(':')[1])), 'debug_set')
    if os.environ.get('ITM_NO_DIRECTORY'):
        be.check(be.fn['debug_set'](5, 1), 'debug_set')   # A/B: ray cast through the table walk instead of the block directory
    if cfg == 3:
        W, H, vox, idx, vs = 640, 480, capi.VOXEL_S, capi.INDEX_DENSE, 0.004
        prm = capi.default_params(voxelSize=vs, stopIntegratingAtMaxW=True)
        scene = be.create_scene(vox, idx, prm)          # 512^3, offset (-256,-256,0)
        colour = False
    elif cfg == 5:
        W, H, vox, idx, vs = 1280, 960, capi.VOXEL_F_RGB, capi.INDEX_HASH, 0.002
        prm = capi.default_params(voxelSize=vs)
        scene = be.create_scene(vox, idx, prm, localBlockNum=0x40000)
        colour = True
    else:
        W, H, vox, idx, vs = 640, 480, capi.VOXEL_S, capi.INDEX_HASH, 0.004
        prm = capi.default_params(voxelSize=vs)
        scene = be.create_scene(vox, idx, prm, localBlockNum=0x40000)
        colour = False
    scene.reco.ResetScene()
    rs = scene.vis.CreateRenderState((W, H))
    intr = synth.intrinsics_for(W, H)
    P = W * H
    pts = capi.DevBuffer(be, P * 16)
    nrm = capi.DevBuffer(be, P * 16)
    rgb = be.to_backend(synth.rgb_frame(W, H)) if colour else None
    n_distinct = min(frames, 20)
    views = []
    for k in range(n_distinct):
        t = synth.parity_position(k) if cfg == 5 else synth.bench_position(k)
        d = be.to_backend(synth.depth_frame(W, H, t, intr))
        views.append(capi.View(d, W, H, M_d=synth.pose_matrix(t), intr_d=intr, rgb=rgb, w_rgb=W, h_rgb=H, intr_rgb=intr))
    warm = 3
    for k in range(warm):
        scene.process_frame(views[k % n_distinct], rs, pts, nrm)
    be.sync()
    scene.profile_read(reset=True)
    scene.profile_enable(0x7f)
    t0 = time.perf_counter()
    for k in range(warm, warm + frames):
        scene.process_frame(views[k % n_distinct], rs, pts, nrm)
    be.sync()
    dt = time.perf_counter() - t0
    prof = scene.profile_read()
    c = scene.counters(rs)
    out = {"config": cfg, "frames": frames, "fps_with_timers": round(frames / dt, 1),
           "kernels_us": {k: round(1e3 * v["total_ms"] / max(1, v["calls"]), 2) for k, v in prof.items() if v["calls"]},
           "visible_blocks": c["noVisibleEntries"], "lastFreeBlockId": c["lastFreeBlockId"]}
    V = be.fn["voxel_size_bytes"](vox)
    t_int = prof["integrate"]["total_ms"] / max(1, prof["integrate"]["calls"]) * 1e-3
    if cfg == 3:
        vol = scene.download(capi.BUF_VOXEL_BLOCKS)
        updated = int((vol["w_depth"] > 0).sum())
        alg = 512 ** 3 * V + 4 * P            # full-volume read (+ writes of updated voxels, counted below)
        out["integrate_alg_bytes_read"] = alg
        out["integrate_GBps_read_only"] = round(alg / t_int / 1e9, 1)
        out["voxels_touched_total"] = updated
    else:
        nv = c["noVisibleEntries"]
        alg = nv * (512 * V * 2 + 20) + 4 * P + (4 * P if colour else 0)
        out["integrate_alg_bytes"] = alg
        out["integrate_GBps"] = round(alg / t_int / 1e9, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    cfgs = [int(sys.argv[1])] if len(sys.argv) > 1 else [2, 3, 5]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    for c in cfgs:
        run(c, n)
