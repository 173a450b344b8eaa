#!/usr/bin/env python3
"""The fused frame at 640x480 with finer voxels -- 9.6 k / 53.5 k / 120.8 k visible blocks -- per-kernel event timers on (measurement tool: how launch
constants chosen on BASELINE configs[1] behave when the list is 5-12 times as long).  usage: python tools/fine_voxel_bench.py <libitmhip.so>"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from infinitam_amd import capi, synth
be = capi.Backend(sys.argv[1], "itm_")
W, H = 640, 480
for vs, pool in ((0.004, 0x40000), (0.002, 0x80000), (0.0015, 0xC0000)):
    scene = be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=vs), localBlockNum=pool)
    scene.reco.ResetScene()
    rs = scene.vis.CreateRenderState((W, H))
    intr = synth.intrinsics_for(W, H)
    pts = capi.DevBuffer(be, W * H * 16); nrm = capi.DevBuffer(be, W * H * 16)
    frames = [be.to_backend(synth.depth_frame(W, H, synth.bench_position(k), intr)) for k in range(50)]
    views = [capi.View(frames[k], W, H, M_d=synth.pose_matrix(synth.bench_position(k)), intr_d=intr) for k in range(50)]
    for k in range(30): scene.process_frame(views[k % 50], rs, pts, nrm)
    be.sync()
    scene.profile_enable(0x7f); scene.profile_sample(1)
    t0 = time.perf_counter()
    N = 200
    for k in range(30, 30 + N): scene.process_frame(views[k % 50], rs, pts, nrm)
    be.sync()
    dt = time.perf_counter() - t0
    prof = scene.profile_read()
    nv = scene.counters(rs)["noVisibleEntries"]
    print("voxel %.4f visible %6d  %8.1f fps (timers on)  integrate %.1f us  raycast %.1f us" % (vs, nv, N / dt, prof["integrate"]["total_ms"] * 1e3 / prof["integrate"]["calls"], prof["raycast"]["total_ms"] * 1e3 / prof["raycast"]["calls"]), flush=True)
    rs.close(); scene.close()
