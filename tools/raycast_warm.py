#!/usr/bin/env python3
"""How much of the in-frame ray cast is the price of voxel / mirror lines written by other XCDs: the ray cast of the bench frame
in-frame (after allocate + integrate) against the same ray cast repeated on the unchanged scene (its lines then sit in the L2 of the
XCD that casts them).  usage: python tools/raycast_warm.py [frames]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import infinitam_amd as itm  # noqa: E402
from infinitam_amd import capi, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W, H = 640, 480
be = itm.load()
intr = synth.intrinsics_for(W, H)
scene = be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.004), localBlockNum=0x40000)
scene.reco.ResetScene()
rs = scene.vis.CreateRenderState((W, H))
pts = capi.DevBuffer(be, W * H * 16); nrm = capi.DevBuffer(be, W * H * 16)
frames = [be.to_backend(synth.depth_frame(W, H, synth.bench_position(k), intr)) for k in range(100)]
views = [capi.View(frames[k], W, H, M_d=synth.pose_matrix(synth.bench_position(k)), intr_d=intr) for k in range(100)]
scene.profile_enable(1 << 5)
for k in range(20):
    scene.process_frame(views[k], rs, pts, nrm)
be.sync(); scene.profile_read(reset=True)
inframe, warm1, warm2 = 0.0, 0.0, 0.0
for k in range(20, 20 + N):
    scene.process_frame(views[k % 100], rs, pts, nrm); be.sync()
    inframe += scene.profile_read(reset=True)["raycast"]["total_ms"]
    scene.vis.FindSurface(views[k % 100].M_d, intr, rs); be.sync()
    warm1 += scene.profile_read(reset=True)["raycast"]["total_ms"]
    scene.vis.FindSurface(views[k % 100].M_d, intr, rs); be.sync()
    warm2 += scene.profile_read(reset=True)["raycast"]["total_ms"]
print(json.dumps({"raycast_in_frame_us": round(inframe / N * 1e3, 1), "repeated_once_us": round(warm1 / N * 1e3, 1), "repeated_twice_us": round(warm2 / N * 1e3, 1),
                  "note": "FindSurface = the same ray cast over the range image the frame left (no range reduction in its prologue)"}))
