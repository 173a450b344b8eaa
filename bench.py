#!/usr/bin/env python3
"""bench.py -- fused depth frames/s of the TSDF hot path on MI355X (BASELINE.json metric).

One "step" = one 640x480 depth frame through the per-frame call sequence of
ITMMainEngine::ProcessFrame after the view is built (reference Engine/ITMMainEngine.cpp:123-126):
AllocateSceneFromDepth + IntegrateIntoScene + CreateExpectedDepths + CreateICPMaps, issued through
the C-ABI (itm_process_frame) with the float depth frames already resident in HBM.

Workload (config.workload): BASELINE.json configs[1] -- synthetic 640x480 depth (sphere + wall,
SURVEY.md section 8d benchmark trajectory), hash TSDF, ITMVoxel_s, 4 mm voxels, mu 0.02,
512^3-equivalent pool (0x40000 blocks).  Multi-GPU (config 4): one independent stream per rank
(stream g offset 0.05*g m), weak scaling, with an RCCL all-gather of the per-stream
{pose, visible-block list} record every frame on a side stream.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 640, 480
VOXEL_SIZE, MU = 0.004, 0.02
LOCAL_BLOCKS = 0x40000
PERIOD = 100                 # the benchmark trajectory repeats every 100 frames
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MAX_IDS = 16384              # ids per exchanged visible-block record
EXCHANGE_BATCH = 8           # frames per all-gather (8 x 64 KB records per rank and collective)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=40)
    ap.add_argument("--no-exchange", action="store_true", help="skip the visible-list all-gather at N>1")
    ap.add_argument("--force-exchange", action="store_true", help="run the all-gather path even with one rank (self-test)")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # self-test hook (never set by the driver): ITM_BENCH_SHARED_GPU=1 lets several ranks share GPU 0 with the gloo
    # backend, to exercise the multi-rank control flow on a one-GPU box; RCCL needs one GPU per rank
    shared = os.environ.get("ITM_BENCH_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import infinitam_amd as itm
    from infinitam_amd import capi, synth
    be = itm.load()
    be.check(be.fn["set_device"](local_rank), "set_device")

    # ---- scene + inputs resident in HBM -------------------------------------------------------
    params = capi.default_params(voxelSize=VOXEL_SIZE, mu=MU)
    scene = be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, params, localBlockNum=LOCAL_BLOCKS)
    scene.reco.ResetScene()
    rs = scene.vis.CreateRenderState((W, H))
    intr = synth.intrinsics_for(W, H)
    frames = np.stack([synth.depth_frame(W, H, synth.bench_position(k, rank), intr) for k in range(PERIOD)])
    depth_dev = torch.from_numpy(frames).cuda()
    points = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    normals = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    poses = [synth.pose_matrix(synth.bench_position(k, rank)) for k in range(PERIOD)]
    views = []
    for k in range(PERIOD):
        v = capi.View(depth_dev[k].data_ptr(), W, H, M_d=poses[k], intr_d=intr).struct()
        views.append(v)
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    fn = be.fn["process_frame"]
    sh, rh = C.c_void_p(scene.h), C.c_void_p(rs.h)
    pp, np_ = C.c_void_p(points.data_ptr()), C.c_void_p(normals.data_ptr())

    exchange = (world > 1 and not args.no_exchange) or args.force_exchange
    if exchange:
        from infinitam_amd.streams import VisibleListExchange
        ex = VisibleListExchange(be, world, rank, MAX_IDS, device="cuda", batch=EXCHANGE_BATCH)

    def step(k):
        v = views[k % PERIOD]
        rc = fn(sh, C.byref(v), rh, pp, np_, sp)
        if rc:
            be.check(rc, "process_frame")
        if exchange:
            # record copy on the frame stream, RCCL all-gather on a side stream (off the critical path)
            ex.step(rs.h, poses[k % PERIOD], stream)

    TK_RAYCAST = 5
    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if rank == 0:
        scene.profile_read(reset=True)
        scene.profile_enable(1 << TK_RAYCAST)   # two hipEventRecord per frame around the dominant kernel
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    counters = scene.counters(rs)
    roofline = None
    if rank == 0:
        roofline = read_roofline(be, scene, rs, args.steps, counters)
        scene.profile_enable(0)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(args.cpu_frames)

    if rank == 0:
        fps = world * args.steps / elapsed
        out = {
            "metric": "fused depth frames/sec (640x480, hash TSDF)",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic 640x480 depth (sphere+wall, bench trajectory), "
                                   "hash TSDF ITMVoxel_s, 4 mm voxels, mu 0.02, 0x40000-block pool; "
                                   "allocate+integrate+expected-depths+ICP raycast per frame",
                       "streams": world, "exchange": f"rccl all_gather of visible-block records, {EXCHANGE_BATCH} frames per collective" if exchange else "none",
                       "visible_blocks_last_frame": counters["noVisibleEntries"]},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
    if world > 1 or args.force_exchange:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which a pipe only delivers at exit: flush it first so that the
        # JSON line is the last line on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def algorithmic_bytes():
    """Algorithmic bytes per launch of the kernels on the path (DESIGN.md 'Algorithmic bytes',
    SURVEY.md section 8d), priced from the oracle's work counters on frames 20..59 of this workload
    (tests/golden/algbytes_config2.json, regenerated by tests/golden/make_algbytes.py)."""
    with open(os.path.join(ROOT, "tests", "golden", "algbytes_config2.json")) as f:
        c = json.load(f)
    P, V, E = W * H, 4, 16
    found_nearest = c["nearest_reads"] - c["nearest_misses"]
    raycast = 8 * (P / 64) + V * (found_nearest + 8 * c["trilinear_reads"]) + E * c["hash_probes"] + 16 * P
    integrate = c["visible_blocks"] * (512 * V * 2 + E + 4) + 4 * P
    return {"raycast": raycast, "integrate": integrate}, c


def read_roofline(be, scene, rs, steps, counters):
    """Dominant kernel = the ray-cast kernel (rocprofv3: 45-55 % of the frame, profiles/).  Its average
    duration is measured with hipEvents recorded around every launch inside the timed region, on the
    stream the kernel runs on (itm_profile_enable / itm_profile_read)."""
    prof = scene.profile_read(reset=True)
    alg, _ = algorithmic_bytes()
    r = prof["raycast"]
    if not r["calls"]:
        return None
    avg_s = r["total_ms"] * 1e-3 / r["calls"]
    achieved = alg["raycast"] / avg_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "raycast_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f).get("hbm_bytes_per_launch")
    return {"bound": "hbm", "kernel": "raycast_kernel<VoxelS,hash>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": round(alg["raycast"]), "avg_kernel_us": round(avg_s * 1e6, 2),
            "launches_timed": r["calls"]}


def _time_cpu(be, sc, nframes, T):
    ses = T.Session(be, sc)
    depth = [be.to_backend(sc.depth(k)) for k in range(nframes)]
    t0 = time.perf_counter()
    for k in range(nframes):
        v = T.View(depth[k], sc.w, sc.h, M_d=sc.pose(k), intr_d=sc.intr())
        ses.scene.process_frame(v, ses.rs, ses.points, ses.normals)
    dt = time.perf_counter() - t0
    ses.close()
    return nframes / dt


def run_cpu_baseline(nframes):
    """The CPU oracle (a port of the reference CPU engines, bit-equal to them) timed on ONE host
    core on the first `nframes` frames of the same workload; beside it, where the prebuilt oracle/_ref library
    travelled along, the reference's OWN CPU engines on the same frames (their voxel pool is the fork's compile-time
    0x10000 blocks instead of 0x40000, which this workload never exhausts).  Checker code, used only here."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import itm_testlib as T
    ob = T.oracle_backend()
    sc = T.Scenario(name="bench_cpu", voxelSize=VOXEL_SIZE, mu=MU, localBlockNum=LOCAL_BLOCKS, trajectory="bench", frames=nframes)
    _time_cpu(ob, T.Scenario(name="warm", voxelSize=VOXEL_SIZE, mu=MU, localBlockNum=LOCAL_BLOCKS, trajectory="bench", frames=2), 2, T)
    out = {"value": round(_time_cpu(ob, sc, nframes, T), 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"first {nframes} frames of the same workload, oracle/libitm_oracle.so single thread"}
    if os.path.exists(T.REF_LIB):
        rb = T.Backend(T.REF_LIB, "itmr_")
        sc_ref = T.Scenario(name="bench_ref", voxelSize=VOXEL_SIZE, mu=MU, trajectory="bench", frames=nframes)
        out["reference_engines"] = {"value": round(_time_cpu(rb, sc_ref, nframes, T), 3), "unit": "frames/s", "cores": 1,
                                    "note": "ITMSceneReconstructionEngine_CPU / ITMVisualisationEngine_CPU compiled from the reference sources (oracle/_ref), same frames"}
    return out


if __name__ == "__main__":
    main()
