#!/usr/bin/env python3
"""bench.py -- fused depth frames/s of the TSDF hot path on MI355X (BASELINE.json metric).

One "step" = one depth frame through the per-frame call sequence of ITMMainEngine::ProcessFrame after the
view is built (reference Engine/ITMMainEngine.cpp:123-126): AllocateSceneFromDepth + IntegrateIntoScene
(Engine/ITMDenseMapper.cpp:50-57) + CreateExpectedDepths + CreateICPMaps (Engine/ITMTrackingController.cpp:30-46),
issued as those FOUR calls through the C-ABI -- what a drop-in back-end is called with -- with the float depth
frames already resident in HBM.  `value` is that figure.  Beside it, never as `value`: the same frames through the
single entry point itm_process_frame and through itm_process_frame_ahead (which needs the NEXT frame's pose before
this frame is done: fine for an offline sequence, impossible for a closed tracking loop).

Short runs: with --steps < 100 the timed region (exactly --steps frames between two barriers) is repeated until at
least 0.25 s have been measured; `value` / `ms_per_step` are those of the MEDIAN repetition, `repetitions` says how
many there were.  The roofline kernel is never timed inside the timed region: one extra, untimed repetition brackets
EVERY launch of it (and of the integration and the visible-list launch) with HIP events, at least 64 samples.

Workloads (--config, names in config.workload):
  2 (default, the headline): BASELINE configs[1] -- synthetic 640x480 depth (sphere + wall, SURVEY 8d bench
     trajectory), hash TSDF, ITMVoxel_s, 4 mm voxels, mu 0.02, 0x40000-block pool.  roofline = ray cast.
  3: BASELINE configs[2] -- dense ITMPlainVoxelArray 512^3, ITMVoxel_s, 4 mm, stopIntegratingAtMaxW.
     roofline = dense integrate (the pure HBM-bound kernel).
  5: BASELINE configs[4] -- 1280x960, ITMVoxel_f_rgb, 2 mm, hash pool 0x40000.  roofline = hash integrate.
Multi-GPU (BASELINE configs[3]): one independent stream per rank (stream g offset 0.05*g m in y), weak
scaling, with an RCCL all-gather of the per-stream {pose, visible-block list} record on a side stream.

Launching:  `python bench.py --gpus N` spawns N rank processes itself (before anything touches the GPU);
under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (WORLD_SIZE set) it is one rank.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MAX_IDS = 16384              # ids per exchanged visible-block record
PERIOD = 100                 # the benchmark trajectory repeats every 100 frames
DEBUG_KEYS_FALLBACK = {1: 5, 2: 12}          # include/itm_hip.h: ITM_DEBUG_NO_DIRECTORY, ITM_DEBUG_NO_SDF_MIRROR
TK = {"request": 0, "alloc_sweep": 1, "visible_list": 2, "integrate": 3, "range": 4, "raycast": 5, "icp_maps": 6}

WORKLOADS = {
    2: dict(w=640, h=480, voxel="s", index="hash", voxelSize=0.004, mu=0.02, blocks=0x40000, stopAtMax=False, colour=False,
            kernel="raycast", distinct=PERIOD,
            name="BASELINE configs[1]: synthetic 640x480 depth (sphere+wall, bench trajectory), hash TSDF ITMVoxel_s, "
                 "4 mm voxels, mu 0.02, 0x40000-block pool; allocate+integrate+expected-depths+ICP raycast per frame"),
    3: dict(w=640, h=480, voxel="s", index="dense", voxelSize=0.004, mu=0.02, blocks=0, stopAtMax=True, colour=False,
            kernel="integrate", distinct=PERIOD,
            name="BASELINE configs[2]: synthetic 640x480 depth (bench trajectory), dense ITMPlainVoxelArray 512^3 ITMVoxel_s, "
                 "4 mm voxels, stopIntegratingAtMaxW; integrate+expected-depths+ICP raycast per frame"),
    5: dict(w=1280, h=960, voxel="f_rgb", index="hash", voxelSize=0.002, mu=0.02, blocks=0x40000, stopAtMax=False, colour=True,
            kernel="integrate", distinct=25,
            name="BASELINE configs[4]: synthetic 1280x960 depth+rgb (bench trajectory), hash TSDF ITMVoxel_f_rgb, 2 mm voxels, "
                 "mu 0.02, 0x40000-block pool; allocate+integrate(colour)+expected-depths+ICP raycast per frame"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU baseline sample (0 = per-config default)")
    ap.add_argument("--no-exchange", action="store_true", help="skip the visible-list all-gather at N>1")
    ap.add_argument("--force-exchange", action="store_true", help="run the all-gather path even with one rank (self-test)")
    ap.add_argument("--exchange-batch", type=int, default=8,
                    help="frames per all-gather; 1 = every frame (measured on MI355X through torch.distributed: per-frame costs 24 percent of the frame rate, the host-side collective call being the bound; 8 costs 5 percent, see DESIGN.md section 6)")
    ap.add_argument("--exchange-impl", choices=["library", "torch"], default="library",
                    help="who issues the collective: the library (RCCL from C++) or torch.distributed (always used by the CPU self-tests)")
    ap.add_argument("--streams-per-gpu", type=int, default=1,
                    help="independent scenes co-scheduled on one GPU, each on its own HIP stream (separate figure; headline is 1)")
    ap.add_argument("--no-host-threads", dest="host_threads", action="store_false",
                    help="feed the k streams of --streams-per-gpu from one host thread instead of one thread per stream")
    ap.add_argument("--timer-frames", type=int, default=0,
                    help="frames of the extra, untimed repetition in which every launch of the roofline kernel is bracketed by HIP events (0 = max(--steps, 64))")
    ap.add_argument("--frame-call", choices=["four", "process_frame", "ahead"], default="four",
                    help="how the timed region issues a frame: the reference's four engine calls (default, the headline), itm_process_frame, or "
                         "itm_process_frame_ahead (a marked line: the next frame's pose is handed over before this frame is done)")
    ap.add_argument("--min-measured-s", type=float, default=0.25, help="runs with --steps < 100 repeat the timed region until this much has been measured")
    ap.add_argument("--raw-serial", action="store_true", help="raw-depth legs: the upload on the frame's own stream instead of the library's stager (A/B)")
    ap.add_argument("--raw-depth", action="store_true",
                    help="SURVEY 8d second figure as the run's value: each step uploads the 16-bit raw frame from pinned host memory "
                         "(w*h*2 bytes over PCIe) and converts it with itm_update_view inside the timed region (never the headline)")
    ap.add_argument("--origin-offset", default="0,0,0",
                    help="X,Y,Z metres added to every camera position (scene and trajectory translated together: same depth images, "
                         "block coordinates far from the world origin)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the child runs of BASELINE configs[2] (dense 512^3) and configs[4] (1280x960 colour) that the default single-GPU run appends as `other_configs`")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the secondary measurements of the default run (PCIe-inclusive rate, table-walk ray cast, stream-copy peak)")
    ap.add_argument("--lib", default=None, help="alternative shared library exporting the same C-ABI (tests: a host-memory backend)")
    ap.add_argument("--lib-prefix", default="itm_")
    ap.add_argument("--debug-keys", default="", help="comma-separated itm_debug_set keys switched on for the run (A/B of a replaced code path; a marked line, not the headline)")
    return ap.parse_args()


# -------------------------------------------------------------------------------------------------------------
# parent: spawn one process per GPU.  Nothing here imports torch or touches the GPU.
# -------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _parse_cpulist(text: str):
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def rank_cpu_set(local_rank: int, local_world: int):
    """The host cores of one rank: a DISJOINT share of the cores this process may run on -- of the NUMA node of the rank's GPU where sysfs
    names one (/sys/class/drm/renderD<128 + i>/device/numa_node, the node's cores split among the ranks whose GPUs sit on it), of the
    whole set otherwise.  The per-frame path is host-issued (four C-ABI calls + the exchange's issuer thread per rank): eight ranks
    that wander over all cores migrate between NUMA nodes and share cores with each other's issuer threads.  Reads sysfs only --
    nothing here touches a GPU.  Returns None when there is nothing to restrict (one rank, or fewer cores than ranks)."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if local_world <= 1 or len(avail) < local_world:
        return None

    def node_of(i):
        try:
            with open(f"/sys/class/drm/renderD{128 + i}/device/numa_node") as f:
                n = int(f.read().strip())
            return n if n >= 0 else None
        except (OSError, ValueError):
            return None
    nodes = [node_of(i) for i in range(local_world)]
    mine = nodes[local_rank]
    if mine is not None:
        try:
            with open(f"/sys/devices/system/node/node{mine}/cpulist") as f:
                cores = sorted(_parse_cpulist(f.read()) & set(avail))
        except (OSError, ValueError):
            cores = []
        peers = [i for i in range(local_world) if nodes[i] == mine]
        if len(cores) >= len(peers):
            k, per = peers.index(local_rank), len(cores) // len(peers)
            return set(cores[k * per:(k + 1) * per])
    per = len(avail) // local_world
    return set(avail[local_rank * per:(local_rank + 1) * per])


def launch_ranks(args) -> int:
    n = args.gpus
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "ITM_BENCH_SPAWNED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if env.get("ITM_BENCH_SHARED_GPU") == "1":
            # several PROCESSES on one GPU: each library sees one scene and would pick the one-launch visible list, whose workgroups wait
            # for each other -- between processes the queues are time-sliced and those waits collapse the frame rate (36 frames/s measured)
            env.setdefault("ITM_ONE_PASS_LIST", "0")
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    text = procs[0].communicate()[0].decode(errors="replace")
    codes = [procs[0].returncode]
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            codes.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()                       # exact child, never a pattern
            codes.append(-9)
    lines = [ln for ln in text.splitlines() if ln.strip()]
    for ln in lines[:-1]:
        print(ln, file=sys.stderr)
    if any(codes) or not lines:
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        if lines:
            print(lines[-1], file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


# -------------------------------------------------------------------------------------------------------------
# one rank
# -------------------------------------------------------------------------------------------------------------
class Stream:
    """One depth stream: scene + render state + its frames resident in the backend's memory."""

    def __init__(self, be, capi, synth, torch, wl, stream_id, device, hip_stream, offset=(0.0, 0.0, 0.0), raw=False):
        import numpy as np
        w, h = wl["w"], wl["h"]
        self.w, self.h, self.torch, self.be = w, h, torch, be
        vox = {"s": capi.VOXEL_S, "f_rgb": capi.VOXEL_F_RGB}[wl["voxel"]]
        idx = capi.INDEX_HASH if wl["index"] == "hash" else capi.INDEX_DENSE
        params = capi.default_params(voxelSize=wl["voxelSize"], mu=wl["mu"], stopIntegratingAtMaxW=wl["stopAtMax"])
        self.scene = be.create_scene(vox, idx, params, localBlockNum=wl["blocks"])
        # the four-call path of the timed region relies on the library recording the first three calls (include/itm_hip.h,
        # itm_scene_set_deferred_fusion): this host accepts that contract -- it never touches the stream between the four calls
        self.scene.set_deferred_fusion(True)
        self.scene.reco.ResetScene()
        self.rs = self.scene.vis.CreateRenderState((w, h))
        intr = synth.intrinsics_for(w, h)
        nd = wl["distinct"]
        # the trajectory has period 100; with fewer resident frames (config 5) every 100/nd-th pose is kept
        ks = [k * (PERIOD // nd) for k in range(nd)]
        frames = np.stack([synth.depth_frame(w, h, synth.bench_position(k, stream_id), intr) for k in ks])
        self.depth = torch.from_numpy(frames).to(device)
        self.points = torch.empty((h, w, 4), dtype=torch.float32, device=device)
        self.normals = torch.empty((h, w, 4), dtype=torch.float32, device=device)
        self.rgb = torch.from_numpy(synth.rgb_frame(w, h)).to(device) if wl["colour"] else None
        off = [np.float32(v) for v in offset]
        self.poses = [synth.pose_matrix([np.float32(p) + o for p, o in zip(synth.bench_position(k, stream_id), off)]) for k in ks]
        self.views = [capi.View(self.depth[i].data_ptr(), w, h, M_d=self.poses[i], intr_d=intr,
                                rgb=(self.rgb.data_ptr() if self.rgb is not None else None), w_rgb=w, h_rgb=h, intr_rgb=intr).struct()
                      for i in range(nd)]
        self.poses_c = [(C.c_float * 16)(*[float(x) for x in m]) for m in self.poses]   # built once, not per frame
        self.nd = nd
        self.hip_stream = hip_stream
        self.sp = C.c_void_p(hip_stream.cuda_stream if hip_stream is not None else None)
        self.sh, self.rh = C.c_void_p(self.scene.h), C.c_void_p(self.rs.h)
        self.pp, self.np_ = C.c_void_p(self.points.data_ptr()), C.c_void_p(self.normals.data_ptr())
        self.raw_ready = False
        self.intr_c = (C.c_float * 4)(*intr)
        self.synth, self.capi, self.intr, self.ks, self.device, self.stream_id = synth, capi, intr, ks, device, stream_id
        if raw:
            self.enable_raw()

    def enable_raw(self):
        """Raw 16-bit frames in pinned host memory + the buffers of itm_update_view; the views of this mode read the converted image."""
        if self.raw_ready:
            return
        import numpy as np
        torch, synth, capi, w, h = self.torch, self.synth, self.capi, self.w, self.h
        raws = np.stack([synth.raw_depth_mm(w, h, synth.bench_position(k, self.stream_id), self.intr) for k in self.ks])
        self.raw_host = torch.from_numpy(raws)
        if self.device != "cpu":
            self.raw_host = self.raw_host.pin_memory()
        self.raw_ptrs = [self.raw_host[i].data_ptr() for i in range(self.nd)]
        self.raw_dev = torch.empty((2, h, w), dtype=torch.int16, device=self.device)      # two upload slots (CPU shims; --raw-serial)
        # the product uploads through the library's stager: a copy stream of its own, one frame ahead of the frame being fused
        self.stager, self.staged = None, -1
        self.converting, self.holding = False, False
        if self.device != "cpu" and "depth_stager_create" in self.be.fn and self.be.prefix == "itm_":
            g = C.c_void_p()
            self.ahead = max(1, int(os.environ.get("ITM_BENCH_RAW_AHEAD", "1")))
            self.be.check(self.be.fn["depth_stager_create"](w, h, self.ahead + 3, C.byref(g)), "depth_stager_create")
            # the copy converts as well (convertDepthAffineToFloat: the view has neither bilateral filter nor noise model): the slot's
            # float image IS the view's depth, no conversion launch on the frame's stream; held until the next frame has been submitted
            self.converting = "depth_stager_set_conversion" in self.be.fn
            if self.converting:
                self.be.check(self.be.fn["depth_stager_set_conversion"](g, 1, 0.001, 0.0, float(self.intr[0])), "depth_stager_set_conversion")
            self.stager, self.holding = g, False
        self.depth_conv = torch.empty((h, w), dtype=torch.float32, device=self.device)
        self.scratch = torch.empty((h, w), dtype=torch.float32, device=self.device)
        self.raw_views = [self.capi.View(self.depth_conv.data_ptr(), w, h, M_d=self.poses[i], intr_d=self.intr,
                                         rgb=(self.rgb.data_ptr() if self.rgb is not None else None), w_rgb=w, h_rgb=h, intr_rgb=self.intr).struct()
                          for i in range(self.nd)]
        self.raw_ready = True


def worker(args) -> int:
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with matching values", file=sys.stderr)
        return 2
    # host cores of this rank, before anything touches the GPU (under torchrun as well as under bench.py's own launcher)
    cpus = None
    if world > 1 and os.environ.get("ITM_BENCH_NO_AFFINITY") != "1":
        cpus = rank_cpu_set(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
        if cpus:
            try:
                os.sched_setaffinity(0, cpus)
                os.environ.setdefault("OMP_NUM_THREADS", str(len(cpus)))
            except OSError as e:
                print(f"bench.py: rank {rank}: sched_setaffinity failed ({e})", file=sys.stderr)
                cpus = None
    # stdout carries ONE line, the JSON of rank 0.  RCCL prints a version banner through C stdio when a communicator is created:
    # from here on file descriptor 1 is stderr for everything (C libraries and Python alike), the JSON goes to the saved descriptor.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    import infinitam_amd as itm
    from infinitam_amd import capi, synth
    be = capi.Backend(args.lib, args.lib_prefix) if args.lib else itm.load()
    product = args.lib is None
    for key in [k for k in args.debug_keys.split(",") if k.strip()]:
        key, _, val = key.partition("=")                      # "9" switches key 9 on, "16=2" sets key 16 to 2
        be.check(be.fn["debug_set"](int(key), int(val) if val else 1), "debug_set")
    on_gpu = be.on_device
    if on_gpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # self-test hook (never set by the driver): ITM_BENCH_SHARED_GPU=1 lets several ranks share GPU 0 with the gloo
    # backend, to exercise the multi-rank control flow on a one-GPU box; RCCL needs one GPU per rank
    shared = os.environ.get("ITM_BENCH_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    device = "cpu"
    if on_gpu:
        torch.cuda.set_device(local_rank)
        device = "cuda"
    backend_name = "none"
    if world > 1 or args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if shared or not on_gpu:
            backend_name = "gloo"
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            backend_name = "nccl (RCCL)"
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if on_gpu:
        be.check(be.fn["set_device"](local_rank), "set_device")
    ctl_device = "cpu" if backend_name.startswith("gloo") else device      # where the control plane's small tensors live

    wl = WORKLOADS[args.config]
    offset = tuple(float(v) for v in args.origin_offset.split(","))
    assert len(offset) == 3, "--origin-offset X,Y,Z"
    k_streams = max(1, args.streams_per_gpu)
    streams = []
    for j in range(k_streams):
        hs = None
        if on_gpu:
            hs = torch.cuda.current_stream() if k_streams == 1 else torch.cuda.Stream()
        streams.append(Stream(be, capi, synth, torch, wl, rank * k_streams + j, device, hs, offset=offset, raw=args.raw_depth))
    fn = be.fn["process_frame"]
    fn_ahead = be.fn["process_frame_ahead"]
    fn_alloc, fn_integrate, fn_expected, fn_icp = (be.fn[n] for n in ("allocate_scene_from_depth", "integrate_into_scene", "create_expected_depths", "create_icp_maps"))
    can_ahead = product and wl["index"] == "hash"
    fn_view = be.fn["update_view"]

    exchange = (world > 1 and not args.no_exchange) or args.force_exchange
    exs = []
    # the exchange is issued by the library (RCCL called from C++, infinitam_amd/csrc/exchange.hip) whenever the product runs with one
    # GPU per rank; the torch.distributed path remains for the CPU / shared-GPU self-tests and for --exchange-impl torch
    # (shared-GPU self-test: RCCL refuses two ranks on one device, so the library's exchange runs there only over a stand-in transport
    # named by ITM_RCCL_LIBRARY -- tests/cpp/rccl_standin.cpp -- which exercises this file's N-rank bootstrap and the library's N-rank code)
    standin = os.environ.get("ITM_RCCL_LIBRARY") if shared else None
    native = exchange and product and on_gpu and (not shared or bool(standin)) and args.exchange_impl == "library"
    if exchange and native:
        from infinitam_amd.streams import NativeExchange
        for _ in streams:
            uid = None
            if world > 1:
                t = torch.zeros(128, dtype=torch.uint8, device=ctl_device)
                if rank == 0:
                    try:
                        t.copy_(torch.tensor(list(NativeExchange.unique_id(be)), dtype=torch.uint8))
                    except Exception as e:      # noqa: BLE001 -- an all-zero id tells every rank to take the torch path
                        print(f"bench.py: no RCCL unique id ({e})", file=sys.stderr)
                dist.broadcast(t, 0)
                uid = bytes(t.cpu().tolist())
                if not any(uid):
                    break
            try:
                # communicator creation is a rendezvous of all ranks inside RCCL: bounded here (a daemon thread with this rank's device
                # current), so that a rank that cannot complete it reports failure and every rank takes the torch path instead of
                # the whole run hanging
                box = {}

                def make_exchange(uid=uid, box=box):
                    try:
                        if on_gpu:
                            torch.cuda.set_device(local_rank)
                        box["ex"] = NativeExchange(be, world, rank, MAX_IDS, batch=max(1, args.exchange_batch), unique_id=uid)
                    except Exception as e:      # noqa: BLE001
                        box["err"] = e

                th = threading.Thread(target=make_exchange, daemon=True)
                th.start()
                th.join(float(os.environ.get("ITM_EXCHANGE_INIT_TIMEOUT", "120")))
                if th.is_alive():
                    # the thread may be inside ncclCommInitRank holding RCCL and device state: this rank cannot go on beside it.
                    # Say so and leave with a failure (the launcher then fails the run) instead of tearing down around a stuck thread.
                    print(f"bench.py: rank {rank}: no RCCL communicator within the time limit", file=sys.stderr, flush=True)
                    os._exit(3)
                if "err" in box:
                    raise box["err"]
                exs.append(box["ex"])
            except Exception as e:      # noqa: BLE001 -- reported below, every rank then takes the torch path together
                print(f"bench.py: rank {rank}: library exchange unavailable ({e})", file=sys.stderr)
                break
        ok = torch.tensor([1 if len(exs) == len(streams) else 0], dtype=torch.int32, device=ctl_device)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)      # all ranks or none
        if int(ok.item()) == 0:
            for ex in exs:
                ex.close()
            exs, native = [], False
    if exchange and not native:
        from infinitam_amd.streams import VisibleListExchange
        exs = [VisibleListExchange(be, world, rank, MAX_IDS, device=device, batch=max(1, args.exchange_batch)) for _ in streams]

    # "call": how a frame is issued -- "four" = the reference's four engine calls (the library records three and launches the fused
    # frame at the fourth, infinitam_amd/csrc/pending.hip), "process_frame" = the single entry point, "ahead" = itm_process_frame_ahead
    mode = {"raw": args.raw_depth, "call": (args.frame_call if (args.frame_call != "ahead" or can_ahead) else "process_frame")}

    def issue(s, v, view, nxt):
        """One frame of stream s for view struct `view` (a ViewStruct); nxt = the view that follows, or None."""
        c = mode["call"]
        if c == "four":
            return (fn_alloc(s.sh, v, s.rh, 0, s.sp) or fn_integrate(s.sh, v, s.rh, s.sp) or fn_expected(s.sh, view.M_d, view.intr_d, s.rh, s.sp)
                    or fn_icp(s.sh, v, s.rh, s.pp, s.np_, s.sp))
        if c == "ahead":
            return fn_ahead(s.sh, v, (C.byref(nxt) if nxt is not None else None), s.rh, s.pp, s.np_, s.sp)
        return fn(s.sh, v, s.rh, s.pp, s.np_, s.sp)

    def step_stream(j, k, last=False):
        s = streams[j]
        i = k % s.nd
        if mode["raw"]:
            # the frame as the sensor delivers it: 16-bit raw over PCIe from pinned memory, convertDepthAffineToFloat on the device,
            # the upload through the library's stager (copy stream, one frame ahead); conversion and fusion on the frame's stream
            raw_ptr = None
            if s.stager is not None and not args.raw_serial:
                def upload(step):
                    be.check(be.fn["depth_stager_upload"](s.stager, C.c_void_p(s.raw_ptrs[step % s.nd])), "depth_stager_upload")
                    s.staged = step
                if s.holding:                       # the previous frame's slot: everything that read its depth has been submitted
                    be.check(be.fn["depth_stager_release"](s.stager, s.sp), "depth_stager_release")
                    s.holding = False
                if s.staged < k or s.staged > k + s.ahead:
                    s.staged = k - 1                # first frame of a leg (nothing of it is in flight)
                while s.staged < k + (0 if last else s.ahead):
                    upload(s.staged + 1)            # the next frames travel while this one is fused
                dev = C.c_void_p()
                if s.converting:
                    depth_dev = C.c_void_p()
                    be.check(be.fn["depth_stager_acquire_depth"](s.stager, s.sp, None, C.byref(depth_dev)), "depth_stager_acquire_depth")
                    s.raw_views[i].depth = depth_dev.value
                    s.holding = True
                else:
                    be.check(be.fn["depth_stager_acquire"](s.stager, s.sp, C.byref(dev)), "depth_stager_acquire")
                    raw_ptr = dev
            else:
                slot = s.raw_dev[k & 1]
                if on_gpu:
                    with torch.cuda.stream(s.hip_stream):
                        slot.copy_(s.raw_host[i], non_blocking=True)
                else:
                    slot.copy_(s.raw_host[i])
                raw_ptr = C.c_void_p(slot.data_ptr())
            converted = s.stager is not None and not args.raw_serial and s.converting
            if not converted:
                s.raw_views[i].depth = s.depth_conv.data_ptr()
                rc = fn_view(raw_ptr, s.w, s.h, 1, 0.001, 0.0, s.intr_c, 0, 0, C.c_void_p(s.depth_conv.data_ptr()),
                             C.c_void_p(s.scratch.data_ptr()), None, None, s.sp)
                if rc:
                    be.check(rc, "update_view")
                if s.stager is not None and not args.raw_serial:
                    be.check(be.fn["depth_stager_release"](s.stager, s.sp), "depth_stager_release")      # (the conversion is what read the slot)
            rc = issue(s, C.byref(s.raw_views[i]), s.raw_views[i], None)       # (no frame can be announced ahead: its depth image does not exist yet)
            if converted and last and not rc:          # the leg ends here: nothing stays held across legs
                be.check(be.fn["depth_stager_release"](s.stager, s.sp), "depth_stager_release")
                s.holding = False
        else:
            # the last frame of a run() names no successor: the next leg may start anywhere
            rc = issue(s, C.byref(s.views[i]), s.views[i], (None if last else s.views[(k + 1) % s.nd]))
        if rc:
            be.check(rc, "frame (" + mode["call"] + ")")
        if exchange:
            # record copy on the frame stream, all-gather on a side stream (off the critical path)
            exs[j].step(s.rs.h, s.poses_c[i], s.hip_stream)

    def step(k, last=False):
        for j in range(len(streams)):
            step_stream(j, k, last)

    def run(first, last):
        """Frames [first, last) of every stream.  With k > 1 streams per GPU each stream is fed by its own host thread (the
        library call releases the GIL): one thread issues ~6 launches per frame at ~10 us each, which is what bounded k = 4."""
        if len(streams) == 1 or exchange or not args.host_threads:
            for k in range(first, last):
                step(k, k == last - 1)
            return
        import threading
        errors = []

        def feed(j):
            try:
                if on_gpu:
                    torch.cuda.set_device(local_rank)
                for k in range(first, last):
                    step_stream(j, k, k == last - 1)
            except BaseException as e:      # noqa: BLE001 -- re-raised on the main thread
                errors.append(e)
        threads = [threading.Thread(target=feed, args=(j,)) for j in range(len(streams))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def drain():
        if exchange and native:
            # every collective the library's issuer thread still holds is put on its stream and completes (the synchronisation below
            # waits for streams, not for that thread's queue): no all-gather of the library's communicator is left to overlap -- in an
            # order that could differ from rank to rank -- with the control plane's collective on the other communicator
            for ex in exs:
                ex.self_check()
        sync()

    def barrier():
        drain()
        if world > 1:
            dist.barrier()
        sync()

    sync()
    run(0, args.warmup)
    barrier()
    timed_kernel = TK[wl["kernel"]]

    def timed_region(first):
        """EXACTLY --steps frames: a barrier + synchronisation in front, each rank's own synchronisation (frames and exchange) behind; returns the MAX
        over ranks of the elapsed times -- the job's time; the all-reduce that forms it is the closing barrier -- and this rank's own."""
        barrier()
        t0 = time.perf_counter()
        run(first, first + args.steps)
        drain()                                   # this rank's frames AND its share of the exchange are done ...
        mine = time.perf_counter() - t0           # ... at its own clock; the MAX over ranks below is the job's time (and the closing barrier)
        if world == 1:
            return mine, mine
        tt = torch.tensor([mine], dtype=torch.float64, device=ctl_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()), mine

    # Short runs are repeated: 20 frames are 1.7 ms, less than the jitter of one host wake-up.  The number of repetitions follows from
    # the first one (the same on every rank: it is computed from the all-reduced time), the MEDIAN repetition is the one reported.
    reps = [timed_region(args.warmup)]
    if args.steps < 100 and args.min_measured_s > 0:
        want = int(min(2000, max(0, -(-args.min_measured_s // max(reps[0][0], 1e-6)) - 1)))
        for r in range(want):
            reps.append(timed_region(args.warmup + (r + 1) * args.steps))
    order = sorted(range(len(reps)), key=lambda r: reps[r][0])
    med = order[len(order) // 2]
    elapsed, elapsed_local = reps[med]
    fps_minmax = None
    if world > 1:
        tn = torch.tensor([elapsed_local], dtype=torch.float64, device=ctl_device)
        dist.all_reduce(tn, op=dist.ReduceOp.MIN)
        per = k_streams * args.steps
        fps_minmax = [round(per / elapsed, 1), round(per / float(tn.item()), 1)]
    rep_stats = {"count": len(reps), "measured_s": round(sum(t for t, _ in reps), 4),
                 "fps_min_median_max": [round(world * k_streams * args.steps / reps[order[-1]][0], 1), round(world * k_streams * args.steps / elapsed, 1),
                                        round(world * k_streams * args.steps / reps[order[0]][0], 1)]}

    # ---- the roofline kernel: one extra, UNTIMED repetition with an event pair around EVERY launch of it (and of the integration and
    # the visible-list launch), then empty brackets on the same stream for what an event pair itself adds to an interval ----
    kernel_times = None
    if rank == 0 and product and on_gpu:
        s0 = streams[0]
        nt = args.timer_frames if args.timer_frames > 0 else max(args.steps, 64)
        timed_set = sorted({timed_kernel, TK["request"], TK["integrate"], TK["visible_list"], TK["raycast"]} if wl["index"] == "hash" else {timed_kernel, TK["raycast"]})
        # the SAME frames on the SAME state as the timed region saw them: the scene is reset and warmed up again first (what a launch
        # costs depends on the scene -- BASELINE configs[2]'s dense integration gets cheaper as weights saturate: 83 us over frames
        # 20..219 of a fresh scene, 61 us when the same frames are fused a second time)
        was, exchange = exchange, False         # rank 0 alone: no collective may be issued here (the other ranks wait at the barrier below)
        for s_ in streams:
            s_.scene.reco.ResetScene(stream=(s_.hip_stream.cuda_stream if s_.hip_stream is not None else None))
        run(0, args.warmup)
        sync()
        s0.scene.profile_read(reset=True)
        s0.scene.profile_enable(sum(1 << t for t in timed_set))
        s0.scene.profile_sample(1)
        run(args.warmup, args.warmup + nt)
        exchange = was
        sync()
        s0.scene.profile_calibrate(64, s0.hip_stream.cuda_stream if s0.hip_stream is not None else None)
        sync()
        kernel_times = s0.scene.profile_read(reset=True)
        s0.scene.profile_enable(0)
    if world > 1:
        barrier()

    # what the exchange costs THIS run: the same frames once more without it (every rank, no collective involved)
    exchange_cost = None
    if exchange and (on_gpu or world > 1):
        n3 = min(args.steps, 200)
        was, exchange = exchange, False
        run(0, min(args.warmup, 20)); sync()
        t2 = time.perf_counter()
        run(args.warmup, args.warmup + n3); sync()
        without = k_streams * n3 / (time.perf_counter() - t2)
        exchange = was
        with_ = k_streams * args.steps / elapsed_local
        exchange_cost = {"frames_per_collective": max(1, args.exchange_batch), "rank0_fps_with_exchange": round(with_, 1),
                         "rank0_fps_without_exchange": round(without, 1), "cost_percent": round(100.0 * (1.0 - with_ / without), 1),
                         "north_star_says": "per-frame all-gather; the default batches 8 frames per collective (same records, same table, delivered up to 7 frames later)",
                         "why_not_every_frame": "one collective per frame was measured at 8.6 percent of the frame rate issued from a C loop and 28 percent from "
                                                "this harness on one MI355X (DESIGN.md section 6); --exchange-batch 1 selects it"}

    # what the exchange delivered: rank 0 decodes the newest gathered table -- one block per rank: pose + visible ids
    exchange_table = None
    if exchange and exs:
        drain()
        try:
            tab = exs[0].table()
            exchange_table = {"blocks": len(tab), "blocks_with_ids": sum(1 for _, ids in tab if len(ids) > 0),
                              "visible_counts": [int(len(ids)) for _, ids in tab],
                              "camera_y_m": [round(-float(M[13]), 4) for M, _ in tab],
                              "what": "newest gathered table decoded on rank 0 (infinitam_amd.streams.decode_table): stream g's camera is offset 0.05 g m in y"}
        except Exception as e:      # noqa: BLE001 -- a missing figure, not a failed run
            exchange_table = {"error": str(e)[:200]}
    # host cores: every rank's share (count, first core), gathered with one all-reduce
    host_cores = None
    if world > 1:
        hc = torch.zeros(world * 2, dtype=torch.int32, device=ctl_device)
        if cpus:
            hc[2 * rank], hc[2 * rank + 1] = len(cpus), min(cpus)
        dist.all_reduce(hc)
        hc = hc.cpu().tolist()
        host_cores = {"per_rank_count": hc[0::2], "per_rank_first_core": hc[1::2],
                      "what": "os.sched_setaffinity per rank before anything touches the GPU: a disjoint share of the cores of the GPU's NUMA node (sysfs), "
                              "or of all cores where sysfs names none; count 0 = not restricted"}

    counters = streams[0].scene.counters(streams[0].rs)
    roofline = None
    if rank == 0 and product and kernel_times is not None:
        roofline = read_roofline(args.config, wl, kernel_times, counters, streams[0].scene)

    # ---- secondary figures of the default single-GPU run (after the timed region; none of them is `value`) --------------------
    extra = {}
    if rank == 0 and world == 1 and product and k_streams == 1 and on_gpu and not exchange and not args.no_extra_legs and not args.raw_depth:
        s0 = streams[0]
        n2 = max(args.steps, 100)

        def fresh():
            """Every leg starts where the timed region started: a reset scene + the same warm-up (what a frame costs depends on the
            scene's state -- config 3's dense integration gets cheaper once weights saturate and stop integrating)."""
            for s_ in streams:
                s_.scene.reco.ResetScene(stream=(s_.hip_stream.cuda_stream if s_.hip_stream is not None else None))

        def leg(call):
            """The same frames through another entry point: reset, warm-up, then n2 frames between two synchronisations."""
            was = mode["call"]
            mode["call"] = call
            fresh()
            run(0, args.warmup); sync()
            t1 = time.perf_counter()
            run(args.warmup, args.warmup + n2); sync()
            dt = time.perf_counter() - t1
            mode["call"] = was
            return round(n2 / dt, 2)
        # (0) the same frames through the library's own single entry point, and with the next frame announced ahead.  Neither is a call
        #     the reference makes: the first shows what the four-call path costs over it, the second needs the NEXT frame's pose before
        #     this frame is done (an offline sequence has it, a closed tracking loop cannot)
        if mode["call"] == "four":
            extra["process_frame_entry_point"] = {"value": leg("process_frame"), "unit": "frames/s", "steps": n2,
                                                  "what": "itm_process_frame: one call per frame instead of the reference's four (same launches)"}
            if can_ahead:
                extra["with_lookahead"] = {"value": leg("ahead"), "unit": "frames/s", "steps": n2,
                                           "what": "itm_process_frame_ahead: the next resident frame's block requests ride in this frame's ray-cast launch (4 launches per frame); "
                                                   "needs the next frame's pose before this frame is done -- never the headline"}
        # (1) SURVEY 8d's second figure: the frame arrives as 16-bit raw depth in pinned host memory; H2D copy + itm_update_view
        #     (convertDepthAffineToFloat) inside the timed region
        s0.enable_raw()
        mode["raw"] = True
        fresh()
        run(0, args.warmup); sync()
        t1 = time.perf_counter()
        run(args.warmup, args.warmup + n2); sync()
        dt = time.perf_counter() - t1
        mode["raw"] = False
        extra["with_h2d_raw_depth"] = {"value": round(n2 / dt, 2), "unit": "frames/s", "steps": n2,
                                       "what": f"per frame {wl['w'] * wl['h'] * 2} bytes of raw short depth from pinned host memory over PCIe (itm_depth_stager: a copy stream, one frame ahead; the copy kernel also does convertDepthAffineToFloat -- itm_update_view's work for a view without filter and noise model -- so the slot's float image is the view's depth) + the same fused frame"}
        # (2) the same roofline kernel when the acceleration structures (block directory, sdf mirror) do not answer -- blocks whose
        #     cells are owned by other blocks, or a device without room for them -- i.e. on the reference's own table walk
        if wl["index"] == "hash":
            for key in (1, 2):                 # ITM_DEBUG_NO_DIRECTORY, ITM_DEBUG_NO_SDF_MIRROR
                be.check(be.fn["debug_set"](DEBUG_KEYS_FALLBACK[key], 1), "debug_set")
            fresh()
            run(0, args.warmup); sync()
            s0.scene.profile_read(reset=True)
            s0.scene.profile_enable(1 << timed_kernel)
            s0.scene.profile_sample(1)
            t1 = time.perf_counter()
            run(args.warmup, args.warmup + n2); sync()
            dt = time.perf_counter() - t1
            prof = s0.scene.profile_read(reset=True)[wl["kernel"]]
            s0.scene.profile_enable(0)
            for key in (1, 2):
                be.check(be.fn["debug_set"](DEBUG_KEYS_FALLBACK[key], 0), "debug_set")
            extra["table_walk_fallback"] = {"value": round(n2 / dt, 2), "unit": "frames/s", "steps": n2,
                                            "kernel_us": round(prof["total_ms"] * 1e3 / max(1, prof["calls"]), 2),
                                            "what": "block directory and sdf mirror switched off: every look-up walks the hash table as the reference does (event pair on every launch)"}
        # (3) the same frames on a scene whose sdf mirror is PAGED (a 768 MB pool behind a page table instead of the dense 17 GB cube:
        #     what every scene gets once the device no longer has 3 x 17 GB to spare, or with ITM_MIRROR=paged)
        if wl["index"] == "hash" and wl["voxel"] == "s" and s0.scene.accel_info()["mirror_pages"] == 0 and s0.scene.accel_info()["mirror_bytes"] > 0:
            os.environ["ITM_MIRROR"] = "paged"
            try:
                sp = Stream(be, capi, synth, torch, wl, rank * k_streams, device, s0.hip_stream, offset=offset)
            finally:
                del os.environ["ITM_MIRROR"]
            streams.append(sp)
            keep, streams[0] = streams[0], sp

            def run_paged(a, b):
                for k in range(a, b):
                    step_stream(0, k, k == b - 1)
            run_paged(0, args.warmup); sync()
            t1 = time.perf_counter()
            run_paged(args.warmup, args.warmup + n2); sync()
            dt = time.perf_counter() - t1
            sp.scene.profile_read(reset=True)
            sp.scene.profile_enable(1 << timed_kernel); sp.scene.profile_sample(1)
            run_paged(args.warmup, args.warmup + 64); sync()
            sp.scene.profile_calibrate(64, sp.hip_stream.cuda_stream if sp.hip_stream is not None else None); sync()
            pr = sp.scene.profile_read(reset=True)
            sp.scene.profile_enable(0)
            ai = sp.scene.accel_info()
            half_pair = 0.5 * pr["empty"]["total_ms"] / max(1, pr["empty"]["calls"])
            extra["paged_mirror"] = {"value": round(n2 / dt, 2), "unit": "frames/s", "steps": n2,
                                     "kernel_us": round((pr[wl["kernel"]]["total_ms"] / max(1, pr[wl["kernel"]]["calls"]) - half_pair) * 1e3, 2),
                                     "mirror_bytes": ai["mirror_bytes"], "mirror_pages_mapped": ai["mirror_pages_mapped"], "mirror_pages": ai["mirror_pages"],
                                     "what": "the sdf mirror as 4 MB pages from a 768 MB pool behind a 16 KB table instead of the dense 17 GB cube of the headline scene"}
            streams[0] = keep
            streams.pop()
    # (4) an EXPLORING camera: the headline trajectory is 100-periodic, so from frame 100 on nothing is allocated; here the camera of the
    #     parity trajectory (1 cm per frame along x, never turning back) starts on a reset scene -- every frame requests and allocates
    #     blocks, the allocation sweep of _CPU.cpp:175-227 and the visible-list re-tests have work in every frame
    if (rank == 0 and world == 1 and product and k_streams == 1 and on_gpu and not exchange and not args.no_extra_legs and not args.raw_depth
            and args.config == 2 and mode["call"] == "four"):
        extra["exploring"] = run_exploring(streams[0], be, capi, synth, torch, wl, issue, sync, timed_set=("request", "visible_list", "integrate", "raycast"))
    if rank == 0 and world == 1 and product and on_gpu and roofline is not None and not args.no_extra_legs:
        roofline["peak_measured"] = measured_stream_peak()
        pm = roofline["peak_measured"] or {}
        if pm.get("copy_GBs"):
            # the same achieved rate against what THIS box's memory system delivers to a streaming read+write kernel (the integration
            # kernels read and write every voxel of every visible block), beside the fraction of the 8 TB/s vendor peak
            roofline["frac_of_measured_copy_peak"] = round(roofline["achieved"] / pm["copy_GBs"], 4)
            for o in (roofline.get("other_kernels") or {}).values():
                o["frac_of_measured_copy_peak"] = round(o["achieved"] / pm["copy_GBs"], 4)

    cpu_baseline, parity_check = None, None
    if rank == 0 and world == 1 and product and k_streams == 1 and not args.no_cpu_baseline:
        cpu_baseline, parity_check = run_cpu_baseline(args.config, wl, args.cpu_frames, on_gpu)
        if "exploring" in extra and on_gpu:
            extra["exploring"].update(run_exploring_check(wl))

    # the other single-GPU configurations of BASELINE.json, each as its own bench.py process (its own scene, its own parity check and
    # CPU sample), after everything of this run has been measured: the driver's ONE line then carries all three
    other_configs = None
    if (rank == 0 and world == 1 and product and k_streams == 1 and on_gpu and not args.no_extra_legs and not args.no_other_configs
            and not args.raw_depth and args.config == 2 and not args.debug_keys):
        sync()
        other_configs = {f"config{c}": run_other_config(c, args) for c in (3, 5)}

    out = None
    if rank == 0:
        total_frames = world * k_streams * args.steps
        fps = total_frames / elapsed
        out = {
            "metric": "fused depth frames/sec (640x480, hash TSDF)" if args.config == 2 else f"fused depth frames/sec (config {args.config})",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "repetitions": rep_stats,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if product else "synthetic; ALTERNATIVE BACKEND (--lib) -- control-flow check, not a measurement of the product",
            "config": {"workload": wl["name"], "streams": world * k_streams, "streams_per_gpu": k_streams,
                       "host_threads": (k_streams if (k_streams > 1 and args.host_threads and not exchange) else 1),
                       "world_size_seen": (dist.get_world_size() if dist.is_initialized() else 1), "collective_backend": backend_name,
                       "exchange": (f"all_gather of {17 + MAX_IDS}-word visible-block records, {max(1, args.exchange_batch)} frame(s) per collective, side stream, "
                                    + (("issued by the library over the STAND-IN transport of the shared-GPU self-test (not RCCL)" if standin else "issued by the library (RCCL from C++)") if native else "issued through torch.distributed")
                                    if exchange else "none"),
                       "exchange_cost_measured": exchange_cost,
                       "exchange_frames_per_collective": (max(1, args.exchange_batch) if exchange else None),
                       "exchange_table": exchange_table,
                       "host_cores": host_cores,
                       "per_rank_fps_min_max": fps_minmax,
                       "visible_blocks_last_frame": counters["noVisibleEntries"],
                       **({"acceleration_structures": streams[0].scene.accel_info()} if (product and wl["index"] == "hash") else {}),
                       "origin_offset_m": list(offset),
                       "frame_call": {"four": "the reference's four engine calls per frame: itm_allocate_scene_from_depth, itm_integrate_into_scene, itm_create_expected_depths, "
                                              "itm_create_icp_maps (recorded and launched as one fused frame by the library)",
                                      "process_frame": "itm_process_frame (NOT the headline call sequence)",
                                      "ahead": "itm_process_frame_ahead (NOT the headline: the next frame's pose is handed over before this frame is done)"}[mode["call"]],
                       "input": ("16-bit raw depth from pinned host memory: H2D + itm_update_view inside the timed region" if args.raw_depth
                                 else "float depth frames resident in HBM"),
                       "timing_note": "no event timers inside the timed region; kernels are timed in one extra, untimed repetition (roofline.launches_timed)",
                       **({"debug_keys": args.debug_keys} if args.debug_keys else {}),
                       "backend": be.version()},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity_check": parity_check,
            **({"other_configs": other_configs} if other_configs is not None else {}),
            **extra,
        }
    for ex in exs:                     # communicators of the library exchange go before the process group they were bootstrapped over
        if hasattr(ex, "close"):
            sync()
            ex.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        C.CDLL(None).fflush(None)
        sys.stdout.flush()
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
        failed = [("parity_check", parity_check)] if (parity_check is not None and not parity_check["equal"]) else []
        ex_par = (out.get("exploring") or {}).get("parity_check")
        if ex_par is not None and not ex_par["equal"]:
            failed.append(("exploring.parity_check", ex_par))
        for name, oc in (other_configs or {}).items():
            if (oc.get("parity_check") or {}).get("equal") is False:
                failed.append((name + ".parity_check", oc["parity_check"]))
        if failed:
            print(f"bench.py: PARITY CHECK FAILED: {failed}", file=sys.stderr)
            return 4
    return 0


# -------------------------------------------------------------------------------------------------------------
# secondary legs of the default single-GPU run
# -------------------------------------------------------------------------------------------------------------
EXPLORING_FRAMES = 200


def run_exploring(s0, be, capi, synth, torch, wl, issue, sync, timed_set):
    """BASELINE configs[1]'s scene under a camera that EXPLORES: frames 0..199 of the parity trajectory (camera k at x = 0.01 k m)
    from a reset scene, through the same four engine calls as the timed region.  Three passes over the same frames, each from a reset
    scene: (A) timed between two synchronisations -> frames/s; (B) an event pair around every launch of the request, visible-list (the
    allocation sweep runs inside it), integration and ray-cast kernels -> their mean durations; (C) counters read back after every frame
    -> blocks allocated per frame."""
    import numpy as np
    w, h, n = s0.w, s0.h, EXPLORING_FRAMES
    intr = s0.intr
    pos = [synth.parity_position(k) for k in range(n)]
    depth = torch.from_numpy(np.stack([synth.depth_frame(w, h, t, intr) for t in pos])).to(s0.device)
    views = [capi.View(depth[k].data_ptr(), w, h, M_d=synth.pose_matrix(pos[k]), intr_d=intr, rgb=None, w_rgb=w, h_rgb=h, intr_rgb=intr).struct() for k in range(n)]
    stream = s0.hip_stream.cuda_stream if s0.hip_stream is not None else None

    def reset():
        s0.scene.reco.ResetScene(stream=stream)

    def frames(per_frame=None):
        for k in range(n):
            rc = issue(s0, C.byref(views[k]), views[k], None)
            if rc:
                be.check(rc, "frame (exploring)")
            if per_frame:
                per_frame(k)
    # (A)
    reset(); sync()
    t0 = time.perf_counter()
    frames(); sync()
    dt = time.perf_counter() - t0
    end = s0.scene.counters(s0.rs)
    # (B)
    reset(); sync()
    s0.scene.profile_read(reset=True)
    s0.scene.profile_enable(sum(1 << TK[t] for t in timed_set))
    s0.scene.profile_sample(1)
    frames(); sync()
    s0.scene.profile_calibrate(64, stream); sync()
    prof = s0.scene.profile_read(reset=True)
    s0.scene.profile_enable(0)
    half_pair = 0.5 * prof["empty"]["total_ms"] / max(1, prof["empty"]["calls"])
    kernels = {t: round((prof[t]["total_ms"] / max(1, prof[t]["calls"]) - half_pair) * 1e3, 2) for t in timed_set}
    # (C)
    reset(); sync()
    free, vis = [wl["blocks"] - 1], []

    def after(k):
        c = s0.scene.counters(s0.rs)
        free.append(c["lastFreeBlockId"]); vis.append(c["noVisibleEntries"])
    frames(after)
    alloc = [a - b for a, b in zip(free[:-1], free[1:])]
    steady = alloc[20:]
    del depth
    return {"value": round(n / dt, 2), "unit": "frames/s", "steps": n, "ms_per_step": round(1e3 * dt / n, 4),
            "what": f"frames 0..{n - 1} of the parity trajectory (camera k at x = 0.01 k m: 1 cm per frame, never turning back) on a RESET scene, the reference's four engine "
                    "calls per frame, depth resident in HBM: every frame requests, allocates (sweep of ITMSceneReconstructionEngine_CPU.cpp:175-227 inside the visible-list launch) "
                    "and re-tests blocks, unlike the periodic headline trajectory",
            "blocks_allocated_per_frame": {"first_frame": alloc[0], "mean_frames_1_19": round(sum(alloc[1:20]) / 19.0, 1),
                                           "mean_frames_20_on": round(sum(steady) / max(1, len(steady)), 1), "min_frames_20_on": min(steady), "max_frames_20_on": max(steady)},
            "blocks_allocated_total": (wl["blocks"] - 1) - end["lastFreeBlockId"],
            "excess_entries_in_use_at_end": (0x20000 - 1) - end["lastFreeExcessListId"],
            "visible_blocks": {"mean": round(sum(vis) / len(vis), 1), "last": vis[-1]},
            "kernel_us": kernels,
            "kernel_note": "mean over the 200 launches of each, an event pair around every launch in a separate pass (half an empty pair subtracted); "
                           "visible_list includes the allocation sweep",
            "statusFlags": end.get("statusFlags", 0)}


def run_exploring_check(wl):
    """The exploring leg's own certificate: its first frames on a fresh product scene against the CPU oracle on the same frames (SHA-256
    of every buffer), and the oracle's rate on them as the CPU figure for an allocating workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import itm_testlib as T
    from infinitam_amd import capi
    nframes = 30
    sc = T.Scenario(name="exploring", w=wl["w"], h=wl["h"], voxelType=capi.VOXEL_S, indexType=capi.INDEX_HASH, voxelSize=wl["voxelSize"], mu=wl["mu"],
                    trajectory="parity", frames=nframes, localBlockNum=wl["blocks"])
    one, dig = _time_cpu(T.oracle_backend(), sc, nframes, T, digests=True)
    par = run_parity_check(sc, nframes, dig, T)
    return {"parity_check": par,
            "cpu_baseline": {"value": round(one, 3), "unit": "frames/s", "cores": 1, "kind": "port", "sample": f"first {nframes} frames of the exploring trajectory, oracle/libitm_oracle.so single thread"}}


def run_other_config(c, args):
    """`bench.py --config c` as a child process (this process's GPU work is done and synchronised; the child is started, never exec'd
    into): the same measurement the flag gives on its own -- timed region, kernel brackets, parity check against the oracle, CPU sample --
    reduced to the figures a reader of the driver's line needs."""
    cmd = [sys.executable, os.path.abspath(__file__), "--config", str(c), "--gpus", "1", "--steps", "100", "--warmup", "20", "--no-extra-legs", "--no-other-configs"]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ITM_BENCH_SPAWNED")}
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
        if not lines:
            return {"error": f"exit code {r.returncode}, no JSON line", "stderr_tail": r.stderr[-400:]}
        d = json.loads(lines[-1])
    except Exception as e:      # noqa: BLE001 -- a missing figure, reported as such
        return {"error": str(e)[:300]}
    rf = d.get("roofline") or {}
    cb = d.get("cpu_baseline") or {}
    out = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
           "repetitions": (d.get("repetitions") or {}).get("count"), "dtype": d["dtype"], "exit_code": r.returncode,
           "roofline": {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "algorithmic_bytes_per_launch",
                                               "avg_kernel_us", "launches_timed")},
           "other_kernels_us": {k: v.get("avg_kernel_us") for k, v in (rf.get("other_kernels") or {}).items()},
           "parity_check": d.get("parity_check"),
           "cpu_baseline": ({k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")} | ({"all_cores": cb["all_cores"]} if "all_cores" in cb else {})) if cb else None,
           "visible_blocks_last_frame": d["config"].get("visible_blocks_last_frame"),
           "child_wall_s": round(time.perf_counter() - t0, 1),
           "how": "python bench.py --config %d --steps 100 --warmup 20 --no-extra-legs, run as a child of this process after its own measurements" % c}
    return out


# -------------------------------------------------------------------------------------------------------------
# roofline of the dominant kernel
# -------------------------------------------------------------------------------------------------------------
def algorithmic_bytes(config, wl, counters):
    """Algorithmic bytes per launch (DESIGN.md 'Algorithmic bytes', SURVEY.md section 8d formulas).

    config 2, ray cast: 8*P/64 + V*(R_found + 8*R_t) + E*H + 16*P with the counts of the reference algorithm on this
      workload taken from the CPU oracle's work counters (tests/golden/algbytes_config2.json, frames 20..59):
      R_found = nearest reads that find a voxel, R_t = trilinear reads, H = 16-byte hash entries the reference's
      readVoxel dereferences (every cache-miss probe incl. chain links and the misses of empty-space steps).
    config 5, hash integrate: Nv*(512*V*2 + E + 4) + 4*P (+ 4*P rgb), Nv live from the device counters.
    config 3, dense integrate: 512^3*V read + 4*P (the reference reads every voxel's weight in every frame; the writes of updated
      voxels are NOT counted.  The kernel itself fetches far less -- the frustum cull decides before fetching, roofline.frac_traffic
      prices what the PMC counters saw)."""
    P = wl["w"] * wl["h"]
    E = 16
    if config == 2:
        with open(os.path.join(ROOT, "tests", "golden", "algbytes_config2.json")) as f:
            c = json.load(f)
        V = 4
        found_nearest = c["nearest_reads"] - c["nearest_misses"]
        return 8 * (P / 64) + V * (found_nearest + 8 * c["trilinear_reads"]) + E * c["hash_probes"] + 16 * P
    if config == 5:
        V = 12
        return counters["noVisibleEntries"] * (512 * V * 2 + E + 4) + 4 * P + 4 * P
    return 512 ** 3 * 4 + 4 * P


def algorithmic_bytes_secondary(config, wl, counters, which, n_entries):
    """Algorithmic bytes of the other large launches of a frame (SURVEY 8d formulas; work counts of the reference algorithm on this
    workload from the CPU oracle's counters, tests/golden/algbytes_config<N>.json):
    integrate (hash): Nv*(512*V*2 + E + 4) + 4*P (+ 4*P_rgb)  -- every visible block read and written once, its entry and list id, the depth image;
    visible_list: T + (E + 4)*Nv  -- one type byte per table slot (the reference's own sweep, _CPU.cpp:229-269, visits every slot), the
      entry of every re-tested block and the id written for it.  A latency-bound sweep: its fraction is tiny by construction;
    raycast: 8*P/64 + V*(R_found + 8*R_t) + E*H + 16*P  -- V is the WHOLE voxel as the reference's readVoxel fetches it (the kernels
      read only the sdf field: 2 of 4 bytes for ITMVoxel_s, 4 of 12 for ITMVoxel_f_rgb), H = 0 for a plain voxel array;
    request: 4*P + E*S + 9*A  -- depth image, S hash probes of buildHashAllocAndVisibleTypePP, A allocation requests."""
    P = wl["w"] * wl["h"]
    V = {"s": 4, "f_rgb": 12}[wl["voxel"]]
    nv = counters["noVisibleEntries"]
    if which == "integrate":
        return nv * (512 * V * 2 + 16 + 4) + 4 * P + (4 * P if wl["colour"] else 0)
    if which == "visible_list":
        return n_entries + (16 + 4) * nv
    with open(os.path.join(ROOT, "tests", "golden", f"algbytes_config{config}.json")) as f:
        c = json.load(f)
    if which == "raycast":
        return 8 * (P / 64) + V * ((c["nearest_reads"] - c["nearest_misses"]) + 8 * c["trilinear_reads"]) + 16 * c["hash_probes"] + 16 * P
    if which == "request":
        return 4 * P + 16 * c["alloc_probes"] + 9 * counters.get("noAllocRequests", 0)
    raise KeyError(which)


def read_roofline(config, wl, prof, counters, scene):
    """Average duration of the roofline kernel from HIP events around EVERY launch of it in one extra, untimed repetition (at least 64
    launches, steady state: no synchronisation in front of any of them), on the stream the kernel runs on.  `event_pair_us` is what an
    event pair adds to an interval (half of what 64 EMPTY brackets on the same stream measure, see below): the part of every bracketed
    interval that is not the kernel; `avg_kernel_us` = bracket - that, which is what rocprofv3 reports for the kernel
    (profiles/r5_*_kernel_stats.csv), `avg_bracket_us` the raw figure and `frac_raw_bracket` the fraction it would give.  `traffic` =
    HBM bytes per launch from the PMC counters of a separate rocprofv3 pass (profiles/traffic_r0N.json, stamped with the commit it was
    collected on).  `other_kernels`: every other launch that is a sizeable part of the frame, priced the same way."""
    r = prof[wl["kernel"]]
    if not r["calls"]:
        return None
    empty = prof.get("empty", {"calls": 0, "total_ms": 0.0})
    # An EMPTY bracket is two marker packets back to back: it measures two marker latencies.  A bracket around a kernel contains ONE of
    # them (the first marker's stamp is taken when it retires, the kernel starts right behind it; the second marker's own latency follows
    # the kernel).  Half the empty bracket is therefore what the pair adds -- measured: bracket 40.8 us, empty 4.9 us, rocprofv3 of the
    # same build 38.5 us (profiles/r4_config2_kernel_stats.csv) = bracket - 2.3.  The correction is bounded: never more than 10 % of
    # the bracket it is taken from (a calibration gone wrong must not buy a fraction).
    raw_s = r["total_ms"] * 1e-3 / r["calls"]
    pair_s = min(0.5 * ((empty["total_ms"] * 1e-3 / empty["calls"]) if empty["calls"] else 0.0), 0.1 * raw_s)
    avg_s = max(raw_s - pair_s, 1e-9)
    alg = algorithmic_bytes(config, wl, counters)
    achieved = alg / avg_s / 1e9
    traffic_all = {}
    for tname in ("traffic_r06.json", "traffic_r05.json", "traffic_r04.json", "traffic_r03.json", "traffic_r02.json"):        # the newest PMC collection that has this config
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath):
            with open(tpath) as f:
                t = json.load(f).get(f"config{config}")
            if t:
                traffic_all = t
                break
    traffic, traffic_src = traffic_all.get("hbm_bytes_per_launch"), traffic_all.get("source")
    kname = {2: "raycast_kernel<VoxelS,hash>", 3: "integrate_dense_strip_kernel", 5: "integrate_hash_kernel<VoxelFRgb>"}[config]
    out = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
           # the same fraction on the bytes the kernel really moved (PMC counters of a separate pass) instead of the reference algorithm's
           "frac_traffic": (round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
           "frac_raw_bracket": round(alg / raw_s / 1e9 / HBM_PEAK_GBS, 4),
           "algorithmic_bytes_per_launch": round(alg), "avg_kernel_us": round(avg_s * 1e6, 2), "avg_bracket_us": round(raw_s * 1e6, 2),
           "event_pair_us": round(pair_s * 1e6, 2), "launches_timed": r["calls"],
           "timed_in": "an extra untimed repetition, every launch bracketed (no sampling, no synchronisation between frames)"}
    other = {}
    n_entries = (scene.be.fn["buffer_bytes"](C.c_void_p(scene.h), None, 0) // 16) if wl["index"] == "hash" else 0
    for which in (("request", "visible_list", "integrate", "raycast") if wl["index"] == "hash" else ("raycast",)):
        q = prof[which]
        if which == wl["kernel"] or not q["calls"]:
            continue
        qraw = q["total_ms"] * 1e-3 / q["calls"]
        t = max(qraw - min(pair_s, 0.1 * qraw), 1e-9)
        ab = algorithmic_bytes_secondary(config, wl, counters, which, n_entries)
        # (bracket - half a pair: within ~2 us of rocprofv3, which profiles/ holds)
        other[which] = {"avg_kernel_us": round(t * 1e6, 2), "algorithmic_bytes_per_launch": int(ab), "achieved": round(ab / t / 1e9, 1),
                        "frac": round(ab / t / 1e9 / HBM_PEAK_GBS, 4), "launches_timed": q["calls"]}
        tr = (traffic_all.get("other_kernels") or {}).get(which)
        if tr:
            other[which]["traffic"] = tr.get("hbm_bytes_per_launch")
            other[which]["frac_traffic"] = round(tr["hbm_bytes_per_launch"] / t / 1e9 / HBM_PEAK_GBS, 4) if tr.get("hbm_bytes_per_launch") else None
    out["other_kernels"] = other
    return out


def measured_stream_peak():
    """The device-stream-copy peak of THIS box beside the 8.0 TB/s vendor figure: tools/microbench/stream_copy (built by
    __graft_entry__.build(), a child process) streams 2 GiB buffers; None when the binary is not there."""
    exe = os.path.join(ROOT, "tools", "microbench", "stream_copy")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "2048"], capture_output=True, timeout=120, text=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"read_GBs": d["read_GBs"], "copy_GBs": d["copy_GBs"], "fill_GBs": d["fill_GBs"], "unit": "GB/s",
                "what": "tools/microbench/stream_copy: best of 5 streaming launches over 2 GiB buffers (read-only / read+write / write-only)"}
    except Exception as e:      # noqa: BLE001 -- a missing figure, not a failed run
        return {"error": str(e)[:200]}


# -------------------------------------------------------------------------------------------------------------
# CPU baseline (checker code: the only place outside tests/ and smoke() that loads anything from oracle/)
# -------------------------------------------------------------------------------------------------------------
def _state_digests(ses, sc):
    """SHA-256 of everything a frame sequence leaves behind in a scene + render state: what the parity check compares."""
    import hashlib
    import numpy as np
    from infinitam_amd import capi
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()     # noqa: E731
    s, rs = ses.scene, ses.rs
    c = s.counters(rs)
    d = {"counters": {k: c[k] for k in ("lastFreeBlockId", "lastFreeExcessListId", "noVisibleEntries")}}
    if s.is_hash:
        d["hash_table"] = sha(s.download(capi.BUF_HASH_ENTRIES))
        d["excess_list"] = sha(s.download(capi.BUF_EXCESS_LIST))
        d["visible_ids"] = sha(s.download(capi.BUF_VISIBLE_IDS, rs)[: c["noVisibleEntries"]])
        d["visible_types"] = sha(s.download(capi.BUF_VISIBLE_TYPE, rs))
    d["allocation_list"] = sha(s.download(capi.BUF_ALLOCATION_LIST))
    d["voxels"] = sha(s.download(capi.BUF_VOXEL_BLOCKS))
    ray = s.download(capi.BUF_RAYCAST_RESULT, rs).copy()
    ray[ray[..., 3] <= 0, :3] = 0          # a ray that found nothing keeps an unspecified position in the reference; its w is compared
    d["raycast_result"] = sha(ray)
    d["icp_points"], d["icp_normals"] = sha(ses.points.numpy()), sha(ses.normals.numpy())
    d["raycast_image"] = sha(s.download(capi.BUF_RAYCAST_IMAGE, rs))
    return d


def _time_cpu(be, sc, nframes, T, digests=False):
    ses = T.Session(be, sc)
    depth = [be.to_backend(sc.depth(k)) for k in range(nframes)]
    rgb = ses.rgb
    t0 = time.perf_counter()
    for k in range(nframes):
        v = T.View(depth[k], sc.w, sc.h, M_d=sc.pose(k), intr_d=sc.intr(), rgb=rgb, w_rgb=sc.w, h_rgb=sc.h, intr_rgb=sc.intr())
        ses.scene.process_frame(v, ses.rs, ses.points, ses.normals)
    dt = time.perf_counter() - t0
    dig = _state_digests(ses, sc) if digests else None
    ses.close()
    return (nframes / dt, dig) if digests else nframes / dt


def run_parity_check(sc, nframes, oracle_digests, T):
    """Outside every timed region: the first `nframes` frames of the workload on a FRESH scene of the product, through the headline
    call sequence (the reference's four engine calls, recorded and launched as the fused frame), compared -- SHA-256 of every buffer --
    with the oracle scene the cpu_baseline leg has just built on those same frames.  Every bench run thereby certifies that the
    launches it timed do the reference's work; a mismatch makes the run fail."""
    hip = T.hip_backend()
    ses = T.Session(hip, sc)                 # (Session switches the recording on: itm_scene_set_deferred_fusion)
    for k in range(nframes):
        ses.frame(k, fused="four")
    got = _state_digests(ses, sc)
    status = ses.scene.counters(ses.rs)["statusFlags"]
    ses.close()
    differing = sorted(k for k in oracle_digests if got.get(k) != oracle_digests[k])
    return {"frames": nframes, "equal": (not differing and status == 0), "what": sorted(oracle_digests),
            "against": "oracle/libitm_oracle.so (bit-equal to the reference's CPU engines) on the same frames, same fresh scene",
            "through": "itm_allocate_scene_from_depth + itm_integrate_into_scene + itm_create_expected_depths + itm_create_icp_maps per frame",
            **({"differing": differing, "statusFlags": status} if (differing or status) else {})}


def run_cpu_baseline(config, wl, nframes, on_gpu=True):
    """The CPU oracle (a port of the reference CPU engines, bit-equal to them) on the first `nframes` frames of the same
    workload: single thread, and with OpenMP on all host cores over the loops the reference parallelises
    (ITMSceneReconstructionEngine_CPU.cpp:80,164,348; ITMVisualisationEngine_CPU.cpp:168,211,283).  Beside it, where
    the prebuilt oracle/_ref libraries travelled along, the reference's OWN CPU engines on the same frames (their voxel
    pool is the fork's compile-time 0x10000 blocks, which config 2 never exhausts; configs 3/5 need other template
    instantiations and are timed on the port only)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import itm_testlib as T
    from infinitam_amd import capi
    cores = os.cpu_count() or 1
    if not nframes:
        nframes = {2: 40, 3: 3, 5: 8}[config]
    vox = {"s": capi.VOXEL_S, "f_rgb": capi.VOXEL_F_RGB}[wl["voxel"]]
    idx = capi.INDEX_HASH if wl["index"] == "hash" else capi.INDEX_DENSE
    mk = lambda name, n, **kw: T.Scenario(name=name, w=wl["w"], h=wl["h"], voxelType=vox, indexType=idx, voxelSize=wl["voxelSize"], mu=wl["mu"],  # noqa: E731
                                          stopIntegratingAtMaxW=wl["stopAtMax"], colour=wl["colour"], trajectory="bench", frames=n, **kw)
    ob = T.oracle_backend()
    _time_cpu(ob, mk("warm", 1, localBlockNum=wl["blocks"]), 1, T)
    sc_cpu = mk("bench_cpu", nframes, localBlockNum=wl["blocks"])
    one, oracle_digests = _time_cpu(ob, sc_cpu, nframes, T, digests=True)
    out = {"value": round(one, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"first {nframes} frames of the same workload, oracle/libitm_oracle.so single thread"}
    parity = run_parity_check(sc_cpu, nframes, oracle_digests, T) if on_gpu else None
    omp = T.oracle_omp_backend()
    if omp is not None:
        n2 = nframes * 2
        _time_cpu(omp, mk("warm", 1, localBlockNum=wl["blocks"]), 1, T)
        out["all_cores"] = {"value": round(_time_cpu(omp, mk("bench_cpu_omp", n2, localBlockNum=wl["blocks"]), n2, T), 3), "unit": "frames/s",
                            "cores": cores, "kind": "port", "sample": f"first {n2} frames, oracle/libitm_oracle_omp.so (OpenMP, {cores} threads; timing only)"}
    if config == 2 and os.path.exists(T.REF_LIB):
        rb = T.Backend(T.REF_LIB, "itmr_")
        out["reference_engines"] = {"value": round(_time_cpu(rb, mk("bench_ref", nframes), nframes, T), 3), "unit": "frames/s", "cores": 1, "kind": "reference",
                                    "note": "ITMSceneReconstructionEngine_CPU / ITMVisualisationEngine_CPU compiled from the reference sources (oracle/_ref), same frames"}
        if os.path.exists(T.REF_OMP_LIB):
            ro = T.Backend(T.REF_OMP_LIB, "itmr_")
            n2 = nframes * 2
            out["reference_engines_all_cores"] = {"value": round(_time_cpu(ro, mk("bench_ref_omp", n2), n2, T), 3), "unit": "frames/s", "cores": cores,
                                                  "kind": "reference", "note": "the same engines built with -fopenmp -DWITH_OPENMP (timing only: the reference's OpenMP allocation loop is racy)"}
    return out, parity


def main() -> int:
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
