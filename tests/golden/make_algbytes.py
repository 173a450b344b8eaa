#!/usr/bin/env python3
"""Work counters of the bench workloads (bench.py --config 2 / 3 / 5 = BASELINE configs[1] / [2] / [4]) from the CPU oracle's
instrumentation (oracle/itm_oracle.cpp, Stats) -> tests/golden/algbytes_config{2,3,5}.json.  bench.py prices the
algorithmic bytes of the dominant kernel from these counts ('Algorithmic bytes').
Run here (CPU only): python tests/golden/make_algbytes.py [2|3|5 ...]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import itm_testlib as T

NAMES = ("rays ray_hits ray_steps max_ray_steps nearest_reads nearest_misses trilinear_reads voxel_reads hash_probes "
         "alloc_pixels alloc_steps alloc_probes fuse_blocks fuse_voxels_visited fuse_voxels_updated").split()
from infinitam_amd import capi
# (warm-up frames, counted frames): the oracle runs config 3 at ~1 frame/s and config 5 at ~1.5, so those average fewer frames
CONFIGS = {
    2: dict(warm=20, n=40, kw=dict(voxelSize=0.004, localBlockNum=0x40000),
            workload="640x480, hash, ITMVoxel_s, 4 mm, mu 0.02, pool 0x40000, bench trajectory stream 0"),
    3: dict(warm=4, n=8, kw=dict(voxelSize=0.004, indexType=capi.INDEX_DENSE, stopIntegratingAtMaxW=True),
            workload="640x480, dense 512^3, ITMVoxel_s, 4 mm, mu 0.02, stopIntegratingAtMaxW, bench trajectory stream 0"),
    5: dict(warm=4, n=8, kw=dict(w=1280, h=960, voxelSize=0.002, localBlockNum=0x40000, voxelType=capi.VOXEL_F_RGB, colour=True),
            workload="1280x960, hash, ITMVoxel_f_rgb, 2 mm, mu 0.02, pool 0x40000, bench trajectory stream 0 (every 4th pose, as bench.py keeps 25 frames resident)"),
}

def main(config):
    WARM, N = CONFIGS[config]["warm"], CONFIGS[config]["n"]
    ob = T.oracle_backend()
    def stats(clear=True):
        buf = (ctypes.c_longlong * 15)(); ob.lib.itmo_debug_stats(buf, int(clear)); return dict(zip(NAMES, list(buf)))
    stride = 4 if config == 5 else 1          # bench.py --config 5 replays every 4th pose of the trajectory
    sc = T.Scenario(name="bench", trajectory="bench", frames=(WARM + N) * stride, **CONFIGS[config]["kw"])
    ses = T.Session(ob, sc)
    acc = {k: 0 for k in NAMES}; nv = 0; maxsteps = 0
    for k in range(WARM + N):
        stats()
        ses.frame(k * stride, fused=True)
        st = stats()
        if k >= WARM:
            for n in NAMES: acc[n] += st[n]
            maxsteps = max(maxsteps, st["max_ray_steps"])
            nv += ses.scene.counters(ses.rs)["noVisibleEntries"]
    out = {n: acc[n] / N for n in NAMES}
    out["max_ray_steps"] = maxsteps
    out["visible_blocks"] = nv / N
    out["frames"] = [WARM, WARM + N]
    out["workload"] = CONFIGS[config]["workload"]
    with open(os.path.join(ROOT, "tests", "golden", f"algbytes_config{config}.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    for c in ([int(a) for a in sys.argv[1:]] or [2, 3, 5]):
        main(c)
