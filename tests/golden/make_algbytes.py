#!/usr/bin/env python3
"""Work counters of the bench workload (BASELINE configs[1]) from the CPU oracle's instrumentation
(oracle/itm_oracle.cpp, Stats) -> tests/golden/algbytes_config2.json.  bench.py prices the
algorithmic bytes of the dominant kernel from these counts ('Algorithmic bytes').
Run here (CPU only): python tests/golden/make_algbytes.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import itm_testlib as T

NAMES = ("rays ray_hits ray_steps max_ray_steps nearest_reads nearest_misses trilinear_reads voxel_reads hash_probes "
         "alloc_pixels alloc_steps alloc_probes fuse_blocks fuse_voxels_visited fuse_voxels_updated").split()
WARM, N = 20, 40

def main():
    ob = T.oracle_backend()
    def stats(clear=True):
        buf = (ctypes.c_longlong * 15)(); ob.lib.itmo_debug_stats(buf, int(clear)); return dict(zip(NAMES, list(buf)))
    sc = T.Scenario(name="bench", voxelSize=0.004, localBlockNum=0x40000, trajectory="bench", frames=WARM + N)
    ses = T.Session(ob, sc)
    acc = {k: 0 for k in NAMES}; nv = 0; maxsteps = 0
    for k in range(sc.frames):
        stats()
        ses.frame(k, fused=True)
        st = stats()
        if k >= WARM:
            for n in NAMES: acc[n] += st[n]
            maxsteps = max(maxsteps, st["max_ray_steps"])
            nv += ses.scene.counters(ses.rs)["noVisibleEntries"]
    out = {n: acc[n] / N for n in NAMES}
    out["max_ray_steps"] = maxsteps
    out["visible_blocks"] = nv / N
    out["frames"] = [WARM, WARM + N]
    out["workload"] = "640x480, hash, ITMVoxel_s, 4 mm, mu 0.02, pool 0x40000, bench trajectory stream 0"
    with open(os.path.join(ROOT, "tests", "golden", "algbytes_config2.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
