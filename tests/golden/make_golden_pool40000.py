#!/usr/bin/env python3
"""Golden vectors for BASELINE configs[1] (640x480, 4 mm, ITMVoxel_s) and configs[4] (1280x960, 2 mm, ITMVoxel_f_rgb) at
their real pool size of 0x40000 voxel blocks, produced by the REFERENCE's own CPU engines.

The fork's header fixes SDF_LOCAL_BLOCK_NUM at 0x10000 at compile time, so the reference is rebuilt for this purpose
with -DSDF_LOCAL_BLOCK_NUM=0x40000 (oracle/Makefile target `ref40000`: SURVEY.md Appendix B recipe on a scratch copy
outside the repository that is deleted again; container only).  Only data is stored here: digests, counters, the
occupied hash entries, the visible list and stride-4 samples of the maps.

Also here (GOLDEN_LONG): the headline workload after 200 frames of the bench trajectory -- the saturated steady state bench.py times.

Run in the development container:  python tests/golden/make_golden_pool40000.py [scenario names]
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import itm_testlib as T  # noqa: E402
from golden_scenarios import GOLDEN_LONG, GOLDEN_POOL40000, summarise  # noqa: E402


def main():
    ref = T.reference_pool40000_backend()
    if ref is None:
        raise SystemExit("reference build not available (needs /root/reference)")
    only = sys.argv[1:]
    for sc, full in GOLDEN_POOL40000 + GOLDEN_LONG:
        if only and sc.name not in only:
            continue
        depth_sha = []

        def hook(k, ses):
            if k < 100:          # the bench trajectory repeats after 100 frames: one period pins the generator
                depth_sha.append(hashlib.sha256(np.ascontiguousarray(sc.depth(k)).tobytes()).hexdigest())

        res = T.run_scenario(ref, sc, per_frame_hook=hook)
        arrays, meta = summarise(res, sc, full)
        meta["depth_sha256"] = depth_sha
        meta["counters"] = [{k: c[k] for k in ("lastFreeBlockId", "lastFreeExcessListId", "noVisibleEntries")} for c in res.counters]
        meta["blocks_allocated"] = int((res.hash["ptr"] >= 0).sum())
        meta["excess_entries_used"] = int((res.hash["ptr"][0x100000:] >= 0).sum())
        meta["voxels_at_maxW"] = int((res.voxels["w_depth"] == sc.maxW).sum())
        if "w_color" in res.voxels.dtype.names:
            meta["voxels_coloured"] = int((res.voxels["w_color"] > 0).sum())
        meta["generator"] = "reference CPU engines compiled with -DSDF_LOCAL_BLOCK_NUM=0x40000 (oracle/_ref/libitm_ref_pool40000.so, " + ref.version() + ")"
        np.savez_compressed(os.path.join(T.GOLDEN_DIR, sc.name + ".npz"), **arrays)
        with open(os.path.join(T.GOLDEN_DIR, sc.name + ".json"), "w") as f:
            json.dump(meta, f, indent=1)
        print(sc.name, meta["counters"], {k: meta[k] for k in ("blocks_allocated", "excess_entries_used", "voxels_touched")}, meta.get("voxels_coloured"))


if __name__ == "__main__":
    main()
