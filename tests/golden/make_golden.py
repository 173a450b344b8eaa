#!/usr/bin/env python3
"""Generates the golden vectors in tests/golden/ from the REFERENCE's own CPU engines
(oracle/_ref/libitm_ref.so, built by `make -C oracle ref` from /root/reference).  The reference
ships no tests or fixtures, so these files are what pins parity on machines where
the reference is absent (the GPU box).  Only data is stored: inputs are regenerated from
infinitam_amd.synth (their SHA-256 is stored), outputs are the reference's.

Run in the development container:  python tests/golden/make_golden.py
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
from golden_scenarios import GOLDEN_SCENARIOS, summarise  # noqa: E402


def main():
    ref = T.reference_backend()
    if ref is None:
        raise SystemExit("reference build not available (needs /root/reference)")
    for sc, full in GOLDEN_SCENARIOS:
        depth_sha = []
        snaps = {}

        def hook(k, ses):
            depth_sha.append(hashlib.sha256(np.ascontiguousarray(sc.depth(k)).tobytes()).hexdigest())
            if k in (0, sc.frames - 1):
                snaps[k] = ses.snapshot()
                snaps[k].counters = [ses.scene.counters(ses.rs)]

        res = T.run_scenario(ref, sc, per_frame_hook=hook)
        arrays, meta = summarise(res, sc, full)
        meta["depth_sha256"] = depth_sha
        meta["counters"] = [{k: c[k] for k in ("lastFreeBlockId", "lastFreeExcessListId", "noVisibleEntries")} for c in res.counters]
        meta["generator"] = "reference CPU engines via oracle/_ref/libitm_ref.so (" + ref.version() + ")"
        np.savez_compressed(os.path.join(T.GOLDEN_DIR, sc.name + ".npz"), **arrays)
        with open(os.path.join(T.GOLDEN_DIR, sc.name + ".json"), "w") as f:
            json.dump(meta, f, indent=1)
        print(sc.name, {k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    main()
