#!/usr/bin/env python3
"""tests/golden/g_main_engine.json: the sequences of tests/test_main_engine.py run on the REFERENCE's objects (its view builder,
ITMTrackingState::TrackerFarFromPointCloud, CPU engines and ITMDepthTracker_CPU behind oracle/_ref/libitm_ref.so; the few statements
of ITMMainEngine::ProcessFrame / ITMTrackingController / ITMDenseMapper around them are restated in oracle/ref_driver.cpp because their
translation units need glog).  Run in the development container:  python tests/golden/make_golden_main_engine.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import itm_testlib as T  # noqa: E402
import test_main_engine as tm  # noqa: E402


def main():
    ref = T.reference_backend()
    if ref is None:
        raise SystemExit("reference build not available (needs /root/reference)")
    out = {kind: tm.run_reference(ref, kind) for kind in tm.TRACKERS}
    with open(tm.GOLDEN, "w") as f:
        json.dump(out, f, indent=0)
    for kind, rows in out.items():
        print(kind, [(r["age"], r["full"]) for r in rows])


if __name__ == "__main__":
    main()
