#!/usr/bin/env python3
"""More seeds of tests/test_random_stress.py than the suite runs (HIP vs oracle, bit-exact): usage: python tests/stress_sweep.py [first=1000] [count=100]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import itm_testlib as T  # noqa: E402
import test_random_stress as S  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
hip, oracle = T.hip_backend(), T.oracle_backend()
bad = []
for seed in range(first, first + count):
    c = S.make_case(seed)
    try:
        S.assert_same(S.run_case(hip, c), S.run_case(oracle, c), f"seed{seed}")
    except AssertionError as e:
        bad.append(seed)
        print("seed", seed, "FAIL", str(e)[:300], flush=True)
print(f"{count} seeds from {first}: {len(bad)} failures {bad}")
sys.exit(1 if bad else 0)
