"""BASELINE configs[1] and configs[4] at their real pool size (0x40000 voxel blocks) against vectors produced by the
REFERENCE's CPU engines compiled with SDF_LOCAL_BLOCK_NUM=0x40000 (tests/golden/make_golden_pool40000.py), plus the
reference-produced known answers SURVEY.md section 8d records for exactly these configurations.
 - not gpu: pins the CPU oracle's generalisation to pool sizes other than the fork's 0x10000;
 - gpu: the HIP path against the same reference vectors.  All comparisons are bit-exact."""
import json
import os

import numpy as np
import pytest

import itm_testlib as T
from golden_scenarios import GOLDEN_POOL40000, check_against_golden

# config 2 on the parity trajectory, frame 2; config 5 frames 0/1/2 and totals after 3 frames
SURVEY_CFG2 = {"Nv_frame2": 9085}
SURVEY_CFG5 = {"Nv": [41755, 50337, 50863], "blocks_allocated": 51238, "excess_entries_used": 8639}


def _check(be, sc, full, fused):
    res = T.run_scenario(be, sc, fused=fused)
    check_against_golden(res, sc, full)
    with open(os.path.join(T.GOLDEN_DIR, sc.name + ".json")) as f:
        meta = json.load(f)
    assert int((res.hash["ptr"] >= 0).sum()) == meta["blocks_allocated"]
    assert int((res.hash["ptr"][0x100000:] >= 0).sum()) == meta["excess_entries_used"]
    if "voxels_coloured" in meta:
        assert int((res.voxels["w_color"] > 0).sum()) == meta["voxels_coloured"]
    if sc.name == "g_cfg2_pool40000":
        assert res.counters[2]["noVisibleEntries"] == SURVEY_CFG2["Nv_frame2"]
    if sc.name == "g_cfg5_pool40000":
        assert [c["noVisibleEntries"] for c in res.counters] == SURVEY_CFG5["Nv"]
        assert meta["blocks_allocated"] == SURVEY_CFG5["blocks_allocated"] and meta["excess_entries_used"] == SURVEY_CFG5["excess_entries_used"]
        assert 4.7e6 < meta["voxels_coloured"] < 4.85e6      # "4.77 M voxels coloured"


@pytest.mark.parametrize("sc,full", GOLDEN_POOL40000, ids=lambda v: getattr(v, "name", str(v)))
def test_oracle_reproduces_reference_at_pool_40000(oracle, sc, full):
    _check(oracle, sc, full, fused=False)


def test_config2_frame2_updates_the_surveys_voxel_count(oracle):
    """U = 3 533 546 voxels updated by IntegrateIntoScene on frame 2 of config 2 (oracle work counter)."""
    import ctypes
    sc = GOLDEN_POOL40000[0][0]
    ses = T.Session(oracle, sc)
    buf = (ctypes.c_longlong * 15)()
    for k in range(3):
        oracle.lib.itmo_debug_stats(buf, 1)
        ses.frame(k)
    oracle.lib.itmo_debug_stats(buf, 1)
    assert buf[14] == 3533546 and ses.scene.counters(ses.rs)["noVisibleEntries"] == 9085
    ses.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("sc,full", GOLDEN_POOL40000, ids=lambda v: getattr(v, "name", str(v)))
def test_hip_reproduces_reference_at_pool_40000(hip, sc, full, fused):
    _check(hip, sc, full, fused)


def test_reference_build_still_matches_its_goldens():
    """Where the reference is present, re-run it (guards the generator and the committed files against drift)."""
    ref = T.reference_pool40000_backend()
    if ref is None:
        pytest.skip("reference build (oracle/_ref/libitm_ref_pool40000.so) not available on this machine")
    sc, full = GOLDEN_POOL40000[1]
    check_against_golden(T.run_scenario(ref, sc), sc, full)
