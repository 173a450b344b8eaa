"""Pins the oracle: oracle/itm_oracle.cpp against the reference's own CPU engines
(oracle/_ref/libitm_ref.so, compiled from /root/reference by oracle/Makefile).  Bit-exact.
Skipped on machines without the reference build (the committed goldens cover those)."""
import numpy as np
import pytest

import itm_testlib as T
from itm_testlib import Scenario

SCENARIOS = [
    Scenario(name="micro_hash_s", w=160, h=120, voxelSize=0.01, frames=3),
    Scenario(name="hash_s_5mm", frames=2),
    Scenario(name="hash_s_yaw", voxelSize=0.005, frames=2, trajectory="yaw", w=320, h=240),
    Scenario(name="hash_f", voxelType=T.VOXEL_F, frames=2, w=320, h=240),
    Scenario(name="hash_s_rgb", voxelType=T.VOXEL_S_RGB, frames=2, colour=True, w=320, h=240),
    Scenario(name="hash_f_rgb", voxelType=T.VOXEL_F_RGB, frames=2, colour=True, w=320, h=240),
    Scenario(name="dense_s_64", indexType=T.INDEX_DENSE, denseSize=(64, 64, 64), denseOffset=(-32, -32, 118),
             voxelSize=0.01, frames=2, w=160, h=120),
    Scenario(name="dense_f_rgb_64", indexType=T.INDEX_DENSE, voxelType=T.VOXEL_F_RGB, denseSize=(64, 64, 64),
             denseOffset=(-32, -32, 118), voxelSize=0.01, frames=2, w=160, h=120, colour=True),
]


@pytest.mark.parametrize("sc", SCENARIOS, ids=lambda s: s.name)
def test_oracle_matches_reference(oracle, reference, sc):
    a = T.run_scenario(oracle, sc)
    b = T.run_scenario(reference, sc)
    T.compare_results(a, b, sc)
