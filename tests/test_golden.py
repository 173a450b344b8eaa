"""Golden vectors generated from the reference's CPU engines (tests/golden/make_golden.py):
 - not gpu: pins the CPU oracle on machines without /root/reference;
 - gpu: the HIP path against the same reference vectors.  All comparisons are bit-exact."""
import pytest

import itm_testlib as T
from golden_scenarios import GOLDEN_SCENARIOS, check_against_golden


@pytest.mark.parametrize("sc,full", GOLDEN_SCENARIOS, ids=lambda v: getattr(v, "name", str(v)))
def test_oracle_reproduces_reference_goldens(oracle, sc, full):
    check_against_golden(T.run_scenario(oracle, sc), sc, full)


@pytest.mark.gpu
@pytest.mark.parametrize("sc,full", GOLDEN_SCENARIOS, ids=lambda v: getattr(v, "name", str(v)))
def test_hip_reproduces_reference_goldens(hip, sc, full):
    check_against_golden(T.run_scenario(hip, sc), sc, full)


@pytest.mark.gpu
@pytest.mark.parametrize("sc,full", GOLDEN_SCENARIOS[:2] + GOLDEN_SCENARIOS[-1:], ids=lambda v: getattr(v, "name", str(v)))
def test_hip_fused_frame_reproduces_reference_goldens(hip, sc, full):
    check_against_golden(T.run_scenario(hip, sc, fused=True), sc, full)
