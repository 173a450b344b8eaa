"""Conditions after which a scene is not what the reference would hold are FATAL, and loud.

The reference never drops a frame (DeviceSpecific/CPU/ITMSceneReconstructionEngine_CPU.cpp:229-290 always builds its list, :47-114
always fuses through it).  The product's one-launch visible list hands counts from workgroup to workgroup; it is only taken on a device
that holds all its workgroups at once (asked of the runtime: alloc.hip, one_pass_list_is_safe) and its waits are bounded all the same.  A
wait that does expire -- or a depth pixel with more ray steps than the allocation key can number -- raises itm_counters::statusFlags in device memory
AND in a word of page-locked host memory every entry point reads: from then on every call that names the scene fails with
ITM_ERR_DEVICE until ResetScene, instead of silently continuing with a scene the reference can never be in."""
import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import capi

pytestmark = pytest.mark.gpu


def test_an_expired_wait_in_the_visible_list_launch_is_fatal_until_reset(hip, oracle):
    sc = T.Scenario(name="stuck", voxelSize=0.005, frames=3, trajectory="bench")
    ses = T.Session(hip, sc)
    ses.frame(0, fused=True)
    assert ses.scene.counters(ses.rs)["statusFlags"] == 0
    hip.check(hip.fn["debug_set"](20, 7 + 1), "debug_set")                 # chunk 7 gives up
    try:
        ses.frame(1, fused="four")                                         # enqueued; the condition is raised when the launch runs
        with pytest.raises(capi.ItmError, match="fatal status"):
            ses.scene.counters(ses.rs)
    finally:
        hip.check(hip.fn["debug_set"](20, 0), "debug_set")
    c = capi.Counters()                                                     # the counters are still filled in: they say what happened
    rc = hip.fn["get_counters"](capi._P(ses.scene.h), capi._P(ses.rs.h), capi.C.byref(c), None)
    assert rc == capi.ERR_DEVICE and (c.statusFlags & 2)
    for call in (lambda: ses.frame(2, fused=True), lambda: ses.frame(2, fused="four"), lambda: ses.scene.download(capi.BUF_HASH_ENTRIES),
                 lambda: ses.scene.vis.FindSurface(sc.pose(0), sc.intr(), ses.rs)):
        with pytest.raises(capi.ItmError, match="fatal status"):
            call()
    ses.scene.reco.ResetScene()                                            # clears the condition: the scene is usable again and exact
    for k in range(sc.frames):
        ses.frame(k, fused="four")
    a = ses.snapshot(); a.counters = [ses.scene.counters(ses.rs)]
    ref = T.Session(oracle, sc)
    for k in range(sc.frames):
        ref.frame(k)
    b = ref.snapshot(); b.counters = [ref.scene.counters(ref.rs)]
    T.compare_results(a, b, sc, what="after the fatal status was reset")
    ses.close(); ref.close()


def test_more_ray_steps_than_the_key_can_number_is_fatal(hip):
    """1280 x 960 leaves 10 bits for the step number; a band of 2 mu = 4 cm in blocks of 40 um is ~2 000 steps."""
    sc = T.Scenario(name="key_overflow", w=1280, h=960, voxelSize=0.000005, frames=1, localBlockNum=0x1000)
    ses = T.Session(hip, sc)
    v = ses.view(0)
    ses.scene.reco.AllocateSceneFromDepth(v, ses.rs)                        # (the allocation alone: rays through 5 um voxels would march for minutes)
    with pytest.raises(capi.ItmError, match="ray steps"):
        ses.scene.counters(ses.rs)
    with pytest.raises(capi.ItmError, match="fatal status"):
        ses.scene.reco.AllocateSceneFromDepth(v, ses.rs)
    ses.scene.reco.ResetScene()
    assert ses.scene.counters(ses.rs)["statusFlags"] == 0
    ses.close()


@pytest.mark.parametrize("sc", [T.Scenario(name="early_sweeps_tiny_table", frames=5, bucketNum=0x1000, excessNum=0x1000, w=320, h=240, voxelSize=0.01, trajectory="yaw"),
                                T.Scenario(name="early_sweeps_chains_of_chains", frames=4, bucketNum=0x800, excessNum=0x4000, w=320, h=240, voxelSize=0.008, trajectory="yaw")],
                         ids=lambda s: s.name)
def test_excess_region_chunks_wait_for_each_others_sweeps(hip, oracle, sc):
    """Tables so small that most requests hang off excess entries (chain tails IN the excess region): the workgroups of the excess region
    wait for the stamps of each other's sweeps -- the one wait of the launch that is not a look-back."""
    a = T.run_scenario(hip, sc, fused="four")
    b = T.run_scenario(oracle, sc)
    T.compare_results(a, b, sc, what=sc.name)
    assert a.counters[-1]["statusFlags"] == 0
    assert a.counters[-1]["lastFreeExcessListId"] < sc.excessNum - 1 - 500, "the scenario makes no excess allocations to speak of"
