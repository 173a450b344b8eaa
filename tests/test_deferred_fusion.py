"""The reference's four per-frame engine calls behind the fused frame (infinitam_amd/csrc/pending.hip).

ITMMainEngine::ProcessFrame reaches the engines as AllocateSceneFromDepth -> IntegrateIntoScene (Engine/ITMDenseMapper.cpp:50-57) ->
CreateExpectedDepths -> CreateICPMaps (Engine/ITMTrackingController.cpp:30-46).  The product records the first three and launches the
fused frame at the fourth; whatever else happens in between launches what was recorded, call by call.  Every test compares with the
oracle, which executes each call when it is made: the recording must never be observable."""
import numpy as np
import pytest

import itm_testlib as T
from itm_testlib import Scenario
from infinitam_amd import capi, synth

pytestmark = pytest.mark.gpu


def both(hip, oracle, sc, drive, what):
    """Runs `drive(session, k)` for every frame on both backends and compares everything after the last frame."""
    out = []
    for be in (hip, oracle):
        ses = T.Session(be, sc)
        counters = []
        for k in range(sc.frames):
            drive(ses, k)
            counters.append(ses.scene.counters(ses.rs))
        r = ses.snapshot()
        r.counters = counters
        out.append(r)
        ses.close()
    T.compare_results(out[0], out[1], sc, what=what)
    return out


@pytest.mark.parametrize("sc", [Scenario(name="four_s", voxelSize=0.005, frames=4, trajectory="bench"),
                                Scenario(name="four_f_rgb", voxelType=capi.VOXEL_F_RGB, colour=True, w=320, h=240, voxelSize=0.008, frames=3),
                                Scenario(name="four_tiny_table", frames=4, bucketNum=0x1000, excessNum=0x1000, w=320, h=240, voxelSize=0.01, trajectory="yaw"),
                                Scenario(name="four_dense", indexType=capi.INDEX_DENSE, denseSize=(128, 128, 128), denseOffset=(-64, -64, 100), voxelSize=0.01, w=320, h=240, frames=3)],
                         ids=lambda s: s.name)
def test_four_calls_back_to_back_equal_the_oracle(hip, oracle, sc):
    both(hip, oracle, sc, lambda ses, k: ses.frame(k, fused="four"), sc.name + "/four calls")


def test_four_calls_take_the_fused_launches(hip):
    """The sequence really is fused: with per-kernel timers on, four calls back to back time NO separate range launches
    (CreateExpectedDepths rides in the integration and the ray cast), the same calls with a flush in between time them."""
    sc = Scenario(name="launches", voxelSize=0.005, frames=3, trajectory="bench")
    for mode, separate in (("four", False), (False, True)):
        ses = T.Session(hip, sc)
        ses.scene.profile_enable(0x7f)
        for k in range(sc.frames):
            ses.frame(k, fused=mode)
        prof = ses.scene.profile_read()
        ses.close()
        assert prof["raycast"]["calls"] == sc.frames and prof["integrate"]["calls"] == sc.frames, prof
        assert (prof["range"]["calls"] > 0) == separate, (mode, prof)


def test_a_scene_that_never_asked_launches_every_call_at_once(hip, oracle):
    """The default (no itm_scene_set_deferred_fusion): every call has enqueued its kernels when it returns -- the stream order a host
    with its own events and kernels relies on -- so four calls back to back time the SEPARATE range launches; results are the same."""
    sc = Scenario(name="not_asked", voxelSize=0.005, frames=3, trajectory="bench")
    ses = T.Session(hip, sc, deferred_fusion=False)
    ses.scene.profile_enable(0x7f)
    for k in range(sc.frames):
        ses.frame(k, fused="four")
    prof = ses.scene.profile_read()
    a = ses.snapshot()
    ses.close()
    assert prof["range"]["calls"] > 0 and prof["raycast"]["calls"] == sc.frames, prof
    b = T.run_scenario(oracle, sc)
    b.counters = b.counters[-1:]
    T.compare_results(a, b, sc, what="never asked for the recording")


def test_scene_destroyed_before_its_render_state(hip):
    """Handles are destroyed in any order (garbage-collected hosts): a render state that outlives its scene -- with calls recorded on
    it, and with requests issued ahead -- is destroyed without touching the scene's memory."""
    sc = Scenario(name="order", voxelSize=0.01, w=160, h=120, frames=2)
    for ahead in (False, True):
        ses = T.Session(hip, sc)
        v = ses.view(0)
        if ahead:
            ses.scene.process_frame_ahead(v, ses.view(1), ses.rs, ses.points, ses.normals)
        else:
            ses.scene.reco.AllocateSceneFromDepth(v, ses.rs)          # recorded, never launched
        ses.scene.close()
        ses.rs.close()
        # the library is still healthy
        b = T.Session(hip, sc)
        b.frame(0, fused="four")
        assert b.scene.counters(b.rs)["noVisibleEntries"] > 0
        b.close()


def test_observers_between_the_calls_see_what_the_reference_would(hip, oracle):
    """A read between two of the calls launches what was recorded: the table after the allocation alone, the voxels after the
    integration alone, the range image after CreateExpectedDepths alone -- all equal to the oracle's at that point."""
    sc = Scenario(name="observed", voxelSize=0.005, frames=3, trajectory="bench")
    seen = {}

    def drive(ses, k):
        s, rs, v = ses.scene, ses.rs, ses.view(k)
        log = seen.setdefault(id(ses.be), [])
        s.reco.AllocateSceneFromDepth(v, rs)
        log.append(("hash", s.download(capi.BUF_HASH_ENTRIES), s.counters(rs)))
        s.reco.IntegrateIntoScene(v, rs)
        log.append(("voxels", s.download(capi.BUF_VOXEL_BLOCKS)))
        s.vis.CreateExpectedDepths(v.M_d, v.intr_d, rs)
        log.append(("range", T.range_region(s.download(capi.BUF_RANGE_IMAGE, rs), sc.w, sc.h).copy()))
        s.vis.CreateICPMaps(v, rs, ses.points, ses.normals)

    both(hip, oracle, sc, drive, "observed")
    a, b = seen[id(hip)], seen[id(oracle)]
    assert len(a) == len(b) == 3 * sc.frames
    for i, (x, y) in enumerate(zip(a, b)):
        if x[0] == "hash":
            T.assert_fields_equal(x[1], y[1], "frame %d: table after the allocation" % (i // 3))
            assert x[2]["lastFreeBlockId"] == y[2]["lastFreeBlockId"] and x[2]["noVisibleEntries"] == y[2]["noVisibleEntries"]
        elif x[0] == "voxels":
            T.assert_fields_equal(x[1], y[1], "frame %d: voxels after the integration" % (i // 3))
        else:
            assert np.array_equal(x[1], y[1]), "frame %d: range image after CreateExpectedDepths" % (i // 3)


def test_partial_sequences_and_other_calls_in_between(hip, oracle):
    """Frame by frame a different way of NOT completing the sequence: a counter read after the integration, a free-view render from
    another pose through a second render state, FindSurface instead of CreateICPMaps, expected depths for another pose, the
    integration left out, itm_process_frame right behind a recorded allocation of the same view."""
    sc = Scenario(name="partial", voxelSize=0.005, frames=7, trajectory="bench")
    extra = {}

    def drive(ses, k):
        s, rs, v = ses.scene, ses.rs, ses.view(k)
        log = extra.setdefault(id(ses.be), [])
        if k == 0:
            ses.frame(k, fused="four")
        elif k == 1:
            s.reco.AllocateSceneFromDepth(v, rs); s.reco.IntegrateIntoScene(v, rs)
            log.append(s.counters(rs)["noVisibleEntries"])
            s.vis.CreateExpectedDepths(v.M_d, v.intr_d, rs); s.vis.CreateICPMaps(v, rs, ses.points, ses.normals)
        elif k == 2:
            free = s.vis.CreateRenderState((sc.w, sc.h))
            M = synth.pose_matrix_yaw((0.05, 0.0, 0.0), 0.1)
            s.reco.AllocateSceneFromDepth(v, rs); s.reco.IntegrateIntoScene(v, rs); s.vis.CreateExpectedDepths(v.M_d, v.intr_d, rs)
            s.vis.FindVisibleBlocks(M, v.intr_d, free); s.vis.CreateExpectedDepths(M, v.intr_d, free)
            s.vis.RenderImage(M, v.intr_d, free, None, capi.RENDER_SHADED_GREYSCALE)
            log.append(s.download(capi.BUF_RAYCAST_IMAGE, free).copy())
            s.vis.CreateICPMaps(v, rs, ses.points, ses.normals)
            free.close()
        elif k == 3:
            s.reco.AllocateSceneFromDepth(v, rs); s.reco.IntegrateIntoScene(v, rs); s.vis.CreateExpectedDepths(v.M_d, v.intr_d, rs)
            s.vis.FindSurface(v.M_d, v.intr_d, rs)
            log.append(s.download(capi.BUF_RAYCAST_RESULT, rs)[..., 3].copy())
            s.vis.CreateICPMaps(v, rs, ses.points, ses.normals)
        elif k == 4:
            other = synth.pose_matrix(synth.bench_position(k + 3))
            s.reco.AllocateSceneFromDepth(v, rs); s.reco.IntegrateIntoScene(v, rs)
            s.vis.CreateExpectedDepths(other, v.intr_d, rs)              # not the view's pose: nothing may be fused
            s.vis.CreateExpectedDepths(v.M_d, v.intr_d, rs)
            s.vis.CreateICPMaps(v, rs, ses.points, ses.normals)
        elif k == 5:
            s.reco.AllocateSceneFromDepth(v, rs)                          # fusion switched off for a frame: no integration
            s.vis.CreateExpectedDepths(v.M_d, v.intr_d, rs); s.vis.CreateICPMaps(v, rs, ses.points, ses.normals)
        else:
            s.reco.AllocateSceneFromDepth(v, rs, onlyUpdateVisibleList=True)
            s.process_frame(v, rs, ses.points, ses.normals)

    both(hip, oracle, sc, drive, "partial sequences")
    a, b = extra[id(hip)], extra[id(oracle)]
    assert a[0] == b[0]
    assert np.array_equal(a[1], b[1]), "free-view image rendered between CreateExpectedDepths and CreateICPMaps"
    assert np.array_equal(a[2], b[2]), "FindSurface hit mask"


def test_an_image_overwritten_between_the_calls_is_read_as_it_was(hip, oracle):
    """The recorded calls read the view's depth image when they are launched -- so a copy into that image through the library must
    launch them first: the allocation sees the OLD image, the integration the new one, exactly as with immediate calls."""
    sc = Scenario(name="overwrite", voxelSize=0.005, frames=3, trajectory="bench")

    def drive(ses, k):
        s, rs = ses.scene, ses.rs
        v = ses.view(k)
        s.reco.AllocateSceneFromDepth(v, rs)
        newer = np.ascontiguousarray(sc.depth(k + 1))
        ses.be.check(ses.be.fn["memcpy_h2d"](capi._P(ses._depth.ptr), newer.ctypes.data_as(capi._P), newer.nbytes, None), "memcpy_h2d")
        ses.be.sync()
        s.reco.IntegrateIntoScene(v, rs)
        s.vis.CreateExpectedDepths(v.M_d, v.intr_d, rs)
        s.vis.CreateICPMaps(v, rs, ses.points, ses.normals)

    both(hip, oracle, sc, drive, "depth overwritten between allocation and integration")


def test_a_view_that_changes_between_the_calls_is_not_fused(hip, oracle):
    """IntegrateIntoScene for ANOTHER view than the recorded allocation: both happen, in order, each with its own view."""
    sc = Scenario(name="changing_view", voxelSize=0.005, frames=3, trajectory="bench")

    def drive(ses, k):
        s, rs = ses.scene, ses.rs
        v = ses.view(k)
        keep = ses._depth
        w = ses.view(k + 1)          # another image, another pose
        s.reco.AllocateSceneFromDepth(v, rs)
        s.reco.IntegrateIntoScene(w, rs)
        s.vis.CreateExpectedDepths(w.M_d, w.intr_d, rs)
        s.vis.CreateICPMaps(w, rs, ses.points, ses.normals)
        del keep

    both(hip, oracle, sc, drive, "view changed between the calls")


def test_recorded_calls_survive_synchronize_and_destruction(hip, oracle):
    """itm_stream_synchronize launches what was recorded on the stream; destroying the render state does not drop an allocation."""
    sc = Scenario(name="sync_destroy", voxelSize=0.008, w=320, h=240, frames=2)
    tables = []
    for be in (hip, oracle):
        ses = T.Session(be, sc)
        v = ses.view(0)
        ses.scene.reco.AllocateSceneFromDepth(v, ses.rs)
        ses.scene.reco.IntegrateIntoScene(v, ses.rs)
        be.sync()
        vox = ses.scene.download(capi.BUF_VOXEL_BLOCKS)
        v1 = ses.view(1)
        ses.scene.reco.AllocateSceneFromDepth(v1, ses.rs)
        ses.rs.close()
        tables.append((vox, ses.scene.download(capi.BUF_HASH_ENTRIES), ses.scene.counters(None)["lastFreeBlockId"]))
        ses.rs = ses.scene.vis.CreateRenderState((sc.w, sc.h))
        ses.close()
    T.assert_fields_equal(tables[0][0], tables[1][0], "voxels after allocate + integrate + synchronize")
    T.assert_fields_equal(tables[0][1], tables[1][1], "table after the render state was destroyed with a recorded allocation")
    assert tables[0][2] == tables[1][2]


def test_refusals_are_reported_by_the_call_that_causes_them(hip):
    """What a recorded call could be refused for is checked when it is made, not when it is launched."""
    sc = Scenario(name="refusals", voxelType=capi.VOXEL_S_RGB, colour=True, voxelSize=0.01, w=160, h=120, frames=1)
    ses = T.Session(hip, sc)
    v = ses.view(0)
    singular = capi.View(v.depth, sc.w, sc.h, M_d=np.zeros((4, 4), np.float32), intr_d=sc.intr(), rgb=ses.rgb, w_rgb=sc.w, h_rgb=sc.h, intr_rgb=sc.intr())
    with pytest.raises(capi.ItmError, match="singular"):
        ses.scene.reco.AllocateSceneFromDepth(singular, ses.rs)
    ses.scene.reco.AllocateSceneFromDepth(v, ses.rs)
    no_rgb = capi.View(v.depth, sc.w, sc.h, M_d=sc.pose(0), intr_d=sc.intr())
    with pytest.raises(capi.ItmError, match="rgb"):
        ses.scene.reco.IntegrateIntoScene(no_rgb, ses.rs)
    ses.scene.reco.IntegrateIntoScene(v, ses.rs)
    ses.scene.vis.CreateExpectedDepths(v.M_d, v.intr_d, ses.rs)
    ses.scene.vis.CreateICPMaps(v, ses.rs, ses.points, ses.normals)
    assert ses.scene.counters(ses.rs)["noVisibleEntries"] > 0
    ses.close()


def test_four_calls_from_several_host_threads(hip, oracle):
    """Three scenes driven by three host threads at once (ctypes releases the interpreter lock inside every call): the recording of one
    thread's calls, the flushes its uploads trigger and the fused launches -- made outside the library's lock -- must not disturb the
    others.  Every scene equals the oracle's sequential run of the same frames."""
    import threading
    scs = [Scenario(name="thr%d" % g, w=320, h=240, voxelSize=0.008, frames=8, trajectory="bench", stream=g) for g in range(3)]
    sessions = [T.Session(hip, sc) for sc in scs]
    errors = []

    def drive(ses):
        try:
            for k in range(ses.sc.frames):
                ses.frame(k, fused="four")
                if k % 3 == 2:
                    ses.scene.counters(ses.rs)            # an observer now and then (flushes nothing: the frame is complete)
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=drive, args=(ses,)) for ses in sessions]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for ses, sc in zip(sessions, scs):
        got = ses.snapshot()
        ses.close()
        ref = T.Session(oracle, sc)
        for k in range(sc.frames):
            ref.frame(k, fused="four")
        want = ref.snapshot()
        ref.close()
        T.compare_results(got, want, sc, what=sc.name + "/threads")
