"""itm_process_frame_ahead: the next frame's per-pixel block requests ride in this frame's last launch, beside the ICP maps (the request
stage of AllocateSceneFromDepth -- buildHashAllocAndVisibleTypePP, DeviceAgnostic/ITMSceneReconstructionEngine.h:141-241 -- only reads
the table as this frame's allocation left it).  Frame by frame the results must be those of the four reference calls
(Engine/ITMMainEngine.cpp:123-126), whatever mixture of calls with and without a successor the host makes."""
import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import capi, synth
from test_accel_origin import walk_poses


def run_ahead(be, sc, pattern, poses=None):
    """pattern[k]: True = frame k is issued with frame k + 1 as its successor."""
    ses = T.Session(be, sc)
    n = sc.frames
    depth = [be.to_backend(sc.depth(k)) for k in range(n)]
    views = [capi.View(depth[k], sc.w, sc.h, M_d=(poses[k] if poses is not None else sc.pose(k)), intr_d=sc.intr(), rgb=ses.rgb, w_rgb=sc.w, h_rgb=sc.h, intr_rgb=sc.intr())
             for k in range(n)]
    per_frame = []
    for k in range(n):
        nxt = views[k + 1] if (k + 1 < n and pattern[k] and be.on_device) else None
        if be.on_device:
            ses.scene.process_frame_ahead(views[k], nxt, ses.rs, ses.points, ses.normals)
        else:
            ses.scene.process_frame(views[k], ses.rs, ses.points, ses.normals)
        per_frame.append((ses.scene.counters(ses.rs), ses.points.numpy().copy()))
    res = ses.snapshot()
    res.counters = [c for c, _ in per_frame]
    return ses, res, per_frame


@pytest.mark.gpu
@pytest.mark.parametrize("pattern", ["always", "alternate", "never"])
@pytest.mark.parametrize("name,kw", [("hash_s", dict(voxelSize=0.005)), ("hash_f_rgb", dict(voxelSize=0.01, voxelType=capi.VOXEL_F_RGB, colour=True)),
                                     ("large", dict(w=1280, h=960, voxelSize=0.01))])
def test_frames_issued_ahead_equal_the_reference_sequence(hip, oracle, pattern, name, kw):
    sc = T.Scenario(name="ahead_" + name, frames=6, trajectory="bench", **kw)
    pat = {"always": [True] * 6, "alternate": [k % 2 == 0 for k in range(6)], "never": [False] * 6}[pattern]
    hs, a, pa = run_ahead(hip, sc, pat)
    os_, b, pb = run_ahead(oracle, sc, pat)
    for k, ((ca, xa), (cb, xb)) in enumerate(zip(pa, pb)):
        for key in ("lastFreeBlockId", "lastFreeExcessListId", "noVisibleEntries"):
            assert ca[key] == cb[key], (k, key, ca, cb)
        assert np.array_equal(xa, xb), "frame %d: ICP points" % k
    T.compare_results(a, b, sc, what="ahead " + pattern)
    hs.close(); os_.close()


@pytest.mark.gpu
def test_requests_ahead_while_the_cubes_move(hip, oracle):
    """The successor's view may need the acceleration cubes somewhere else: they move between this frame's ray cast and the fused launch."""
    poses = walk_poses()
    sc = T.Scenario(name="ahead_walk", w=160, h=120, voxelSize=0.005, localBlockNum=0x40000, frames=len(poses))
    hs, a, pa = run_ahead(hip, sc, [True] * len(poses), poses)
    os_, b, pb = run_ahead(oracle, sc, [False] * len(poses), poses)
    for k, ((ca, xa), (cb, xb)) in enumerate(zip(pa, pb)):
        assert ca["noVisibleEntries"] == cb["noVisibleEntries"] and ca["lastFreeBlockId"] == cb["lastFreeBlockId"], (k, ca, cb)
        assert np.array_equal(xa, xb), "frame %d: ICP points" % k
    T.compare_results(a, b, sc, what="ahead while walking")
    assert hs.scene.accel_info()["moves"] >= 3
    hs.close(); os_.close()


@pytest.mark.gpu
def test_a_frame_other_than_the_announced_one_is_refused(hip):
    sc = T.Scenario(name="ahead_contract", w=160, h=120, voxelSize=0.01, frames=3)
    ses = T.Session(hip, sc)
    d = [hip.to_backend(sc.depth(k)) for k in range(3)]
    v = [capi.View(d[k], sc.w, sc.h, M_d=sc.pose(k), intr_d=sc.intr()) for k in range(3)]
    ses.scene.process_frame_ahead(v[0], v[1], ses.rs, ses.points, ses.normals)
    with pytest.raises(capi.ItmError):
        ses.scene.process_frame(v[2], ses.rs, ses.points, ses.normals)           # not the announced view
    ses.scene.process_frame(v[1], ses.rs, ses.points, ses.normals)               # the announced one goes through
    ses.scene.process_frame_ahead(v[2], v[0], ses.rs, ses.points, ses.normals)
    # while the requests are pending the visible types carry their marks: the list cannot be read, saved, rewritten or rebuilt,
    # and no other render state of the scene may allocate
    with pytest.raises(capi.ItmError, match="issued ahead"):
        ses.scene.download(capi.BUF_VISIBLE_TYPE, ses.rs)
    with pytest.raises(capi.ItmError, match="issued ahead"):
        ses.scene.vis.FindVisibleBlocks(sc.pose(0), sc.intr(), ses.rs)
    with pytest.raises(capi.ItmError, match="issued ahead"):
        ses.scene.upload(capi.BUF_VISIBLE_IDS, np.zeros(16, np.int32), ses.rs)
    other = ses.scene.vis.CreateRenderState((sc.w, sc.h))
    with pytest.raises(capi.ItmError, match="another render state"):
        ses.scene.reco.AllocateSceneFromDepth(v[0], other)
        ses.scene.flush(other)
    other.close()
    ses.scene.reco.ResetScene()                                                  # the table the requests were made against is gone, and so are they
    ses.scene.process_frame(v[1], ses.rs, ses.points, ses.normals)               # any view may follow
    assert ses.scene.counters(ses.rs)["noVisibleEntries"] > 0
    ses.close()


@pytest.mark.gpu
def test_cancelled_requests_leave_no_trace(hip, oracle):
    """itm_cancel_ahead: the frames after it -- for ANOTHER view than the announced one, through the four separate calls -- equal the
    oracle's, and the visible types read like the reference's after its "previous list -> 3" loop."""
    sc = T.Scenario(name="ahead_cancel", voxelSize=0.005, frames=6, trajectory="bench")
    ses, ref = T.Session(hip, sc), T.Session(oracle, sc)
    d = [hip.to_backend(sc.depth(k)) for k in range(sc.frames)]
    v = [capi.View(d[k], sc.w, sc.h, M_d=sc.pose(k), intr_d=sc.intr()) for k in range(sc.frames)]
    ses.scene.process_frame_ahead(v[0], v[1], ses.rs, ses.points, ses.normals)
    ses.scene.process_frame_ahead(v[1], v[2], ses.rs, ses.points, ses.normals)
    ses.scene.cancel_ahead(ses.rs)
    types = ses.scene.download(capi.BUF_VISIBLE_TYPE, ses.rs)
    ref.frame(0); ref.frame(1)
    want = np.where(ref.scene.download(capi.BUF_VISIBLE_TYPE, ref.rs) > 0, 3, 0).astype(np.uint8)
    assert np.array_equal(types, want), "types after the cancellation"
    for k in (4, 5, 3):                                                          # not the announced frame 2
        ses.view = lambda kk, _v=v: _v[kk]
        ses.frame(k, fused=(k == 5))
        ref.frame(k)
    a, b = ses.snapshot(), ref.snapshot()
    a.counters = [ses.scene.counters(ses.rs)]; b.counters = [ref.scene.counters(ref.rs)]
    T.compare_results(a, b, sc, what="after itm_cancel_ahead")
    ses.close(); ref.close()
