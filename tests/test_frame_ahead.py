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
    ses.scene.reco.ResetScene()                                                  # the table the requests were made against is gone
    with pytest.raises(capi.ItmError):
        ses.scene.process_frame(v[0], ses.rs, ses.points, ses.normals)
    ses.close()
