"""itm_depth_stager: raw frames from host memory on a copy stream, ahead of the frame being fused -- the MI355X counterpart of the
synchronous shortImage->SetFrom(rawDepthImage, CPU_TO_CUDA) that opens ITMViewBuilder_CUDA::UpdateView
(Engine/DeviceSpecific/CUDA/ITMViewBuilder_CUDA.cu:53).  What reaches itm_update_view must be the frames, in order."""
import ctypes as C

import numpy as np
import pytest

from infinitam_amd import capi, synth

W, H = 160, 120


def frames(n):
    intr = synth.intrinsics_for(W, H)
    return [np.ascontiguousarray(synth.raw_depth_mm(W, H, synth.bench_position(3 * k), intr)) for k in range(n)], intr


@pytest.mark.gpu
def test_frames_leave_the_stager_in_order_and_converted_like_direct_uploads(hip):
    raws, intr = frames(7)
    intr_c = (C.c_float * 4)(*intr)
    g = C.c_void_p()
    hip.check(hip.fn["depth_stager_create"](W, H, 3, C.byref(g)), "create")
    st = C.c_void_p()
    hip.check(hip.fn["stream_create"](C.byref(st)), "stream_create")
    out = capi.DevBuffer(hip, W * H * 4, np.float32, (H, W))
    direct = capi.DevBuffer(hip, W * H * 4, np.float32, (H, W))
    try:
        hip.check(hip.fn["depth_stager_upload"](g, raws[0].ctypes.data_as(C.c_void_p)), "upload")
        for k in range(len(raws)):
            if k + 1 < len(raws):
                hip.check(hip.fn["depth_stager_upload"](g, raws[k + 1].ctypes.data_as(C.c_void_p)), "upload")      # one frame ahead
            dev = C.c_void_p()
            hip.check(hip.fn["depth_stager_acquire"](g, st, C.byref(dev)), "acquire")
            hip.check(hip.fn["update_view"](dev, W, H, 1, 0.001, 0.0, intr_c, 0, 0, C.c_void_p(out.ptr), None, None, None, st), "update_view")
            hip.check(hip.fn["depth_stager_release"](g, st), "release")
            hip.sync(st.value)
            got = out.numpy().copy()
            d = hip.to_backend(raws[k])
            hip.check(hip.fn["update_view"](C.c_void_p(d.ptr), W, H, 1, 0.001, 0.0, intr_c, 0, 0, C.c_void_p(direct.ptr), None, None, None, None), "update_view")
            hip.sync()
            assert np.array_equal(got, direct.numpy()), k
            assert np.count_nonzero(got > 0) > W * H // 2
    finally:
        hip.check(hip.fn["depth_stager_destroy"](g), "destroy")
        hip.check(hip.fn["stream_destroy"](st), "stream_destroy")


@pytest.mark.gpu
def test_the_stager_refuses_what_would_lose_or_reorder_a_frame(hip):
    raws, _ = frames(4)
    g = C.c_void_p()
    hip.check(hip.fn["depth_stager_create"](W, H, 2, C.byref(g)), "create")
    dev = C.c_void_p()
    try:
        assert hip.fn["depth_stager_acquire"](g, None, C.byref(dev)) == capi.ERR_INVALID          # nothing uploaded
        assert hip.fn["depth_stager_release"](g, None) == capi.ERR_INVALID                        # nothing held
        for k in range(2):
            hip.check(hip.fn["depth_stager_upload"](g, raws[k].ctypes.data_as(C.c_void_p)), "upload")
        assert hip.fn["depth_stager_upload"](g, raws[2].ctypes.data_as(C.c_void_p)) == capi.ERR_INVALID   # both slots hold unreleased frames
        hip.check(hip.fn["depth_stager_acquire"](g, None, C.byref(dev)), "acquire")
        assert hip.fn["depth_stager_acquire"](g, None, C.byref(dev)) == capi.ERR_INVALID          # one at a time
        hip.check(hip.fn["depth_stager_release"](g, None), "release")
        hip.check(hip.fn["depth_stager_upload"](g, raws[2].ctypes.data_as(C.c_void_p)), "upload")
        hip.sync()
        # the upload-done query: two frames wait to be acquired; once the copy stream has drained no host buffer is still being read
        waiting, busy = C.c_int(-1), C.c_int(-1)
        for _ in range(2000):
            hip.check(hip.fn["depth_stager_pending"](g, C.byref(waiting), C.byref(busy)), "pending")
            if busy.value == 0:
                break
        assert waiting.value == 2 and busy.value == 0
        assert hip.fn["depth_stager_create"](W, H, 1, C.byref(dev)) == capi.ERR_INVALID
    finally:
        hip.check(hip.fn["depth_stager_destroy"](g), "destroy")


@pytest.mark.gpu
@pytest.mark.parametrize("calib", [(1, 0.001, 0.0), (1, 0.0005, 0.01), (0, 1135.09, 0.0819141)], ids=["affine mm", "affine with offset", "kinect disparity"])
@pytest.mark.parametrize("pinned", [True, False], ids=["page-locked host memory (copy kernel)", "pageable host memory (runtime copy)"])
def test_the_copy_that_also_converts_gives_update_views_depth_image(hip, calib, pinned):
    """itm_depth_stager_set_conversion: the float depth image a slot carries must be, bit for bit, what itm_update_view makes of the same raw
    frame (no filter, no noise model) -- zeros, negative and out-of-range samples included -- and the raw copy beside it the frame itself."""
    calib_type, c0, c1 = calib
    raws, intr = frames(5)
    rng = np.random.default_rng(7)
    for r in raws:      # samples the conversions reject or clamp
        idx = rng.integers(0, r.size, 400)
        r.reshape(-1)[idx] = rng.choice(np.array([0, -5, 32001, 32767, -32768, 1, 32000], np.int16), 400)
        if calib_type == 0:
            r.reshape(-1)[idx[:50]] = 1135      # c0 - raw == 0.09: huge depths; and the exact zero of the denominator cannot occur for integers
    intr_c = (C.c_float * 4)(*intr)
    g = C.c_void_p()
    hip.check(hip.fn["depth_stager_create"](W, H, 3, C.byref(g)), "create")
    host = []
    try:
        assert hip.fn["depth_stager_acquire_depth"](g, None, None, C.byref(C.c_void_p())) == capi.ERR_INVALID      # no conversion set
        hip.check(hip.fn["depth_stager_set_conversion"](g, calib_type, c0, c1, intr[0]), "set_conversion")
        for r in raws:
            if pinned:
                p = C.c_void_p()
                hip.check(hip.fn["host_malloc"](C.byref(p), r.nbytes), "host_malloc")
                C.memmove(p, r.ctypes.data, r.nbytes)
                host.append(p)
            else:
                host.append(r.ctypes.data_as(C.c_void_p))
        direct = capi.DevBuffer(hip, W * H * 4, np.float32, (H, W))
        hip.check(hip.fn["depth_stager_upload"](g, host[0]), "upload")
        for k in range(len(raws)):
            if k + 1 < len(raws):
                hip.check(hip.fn["depth_stager_upload"](g, host[k + 1]), "upload")
            raw_dev, depth_dev = C.c_void_p(), C.c_void_p()
            hip.check(hip.fn["depth_stager_acquire_depth"](g, None, C.byref(raw_dev), C.byref(depth_dev)), "acquire_depth")
            hip.sync()
            got = np.zeros((H, W), np.float32); got_raw = np.zeros((H, W), np.int16)
            hip.check(hip.fn["memcpy_d2h"](got.ctypes.data_as(C.c_void_p), depth_dev, got.nbytes, None), "d2h"); hip.sync()
            hip.check(hip.fn["memcpy_d2h"](got_raw.ctypes.data_as(C.c_void_p), raw_dev, got_raw.nbytes, None), "d2h"); hip.sync()
            hip.check(hip.fn["depth_stager_release"](g, None), "release")
            d = hip.to_backend(raws[k])
            hip.check(hip.fn["update_view"](C.c_void_p(d.ptr), W, H, calib_type, c0, c1, intr_c, 0, 0, C.c_void_p(direct.ptr), None, None, None, None), "update_view")
            hip.sync()
            want = direct.numpy()
            assert np.array_equal(got_raw, raws[k]), k
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, np.nonzero(got != want)[0][:5])
            assert np.count_nonzero(got > 0) > (W * H // 2 if calib_type == 1 else 1000) and np.count_nonzero(got == -1.0) >= 50
        # a conversion cannot be changed under frames that wait in the ring
        hip.check(hip.fn["depth_stager_upload"](g, host[0]), "upload")
        assert hip.fn["depth_stager_set_conversion"](g, 1, 0.001, 0.0, intr[0]) == capi.ERR_INVALID
    finally:
        hip.sync()
        hip.check(hip.fn["depth_stager_destroy"](g), "destroy")
        if pinned:
            for p in host:
                hip.check(hip.fn["host_free"](p), "host_free")


@pytest.mark.gpu
def test_release_launches_the_recorded_calls_that_read_the_slot(hip, oracle):
    """A slot's float image is a view's depth.  On a scene that records its engine calls (itm_scene_set_deferred_fusion) the allocation and
    the integration of a frame may still be RECORDED when the slot is released -- their CreateICPMaps has not come -- and the release says
    "everything submitted so far is what read it": it must launch them first.  Ring of two slots; after every release the frame after
    next is uploaded into the slot just released (an idle device lets the copy run at once), and only then the frame is completed."""
    import itm_testlib as T
    sc = T.Scenario(name="stager_release", w=W, h=H, voxelSize=0.01, frames=4, noise_seed=4242)
    b = T.run_scenario(oracle, sc)
    # (with a noise seed the scenario's depth images ARE the converted raw frames, itm_testlib.Scenario.depth)
    raws = [np.ascontiguousarray(synth.raw_depth_mm(W, H, sc.position(k), sc.intr(), sc.noise_seed + k)) for k in range(sc.frames + 2)]
    assert np.array_equal(synth.depth_from_raw(raws[1]), sc.depth(1))
    ses = T.Session(hip, sc)
    g = C.c_void_p()
    hip.check(hip.fn["depth_stager_create"](W, H, 2, C.byref(g)), "create")
    try:
        hip.check(hip.fn["depth_stager_set_conversion"](g, 1, 0.001, 0.0, sc.intr()[0]), "set_conversion")
        hip.check(hip.fn["depth_stager_upload"](g, raws[0].ctypes.data_as(C.c_void_p)), "upload")
        hip.check(hip.fn["depth_stager_upload"](g, raws[1].ctypes.data_as(C.c_void_p)), "upload")
        for k in range(sc.frames):
            depth_dev = C.c_void_p()
            hip.check(hip.fn["depth_stager_acquire_depth"](g, None, None, C.byref(depth_dev)), "acquire_depth")
            v = ses.view(k)
            v.depth = depth_dev.value          # (View.struct() takes a DevBuffer or a raw device pointer)
            ses.scene.reco.AllocateSceneFromDepth(v, ses.rs)          # recorded
            ses.scene.reco.IntegrateIntoScene(v, ses.rs)              # recorded
            hip.check(hip.fn["depth_stager_release"](g, None), "release")
            hip.check(hip.fn["depth_stager_upload"](g, raws[k + 2].ctypes.data_as(C.c_void_p)), "upload")      # into the slot just released
            hip.sync()
            ses.scene.vis.CreateExpectedDepths(v.M_d, v.intr_d, ses.rs)
            ses.scene.vis.CreateICPMaps(v, ses.rs, ses.points, ses.normals)
        a = ses.snapshot()
        a.counters = [ses.scene.counters(ses.rs)]
        b.counters = b.counters[-1:]
        T.compare_results(a, b, sc, what="frames whose slot is overwritten right after its release")
    finally:
        hip.sync()
        hip.check(hip.fn["depth_stager_destroy"](g), "destroy")
        ses.close()
