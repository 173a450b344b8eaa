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
