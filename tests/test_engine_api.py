"""Every entry point of the engine interface beyond the per-frame sequence: FindVisibleBlocks,
RenderImage (three types), FindSurface, ForwardRender, CreatePointCloud, onlyUpdateVisibleList,
view-builder conversions, state upload/download and the visible-list record.

The same scripted sequence is run on two implementations of the C-ABI and every output compared
bit-for-bit: oracle vs reference (CPU, pins the oracle) and HIP vs oracle (GPU)."""
import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import capi, synth
from infinitam_amd.capi import (BUF_FORWARD_PROJECTION, BUF_HASH_ENTRIES, BUF_MISSING_POINTS, BUF_RAYCAST_IMAGE,
                                BUF_RAYCAST_RESULT, BUF_VISIBLE_IDS, BUF_VOXEL_BLOCKS, DevBuffer, View)
from itm_testlib import Scenario

API_SCENARIOS = [
    Scenario(name="api_hash_s", w=160, h=120, voxelSize=0.01, frames=3),
    Scenario(name="api_hash_f_rgb", w=160, h=120, voxelSize=0.01, frames=3, voxelType=T.VOXEL_F_RGB, colour=True),
    Scenario(name="api_dense_s_rgb", w=160, h=120, voxelSize=0.01, frames=3, voxelType=T.VOXEL_S_RGB, colour=True,
             indexType=T.INDEX_DENSE, denseSize=(64, 64, 64), denseOffset=(-32, -32, 95)),
]


def api_sequence(be, sc: Scenario) -> dict:
    out = {}
    ses = T.Session(be, sc)
    s, rs = ses.scene, ses.rs
    W, H, P = sc.w, sc.h, sc.w * sc.h
    for k in range(sc.frames):
        v = ses.frame(k)
    out["counters_after_frames"] = s.counters(rs)

    # ---- ForwardRender on the next pose (approximate raycast path) -----------------------------
    v_next = ses.view(sc.frames)
    if s.is_hash:
        s.reco.AllocateSceneFromDepth(v_next, rs, onlyUpdateVisibleList=True)   # ITMDenseMapper::UpdateVisibleList
        out["counters_update_visible"] = s.counters(rs)
        out["visible_ids_update"] = s.download(BUF_VISIBLE_IDS, rs)[: out["counters_update_visible"]["noVisibleEntries"]]
        out["hash_after_update_visible"] = s.download(BUF_HASH_ENTRIES)
    s.vis.CreateExpectedDepths(v_next.M_d, v_next.intr_d, rs)
    s.vis.ForwardRender(v_next, rs)
    c = s.counters(rs)
    out["fwd_missing_count"] = c["noFwdProjMissingPoints"]
    out["fwd_missing"] = s.download(BUF_MISSING_POINTS, rs)[: c["noFwdProjMissingPoints"]]
    out["fwd_projection"] = s.download(BUF_FORWARD_PROJECTION, rs)
    out["fwd_image"] = s.download(BUF_RAYCAST_IMAGE, rs)

    # ---- free-view rendering from a different pose ------------------------------------------------
    free_rs = s.vis.CreateRenderState((W, H))
    t = sc.position(1)
    M = synth.pose_matrix_yaw((float(t[0]) + 0.05, float(t[1]) - 0.03, 0.02), 0.05)
    intr = sc.intr()
    s.vis.FindVisibleBlocks(M, intr, free_rs)
    if s.is_hash:
        cf = s.counters(free_rs)
        out["free_visible_count"] = cf["noVisibleEntries"]
        out["free_visible_ids"] = s.download(BUF_VISIBLE_IDS, free_rs)[: cf["noVisibleEntries"]]
    s.vis.CreateExpectedDepths(M, intr, free_rs)
    out["free_range"] = T.range_region(s.download(capi.BUF_RANGE_IMAGE, free_rs), W, H)
    img = DevBuffer(be, P * 4, np.uint8, (H, W, 4))
    for name, typ in (("grey", capi.RENDER_SHADED_GREYSCALE), ("volume", capi.RENDER_COLOUR_FROM_VOLUME),
                      ("normal", capi.RENDER_COLOUR_FROM_NORMAL)):
        be.check(be.fn["memcpy_h2d"](img.ptr, np.full((H, W, 4), 7, np.uint8).ctypes.data, P * 4, None), "h2d")
        be.sync()
        s.vis.RenderImage(M, intr, free_rs, img, typ)
        out["render_" + name] = img.numpy()
    s.vis.RenderImage(M, intr, free_rs, None, capi.RENDER_SHADED_GREYSCALE)   # into renderState->raycastImage
    out["render_default_target"] = s.download(BUF_RAYCAST_IMAGE, free_rs)
    s.vis.FindSurface(M, intr, free_rs)
    ray = s.download(BUF_RAYCAST_RESULT, free_rs)
    out["find_surface_w"] = ray[..., 3].copy()
    ray[ray[..., 3] <= 0, :3] = 0
    out["find_surface"] = ray

    # ---- colour-tracker point cloud ---------------------------------------------------------------
    loc = DevBuffer(be, P * 16, np.float32, (P, 4))
    col = DevBuffer(be, P * 16, np.float32, (P, 4))
    for skip in (False, True):
        s.vis.CreateExpectedDepths(v.M_d, v.intr_d, rs)
        s.vis.CreatePointCloud(v, rs, loc, col, skipPoints=skip)
        n = s.counters(rs)["noTotalPoints"]
        out[f"pc_count_{int(skip)}"] = n
        out[f"pc_locations_{int(skip)}"] = loc.numpy()[:n]
        out[f"pc_colours_{int(skip)}"] = col.numpy()[:n]
        out[f"pc_image_{int(skip)}"] = s.download(BUF_RAYCAST_IMAGE, rs)

    # ---- visible-list record for the multi-stream exchange -----------------------------------------
    if s.is_hash:
        rec = DevBuffer(be, (17 + 256) * 4, np.int32, (17 + 256,))
        Ma = np.ascontiguousarray(v.M_d, np.float32)
        be.check(be.fn["export_visible_record"](rs.h, Ma.ctypes.data_as(capi.C.POINTER(capi.C.c_float)), 256, rec.ptr, None), "export")
        be.sync()
        out["visible_record"] = rec.numpy()
    ses.close()
    return out


def compare(a: dict, b: dict, tag: str):
    assert a.keys() == b.keys()
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, dict):
            for kk in ("lastFreeBlockId", "lastFreeExcessListId", "noVisibleEntries"):
                assert x[kk] == y[kk], f"{tag}:{k}.{kk}: {x[kk]} vs {y[kk]}"
        elif isinstance(x, np.ndarray) and x.dtype.names:
            T.assert_fields_equal(x, y, f"{tag}:{k}")
        elif isinstance(x, np.ndarray):
            assert x.shape == y.shape, f"{tag}:{k} shape {x.shape} vs {y.shape}"
            assert np.array_equal(x, y), f"{tag}:{k}: {int((x != y).sum())} of {x.size} values differ"
        else:
            assert x == y, f"{tag}:{k}: {x} vs {y}"


@pytest.mark.parametrize("sc", API_SCENARIOS, ids=lambda s: s.name)
def test_oracle_api_matches_reference(oracle, reference, sc):
    compare(api_sequence(oracle, sc), api_sequence(reference, sc), sc.name)


@pytest.mark.gpu
@pytest.mark.parametrize("sc", API_SCENARIOS, ids=lambda s: s.name)
def test_hip_api_matches_oracle(hip, oracle, sc):
    compare(api_sequence(hip, sc), api_sequence(oracle, sc), sc.name)


def _raw_frames():
    rng = np.random.default_rng(7)
    raw = rng.integers(-50, 33000, size=(120, 160), dtype=np.int32).astype(np.int16)
    raw[0, :8] = [0, -1, 1, 32000, 32001, 32767, -32768, 1135]
    return raw


def _convert(be, raw):
    h, w = raw.shape
    src = be.to_backend(raw)
    dst = DevBuffer(be, w * h * 4, np.float32, (h, w))
    be.check(be.fn["convert_depth_affine"](src.ptr, dst.ptr, w, h, 0.001, 0.0, None), "affine")
    a = dst.numpy()
    be.check(be.fn["convert_disparity"](src.ptr, dst.ptr, w, h, 1135.09, 0.0819141, 573.71, None), "disparity")
    return a, dst.numpy()


def test_view_builder_conversions_oracle_vs_reference(oracle, reference):
    raw = _raw_frames()
    for x, y in zip(_convert(oracle, raw), _convert(reference, raw)):
        assert np.array_equal(x, y)


def test_view_builder_affine_semantics(oracle):
    """convertDepthAffineToFloat: raw <= 0 or > 32000 -> -1, else raw * a + b (DeviceAgnostic/ITMViewBuilder.h:22-28)."""
    raw = _raw_frames()
    a, _ = _convert(oracle, raw)
    want = np.where((raw <= 0) | (raw > 32000), np.float32(-1.0), raw.astype(np.float32) * np.float32(0.001) + np.float32(0.0))
    assert np.array_equal(a, want.astype(np.float32))


@pytest.mark.gpu
def test_view_builder_conversions_hip_vs_oracle(hip, oracle):
    raw = _raw_frames()
    for x, y in zip(_convert(hip, raw), _convert(oracle, raw)):
        assert np.array_equal(x, y)


def _roundtrip(be):
    """download -> fresh scene -> upload reproduces the scene (checkpoint / resume)."""
    sc = Scenario(name="rt", w=160, h=120, voxelSize=0.01, frames=2)
    ses = T.Session(be, sc)
    for k in range(2):
        ses.frame(k)
    snap = ses.snapshot()
    c = ses.scene.counters(ses.rs)
    ses2 = T.Session(be, sc)
    s2 = ses2.scene
    s2.upload(BUF_HASH_ENTRIES, snap.hash)
    s2.upload(BUF_VOXEL_BLOCKS, snap.voxels)
    s2.upload(capi.BUF_EXCESS_LIST, snap.excess)
    s2.upload(capi.BUF_ALLOCATION_LIST, snap.alloc_list)
    s2.upload(BUF_VISIBLE_IDS, snap.visible_ids, ses2.rs)
    s2.upload(capi.BUF_VISIBLE_TYPE, snap.visible_type, ses2.rs)
    s2.set_counters(ses2.rs, c["lastFreeBlockId"], c["lastFreeExcessListId"], c["noVisibleEntries"])
    ses.frame(2)
    ses2.frame(2)
    a, b = ses.snapshot(), ses2.snapshot()
    a.counters = [ses.scene.counters(ses.rs)]
    b.counters = [ses2.scene.counters(ses2.rs)]
    T.compare_results(a, b, sc, what="resume")
    ses.close()
    ses2.close()


def test_checkpoint_roundtrip_oracle(oracle):
    _roundtrip(oracle)


@pytest.mark.gpu
def test_checkpoint_roundtrip_hip(hip):
    _roundtrip(hip)


@pytest.mark.gpu
def test_kernel_timers_count_and_sample(hip):
    """itm_profile_enable / itm_profile_sample / itm_profile_read (the reference's NVTimer role): every enabled kernel of
    itm_process_frame is counted once per frame, or once per `every` frames when sampling; disabled kernels are not timed; the
    timers change nothing in the results (the session's frames are compared with an untimed run)."""
    from infinitam_amd import capi
    sc = Scenario(name="timers", voxelSize=0.005, frames=9, trajectory="bench")
    plain = T.Session(hip, sc)
    for k in range(sc.frames):
        plain.frame(k, fused=True)
    want = plain.snapshot()
    ses = T.Session(hip, sc)
    ray, integ = capi.TIMED_KERNELS.index("raycast"), capi.TIMED_KERNELS.index("integrate")
    ses.scene.profile_read(reset=True)
    ses.scene.profile_enable((1 << ray) | (1 << integ))
    for k in range(4):
        ses.frame(k, fused=True)
    p = ses.scene.profile_read(reset=True)
    assert p["raycast"]["calls"] == 4 and p["integrate"]["calls"] == 4 and p["request"]["calls"] == 0
    assert 0.0 < p["raycast"]["total_ms"] < 50.0
    ses.scene.profile_sample(3)                      # launches 0, 3 of the next five
    for k in range(4, 9):
        ses.frame(k, fused=True)
    p = ses.scene.profile_read(reset=True)
    assert p["raycast"]["calls"] == 2 and p["integrate"]["calls"] == 2
    ses.scene.profile_enable(0)
    T.compare_results(ses.snapshot(), want, sc, what="timed vs untimed")
    with pytest.raises(Exception):
        ses.scene.profile_sample(0)
