"""BASELINE configs[0] ("Teddy offline sequence ... reference plumbing"): frames on disk in the reference's formats
(16-bit big-endian PGM holding Kinect DISPARITY, PPM colour, calibration text with non-square pixels, an rgb->depth
extrinsic and `kinect` disparity parameters) driven through the whole call order of ITMMainEngine::ProcessFrame:

    read files -> ITMViewBuilder::UpdateView (convertDisparityToDepth, DeviceAgnostic/ITMViewBuilder.h:7-20;
    DeviceSpecific/CPU/ITMViewBuilder_CPU.cpp:14-64) -> AllocateSceneFromDepth -> IntegrateIntoScene (colour voxels, so the
    rgb intrinsics and the extrinsic matter) -> CreateExpectedDepths -> CreateICPMaps.

The Teddy frames are not in the reference tree, so the scene is the synthetic sphere + wall encoded as disparity, and the
calibration is tests/golden/calib_synthetic.txt (Teddy-like semantics, synthetic numbers).
  * CPU: oracle == the reference's ITMViewBuilder_CPU + engines, bit for bit;
  * GPU: HIP == oracle, bit for bit (the conversion is one IEEE multiply chain and one IEEE division)."""
import ctypes as C
import os

import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import synth
from infinitam_amd.capi import DevBuffer, View

W, H, FRAMES = 640, 480, 3
CALIB = os.path.join(T.GOLDEN_DIR, "calib_synthetic.txt")


def write_sequence(io, tmp_path, calib):
    """The synthetic scene as the files the reference's ImageFileReader would be pointed at."""
    intr_d = tuple(calib.intr_d)
    c0, c1 = calib.disparityParams[0], calib.disparityParams[1]
    files = []
    for k in range(FRAMES):
        z = synth.depth_z(W, H, synth.parity_position(k), intr_d).astype(np.float64)
        disp = c0 - 8.0 * c1 * intr_d[0] / z
        raw = np.clip(np.round(disp), -32768, 32767).astype(np.int16)
        raw[5:9, 7:30] = int(c0)            # c0 - raw = 0.25 -> a "depth" of 1.4 km: valid for the view builder, rejected by the frustum
        raw[20:24, 100:140] = 2047          # beyond the disparity origin -> negative depth -> invalid (-1)
        d, c = str(tmp_path / f"{k:04d}.pgm"), str(tmp_path / f"{k:04d}.ppm")
        io.write_image(d, raw)
        io.write_image(c, synth.rgb_frame(W, H))
        files.append((d, c))
    return files


def run_sequence(be, io, files, calib, fused):
    sc = T.Scenario(name="config1", w=W, h=H, voxelType=T.VOXEL_S_RGB, colour=True, voxelSize=0.005, frames=FRAMES)
    ses = T.Session(be, sc)
    intr_d, intr_rgb = tuple(calib.intr_d), tuple(calib.intr_rgb)
    ip = (C.c_float * 4)(*intr_d)
    depth = DevBuffer(be, W * H * 4, np.float32, (H, W))
    scratch = DevBuffer(be, W * H * 4, np.float32, (H, W))
    depths = []
    for k, (dfile, cfile) in enumerate(files):
        raw = be.to_backend(io.read_depth_image(dfile))
        rgb = be.to_backend(io.read_rgb_image(cfile))
        be.check(be.fn["update_view"](raw.ptr, W, H, calib.disparityType, calib.disparityParams[0], calib.disparityParams[1], ip,
                                      0, 0, depth.ptr, scratch.ptr, None, None, None), "update_view")
        depths.append(depth.numpy())
        v = View(depth, W, H, M_d=sc.pose(k), intr_d=intr_d, rgb=rgb, w_rgb=W, h_rgb=H, intr_rgb=intr_rgb,
                 rgb_to_depth=np.array(calib.rgb_to_depth[:], np.float32), rgb_to_depth_inv=np.array(calib.rgb_to_depth_inv[:], np.float32))
        if fused:
            ses.scene.process_frame(v, ses.rs, ses.points, ses.normals)
        else:
            ses.scene.reco.AllocateSceneFromDepth(v, ses.rs)
            ses.scene.reco.IntegrateIntoScene(v, ses.rs)
            ses.scene.vis.CreateExpectedDepths(v.M_d, v.intr_d, ses.rs)
            ses.scene.vis.CreateICPMaps(v, ses.rs, ses.points, ses.normals)
    res = ses.snapshot()
    ses.close()
    return sc, res, depths


def check_plausible(res, depths):
    d = depths[-1]
    assert (d[5:9, 7:30] > 1000).all() and (d[20:24, 100:140] == -1).all()
    valid = (d > 0) & (d < 1000)
    assert valid.mean() > 0.99 and 0.9 < d[valid].min() < 1.1 and 2.4 < d[valid].max() < 2.6      # sphere front ... wall
    assert (res.voxels["w_color"] > 0).sum() > 100000 and (res.raycast[..., 3] > 0).mean() > 0.9


def test_oracle_matches_reference_on_the_disparity_sequence(oracle, reference, hip_host, tmp_path):
    calib = hip_host.read_rgbd_calib(CALIB)
    files = write_sequence(hip_host, tmp_path, calib)
    sc, a, da = run_sequence(oracle, hip_host, files, calib, fused=False)
    # the reference side reads the files with ITS OWN readers and calibration parser
    rcal = reference.read_rgbd_calib(CALIB)
    _, b, db = run_sequence(reference, reference, files, rcal, fused=False)
    for x, y in zip(da, db):
        assert np.array_equal(x, y)
    T.compare_results(a, b, sc, what="config 1 (disparity) oracle vs reference")
    check_plausible(a, da)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_hip_matches_oracle_on_the_disparity_sequence(hip, oracle, tmp_path, fused):
    calib = hip.read_rgbd_calib(CALIB)
    files = write_sequence(hip, tmp_path, calib)
    sc, a, da = run_sequence(hip, hip, files, calib, fused=fused)
    _, b, db = run_sequence(oracle, hip, files, calib, fused=False)
    for x, y in zip(da, db):
        assert np.array_equal(x, y), "convertDisparityToDepth differs"
    T.compare_results(a, b, sc, what="config 1 (disparity) HIP vs oracle")
    check_plausible(a, da)
