"""Dense integration: the per-column frustum cull (integrate.hip, ColumnCull) must never drop a voxel the exact per-voxel test of
computeUpdatedVoxelDepthInfo (DeviceAgnostic/ITMSceneReconstructionEngine.h:9-50) keeps.  The volume after integrating from cameras
in general position -- rotated about all three axes, inside and outside the volume, looking along and across its faces -- must be
bit-identical to the oracle's, and the per-group test (debug key 9) must give the same volume."""
import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import capi

F = np.float32


def rotation(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def pose(R_cam_to_world, position):
    """World->camera M_d, column-major float32[16]."""
    Rt = R_cam_to_world.T
    m = np.eye(4)
    m[:3, :3] = Rt
    m[:3, 3] = -(Rt @ np.asarray(position, np.float64))
    return np.ascontiguousarray(m.astype(F).T).reshape(16).copy()


# (rotation angles, camera position in metres); the 128^3 volume of 8 mm voxels spans [-0.512, 0.512]^2 x [0.8, 1.824]
CAMERAS = [
    ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),                 # axis aligned: top / bottom planes do not depend on x, left / right not on y
    ((0.0, 0.0, 1e-6), (0.0, 0.0, 0.0)),                # almost aligned: tiny slopes
    ((0.3, -0.2, 0.4), (0.1, -0.05, 0.2)),              # general position
    ((0.0, 1.2, 0.0), (-1.2, 0.0, 1.3)),                # looking across the volume from the side
    ((-1.0, 0.0, 0.7), (0.0, -1.3, 1.3)),               # from above, rolled
    ((0.1, 0.2, -0.3), (0.05, 0.1, 1.3)),               # camera inside the volume
    ((0.0, np.pi, 0.0), (0.0, 0.0, 3.0)),               # looking back at the volume from behind
    ((0.2, 0.1, 0.0), (0.9, 0.4, -4.0)),                # far away: the whole volume inside a narrow part of the frustum
    ((0.0, 0.0, np.pi / 2), (0.0, 0.0, 0.3)),           # rolled by 90 degrees: planes swap roles
]


def integrate_all(be, keys=()):
    for k in keys:
        be.check(be.fn["debug_set"](k, 1), "debug_set")
    try:
        W, H = 160, 120
        s = be.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, capi.default_params(voxelSize=0.008, mu=0.04), denseSize=(128, 128, 128), denseOffset=(-64, -64, 100))
        s.reco.ResetScene()
        rs = s.vis.CreateRenderState((W, H))
        rng = np.random.default_rng(7)
        volumes = []
        for i, (ang, pos) in enumerate(CAMERAS):
            depth = ((5.0 if pos[2] < -1 else 1.0) + 0.5 * rng.random((H, W))).astype(F)          # every pixel valid: every projected voxel is touched or rejected by eta
            depth[::7, ::5] = 0.0                                       # and a few invalid ones
            v = capi.View(be.to_backend(depth), W, H, M_d=pose(rotation(*ang), pos), intr_d=(145.0, 145.0, 80.0, 60.0))
            s.reco.IntegrateIntoScene(v, rs)
            volumes.append(s.download(capi.BUF_VOXEL_BLOCKS).copy())
        return volumes
    finally:
        for k in keys:
            be.check(be.fn["debug_set"](k, 0), "debug_set")


@pytest.mark.gpu
def test_column_cull_equals_exact_test_for_cameras_in_general_position(hip, oracle):
    want = integrate_all(oracle)
    got = integrate_all(hip)
    per_group = integrate_all(hip, keys=(9,))
    touched = 0
    for i, (a, b, c) in enumerate(zip(want, got, per_group)):
        assert np.array_equal(a, b), "camera %d: %d voxels differ" % (i, int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))))
        assert np.array_equal(a, c), "camera %d (per-group cull)" % i
        touched += int(np.count_nonzero(a.view(np.uint32) != want[i - 1].view(np.uint32))) if i else int(np.count_nonzero(a.view(np.uint32) != 32767))
    assert touched > 500000          # the cameras really see the volume
