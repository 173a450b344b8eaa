"""Dense integration: the per-column frustum cull (integrate.hip, ColumnCull) must never drop a voxel the exact per-voxel test of
computeUpdatedVoxelDepthInfo (DeviceAgnostic/ITMSceneReconstructionEngine.h:9-50) keeps.  The volume after integrating from cameras
in general position -- rotated about all three axes, inside and outside the volume, looking along and across its faces -- must be
bit-identical to the oracle's, and the per-group test (debug key 9) must give the same volume."""
import ctypes as C

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


def integrate_all(be, keys=(), values=None):
    values = dict(values or {})
    for k in keys:
        values[k] = 1
    keys = list(values)
    for k, val in values.items():
        be.check(be.fn["debug_set"](k, val), "debug_set")
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
    unclassified = integrate_all(hip, values={16: 1})           # groups not classified against the depth tiles
    after_fetch = integrate_all(hip, values={16: 2, 17: 1})     # classified after the fetch, launch shape of rounds 1-2
    old_shape = integrate_all(hip, values={17: 1})              # four groups per lane instead of strips, classified before the fetch
    old_plain = integrate_all(hip, values={16: 1, 17: 1})       # ... not classified
    touched = 0
    for i, (a, f, g) in enumerate(zip(want, old_shape, old_plain)):
        assert np.array_equal(a, f), "camera %d (four groups per lane)" % i
        assert np.array_equal(a, g), "camera %d (four groups per lane, no classification)" % i
    for i, (a, b, c, d, e) in enumerate(zip(want, got, per_group, unclassified, after_fetch)):
        assert np.array_equal(a, b), "camera %d: %d voxels differ" % (i, int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))))
        assert np.array_equal(a, c), "camera %d (per-group cull)" % i
        assert np.array_equal(a, d), "camera %d (no classification)" % i
        assert np.array_equal(a, e), "camera %d (classification after the fetch)" % i
        touched += int(np.count_nonzero(a.view(np.uint32) != want[i - 1].view(np.uint32))) if i else int(np.count_nonzero(a.view(np.uint32) != 32767))
    assert touched > 500000          # the cameras really see the volume


# ---- the interval itself, on the host: thousands of poses against the exact float test ---------------------------------------------
def exact_keep(M, intr, W, H, vs, off, xs, ys, z):
    """computeUpdatedVoxelDepthInfo's projection test in float32, operation by operation (no FMA): True where the voxel is NOT rejected."""
    f = np.float32
    m = M.astype(f)
    mx = (xs + off[0]).astype(f) * f(vs)
    my = (ys + off[1]).astype(f) * f(vs)
    mz = f(z + off[2]) * f(vs)
    def row(j):
        return ((m[j] * mx[None, :] + m[j + 4] * my[:, None]) + m[j + 8] * mz) + m[j + 12] * f(1.0)
    pcx, pcy, pcz = row(0), row(1), row(2)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        u = f(intr[0]) * pcx / pcz + f(intr[2])
        v = f(intr[1]) * pcy / pcz + f(intr[3])
    return (~(pcz <= 0)) & ~((u < 1) | (u > f(W - 2)) | (v < 1) | (v > f(H - 2)))


def test_column_interval_never_excludes_a_voxel_the_exact_test_keeps(hip_host):
    rng = np.random.default_rng(11)
    W, H = 160, 120
    intr = np.array([145.0, 145.0, 80.0, 60.0], np.float32)
    size = np.array([128, 128, 128], np.int32); off = np.array([-64, -64, 100], np.int32)
    vs = 0.008
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    ys = np.arange(size[1])
    kept_rows = needed_rows = checked = 0
    poses = [pose(rotation(*ang), pos) for ang, pos in CAMERAS]
    for _ in range(300):
        ang = rng.uniform(-np.pi, np.pi, 3) * rng.choice([1.0, 0.2, 1e-3, 1e-6])
        pos = rng.uniform(-2.0, 2.0, 3) + np.array([0.0, 0.0, 1.3])
        poses.append(pose(rotation(*ang), pos))
    for M in poses:
        M = np.ascontiguousarray(M, np.float32)
        for _ in range(40):
            x0 = int(rng.integers(0, size[0] // 4)) * 4
            z = int(rng.integers(0, size[2]))
            lo, hi = C.c_int(), C.c_int()
            rc = hip_host.fn["debug_column_cull_rows"](fp(M), fp(intr), W, H, vs, ip(size), ip(off), x0, z, C.byref(lo), C.byref(hi))
            assert rc == 0
            keep = exact_keep(M, intr, W, H, vs, off, np.arange(x0, x0 + 4), ys, z).any(axis=1)     # per row: some voxel of the group passes
            inside = (ys >= lo.value) & (ys <= hi.value)
            assert not np.any(keep & ~inside), (M.tolist(), x0, z, lo.value, hi.value, np.nonzero(keep & ~inside)[0][:5])
            kept_rows += int(inside.sum()); needed_rows += int(keep.sum()); checked += 1
    assert checked == len(poses) * 40 and needed_rows > 20000
    assert kept_rows < 1.6 * needed_rows + 4 * checked          # and the interval is tight: a few rows of slack per column


def raycast_all(be):
    """The volume of integrate_all seen again from every camera: CreateExpectedDepths + FindSurface, the hit maps of all cameras."""
    W, H = 160, 120
    s = be.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, capi.default_params(voxelSize=0.008, mu=0.04), denseSize=(128, 128, 128), denseOffset=(-64, -64, 100))
    s.reco.ResetScene()
    rs = s.vis.CreateRenderState((W, H))
    rng = np.random.default_rng(7)
    intr = (145.0, 145.0, 80.0, 60.0)
    for ang, pos in CAMERAS:
        depth = ((5.0 if pos[2] < -1 else 1.0) + 0.5 * rng.random((H, W))).astype(F)
        depth[::7, ::5] = 0.0
        s.reco.IntegrateIntoScene(capi.View(be.to_backend(depth), W, H, M_d=pose(rotation(*ang), pos), intr_d=intr), rs)
    hits = []
    for ang, pos in CAMERAS:
        M = pose(rotation(*ang), pos)
        s.vis.CreateExpectedDepths(M, intr, rs)
        s.vis.FindSurface(M, intr, rs)
        hits.append(s.download(capi.BUF_RAYCAST_RESULT, rs).copy())
    return hits


@pytest.mark.gpu
def test_dense_ray_cast_from_cameras_in_general_position(hip, oracle):
    """Rays leave the dense volume through every face, start inside, beside and behind it: the wave-level shortcut for rays that
    have left the volume (raycast_device.h) must reproduce the reference's remaining steps -- the end POSITION of a ray that finds
    nothing is part of the result -- bit for bit."""
    want = raycast_all(oracle)
    got = raycast_all(hip)
    found = 0
    for i, (a, b) in enumerate(zip(want, got)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "camera %d: %d components differ" % (i, int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))))
        found += int(np.count_nonzero(a.reshape(-1, 4)[:, 3] > 0))
    assert found > 20000          # the cameras really hit the surface, and most rays leave without a hit


# ---- free-space / shadow classification of 4-voxel groups against the depth tiles (integrate.hip, classify_group) -------------------
def smooth_scene_volumes(be, stop_at_max, mode, frames=7, check=None, no_strips=0):
    """The sphere + wall scene fused into a 128^3 dense volume around the sphere (smooth depth: most groups lie in observed free
    space or in the sphere's shadow), from the parity trajectory and two rotated cameras; maxW 4, so weights saturate on the way."""
    W, H = 320, 240
    be.check(be.fn["debug_set"](16, mode), "debug_set")
    be.check(be.fn["debug_set"](17, no_strips), "debug_set")
    try:
        prm = capi.default_params(voxelSize=0.008, mu=0.04, maxW=4, stopIntegratingAtMaxW=stop_at_max)
        s = be.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, prm, denseSize=(128, 128, 128), denseOffset=(-64, -64, 100))
        s.reco.ResetScene()
        rs = s.vis.CreateRenderState((W, H))
        intr = T.synth.intrinsics_for(W, H)
        out = []
        for k in range(frames):
            t = T.synth.parity_position(3 * k)
            depth = T.synth.depth_frame(W, H, t, intr)
            if k == 2:
                depth[100:140, 50:90] = -1.0           # a hole: invalid pixels inside a smooth region
            if k == 4:
                depth[5, 7] = np.nan                   # a NaN pixel: its tiles are never classified
            M = T.synth.pose_matrix(t) if k < 5 else pose(rotation(0.05 * k, -0.04 * k, 0.3), (0.01 * k, 0.0, 0.0))
            s.reco.IntegrateIntoScene(capi.View(be.to_backend(depth), W, H, M_d=M, intr_d=intr), rs)
            if check is not None:
                c = (C.c_int32 * 4)()
                be.check(be.fn["debug_dense_classify_check"](c, 1), "classify_check")
                check.append(list(c))
            out.append(s.download(capi.BUF_VOXEL_BLOCKS).copy())
        return out
    finally:
        be.check(be.fn["debug_set"](16, 0), "debug_set")
        be.check(be.fn["debug_set"](17, 0), "debug_set")


@pytest.mark.gpu
@pytest.mark.parametrize("stop_at_max", [False, True])
def test_classified_dense_integration_equals_the_exact_path(hip, oracle, stop_at_max):
    want = smooth_scene_volumes(oracle, stop_at_max, 0)
    # strips (the default launch shape): classified / not; four groups per lane: classified before the fetch / not / after the fetch
    for mode, no_strips in ((0, 0), (1, 0), (0, 1), (1, 1), (2, 1)):
        got = smooth_scene_volumes(hip, stop_at_max, mode, no_strips=no_strips)
        for k, (a, b) in enumerate(zip(want, got)):
            assert np.array_equal(a, b), "mode %d%s frame %d: %d voxels differ" % (mode, " (no strips)" if no_strips else "", k, int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))))
    v = want[-1].view(np.uint32).reshape(-1)
    assert np.count_nonzero((v & 0xffff) != 32767) > 100000 and np.count_nonzero(((v >> 16) & 0xff) == 4) > 500000      # a surface, and saturated weights


@pytest.mark.gpu
def test_every_classified_group_agrees_with_its_exact_per_voxel_outcome(hip, oracle):
    """Check mode: the kernel classifies every group, runs the exact path on it anyway and counts the voxels whose exact outcome
    contradicts the class (a free-space voxel not updated with an observation of exactly 1, a shadow voxel touched) or whose
    shortcut result differs from the exact words."""
    c = (C.c_int32 * 4)()
    hip.check(hip.fn["debug_dense_classify_check"](c, 1), "classify_check")
    checks = []
    got = smooth_scene_volumes(hip, False, 3, check=checks)
    want = smooth_scene_volumes(oracle, False, 0)
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    for k, (free, shadow, mixed, bad) in enumerate(checks):
        assert bad == 0, (k, checks)
        assert free > 20000 and shadow > 2000 and mixed > 0, (k, checks)
    # the noisy cameras of the cull test: hardly any free group (every tile holds an invalid pixel), shadow groups, no violation
    hip.check(hip.fn["debug_dense_classify_check"](c, 1), "classify_check")
    integrate_all(hip, values={16: 3})
    hip.check(hip.fn["debug_dense_classify_check"](c, 1), "classify_check")
    assert c[3] == 0 and c[1] > 1000, list(c)


@pytest.mark.gpu
def test_strip_kernel_equals_the_exact_path_over_random_volumes_and_cameras(hip):
    """A slice of tools/dense_classify_sweep.py (random volume sizes 64 / 96 / 128 -- rows shorter and longer than a strip, not powers
    of two --, voxel sizes, band widths, intrinsics, 6-DoF poses, holes / NaN / noise): strip kernel = unclassified exact path bit
    for bit, and the check mode counts no disagreement.  400 seeds of the same sweep ran clean on the round's final kernels."""
    import importlib, sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    argv, sys.argv = sys.argv, [sys.argv[0]]
    try:
        sweep = importlib.import_module("dense_classify_sweep")
    finally:
        sys.argv = argv
    try:
        classified = 0
        for seed in range(12):
            c = sweep.case(seed)
            exact, _ = sweep.run(c, 1, 1)
            strips, _ = sweep.run(c, 0, 0)
            _, checks = sweep.run(c, 3, 1)
            for k, (a, b) in enumerate(zip(exact, strips)):
                assert np.array_equal(a, b), (seed, k, int(np.count_nonzero(a != b)))
            assert all(ch[3] == 0 for ch in checks), (seed, checks)
            classified += sum(ch[0] + ch[1] for ch in checks)
        assert classified > 100000
    finally:
        hip.check(hip.fn["debug_set"](16, 0), "debug_set"); hip.check(hip.fn["debug_set"](17, 0), "debug_set")
