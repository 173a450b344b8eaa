"""Near bits (infinitam_amd/csrc/itm_types.h): one byte per cell of the mirror's cube, bit k set iff a block was allocated within
Chebyshev distance k.  After a read that found no block the ray caster takes the reference's 8-voxel steps
(DeviceAgnostic/ITMVisualisationEngine.h:129-130,139-141) through provably empty space WITHOUT reading: lowest set bit m -> m - 2 steps.

Here: the arithmetic bound the skip rests on (CPU, float32 exactly as the kernel computes), the contents of the bits against the
table (GPU: after frames, a cube move, an upload, a reset), and ray casts with and without the skip against the oracle."""
import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import capi, synth


def test_k_steps_of_eight_voxels_move_the_looked_up_block_by_at_most_k_plus_one():
    """positions q_k = fl(q_{k-1} + fl(8 d)) per axis, |d| a float32 unit vector's component; looked-up voxel = (int)ROUND(q), block =
    voxel >> 3.  The claim used by the kernel: |block(q_k) - block(q_0)| <= k + 1 on every axis, for every k the bits can grant (<= 6)."""
    rng = np.random.default_rng(7)
    n = 400000
    p = (rng.uniform(-1, 1, (n, 3)) * np.float32(2.0) ** rng.integers(0, 19, (n, 1))).astype(np.float32)
    # adversarial starts: just below / above the .5 boundaries next to a block face, where rounding gains a voxel at both ends
    edge = (rng.integers(-30000, 30000, (n // 4, 3)) * 8 + rng.choice([-0.5, 7.5, 7.4999995, -0.50000006], (n // 4, 3))).astype(np.float32)
    p = np.concatenate([p, edge])
    d = rng.normal(size=p.shape).astype(np.float32)
    axis = rng.integers(0, 3, len(p)); pure = rng.random(len(p)) < 0.2
    d[pure] = 0; d[pure, axis[pure]] = rng.choice([-1.0, 1.0], pure.sum())      # axis-aligned rays take the full 8 voxels per step
    nrm = (np.float32(1.0) / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)).astype(np.float32)
    d = (d * nrm[:, None]).astype(np.float32)                                   # dir *= 1 / sqrt(...), as ray_setup does

    def block(q):
        r = np.where(q < 0, q - np.float32(0.5), q + np.float32(0.5)).astype(np.float32)      # ROUND (ORUtils/MathUtils.h:21-23), then (int)
        return np.trunc(r).astype(np.int64) >> 3

    step = (np.float32(8.0) * d).astype(np.float32)
    b0, q = block(p), p.copy()
    for k in range(1, 8):
        q = (q + step).astype(np.float32)
        assert np.abs(block(q) - b0).max() <= k + 1, k


def expected_near_bits(hash_entries, origin, reach=7, side=256):
    """bit k of cell c <=> an entry with ptr >= -1 lies within Chebyshev distance k of c (numpy, dilation per distance)."""
    occ = np.zeros((side + 2 * reach,) * 3, bool)                               # z, y, x with a margin for blocks outside the cube
    e = hash_entries[hash_entries["ptr"] >= -1]
    c = e["pos"].astype(np.int64) - np.asarray(origin, np.int64)
    ok = ((c >= -reach) & (c < side + reach)).all(axis=1)
    c = c[ok] + reach
    occ[c[:, 2], c[:, 1], c[:, 0]] = True
    out = np.zeros((side, side, side), np.uint8)
    cur = occ
    for k in range(reach + 1):
        if k:
            nxt = cur.copy()
            for ax in range(3):                                                 # one more layer: separable max over +-1
                a = nxt.copy()
                sl_lo = [slice(None)] * 3; sl_hi = [slice(None)] * 3
                sl_lo[ax] = slice(1, None); sl_hi[ax] = slice(None, -1)
                a[tuple(sl_lo)] |= nxt[tuple(sl_hi)]
                a[tuple(sl_hi)] |= nxt[tuple(sl_lo)]
                nxt = a
            cur = nxt
        out |= (cur[reach:-reach, reach:-reach, reach:-reach].astype(np.uint8) << k)
    return out


def check_bits(scene, what):
    info = scene.accel_info()
    assert info["near_bits_bytes"] == 256 ** 3 and info["mirror_bytes"] > 0, info
    got = scene.download(capi.BUF_NEAR_BITS).reshape(256, 256, 256)
    want = expected_near_bits(scene.download(capi.BUF_HASH_ENTRIES), info["origin_mirror"])
    bad = np.argwhere(got != want)
    assert len(bad) == 0, "%s: %d cells differ, first (z, y, x) %s: %#x vs %#x" % (what, len(bad), bad[0], got[tuple(bad[0])], want[tuple(bad[0])])
    return int(np.count_nonzero(want))


@pytest.fixture
def near_bits_on(monkeypatch):
    """Scenes only carry near bits when asked to (a measurement feature: the skip lost, profiles/r4_raycast_notes.md)."""
    monkeypatch.setenv("ITM_NEAR_BITS", "1")
    import itm_testlib
    be = itm_testlib.hip_backend()
    s = be.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params())
    have = s.accel_info()["near_bits_bytes"] > 0
    s.close()
    if not have:
        pytest.skip("library built without near bits (the default: -DITM_NEAR_BITS=1 is a measurement build, tools/build_full_variant.sh)")


@pytest.mark.gpu
def test_near_bits_mirror_the_table_through_frames_moves_uploads_and_resets(hip, near_bits_on):
    from test_accel_origin import walk_poses
    poses = walk_poses()
    sc = T.Scenario(name="near_bits", w=160, h=120, voxelSize=0.005, localBlockNum=0x40000, frames=len(poses))
    ses = T.Session(hip, sc)
    depth = [hip.to_backend(sc.depth(k)) for k in range(sc.frames)]
    moves = 0
    for k in range(sc.frames):
        v = capi.View(depth[k], sc.w, sc.h, M_d=poses[k], intr_d=sc.intr())
        ses.scene.process_frame(v, ses.rs, ses.points, ses.normals)
        if k in (0, 1, sc.frames // 2, sc.frames - 1) or ses.scene.accel_info()["moves"] != moves:
            moves = ses.scene.accel_info()["moves"]
            assert check_bits(ses.scene, "frame %d (%d moves)" % (k, moves)) > 1000
    assert moves >= 3
    table = ses.scene.download(capi.BUF_HASH_ENTRIES)
    ses.scene.reco.ResetScene()
    assert not ses.scene.download(capi.BUF_NEAR_BITS).any(), "near bits after ResetScene"
    ses.scene.upload(capi.BUF_HASH_ENTRIES, table)
    assert check_bits(ses.scene, "after an upload of the table") > 1000
    ses.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sc", [T.Scenario(name="skip_bench", voxelSize=0.004, localBlockNum=0x40000, frames=4, trajectory="bench"),
                                T.Scenario(name="skip_yaw_small_voxels", voxelSize=0.002, localBlockNum=0x40000, w=320, h=240, frames=4, trajectory="yaw"),
                                T.Scenario(name="skip_far_origin", voxelSize=0.005, frames=3, trajectory="bench", origin=(20.0, -12.0, 8.0)),
                                T.Scenario(name="skip_s_rgb", voxelType=capi.VOXEL_S_RGB, colour=True, voxelSize=0.008, w=320, h=240, frames=3)],
                         ids=lambda s: s.name)
def test_ray_casts_with_and_without_the_skip_equal_the_oracle(hip, oracle, sc, near_bits_on):
    """(The default build compiles the skip out -- ITM_RAY_NEAR_SKIP=0, it lost -- and then both runs cast the same rays while the
    sweep maintains the bits; with a measurement build, ITM_TEST_LIB=gpurun_variants/lib_near2.so, the first run skips.)"""
    b = T.run_scenario(oracle, sc)
    a = T.run_scenario(hip, sc, fused="four")
    T.compare_results(a, b, sc, what=sc.name + "/near-bit skip")
    hip.check(hip.fn["debug_set"](21, 1), "debug_set")
    try:
        a = T.run_scenario(hip, sc, fused="four")
    finally:
        hip.check(hip.fn["debug_set"](21, 0), "debug_set")
    T.compare_results(a, b, sc, what=sc.name + "/every position read")


@pytest.mark.gpu
def test_free_view_rays_through_empty_space_equal_the_oracle(hip, oracle, near_bits_on):
    """Rays from poses the scene was never fused from cross long stretches without blocks (the skip's best case) and graze allocated
    shells (its worst): FindSurface from a ring of cameras."""
    sc = T.Scenario(name="skip_freeview", voxelSize=0.005, frames=3, trajectory="bench")
    outs = []
    for be in (hip, oracle):
        ses = T.Session(be, sc)
        for k in range(sc.frames):
            ses.frame(k, fused=True)
        free = ses.scene.vis.CreateRenderState((sc.w, sc.h))
        got = []
        for j in range(6):
            M = synth.pose_matrix_yaw((0.4 * np.cos(j), 0.15 * np.sin(2 * j), -0.6 + 0.2 * j), 0.25 * (j - 2.5))
            ses.scene.vis.FindVisibleBlocks(M, sc.intr(), free)
            ses.scene.vis.CreateExpectedDepths(M, sc.intr(), free)
            ses.scene.vis.FindSurface(M, sc.intr(), free)
            got.append(ses.scene.download(capi.BUF_RAYCAST_RESULT, free).copy())
        outs.append(got)
        free.close(); ses.close()
    for j, (x, y) in enumerate(zip(*outs)):
        assert np.array_equal(x[..., 3], y[..., 3]), "camera %d: hit mask" % j
        hit = x[..., 3] > 0
        assert hit.sum() > 1000 and np.array_equal(x[hit], y[hit]), "camera %d: hits" % j
