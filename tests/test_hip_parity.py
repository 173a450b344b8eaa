"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on identical inputs.

Bar: hash table, free lists, counters, visible set, weights bit-exact; sdf, range
image, raycast result, ICP maps bit-exact as well because both sides are built without FP
contraction (tolerances in itm_testlib.compare_results document the fallback bar).
"""
import numpy as np
import pytest

import itm_testlib as T
from itm_testlib import Scenario
from infinitam_amd import capi

pytestmark = pytest.mark.gpu

SCENARIOS = [
    Scenario(name="micro_hash_s", w=160, h=120, voxelSize=0.01, frames=3),
    Scenario(name="hash_s_5mm", frames=3),
    Scenario(name="hash_s_4mm", voxelSize=0.004, frames=3),
    Scenario(name="hash_s_4mm_bench_traj", voxelSize=0.004, frames=4, trajectory="bench"),
    Scenario(name="hash_s_yaw", voxelSize=0.005, frames=3, trajectory="yaw"),
    Scenario(name="hash_f", voxelType=T.VOXEL_F, frames=2),
    Scenario(name="hash_s_rgb", voxelType=T.VOXEL_S_RGB, frames=2, colour=True),
    Scenario(name="hash_f_rgb", voxelType=T.VOXEL_F_RGB, frames=2, colour=True),
    Scenario(name="hash_s_noise", frames=2, noise_seed=12345, w=160, h=120, voxelSize=0.01),
    Scenario(name="hash_small_pool", frames=3, localBlockNum=4096),          # pool exhaustion
    Scenario(name="hash_tiny_table", frames=3, bucketNum=0x1000, excessNum=0x400, w=320, h=240, voxelSize=0.01),  # many collisions + excess exhaustion
    Scenario(name="dense_s_128", indexType=T.INDEX_DENSE, denseSize=(128, 128, 128), denseOffset=(-64, -64, 100),
             voxelSize=0.01, frames=2, w=320, h=240),
    Scenario(name="dense_f_rgb_64", indexType=T.INDEX_DENSE, voxelType=T.VOXEL_F_RGB, denseSize=(64, 64, 64),
             denseOffset=(-32, -32, 118), voxelSize=0.01, frames=2, w=160, h=120, colour=True),
    Scenario(name="dense_s_stopmax", indexType=T.INDEX_DENSE, denseSize=(64, 64, 64), denseOffset=(-32, -32, 118),
             voxelSize=0.01, frames=4, w=160, h=120, maxW=2, stopIntegratingAtMaxW=True),
    Scenario(name="hash_render_block_cap", w=320, h=240, voxelSize=0.01, frames=3, maxRenderingBlocks=700),
]


# BASELINE.json configurations at full size (oracle: seconds per frame)
FULL_SIZE = [
    Scenario(name="config2_hash_4mm_pool40000", voxelSize=0.004, frames=3, trajectory="bench", localBlockNum=0x40000),
    Scenario(name="config5_1280x960_f_rgb_2mm", w=1280, h=960, voxelType=T.VOXEL_F_RGB, colour=True, voxelSize=0.002,
             frames=2, localBlockNum=0x40000),
]


@pytest.mark.parametrize("sc", FULL_SIZE, ids=lambda s: s.name)
def test_baseline_configs_full_size(hip, oracle, sc):
    a = T.run_scenario(hip, sc, fused=True)
    b = T.run_scenario(oracle, sc)
    T.compare_results(a, b, sc)
    assert a.counters[-1]["statusFlags"] == 0


@pytest.mark.parametrize("sc", SCENARIOS, ids=lambda s: s.name)
def test_engine_calls_match_oracle(hip, oracle, sc):
    a = T.run_scenario(hip, sc)
    b = T.run_scenario(oracle, sc)
    T.compare_results(a, b, sc)


@pytest.mark.parametrize("sc", [SCENARIOS[0], SCENARIOS[2], SCENARIOS[7], SCENARIOS[-1]], ids=lambda s: s.name)
def test_fused_process_frame_matches_oracle(hip, oracle, sc):
    """itm_process_frame: projection inside the integration launch, range reduction inside the ray-cast launch (the last
    scenario reaches the rendering-block cap, so every ray-cast workgroup replays the accept / skip sequence itself)."""
    a = T.run_scenario(hip, sc, fused=True)
    b = T.run_scenario(oracle, sc)
    T.compare_results(a, b, sc, what=sc.name + "/fused")


@pytest.mark.parametrize("sc", [SCENARIOS[2], SCENARIOS[-1]], ids=lambda s: s.name)
def test_fused_frame_with_separate_range_reduction(hip, oracle, sc):
    """ITM_DEBUG_NO_FUSED_RANGE_REDUCE (6): the reduction of the partial range images as its own launch."""
    hip.check(hip.fn["debug_set"](6, 1), "debug_set")
    try:
        a = T.run_scenario(hip, sc, fused=True)
    finally:
        hip.check(hip.fn["debug_set"](6, 0), "debug_set")
    T.compare_results(a, T.run_scenario(oracle, sc), sc, what=sc.name + "/fused, separate reduce")


def test_dense_scene_expected_depths_written_once_or_every_frame(hip, oracle):
    """Plain voxel array: CreateExpectedDepths writes the constant (0.2, 3.0) image; the library writes it on the first frame and
    leaves it alone afterwards (default) or rewrites it every frame like the reference (ITM_DEBUG_DENSE_RANGE_REFILL, 15); an
    upload in between makes the next frame write it again.  Same results either way, equal to the oracle."""
    sc = next(s for s in SCENARIOS if s.indexType == T.INDEX_DENSE)
    want = T.run_scenario(oracle, sc)
    T.compare_results(T.run_scenario(hip, sc, fused=True), want, sc, what="dense, range written once")
    hip.check(hip.fn["debug_set"](15, 1), "debug_set")
    try:
        a = T.run_scenario(hip, sc, fused=True)
    finally:
        hip.check(hip.fn["debug_set"](15, 0), "debug_set")
    T.compare_results(a, want, sc, what="dense, range rewritten every frame")
    # a foreign range image uploaded between two frames must not survive the next frame
    ses = T.Session(hip, sc)
    ses.frame(0, fused=True)
    good = ses.scene.download(T.capi.BUF_RANGE_IMAGE, ses.rs)
    ses.scene.upload(T.capi.BUF_RANGE_IMAGE, np.full_like(good, 7.5), ses.rs)
    ses.frame(1, fused=True)
    assert np.array_equal(ses.scene.download(T.capi.BUF_RANGE_IMAGE, ses.rs), good)
    ses.close()


def test_large_image_projection_beside_and_after_the_integration(hip, oracle):
    """1280x960: the sub-sampled range image is too large for the fused launch, so itm_process_frame projects the visible blocks on
    the render state's own stream beside the integration (default) -- or after it on the frame's stream (ITM_DEBUG_NO_SIDE_PROJECTION,
    14).  Both equal the oracle, frame after frame (5 frames: the side stream and its events are reused)."""
    sc = Scenario(name="large_image_side_projection", w=1280, h=960, voxelSize=0.004, frames=5, trajectory="bench", localBlockNum=0x20000)
    want = T.run_scenario(oracle, sc)
    T.compare_results(T.run_scenario(hip, sc, fused=True), want, sc, what="projection beside the integration")
    hip.check(hip.fn["debug_set"](14, 1), "debug_set")
    try:
        a = T.run_scenario(hip, sc, fused=True)
    finally:
        hip.check(hip.fn["debug_set"](14, 0), "debug_set")
    T.compare_results(a, want, sc, what="projection after the integration")


@pytest.mark.parametrize("sc", [SCENARIOS[0], SCENARIOS[-1]], ids=lambda s: s.name)
def test_range_image_global_atomic_path(hip, oracle, sc):
    """The fallback used when the range image does not fit LDS (and its cap replay)."""
    hip.check(hip.fn["debug_set"](1, 1), "debug_set")
    try:
        a = T.run_scenario(hip, sc)
    finally:
        hip.check(hip.fn["debug_set"](1, 0), "debug_set")
    b = T.run_scenario(oracle, sc)
    T.compare_results(a, b, sc, what=sc.name + "/global-atomics")


def test_div_by_32767_is_ieee(hip):
    """SDF_valueToFloat of the short voxels is computed with a 3-instruction reciprocal sequence; it must
    equal the IEEE division for every value the path can produce."""
    rng = np.random.default_rng(3)
    xs = np.concatenate([
        np.arange(-32768, 32768, dtype=np.float32),                                   # every raw short
        rng.uniform(-40000, 40000, 1 << 22).astype(np.float32),                       # trilinear blends
        (rng.uniform(-1, 1, 1 << 20) * np.float32(2.0) ** rng.integers(-60, 20, 1 << 20)).astype(np.float32),
        # -0.0 and +-inf are excluded: they cannot occur (raw values are shorts or convex blends of shorts) and
        # the sequence returns +0.0 / NaN for them where the division returns -0.0 / +-inf
        np.array([0.0, 1e-30, -1e-30, 32767.0, -32767.0, 3.4e38], np.float32),
    ])
    src = hip.to_backend(xs)
    dst = T.DevBuffer(hip, xs.nbytes, np.float32, xs.shape)
    hip.check(hip.fn["debug_div32767"](src.ptr, dst.ptr, len(xs), None), "debug_div32767")
    got = dst.numpy()
    want = xs / np.float32(32767.0)
    finite = np.isfinite(want)
    assert np.array_equal(got[finite].view(np.uint32), want[finite].view(np.uint32))


def test_fused_and_separate_calls_interleave(hip, oracle):
    """itm_process_frame and the four separate engine calls can be mixed on one scene."""
    sc = Scenario(name="mix", voxelSize=0.005, frames=6, trajectory="bench")
    ses = T.Session(hip, sc)
    ref = T.Session(oracle, sc)
    for k in range(6):
        ses.frame(k, fused=(k % 2 == 0))
        ref.frame(k, fused=False)
    x, y = ses.snapshot(), ref.snapshot()
    x.counters = [ses.scene.counters(ses.rs)]
    y.counters = [ref.scene.counters(ref.rs)]
    T.compare_results(x, y, sc, what="interleaved")


@pytest.mark.parametrize("sc", [SCENARIOS[3], SCENARIOS[10]], ids=lambda s: s.name)
def test_two_pass_visible_list_path(hip, oracle, sc):
    """AllocateSceneFromDepth builds the visible list in one launch (counts handed between workgroups as 8-byte granules);
    ITM_DEBUG_TWO_PASS_VISIBLE_LIST (7) selects the count + compaction launches that FindVisibleBlocks still uses."""
    hip.check(hip.fn["debug_set"](7, 1), "debug_set")
    try:
        a = T.run_scenario(hip, sc, fused=True)
    finally:
        hip.check(hip.fn["debug_set"](7, 0), "debug_set")
    T.compare_results(a, T.run_scenario(oracle, sc), sc, what=sc.name + "/two-pass visible list")
    assert a.counters[-1]["statusFlags"] == 0


EXCESS_HEAVY = Scenario(name="hash_tiny_table_moving", frames=5, bucketNum=0x1000, excessNum=0x1000, w=320, h=240, voxelSize=0.01, trajectory="yaw")


@pytest.mark.parametrize("sc", [SCENARIOS[3], SCENARIOS[10], EXCESS_HEAVY], ids=lambda s: s.name)
def test_sweep_inside_the_visible_list_launch_and_as_its_own_launch(hip, oracle, sc):
    """The allocation sweep normally rides in the visible-list launch (excess allocations are handed to the workgroups of the
    excess region through per-chunk stamps); ITM_DEBUG_SEPARATE_SWEEP (13) runs it as its own launch.  Tables with 4 096
    buckets make most allocations excess allocations, over several frames of a turning camera."""
    b = T.run_scenario(oracle, sc)
    a = T.run_scenario(hip, sc, fused=True)
    T.compare_results(a, b, sc, what=sc.name + "/fused sweep")
    assert a.counters[-1]["statusFlags"] & 2 == 0
    hip.check(hip.fn["debug_set"](13, 1), "debug_set")
    try:
        a = T.run_scenario(hip, sc, fused=True)
    finally:
        hip.check(hip.fn["debug_set"](13, 0), "debug_set")
    T.compare_results(a, b, sc, what=sc.name + "/separate sweep")


@pytest.mark.parametrize("sc", [Scenario(name="turning_away_tiny_table", frames=7, bucketNum=0x1000, excessNum=0x2000, w=320, h=240, voxelSize=0.01, trajectory="yaw", yaw_rate=0.12),
                                Scenario(name="turning_away_one_excess_chunk", frames=6, bucketNum=0x800, excessNum=0x800, w=320, h=240, voxelSize=0.01, trajectory="yaw", yaw_rate=-0.15),
                                Scenario(name="turning_away_ragged_excess", frames=6, bucketNum=0x1000, excessNum=0x1238, w=320, h=240, voxelSize=0.01, trajectory="yaw", yaw_rate=0.15)],
                         ids=lambda s: s.name)
def test_excess_region_re_tests_are_shared_among_its_workgroups(hip, oracle, sc):
    """Visible blocks that no pixel requests any more are re-tested against the frustum; the ones that live in the excess region lie
    side by side (the excess list is handed out from one end), so the visible-list launch deals them out -- 64 slots to a wave, a block
    to a lane -- to all workgroups of the region, and the owners of the slots pick the verdicts up as tagged granules (alloc.hip).  A
    camera that turns 7-9 degrees per frame over tables small enough that most blocks are excess blocks: hundreds of them leave the
    frustum or stay unrequested every frame.  Counters every frame, full state at the end, list launch with and without the sweep inside."""
    b = T.run_scenario(oracle, sc)
    vis = [x["noVisibleEntries"] for x in b.counters]
    free = [x["lastFreeBlockId"] for x in b.counters]
    # blocks that left the list = blocks allocated (all of them visible) minus the list's growth, frame by frame
    dropped = sum((free[k - 1] - free[k]) - (vis[k] - vis[k - 1]) for k in range(1, len(vis)))
    assert dropped > 300, (vis, free)
    ex = b.hash["ptr"][sc.bucketNum:] >= 0
    assert ex.sum() > 300, ex.sum()                                             # and the excess region is where they live
    a = T.run_scenario(hip, sc, fused=True)
    T.compare_results(a, b, sc, what=sc.name)
    assert a.counters[-1]["statusFlags"] == 0
    hip.check(hip.fn["debug_set"](13, 1), "debug_set")                          # the sweep as its own launch: the owners walk their own slots
    try:
        a = T.run_scenario(hip, sc, fused=True)
    finally:
        hip.check(hip.fn["debug_set"](13, 0), "debug_set")
    T.compare_results(a, b, sc, what=sc.name + "/separate sweep")


def test_explicit_mark_previous_path(hip, oracle):
    """The allocation normally folds "mark last frame's list as type 3" into the type encoding; the explicit
    launch (used after FindVisibleBlocks / uploads on the same render state) must give the same scene."""
    sc = Scenario(name="explicit_mark", voxelSize=0.005, frames=4, trajectory="bench")
    hip.check(hip.fn["debug_set"](2, 1), "debug_set")
    try:
        a = T.run_scenario(hip, sc)
    finally:
        hip.check(hip.fn["debug_set"](2, 0), "debug_set")
    T.compare_results(a, T.run_scenario(oracle, sc), sc, what="explicit mark")


def test_find_visible_blocks_then_allocate_on_same_render_state(hip, oracle):
    """FindVisibleBlocks rewrites the list but not the types (reference behaviour); the next allocation on
    that render state must still match the oracle."""
    sc = Scenario(name="fvb_then_alloc", w=160, h=120, voxelSize=0.01, frames=2)
    res = []
    for be in (hip, oracle):
        ses = T.Session(be, sc)
        ses.frame(0)
        v1 = ses.view(1)
        ses.scene.vis.FindVisibleBlocks(T.synth.pose_matrix((0.2, 0.0, 0.0)), sc.intr(), ses.rs)
        ses.scene.reco.AllocateSceneFromDepth(v1, ses.rs)
        ses.scene.reco.IntegrateIntoScene(v1, ses.rs)
        ses.scene.vis.CreateExpectedDepths(v1.M_d, v1.intr_d, ses.rs)
        ses.scene.vis.CreateICPMaps(v1, ses.rs, ses.points, ses.normals)
        snap = ses.snapshot()
        snap.counters = [ses.scene.counters(ses.rs)]
        res.append(snap)
        ses.close()
    T.compare_results(res[0], res[1], sc, what="find-visible then allocate")


def _divide(hip, mode, a, b, r=None):
    da, db = hip.to_backend(a), hip.to_backend(b)
    dr = hip.to_backend(r) if r is not None else None
    out = T.DevBuffer(hip, a.nbytes, np.float32, a.shape)
    hip.check(hip.fn["debug_divide"](mode, da.ptr, db.ptr, dr.ptr if dr else None, out.ptr, len(a), None), "debug_divide")
    return out.numpy()


def test_fast_divisions_are_ieee(hip):
    """The integration kernel replaces the IEEE division macro by shorter FMA chains; they must return the
    correctly rounded quotient (== numpy float32 division) over the operand ranges the kernel guards."""
    rng = np.random.default_rng(11)
    n = 1 << 22
    # (1) projection u = fx*x/z: z in [1e-4, 1e4], numerators up to +-1e6, plus log-uniform magnitudes
    z = (10.0 ** rng.uniform(-4, 4, n)).astype(np.float32)
    num = (rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-6, 6, n)).astype(np.float32)
    got = _divide(hip, 1, num, z)
    assert np.array_equal(got.view(np.uint32), (num / z).view(np.uint32))
    # typical values: pixels*depth over depth
    z2 = rng.uniform(0.3, 4.0, n).astype(np.float32)
    num2 = (rng.uniform(-700, 700, n).astype(np.float32) * z2).astype(np.float32)
    assert np.array_equal(_divide(hip, 1, num2, z2).view(np.uint32), (num2 / z2).view(np.uint32))
    # (2)+(3) weights: every integer divisor 1..256, its refined reciprocal and quotients of running sums
    w = np.arange(1, 257, dtype=np.float32)
    assert np.array_equal(_divide(hip, 3, w, w).view(np.uint32), (np.float32(1.0) / w).view(np.uint32))
    ww = rng.integers(1, 257, n).astype(np.float32)
    acc = (rng.uniform(-1.0, 1.0, n) * ww).astype(np.float32)
    assert np.array_equal(_divide(hip, 2, acc, ww).view(np.uint32), (acc / ww).view(np.uint32))
    # (4) eta / mu with the host-rounded reciprocal, several band widths
    for mu in (0.02, 0.01, 0.005, 0.04, 0.1, 0.03125):
        mu32 = np.float32(mu)
        eta = rng.uniform(-3 * mu, 4.0, n).astype(np.float32)
        b = np.full(n, mu32, np.float32)
        r = np.full(n, np.float32(1.0) / mu32, np.float32)
        assert np.array_equal(_divide(hip, 4, eta, b, r).view(np.uint32), (eta / b).view(np.uint32)), mu
    # (5) colour path: x / 255 with the compile-time reciprocal -- every stored colour, and bilinear blends of colours
    d255 = np.full(n, np.float32(255.0), np.float32)
    r255 = np.full(n, np.float32(1.0) / np.float32(255.0), np.float32)
    xs = np.concatenate([np.arange(0, 256, dtype=np.float32), rng.uniform(0, 255.0, n - 256).astype(np.float32)])
    assert np.array_equal(_divide(hip, 4, xs, d255, r255).view(np.uint32), (xs / d255).view(np.uint32))
    # (6) colour running average c / newW: c = old*oldW + new in [0, 256], integer weights 1..256
    c = (rng.uniform(0, 1.0, n) * ww + rng.uniform(0, 1.0, n)).astype(np.float32)
    assert np.array_equal(_divide(hip, 2, c, ww).view(np.uint32), (c / ww).view(np.uint32))


# ---- block directory (dense mirror of the hash table, infinitam_amd/csrc/itm_types.h) ------------------------------------
DIRECTORY_CASES = [
    SCENARIOS[0],
    Scenario(name="hash_s_4mm_bench_traj6", voxelSize=0.004, frames=6, trajectory="bench"),
    # 1 mm voxels: a block is 8 mm, the directory covers +-2.048 m, so the sphere (1.0-1.5 m) lies inside it and the wall
    # (2.5 m) outside: rays start on directory lookups and finish on the table walk
    Scenario(name="hash_s_1mm_wall_outside_directory", w=160, h=120, voxelSize=0.001, mu=0.004, frames=2),
    Scenario(name="hash_s_rgb_yaw", voxelType=T.VOXEL_S_RGB, colour=True, voxelSize=0.005, frames=3, trajectory="yaw", w=320, h=240),
]


@pytest.mark.parametrize("sc", DIRECTORY_CASES, ids=lambda s: s.name)
def test_directory_and_table_walk_agree_with_oracle(hip, oracle, sc):
    """The ray caster normally looks blocks up in the block directory; ITM_DEBUG_NO_DIRECTORY (5) selects the table walk.
    Both must reproduce the oracle bit for bit, including where rays leave the cube the directory covers."""
    b = T.run_scenario(oracle, sc)
    T.compare_results(T.run_scenario(hip, sc, fused=True), b, sc, what=sc.name + "/directory")
    hip.check(hip.fn["debug_set"](5, 1), "debug_set")
    try:
        a = T.run_scenario(hip, sc, fused=True)
    finally:
        hip.check(hip.fn["debug_set"](5, 0), "debug_set")
    T.compare_results(a, b, sc, what=sc.name + "/table walk")
    # the sdf mirror (voxels addressed by position, short voxel types) is what the default path read; ITM_DEBUG_NO_SDF_MIRROR (12)
    # sends the same rays through the directory and the voxel pool
    hip.check(hip.fn["debug_set"](12, 1), "debug_set")
    try:
        a = T.run_scenario(hip, sc, fused=True)
    finally:
        hip.check(hip.fn["debug_set"](12, 0), "debug_set")
    T.compare_results(a, b, sc, what=sc.name + "/directory without the sdf mirror")


def test_ray_cast_sees_blocks_that_were_allocated_but_never_integrated(hip, oracle):
    """Engine calls in an order the main loop never uses: allocate for a new view, then ray-cast WITHOUT integrating.  The new
    blocks hold the initial voxel value; the ray caster must find them (a block that exists reads 1 and makes the ray step
    mu / voxelSize voxels, a missing block makes it step 8) -- on the sdf mirror too, which the allocation initialises."""
    sc = Scenario(name="alloc_only", w=320, h=240, voxelSize=0.005, frames=2)
    out = []
    for be in (hip, oracle):
        ses = T.Session(be, sc)
        ses.frame(0)
        v = ses.view(1)
        M = T.synth.pose_matrix_yaw((0.3, 0.1, 0.0), 0.25)           # a view that sees new surface
        v2 = T.View(v.depth, sc.w, sc.h, M_d=M, intr_d=sc.intr())
        ses.scene.reco.AllocateSceneFromDepth(v2, ses.rs)
        ses.scene.vis.CreateExpectedDepths(M, sc.intr(), ses.rs)
        ses.scene.vis.FindSurface(M, sc.intr(), ses.rs)
        out.append((ses.scene.download(T.BUF_RAYCAST_RESULT, ses.rs).copy(), ses.scene.counters(ses.rs)["lastFreeBlockId"]))
        ses.close()
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))


@pytest.mark.parametrize("sc", [DIRECTORY_CASES[1], DIRECTORY_CASES[2]], ids=lambda s: s.name)
def test_single_pass_ray_cast_path(hip, oracle, sc):
    """Ray casting normally runs in two passes (rays crossing empty space are parked and finished by a second launch);
    ITM_DEBUG_SINGLE_PASS_RAYCAST (8) casts every ray start to finish in one launch.  Same rays, same results."""
    hip.check(hip.fn["debug_set"](8, 1), "debug_set")
    try:
        a = T.run_scenario(hip, sc, fused=True)
    finally:
        hip.check(hip.fn["debug_set"](8, 0), "debug_set")
    T.compare_results(a, T.run_scenario(oracle, sc), sc, what=sc.name + "/single-pass ray cast")


def test_directory_is_rebuilt_after_table_upload(hip, oracle):
    """A scene whose hash table, voxels and free lists were uploaded (checkpoint restore, itm_upload) must ray-cast like
    the scene that produced them: the upload rebuilds the directory from the table."""
    sc = Scenario(name="dir_rebuild", w=320, h=240, voxelSize=0.005, frames=3, trajectory="bench")
    src = T.Session(hip, sc)
    for k in range(sc.frames):
        src.frame(k, fused=True)
    dst = T.Session(hip, sc)
    for which in (T.BUF_HASH_ENTRIES, T.BUF_EXCESS_LIST, T.BUF_ALLOCATION_LIST, T.BUF_VOXEL_BLOCKS):
        dst.scene.upload(which, src.scene.download(which))
    for which in (T.BUF_VISIBLE_IDS, T.BUF_VISIBLE_TYPE):
        dst.scene.upload(which, src.scene.download(which, src.rs), dst.rs)
    c = src.scene.counters(src.rs)
    dst.scene.set_counters(dst.rs, c["lastFreeBlockId"], c["lastFreeExcessListId"], c["noVisibleEntries"])
    ref = T.Session(oracle, sc)
    for k in range(sc.frames):
        ref.frame(k)
    for ses in (dst, ref):          # a fourth frame on the restored scene and on the oracle's uninterrupted run
        ses.frame(3, fused=False)
    x, y = dst.snapshot(), ref.snapshot()
    T.compare_results(x, y, sc, what="restored scene, next frame")
    # free-view ray cast from another pose through the rebuilt directory
    M = T.synth.pose_matrix_yaw((0.05, -0.02, 0.1), 0.1)
    for ses in (dst, ref):
        ses.scene.vis.FindVisibleBlocks(M, sc.intr(), ses.rs)
        ses.scene.vis.CreateExpectedDepths(M, sc.intr(), ses.rs)
        ses.scene.vis.FindSurface(M, sc.intr(), ses.rs)
    ra, rb = dst.scene.download(T.BUF_RAYCAST_RESULT, dst.rs), ref.scene.download(T.BUF_RAYCAST_RESULT, ref.rs)
    assert np.array_equal(ra[..., 3], rb[..., 3])
    hit = ra[..., 3] > 0
    assert hit.sum() > 1000 and np.array_equal(ra[hit], rb[hit])


@pytest.mark.parametrize("voxel,colour", [(capi.VOXEL_S, False), (capi.VOXEL_F, False), (capi.VOXEL_S_RGB, True), (capi.VOXEL_F_RGB, True)])
def test_hash_integration_with_saturating_weights_for_every_voxel_type(hip, oracle, voxel, colour):
    """The hash integration (slices of a block per wave, one voxel per lane) against the oracle for every voxel type, fused with the
    projection (four calls back to back) and as its own launch, with weights that saturate (maxW 3), with and without
    stopIntegratingAtMaxW, on noisy depth.  (Round 4's block-per-wave kernel with 16 bytes per lane shared this test; it was slower
    and is gone: profiles/r4_integrate_notes.md.)"""
    for stop in (False, True):
        sc = Scenario(name="sat_%d_%d" % (voxel, stop), voxelType=voxel, colour=colour, w=320, h=240, voxelSize=0.006, frames=5, maxW=3,
                      stopIntegratingAtMaxW=stop, trajectory="bench", noise_seed=11)
        b = T.run_scenario(oracle, sc)
        for fused in ("four", False):
            a = T.run_scenario(hip, sc, fused=fused)
            T.compare_results(a, b, sc, what="%s/%s" % (sc.name, fused))


@pytest.mark.parametrize("sc", [Scenario(name="ragged_yaw", w=333, h=211, voxelSize=0.006, frames=5, trajectory="yaw"),
                                Scenario(name="ragged_dense", indexType=capi.INDEX_DENSE, denseSize=(128, 128, 128), denseOffset=(-64, -64, 100), voxelSize=0.01, w=325, h=243, frames=4)],
                         ids=lambda s: s.name)
def test_ragged_image_sizes_over_several_frames(hip, oracle, sc):
    """Image sizes that are no multiple of the 16 x 16 ray-cast tile or the 8 x 8 range cell, through the four calls."""
    b = T.run_scenario(oracle, sc)
    a = T.run_scenario(hip, sc, fused="four")
    T.compare_results(a, b, sc, what=sc.name)
