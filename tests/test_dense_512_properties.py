"""BASELINE config 3 at full size (dense 512^3, 537 MB): too slow for the CPU oracle in a unit test, so it is
checked through size-independent properties: idempotent reset, the touched set equals the oracle's on a
sub-volume computed with the same offsets, weights bounded by maxW, and stopIntegratingAtMaxW freezing."""
import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import capi, synth

pytestmark = pytest.mark.gpu


def test_dense_512_subvolume_matches_oracle(hip, oracle):
    W, H = 640, 480
    prm = capi.default_params(voxelSize=0.004, maxW=2, stopIntegratingAtMaxW=True)
    big = hip.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, prm)                      # 512^3, offset (-256,-256,0)
    big.reco.ResetScene()
    rs = big.vis.CreateRenderState((W, H))
    # the oracle fuses the same frames into a 64-voxel thick slab z in [320, 384) of the same grid
    small = oracle.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, prm, denseSize=(512, 512, 64), denseOffset=(-256, -256, 320))
    small.reco.ResetScene()
    rs_o = small.vis.CreateRenderState((W, H))
    intr = synth.intrinsics_for(W, H)
    for k in range(4):
        t = synth.bench_position(k)
        depth = synth.depth_frame(W, H, t, intr)
        M = synth.pose_matrix(t)
        big.reco.IntegrateIntoScene(capi.View(hip.to_backend(depth), W, H, M_d=M, intr_d=intr), rs)
        small.reco.IntegrateIntoScene(capi.View(oracle.to_backend(depth), W, H, M_d=M, intr_d=intr), rs_o)
    vol = big.download(capi.BUF_VOXEL_BLOCKS).reshape(512, 512, 512)    # [z][y][x]
    slab = small.download(capi.BUF_VOXEL_BLOCKS).reshape(64, 512, 512)
    assert np.array_equal(vol["sdf"][320:384], slab["sdf"])
    assert np.array_equal(vol["w_depth"][320:384], slab["w_depth"])
    assert vol["w_depth"].max() == 2                      # frozen at maxW
    assert (vol["w_depth"] > 0).sum() > 1_000_000
    untouched = vol["w_depth"] == 0
    assert np.all(vol["sdf"][untouched] == 32767)
    # raycast of the dense volume: constant range, every hit lies inside the volume
    pts = capi.DevBuffer(hip, W * H * 16, np.float32, (H, W, 4))
    nrm = capi.DevBuffer(hip, W * H * 16, np.float32, (H, W, 4))
    v = capi.View(hip.to_backend(depth), W, H, M_d=M, intr_d=intr)
    big.vis.CreateExpectedDepths(M, intr, rs)
    big.vis.CreateICPMaps(v, rs, pts, nrm)
    p = pts.numpy()
    ok = p[..., 3] > 0
    assert ok.sum() > 50_000
    assert p[ok][:, 2].min() > 0.2 and p[ok][:, 2].max() < 2.06


def test_dense_512_bench_configuration_slab_matches_oracle(hip, oracle):
    """BASELINE configs[2] as bench.py --config 3 runs it (maxW 100, stopIntegratingAtMaxW, bench trajectory, the fused frame
    call), long enough for the weights to reach maxW and freeze: 104 frames.  The oracle fuses the same frames into the slab
    z in [320, 384) of the same grid (every voxel is updated independently of the others, so slab == region of the full grid)."""
    W, H, FRAMES = 640, 480, 104
    prm = capi.default_params(voxelSize=0.004, maxW=100, stopIntegratingAtMaxW=True)
    big = hip.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, prm)
    big.reco.ResetScene()
    rs = big.vis.CreateRenderState((W, H))
    small = oracle.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, prm, denseSize=(512, 512, 64), denseOffset=(-256, -256, 320))
    small.reco.ResetScene()
    rs_o = small.vis.CreateRenderState((W, H))
    intr = synth.intrinsics_for(W, H)
    pts = capi.DevBuffer(hip, W * H * 16, np.float32, (H, W, 4))
    nrm = capi.DevBuffer(hip, W * H * 16, np.float32, (H, W, 4))
    for k in range(FRAMES):
        t = synth.bench_position(k)
        depth = synth.depth_frame(W, H, t, intr)
        M = synth.pose_matrix(t)
        big.process_frame(capi.View(hip.to_backend(depth), W, H, M_d=M, intr_d=intr), rs, pts, nrm)
        small.reco.IntegrateIntoScene(capi.View(oracle.to_backend(depth), W, H, M_d=M, intr_d=intr), rs_o)
    vol = big.download(capi.BUF_VOXEL_BLOCKS).reshape(512, 512, 512)
    slab = small.download(capi.BUF_VOXEL_BLOCKS).reshape(64, 512, 512)
    assert np.array_equal(vol["sdf"][320:384], slab["sdf"])
    assert np.array_equal(vol["w_depth"][320:384], slab["w_depth"])
    assert slab["w_depth"].max() == 100 and (slab["w_depth"] == 100).sum() > 100_000      # saturated voxels exist and stopped integrating
    assert vol["w_depth"].max() == 100


def test_dense_512_whole_volume_in_the_bench_configuration(hip, oracle):
    """The WHOLE 512^3 volume, not a slab: the strip kernel decides per 4 x 4-voxel patch whether the frame's depth tiles prove it free
    space (near the camera: z 0 .. 250), shadow (behind the surfaces: beyond ~450) or neither -- class boundaries that mostly lie
    outside any one slab.  104 frames of bench.py --config 3 (maxW 100, stopIntegratingAtMaxW, the four engine calls back to back),
    weights saturate and freeze.  Oracle: the OpenMP build over all host cores where there is one (the dense update touches every
    voxel independently of the others: the parallel loop is race-free and equals the sequential one), else four slabs that hold the
    class boundaries, single thread."""
    W, H, FRAMES = 640, 480, 104
    prm = capi.default_params(voxelSize=0.004, maxW=100, stopIntegratingAtMaxW=True)
    omp = T.oracle_omp_backend()
    big = hip.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, prm)
    big.reco.ResetScene()
    rs = big.vis.CreateRenderState((W, H))
    if omp is not None and (T.os.cpu_count() or 1) >= 32:
        parts = [(omp, 0, 512)]
    else:
        parts = [(oracle, 0, 64), (oracle, 192, 64), (oracle, 320, 64), (oracle, 448, 64)]
    refs = []
    for be, z0, nz in parts:
        s = be.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, prm, denseSize=(512, 512, nz), denseOffset=(-256, -256, z0))
        s.reco.ResetScene()
        refs.append((be, s, s.vis.CreateRenderState((W, H)), z0, nz))
    intr = synth.intrinsics_for(W, H)
    pts = capi.DevBuffer(hip, W * H * 16, np.float32, (H, W, 4))
    nrm = capi.DevBuffer(hip, W * H * 16, np.float32, (H, W, 4))
    for k in range(FRAMES):
        t = synth.bench_position(k)
        depth = synth.depth_frame(W, H, t, intr)
        M = synth.pose_matrix(t)
        v = capi.View(hip.to_backend(depth), W, H, M_d=M, intr_d=intr)
        big.reco.AllocateSceneFromDepth(v, rs); big.reco.IntegrateIntoScene(v, rs)
        big.vis.CreateExpectedDepths(M, intr, rs); big.vis.CreateICPMaps(v, rs, pts, nrm)
        for be, s, r, _, _ in refs:
            s.reco.IntegrateIntoScene(capi.View(be.to_backend(depth), W, H, M_d=M, intr_d=intr), r)
    vol = big.download(capi.BUF_VOXEL_BLOCKS).reshape(512, 512, 512)
    for be, s, r, z0, nz in refs:
        ref = s.download(capi.BUF_VOXEL_BLOCKS).reshape(nz, 512, 512)
        for f in ("sdf", "w_depth"):
            same = vol[f][z0:z0 + nz] == ref[f]
            assert same.all(), "slices %d..%d, %s: %d voxels differ, first (z, y, x) %s" % (z0, z0 + nz, f, (~same).sum(), np.argwhere(~same)[0] + [z0, 0, 0])
        r.close(); s.close()
    assert (vol["w_depth"] == 100).sum() > 1_000_000 and (vol["w_depth"] == 0).sum() > 10_000_000


def test_dense_512_every_classified_patch_agrees_with_its_voxels(hip):
    """The classification's check mode (debug key 16 = 3) at full size: every 4-voxel group is classified from the depth tiles, the exact
    per-voxel path runs anyway, and a class that a voxel of the group contradicts is counted.  40 frames of the bench trajectory + 6
    cameras in general position: free and shadow classes both occur by the million, no contradiction."""
    W, H = 640, 480
    prm = capi.default_params(voxelSize=0.004, maxW=100, stopIntegratingAtMaxW=True)
    big = hip.create_scene(capi.VOXEL_S, capi.INDEX_DENSE, prm)
    big.reco.ResetScene()
    rs = big.vis.CreateRenderState((W, H))
    intr = synth.intrinsics_for(W, H)
    out = (capi.C.c_int32 * 4)()
    hip.check(hip.fn["debug_dense_classify_check"](out, 1), "classify_check reset")
    hip.check(hip.fn["debug_set"](16, 3), "debug_set")
    try:
        poses = [(synth.bench_position(k), synth.pose_matrix(synth.bench_position(k))) for k in range(40)]
        poses += [((0.1 * j, -0.05 * j, 0.02 * j), synth.pose_matrix_yaw((0.1 * j, -0.05 * j, 0.02 * j), 0.07 * (j - 3))) for j in range(6)]
        for t, M in poses:
            depth = synth.depth_frame(W, H, t, intr)
            big.reco.IntegrateIntoScene(capi.View(hip.to_backend(depth), W, H, M_d=M, intr_d=intr), rs)
        hip.sync()
        hip.check(hip.fn["debug_dense_classify_check"](out, 0), "classify_check")
    finally:
        hip.check(hip.fn["debug_set"](16, 0), "debug_set")
    free, shadow, mixed, violations = list(out)
    assert violations == 0, list(out)
    assert free > 5_000_000 and shadow > 5_000_000 and mixed > 1_000_000, list(out)


def test_dense_512_raycast_matches_oracle_at_full_size(hip, oracle):
    """BASELINE configs[2] at FULL size through the ray cast as well (castRay over a plain voxel array, reference
    DeviceAgnostic/ITMVisualisationEngine.h:92-158 with readFromSDF_* of ITMRepresentationAccess.h:129-142): frames 0, 17, 34 of
    bench.py --config 3's trajectory through the four engine calls on the whole 512^3 volume, on both sides.  Voxels, the constant
    range image, raycast_result (the position of a ray that found nothing is unspecified in the reference: masked, its w is compared),
    ICP points / normals and the grey image are compared bit for bit -- what bench.py --config 3's parity_check does, inside the suite."""
    sc = T.Scenario(name="dense512_full", w=640, h=480, voxelType=capi.VOXEL_S, indexType=capi.INDEX_DENSE, voxelSize=0.004, mu=0.02,
                    stopIntegratingAtMaxW=True, trajectory="bench", frames=3, frame_stride=17)
    a = T.run_scenario(hip, sc, fused="four")
    b = T.run_scenario(oracle, sc)
    for r in (a, b):
        r.raycast = r.raycast.copy()
        r.raycast[r.raycast[..., 3] <= 0, :3] = 0
    hits = int((a.raycast[..., 3] > 0).sum())
    assert hits > 50_000, hits
    T.compare_results(a, b, sc, what="dense 512^3, full size, HIP vs oracle")
