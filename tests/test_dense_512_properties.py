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
