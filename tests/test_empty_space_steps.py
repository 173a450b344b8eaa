"""Empty-space steps of the ray caster.

The reference's step at a position without a block is 8 voxels along a unit direction (DeviceAgnostic/ITMVisualisationEngine.h:
129-130,139-141).  Kept from round 4's near-bit experiment (one byte per cell saying how far the nearest allocated block is, so that
rays cross proven-empty space on arithmetic alone: built, bit-exact, slower in three forms and removed -- profiles/r4_raycast_notes.md
section 1; the code is in the history before round 5):
 * the arithmetic bound any such structure rests on, proven on float32 exactly as the kernel computes (CPU);
 * free-view ray casts through long empty stretches against the oracle (GPU)."""
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


@pytest.mark.gpu
def test_free_view_rays_through_empty_space_equal_the_oracle(hip, oracle):
    """Rays from poses the scene was never fused from cross long stretches without blocks (the parked rays' look-ahead over directory
    cells) and graze allocated shells: FindSurface from a ring of cameras."""
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
