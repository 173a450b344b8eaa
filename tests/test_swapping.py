"""SURVEY 8f-4, last part: host swapping.  ITMSwappingEngine_CPU (DeviceSpecific/CPU/ITMSwappingEngine_CPU.cpp:21-168), the voxel
combination of DeviceAgnostic/ITMSwappingEngine.h:7-69, ITMGlobalCache (Objects/ITMGlobalCache.h) and the swapping hooks of
AllocateSceneFromDepth (enlarged frustum :243-275, swap states :250-253, re-allocation of swapped-out entries :271-285), in the order
ITMDenseMapper::ProcessFrame calls them (Engine/ITMDenseMapper.cpp:50-64): allocate, integrate, IntegrateGlobalIntoLocal,
SaveToGlobalMemory.

The camera looks at the scene, turns away (the scene's blocks leave the enlarged frustum and go to the host cache, 0x1000 per call),
and turns back (their entries are found with ptr == -1, get new voxel blocks, and the host's copies are combined into them).
Everything is compared after every frame: table, pool, free list, visible list and types, swap states, the cache's flags and blocks."""
import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import capi, synth

W, H = 320, 240


def poses_and_depths():
    intr = synth.intrinsics_for(W, H)
    scene_depth = [synth.depth_frame(W, H, synth.parity_position(k), intr) for k in range(3)]
    wall = np.full((H, W), 2.0, np.float32)
    seq = []
    for k in range(3):
        seq.append((synth.pose_matrix(synth.parity_position(k)), scene_depth[k]))
    for k in range(4):                                        # looking away: a wall 2 m in front of a camera turned by 1.3 rad
        seq.append((synth.pose_matrix_yaw((0.0, 0.0, 0.0), 1.3 + 0.01 * k), wall))
    for k in range(3):                                        # and back
        seq.append((synth.pose_matrix(synth.parity_position(k)), scene_depth[k]))
    return intr, seq


def run(be, voxel=capi.VOXEL_S, colour=False, frames=None, localBlockNum=0):
    intr, seq = poses_and_depths()
    s = be.create_scene(voxel, capi.INDEX_HASH, capi.default_params(voxelSize=0.005), useSwapping=True, localBlockNum=localBlockNum)
    s.reco.ResetScene()
    rs = s.vis.CreateRenderState((W, H))
    P = W * H
    pts = capi.DevBuffer(be, P * 16, np.float32, (H, W, 4)); nrm = capi.DevBuffer(be, P * 16, np.float32, (H, W, 4))
    rgb = be.to_backend(synth.rgb_frame(W, H)) if colour else None
    out = []
    for k, (M, depth) in enumerate(seq[:frames]):
        v = capi.View(be.to_backend(depth), W, H, M_d=M, intr_d=intr, rgb=rgb, w_rgb=W, h_rgb=H, intr_rgb=intr)
        s.reco.AllocateSceneFromDepth(v, rs)
        s.reco.IntegrateIntoScene(v, rs)
        s.swap_integrate_global_into_local(rs)
        s.swap_save_to_global_memory(rs)
        s.vis.CreateExpectedDepths(v.M_d, v.intr_d, rs)
        s.vis.CreateICPMaps(v, rs, pts, nrm)
        c = s.counters(rs)
        flags = s.global_cache_flags()
        stored = np.nonzero(flags)[0]
        sample = stored[:: max(1, len(stored) // 64)][:64]
        out.append(dict(counters={k_: c[k_] for k_ in ("lastFreeBlockId", "lastFreeExcessListId", "noVisibleEntries")},
                        hash=s.download(capi.BUF_HASH_ENTRIES), alloc=s.download(capi.BUF_ALLOCATION_LIST), voxels=s.download(capi.BUF_VOXEL_BLOCKS),
                        vis_ids=s.download(capi.BUF_VISIBLE_IDS, rs)[: c["noVisibleEntries"]].copy(), vis_type=s.download(capi.BUF_VISIBLE_TYPE, rs),
                        swap=s.download(capi.BUF_SWAP_STATES), flags=flags, sample=sample, blocks=[s.global_cache_block(int(e)) for e in sample],
                        raycast=s.download(capi.BUF_RAYCAST_RESULT, rs), points=pts.numpy().copy()))
    rs.close(); s.close()
    return out


def compare(a, b, what):
    for k, (x, y) in enumerate(zip(a, b)):
        tag = "%s frame %d: " % (what, k)
        assert x["counters"] == y["counters"], tag + "%s vs %s" % (x["counters"], y["counters"])
        T.assert_fields_equal(x["hash"], y["hash"], tag + "hash")
        assert np.array_equal(x["swap"], y["swap"]), tag + "swap states (%d differ)" % int(np.count_nonzero(x["swap"] != y["swap"]))
        assert np.array_equal(x["vis_type"], y["vis_type"]), tag + "visible types"
        assert np.array_equal(x["vis_ids"], y["vis_ids"]), tag + "visible ids"
        # the free list: entries above lastFreeBlockId are dead storage in both, below they must agree
        n = x["counters"]["lastFreeBlockId"] + 1
        assert np.array_equal(x["alloc"][:max(n, 0)], y["alloc"][:max(n, 0)]), tag + "allocation list"
        T.assert_fields_equal(x["voxels"], y["voxels"], tag + "voxels")
        assert np.array_equal(x["flags"], y["flags"]), tag + "cache flags"
        for e, p, q in zip(x["sample"], x["blocks"], y["blocks"]):
            T.assert_fields_equal(p, q, tag + "cached block of entry %d" % e)
        assert np.array_equal(x["raycast"][..., 3], y["raycast"][..., 3]), tag + "hit mask"
        hit = x["raycast"][..., 3] > 0
        assert np.array_equal(x["raycast"][hit], y["raycast"][hit]), tag + "ray-cast hits"
        assert np.array_equal(x["points"], y["points"]), tag + "ICP points"


def sanity(frames):
    """The sequence really swaps: blocks go out while the camera looks away (more than one transfer's worth), and come back."""
    out_peak = max(int(np.count_nonzero(f["hash"]["ptr"] == -1)) for f in frames)
    assert out_peak > 0x1000, out_peak
    assert int(np.count_nonzero(frames[-1]["swap"] == 2)) > 1000
    assert int(np.count_nonzero(frames[-1]["flags"])) > 0x1000
    assert int(np.count_nonzero(frames[-1]["hash"]["ptr"] == -1)) < out_peak


def test_oracle_swapping_equals_the_reference_engine(oracle):
    ref = T.reference_backend()
    if ref is None:
        pytest.skip("reference build not available")
    a, b = run(oracle), run(ref)
    compare(a, b, "oracle vs reference")
    sanity(b)


def test_oracle_swapping_colour_voxels_equal_the_reference_engine(oracle):
    ref = T.reference_backend()
    if ref is None:
        pytest.skip("reference build not available")
    compare(run(oracle, capi.VOXEL_S_RGB, True, frames=9), run(ref, capi.VOXEL_S_RGB, True, frames=9), "oracle vs reference (colour)")


@pytest.mark.gpu
@pytest.mark.parametrize("voxel,colour", [(capi.VOXEL_S, False), (capi.VOXEL_F_RGB, True)])
def test_hip_swapping_equals_the_oracle(hip, oracle, voxel, colour):
    a, b = run(hip, voxel, colour), run(oracle, voxel, colour)
    compare(a, b, "hip vs oracle")
    sanity(a)


# ---- a pool that runs dry before the camera turns away ------------------------------------------------------------------------------
# The allocation sweep keeps decrementing lastFreeBlockId for every request it cannot serve (_CPU.cpp:189,206), so an exhausted pool
# leaves the counter BELOW -1; the reference's SaveToGlobalMemory then writes voxelAllocationList[vbaIdx + 1] in front of the list
# (undefined).  Product and oracle count such a counter as -1: the blocks a swap-out frees go back on the list from index 0 and are
# handed out again -- the situation swapping exists for.
SMALL_POOL = 0x800


def sanity_small_pool(frames):
    assert min(f["counters"]["lastFreeBlockId"] for f in frames[:3]) < -1, "the pool never ran dry"
    assert int(np.count_nonzero(frames[2]["hash"]["ptr"] >= 0)) == SMALL_POOL, "every block of the pool is in use before the camera turns"
    away = frames[3:7]
    assert max(f["counters"]["lastFreeBlockId"] for f in away) >= 0, "the swap-out put no block back on the list"
    assert any(int(np.count_nonzero(f["hash"]["ptr"] == -1)) > 0 for f in away)
    for f in frames:
        n = f["counters"]["lastFreeBlockId"] + 1
        free = f["alloc"][:max(n, 0)]
        assert len(np.unique(free)) == len(free) and (free >= 0).all() and (free < SMALL_POOL).all(), "free list corrupt"
        used = f["hash"]["ptr"][f["hash"]["ptr"] >= 0]
        assert len(np.unique(used)) == len(used), "a voxel block is owned by two entries"
        assert not np.intersect1d(used, free).size, "a voxel block is both free and in use"
    # the blocks that came back were allocated again
    assert int(np.count_nonzero(frames[-1]["hash"]["ptr"] >= 0)) > SMALL_POOL // 2


def test_oracle_swapping_with_an_exhausted_pool_recycles_the_freed_blocks(oracle):
    sanity_small_pool(run(oracle, localBlockNum=SMALL_POOL))


@pytest.mark.gpu
def test_hip_swapping_with_an_exhausted_pool_equals_the_oracle(hip, oracle):
    a, b = run(hip, localBlockNum=SMALL_POOL), run(oracle, localBlockNum=SMALL_POOL)
    compare(a, b, "hip vs oracle (small pool)")
    sanity_small_pool(a)
