"""The sdf mirror is PAGED (infinitam_amd/csrc/itm_types.h): a 1 MB table over the mirror's cube of 256^3 blocks says which
4 x 4 x 4-block pages hold blocks; pages come from a pool as blocks are allocated.  The reference's footprint is O(pool)
(ITMLib/Objects/ITMLocalVBA.h:18-59); the mirror's now is too (rounds 2-3: 17 GB per scene whatever it held).

Here: the footprint on the bench scene, a pool too small for the scene (pages that could not be mapped say nothing: their rays use
the block directory, same results), and pages returning to the pool when the cube moves or the scene is reset."""
import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import capi

pytestmark = pytest.mark.gpu


def test_mirror_memory_follows_the_scene_not_the_cube(hip, oracle):
    sc = T.Scenario(name="paged_bench", voxelSize=0.004, localBlockNum=0x40000, frames=6, trajectory="bench")
    ses = T.Session(hip, sc)
    for k in range(sc.frames):
        ses.frame(k, fused="four")
    info = ses.scene.accel_info()
    blocks = int(np.count_nonzero(ses.scene.download(capi.BUF_HASH_ENTRIES)["ptr"] >= 0))
    assert 0 < info["mirror_bytes"] < (1 << 30), info                     # 512 MB pool + 1 MB table (rounds 2-3: 17.18 GB)
    assert info["mirror_pages"] == 8192 and 100 < info["mirror_pages_mapped"] < 4000, info
    assert info["mirror_pages_mapped"] * 64 >= blocks / 2, (info, blocks)   # the pages hold the scene's blocks (all of them lie inside the cube here)
    a = ses.snapshot(); a.counters = [ses.scene.counters(ses.rs)]
    ref = T.Session(oracle, sc)
    for k in range(sc.frames):
        ref.frame(k)
    b = ref.snapshot(); b.counters = [ref.scene.counters(ref.rs)]
    T.compare_results(a, b, sc, what="paged mirror")
    ses.scene.reco.ResetScene()
    assert ses.scene.accel_info()["mirror_pages_mapped"] == 0, "pages after ResetScene"
    ses.close(); ref.close()


@pytest.mark.parametrize("pages", [1, 24, 150])
def test_a_pool_that_runs_dry_costs_speed_not_results(hip, oracle, monkeypatch, pages):
    """ITM_MIRROR_PAGES (read when the scene is created): with 1, 24 or 150 pages most, many or some of the scene's pages cannot be
    mapped; their table entries say "unmappable", rays through them read the block directory.  Five frames of a turning camera, the
    free-view entry points from another pose, then a cube move (everything unmapped and mapped again at the new origin)."""
    monkeypatch.setenv("ITM_MIRROR_PAGES", str(pages))
    sc = T.Scenario(name="paged_dry_%d" % pages, voxelSize=0.005, w=320, h=240, frames=5, trajectory="yaw")
    a = T.run_scenario(hip, sc, fused="four")
    b = T.run_scenario(oracle, sc)
    T.compare_results(a, b, sc, what=sc.name)
    ses = T.Session(hip, sc)
    for k in range(2):
        ses.frame(k, fused=True)
    info = ses.scene.accel_info()
    assert info["mirror_pages"] == pages and info["mirror_pages_mapped"] == pages, info      # (the scene wants ~270)
    ses.close()


def test_pages_return_to_the_pool_when_the_cube_moves(hip, oracle):
    from test_accel_origin import walk_poses
    poses = walk_poses()
    sc = T.Scenario(name="paged_walk", w=160, h=120, voxelSize=0.005, localBlockNum=0x40000, frames=len(poses))
    res, mapped = [], []
    for be in (hip, oracle):
        ses = T.Session(be, sc)
        depth = [be.to_backend(sc.depth(k)) for k in range(sc.frames)]
        for k in range(sc.frames):
            v = capi.View(depth[k], sc.w, sc.h, M_d=poses[k], intr_d=sc.intr())
            ses.scene.process_frame(v, ses.rs, ses.points, ses.normals)
            if be is hip:
                mapped.append(ses.scene.accel_info()["mirror_pages_mapped"])
        r = ses.snapshot(); r.counters = [ses.scene.counters(ses.rs)]
        res.append(r)
        if be is hip:
            assert ses.scene.accel_info()["moves"] >= 3
        ses.close()
    T.compare_results(res[0], res[1], sc, what="paged mirror across cube moves")
    assert max(mapped) < 8192 and min(mapped) > 0
