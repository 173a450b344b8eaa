"""Scene checkpoint / resume (aux subsystem "checkpoint/resume"): save -> load into a fresh scene ->
continue must be bit-identical to the uninterrupted run; block files follow ORUtils/MemoryBlockPersister.h (checked
with the reference's own loader where the reference build exists)."""
import ctypes as C
import os

import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import capi
from itm_testlib import Scenario

SC = Scenario(name="ckpt", w=160, h=120, voxelSize=0.01, frames=6)


def run(be, sc, split, tmp):
    """frames [0, split) -> save; returns (uninterrupted result, resumed result)"""
    ses = T.Session(be, sc)
    for k in range(split):
        ses.frame(k, fused=(k % 2 == 0))
    ses.scene.save(tmp, ses.rs)
    for k in range(split, sc.frames):
        ses.frame(k, fused=True)
    full = ses.snapshot()
    ses.close()
    ses2 = T.Session(be, sc)            # fresh scene + render state (ResetScene done)
    ses2.scene.load(tmp, ses2.rs)
    for k in range(split, sc.frames):
        ses2.frame(k, fused=True)
    resumed = ses2.snapshot()
    ses2.close()
    return full, resumed


@pytest.mark.gpu
@pytest.mark.parametrize("voxel,index", [(capi.VOXEL_S, capi.INDEX_HASH), (capi.VOXEL_F_RGB, capi.INDEX_HASH), (capi.VOXEL_S, capi.INDEX_DENSE)])
def test_resume_is_bit_identical(hip, tmp_path, voxel, index):
    kw = dict(voxelType=voxel, indexType=index, colour=voxel in (capi.VOXEL_S_RGB, capi.VOXEL_F_RGB))
    if index == capi.INDEX_DENSE:
        kw.update(denseSize=(128, 128, 128), denseOffset=(-64, -64, 96), voxelSize=0.02)
    sc = Scenario(name="ckpt", w=160, h=120, frames=6, **{"voxelSize": 0.01, **kw})
    full, resumed = run(hip, sc, 3, str(tmp_path))
    T.compare_results(full, resumed, sc, exact=True, what="resume")
    assert np.array_equal(full.voxels, resumed.voxels)


@pytest.mark.gpu
def test_block_files_and_errors(hip, tmp_path):
    ses = T.Session(hip, SC)
    for k in range(2):
        ses.frame(k)
    d = str(tmp_path)
    ses.scene.save(d, ses.rs)
    table = ses.scene.download(capi.BUF_HASH_ENTRIES)
    raw = open(os.path.join(d, "hash.dat"), "rb").read()
    assert np.frombuffer(raw[:4], np.int32)[0] == table.shape[0]                 # element count, then the raw entries
    assert raw[4:] == table.tobytes()
    for name in ("excess.dat", "alloc.dat", "voxel.dat", "counters.dat", "config.dat", "visible_ids.dat", "visible_type.dat"):
        assert os.path.getsize(os.path.join(d, name)) > 4
    ref = T.reference_backend()
    if ref is not None:   # the reference's MemoryBlockPersister accepts the file
        dst = np.zeros_like(table)
        n = C.CDLL(T.REF_LIB).itmr_debug_load_hash_block(os.path.join(d, "hash.dat").encode(), dst.ctypes.data_as(C.c_void_p), table.shape[0])
        assert n == table.shape[0] and dst.tobytes() == table.tobytes()
    # a scene of another configuration refuses the checkpoint
    other = hip.create_scene(capi.VOXEL_F, capi.INDEX_HASH, SC.params())
    with pytest.raises(capi.ItmError):
        other.load(d)
    other.close()
    os.truncate(os.path.join(d, "voxel.dat"), 100)
    with pytest.raises(capi.ItmError):
        ses.scene.load(d, ses.rs)
    ses.close()
