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
    # different scene parameters (config.dat holds them too) are refused as well
    other = hip.create_scene(capi.VOXEL_S, capi.INDEX_HASH, capi.default_params(voxelSize=0.02))
    with pytest.raises(capi.ItmError):
        other.load(d)
    other.close()
    ses.close()


@pytest.mark.gpu
def test_rejected_checkpoints_leave_the_scene_untouched(hip, tmp_path):
    """itm_scene_load validates every file before the first upload: after a refused load the scene continues as if
    nothing had happened (same next frame as an undisturbed twin)."""
    import shutil
    src = T.Session(hip, SC)
    for k in range(2):
        src.frame(k)
    good = str(tmp_path / "good"); os.makedirs(good)
    src.scene.save(good, src.rs)
    assert not [f for f in os.listdir(good) if f.endswith(".tmp")]          # written through temporaries, all renamed
    victim, twin = T.Session(hip, SC), T.Session(hip, SC)
    for ses in (victim, twin):
        for k in range(3):
            ses.frame(k)

    made = []

    def corrupt(name, fn):
        made.append(name)
        bad = str(tmp_path / ("bad%d_%s" % (len(made), name))); shutil.copytree(good, bad)
        path = os.path.join(bad, name)
        fn(path)
        with pytest.raises(capi.ItmError):
            victim.scene.load(bad, victim.rs)

    corrupt("voxel.dat", lambda p: os.truncate(p, 100))                       # short LATER file: earlier blocks must not be live
    corrupt("hash.dat", lambda p: open(p, "ab").write(b"x"))                   # trailing bytes

    def bad_counter(p):
        raw = bytearray(open(p, "rb").read()); c = np.frombuffer(raw, np.int32).copy()
        c[1 + 2] = 10 ** 9                                                     # noVisibleEntries beyond the id list
        open(p, "wb").write(c.tobytes())
    corrupt("counters.dat", bad_counter)

    def bad_free_id(p):
        c = np.frombuffer(open(p, "rb").read(), np.int32).copy(); c[1 + 0] = 1 << 20   # lastFreeBlockId beyond the pool
        open(p, "wb").write(c.tobytes())
    corrupt("counters.dat", bad_free_id)

    def bad_entry(p):
        raw = np.frombuffer(open(p, "rb").read(), np.uint8).copy()
        e = raw[4:].view(capi.HASH_ENTRY_DTYPE); e["ptr"][5] = 1 << 24            # pointer outside the voxel pool
        open(p, "wb").write(raw.tobytes())
    corrupt("hash.dat", bad_entry)

    def bad_id(p):
        v = np.frombuffer(open(p, "rb").read(), np.int32).copy(); v[1] = -7
        open(p, "wb").write(v.tobytes())
    corrupt("visible_ids.dat", bad_id)

    for ses in (victim, twin):
        ses.frame(3, fused=True)
    a, b = victim.snapshot(), twin.snapshot()
    T.compare_results(a, b, SC, what="after refused loads")
    assert np.array_equal(a.voxels, b.voxels)
    # and the intact checkpoint still loads
    victim.scene.load(good, victim.rs)
    for ses in (src, victim, twin):
        ses.close()
