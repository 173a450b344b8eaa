"""The exchange issued from the library (infinitam_amd/csrc/exchange.hip): record copy on the frame stream + all-gather on a side stream.
On the one-GPU box the communicator has one rank (a device copy stands in for the collective); the record layout, the batch
ping-pong and the stream ordering are the ones every rank runs.  RCCL itself is exercised through itm_exchange_unique_id (the library
is loaded and answers) and, with two ranks, by the driver's multi-GPU bench."""
import numpy as np
import pytest

import itm_testlib as T
from infinitam_amd import capi
from infinitam_amd.streams import NativeExchange


def test_exchange_entry_points_are_exported():
    import infinitam_amd
    be = capi.Backend(infinitam_amd.lib_path(), "itm_") if T.os.path.exists(infinitam_amd.lib_path()) else None
    if be is None:
        pytest.skip("library not built")
    for name in ("exchange_unique_id", "exchange_create", "exchange_destroy", "exchange_step", "exchange_info", "exchange_table"):
        assert name in be.fn


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 2, 3])
def test_single_rank_table_is_the_streams_own_record(hip, batch):
    sc = T.Scenario(name="ex", w=160, h=120, voxelSize=0.01, frames=7)
    ses = T.Session(hip, sc)
    ex = NativeExchange(hip, 1, 0, max_ids=2048, batch=batch)
    try:
        for k in range(sc.frames):
            v = ses.frame(k, fused=True)
            ex.step(ses.rs.h, v.M_d, None)
            if (k + 1) % batch == 0:                      # a collective was issued with this frame as the newest record
                (M, ids), = ex.table()
                assert np.array_equal(M, np.asarray(v.M_d, np.float32).reshape(16))
                nv = ses.scene.counters(ses.rs)["noVisibleEntries"]
                want = ses.scene.download(capi.BUF_VISIBLE_IDS, ses.rs)[:nv]
                assert nv > 100 and np.array_equal(ids, want[:2048])
    finally:
        ex.close()
        ses.close()


@pytest.mark.gpu
def test_rccl_is_loadable_and_hands_out_a_unique_id(hip):
    a, b = NativeExchange.unique_id(hip), NativeExchange.unique_id(hip)
    assert len(a) == 128 and a != b and any(a)
