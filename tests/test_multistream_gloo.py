"""N>1 path on CPU: two processes (gloo, world_size 2), one independent depth stream each, the
per-frame all-gather of visible-block records (infinitam_amd/streams.py).  The scene work runs on
the CPU oracle here; the exchange code is the one bench.py uses with RCCL on the GPU box."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import itm_testlib as T
    from infinitam_amd.streams import VisibleListExchange, stream_of_rank
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ob = T.oracle_backend()
        sc = T.Scenario(name="ms", w=160, h=120, voxelSize=0.01, frames=2, stream=stream_of_rank(rank, world))
        ses = T.Session(ob, sc)
        ex = VisibleListExchange(ob, world, rank, max_ids=2048, device="cpu", batch=2)   # 2 frames per collective
        tables = []
        for k in range(sc.frames):
            v = ses.frame(k, fused=True)
            ex.step(ses.rs.h, v.M_d, None)
            if k % 2 == 1:
                tables.append(ex.table())
        nv = ses.scene.counters(ses.rs)["noVisibleEntries"]
        ids = ses.scene.download(T.BUF_VISIBLE_IDS, ses.rs)[:nv]
        q.put((rank, np.asarray(v.M_d, np.float32), ids, tables[-1]))
    finally:
        dist.destroy_process_group()


def test_two_streams_exchange_visible_lists():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        rank, M, ids, table = q.get(timeout=300)
        results[rank] = (M, ids, table)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank holds every stream's pose and list; the two streams differ (0.05 m apart in y)
    for rank in (0, 1):
        table = results[rank][2]
        assert len(table) == 2
        for src in (0, 1):
            M_src, ids_src, _ = results[src]
            assert np.array_equal(table[src][0], M_src)
            assert np.array_equal(table[src][1], ids_src)
    assert not np.array_equal(results[0][1], results[1][1])
    assert results[0][0][13] != results[1][0][13]


def test_single_rank_exchange_is_identity():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import itm_testlib as T
    from infinitam_amd.streams import VisibleListExchange
    ob = T.oracle_backend()
    sc = T.Scenario(name="ms1", w=160, h=120, voxelSize=0.01, frames=1)
    ses = T.Session(ob, sc)
    v = ses.frame(0)
    ex = VisibleListExchange(ob, 1, 0, max_ids=64, device="cpu")   # list longer than the record: truncated
    ex.publish(ses.rs.h, v.M_d)
    ex.all_gather()
    (M, ids), = ex.table()
    nv = ses.scene.counters(ses.rs)["noVisibleEntries"]
    assert nv > 64 and len(ids) == 64
    assert np.array_equal(ids, ses.scene.download(T.BUF_VISIBLE_IDS, ses.rs)[:64])
    assert np.array_equal(M, np.asarray(v.M_d, np.float32))


def test_record_words_are_the_documented_layout():
    """The format both exchange implementations carry (include/itm_hip.h itm_export_visible_record, exchange.hip, streams.py):
    per record 16 words of M_d float bits, the FULL visible count (also when the list is truncated), ids, then -1 padding;
    table = world x batch records, frame of the batch ascending.  Built by hand from the scene's own buffers here."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import itm_testlib as T
    from infinitam_amd.streams import VisibleListExchange, decode_table, RECORD_HEADER
    ob = T.oracle_backend()
    sc = T.Scenario(name="ms2", w=160, h=120, voxelSize=0.01, frames=3)
    ses = T.Session(ob, sc)
    for max_ids in (4096, 64):
        ex = VisibleListExchange(ob, 1, 0, max_ids=max_ids, device="cpu", batch=3)
        want = np.full((1, 3, RECORD_HEADER + max_ids), -1, np.int32)
        for k in range(3):
            v = ses.frame(k, fused=True)
            ex.step(ses.rs.h, v.M_d, None)
            nv = ses.scene.counters(ses.rs)["noVisibleEntries"]
            ids = ses.scene.download(T.BUF_VISIBLE_IDS, ses.rs)[:nv]
            want[0, k, :16] = np.asarray(v.M_d, np.float32).reshape(16).view(np.int32)
            want[0, k, 16] = nv
            n = min(nv, max_ids)
            want[0, k, 17:17 + n] = ids[:n]
        assert np.array_equal(ex.raw_table(), want)
        (M, ids2), = decode_table(want, 1, 3, max_ids)
        assert np.array_equal(M, np.asarray(v.M_d, np.float32).reshape(16)) and np.array_equal(ids2, ids[:max_ids])
