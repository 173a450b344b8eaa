"""The exchange issued from the library (infinitam_amd/csrc/exchange.hip): record copy on the frame stream + all-gather on a side stream.
On the one-GPU box the communicator has ONE rank, but it is a real RCCL communicator (ncclCommInitRank) and the collective is a real
ncclAllGather on the side stream -- the record layout, the batch ping-pong, the stream ordering and the RCCL calls are the ones every
rank of an 8-GPU run makes.  The torch.distributed implementation (VisibleListExchange) is pinned to the same words."""
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
    for name in ("exchange_unique_id", "exchange_create", "exchange_destroy", "exchange_step", "exchange_info", "exchange_table", "exchange_self_check",
                 "exchange_acquire", "exchange_release"):
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


@pytest.mark.gpu
def test_library_and_torch_exchange_hold_the_same_words(hip):
    """Both implementations fill world x batch x (17 + max_ids) int32 words; word for word equal over several batches."""
    import torch
    from infinitam_amd.streams import VisibleListExchange
    sc = T.Scenario(name="ex2", w=160, h=120, voxelSize=0.01, frames=8)
    ses = T.Session(hip, sc)
    lib = NativeExchange(hip, 1, 0, max_ids=1024, batch=4)
    tor = VisibleListExchange(hip, 1, 0, max_ids=1024, device="cuda", batch=4)
    cur = torch.cuda.current_stream()
    try:
        for k in range(sc.frames):
            v = ses.frame(k, fused=True)
            lib.step(ses.rs.h, v.M_d, None)
            tor.step(ses.rs.h, v.M_d, cur)
            if (k + 1) % 4 == 0:
                torch.cuda.synchronize()
                a, b = lib.raw_table(), tor.raw_table()
                assert a.shape == b.shape == (1, 4, 17 + 1024) and np.array_equal(a, b)
                assert (a[0, :, 16] > 100).all()                        # four different frames' counts, none empty
                assert len({a[0, i, :16].tobytes() for i in range(4)}) == 4   # four different poses travelled
    finally:
        lib.close()
        ses.close()


@pytest.mark.gpu
def test_one_rank_collective_equals_the_device_copy(hip):
    """Debug key 24 (ITM_DEBUG_EXCHANGE_DEVICE_COPY) replaces the one-rank ncclAllGather by a memcpy: same table."""
    sc = T.Scenario(name="ex3", w=160, h=120, voxelSize=0.01, frames=4)
    tables = []
    for key in (0, 1):
        hip.check(hip.fn["debug_set"](24, key), "debug_set")
        ses = T.Session(hip, sc)
        try:
            ex = NativeExchange(hip, 1, 0, max_ids=512, batch=2)
        finally:
            hip.check(hip.fn["debug_set"](24, 0), "debug_set")
        try:
            for k in range(sc.frames):
                v = ses.frame(k, fused=True)
                ex.step(ses.rs.h, v.M_d, None)
            tables.append(ex.raw_table().copy())
        finally:
            ex.close()
            ses.close()
    assert np.array_equal(tables[0], tables[1]) and (tables[0][0, :, 16] > 100).all()


@pytest.mark.gpu
def test_every_collective_is_self_checked_and_corruption_is_loud(hip, monkeypatch):
    # (the corrupted word is a test hook of the library: debug key 25, read when an exchange is created)
    """Behind each collective the rank's own block of the gathered table is compared with what it sent (on the side stream): a healthy
    run counts the checks and no mismatch; a word corrupted on the way (test hook) makes the next step and the table read fail."""
    sc = T.Scenario(name="ex_check", w=160, h=120, voxelSize=0.01, frames=9)
    ses = T.Session(hip, sc)
    ex = NativeExchange(hip, 1, 0, max_ids=512, batch=3)
    try:
        for k in range(sc.frames):
            v = ses.frame(k, fused="four")
            ex.step(ses.rs.h, v.M_d, None)
        assert ex.self_check() == (3, 0)
        ex.table()
    finally:
        ex.close()
    hip.check(hip.fn["debug_set"](25, 40), "debug_set")
    try:
        ex = NativeExchange(hip, 1, 0, max_ids=512, batch=3)
    finally:
        hip.check(hip.fn["debug_set"](25, -1), "debug_set")
    try:
        with pytest.raises(capi.ItmError, match="self-check"):
            for k in range(sc.frames):
                v = ses.frame(k, fused=True)
                ex.step(ses.rs.h, v.M_d, None)
                if (k + 1) % 3 == 0:
                    assert ex.self_check()[1] > 0                # (waits for the side stream: the check has run)
        with pytest.raises(capi.ItmError, match="self-check"):
            ex.table()
    finally:
        ex.close()
    monkeypatch.setenv("ITM_EXCHANGE_SELF_CHECK", "0")
    ex = NativeExchange(hip, 1, 0, max_ids=512, batch=3)
    try:
        assert ex.self_check() == (-1, 0)
    finally:
        ex.close()
        ses.close()


def expected_checksum(table_words: np.ndarray, rounds: int) -> int:
    """What itm_debug_checksum computes over a table (exchange.hip: position-weighted sum of the words as unsigned, `rounds` passes)."""
    w = table_words.reshape(-1).view(np.uint32).astype(np.uint64)
    i = np.arange(len(w), dtype=np.uint64) % np.uint64(1021) + np.uint64(1)
    return int((int((w * i).sum(dtype=np.uint64)) * rounds) & 0xFFFFFFFFFFFFFFFF)


@pytest.mark.gpu
def test_a_device_side_consumer_reads_a_table_no_later_collective_touches(hip):
    """itm_exchange_acquire / _release (the per-GPU global visibility table a merger's kernel reads).  A slow consumer kernel
    (one workgroup, 400 passes over the table) is put on its own stream behind an acquire, then 20 further batches -- more than the ring
    of eight slots -- are stepped and collected while it runs: its checksum must be that of the batch it acquired, word for word.  With
    ONE gathered table for all batches in flight (round 4) the kernel would read a table that later collectives rewrite."""
    sc = T.Scenario(name="ex_acq", w=160, h=120, voxelSize=0.01, frames=3)
    ses = T.Session(hip, sc)
    MAX_IDS, BATCH, ROUNDS = 2048, 2, 400
    ex = NativeExchange(hip, 1, 0, max_ids=MAX_IDS, batch=BATCH)
    stream = capi._P()
    hip.check(hip.fn["stream_create"](capi.C.byref(stream)), "stream_create")
    sums = capi.DevBuffer(hip, 64 * 8, np.uint64, (64,))
    try:
        assert ex.acquire(stream.value) == (0, -1)                       # nothing gathered yet
        for k in range(sc.frames):
            v = ses.frame(k, fused="four")
        nv = ses.scene.counters(ses.rs)["noVisibleEntries"]
        ids = np.full(MAX_IDS, -1, np.int32)
        got = ses.scene.download(capi.BUF_VISIBLE_IDS, ses.rs)[:min(nv, MAX_IDS)]
        ids[:len(got)] = got

        def record(frame_no):
            M = np.asarray(v.M_d, np.float32).reshape(16).copy()
            M[0] = np.float32(frame_no)                                   # (the pose travels as the caller gives it: a frame stamp)
            return M

        frame_no, expected, taken = 0, {}, []
        for rnd in range(6):
            for _ in range(BATCH * (1 if rnd == 0 else 4)):               # one batch, then four batches per round: 21 batches, the ring wraps
                ex.step(ses.rs.h, record(frame_no), None)
                frame_no += 1
            table, first = ex.acquire(stream.value)
            assert table != 0 and first == frame_no - BATCH, (table, first, frame_no)
            hip.check(hip.fn["debug_checksum"](capi._P(table), BATCH * (17 + MAX_IDS), ROUNDS, capi._P(sums.ptr + 8 * rnd), stream), "debug_checksum")
            words = np.empty((BATCH, 17 + MAX_IDS), np.int32)
            for j in range(BATCH):
                words[j, :16] = record(first + j).view(np.int32); words[j, 16] = nv; words[j, 17:] = ids
            expected[rnd] = expected_checksum(words, ROUNDS)
            taken.append(first)
        ex.release(stream.value)
        hip.check(hip.fn["stream_synchronize"](stream), "stream_synchronize")
        got_sums = sums.numpy()
        for rnd, first in enumerate(taken):
            assert int(got_sums[rnd]) == expected[rnd], "the consumer of the batch at frame %d read a table that was not that batch's" % first
        assert ex.self_check() == (21, 0)
    finally:
        ex.close()
        hip.check(hip.fn["stream_destroy"](stream), "stream_destroy")
        ses.close()


# ---- world 2 on ONE GPU: the library's exchange through a stand-in transport ------------------------------------------------------
# RCCL refuses two ranks on one device and the pool has one GPU per box, so the N > 1 code of exchange.hip (rank-major table, the
# self-check's own-block offset, the ring of batch buffers with a peer at another pace, the bootstrap through the 128-byte id) runs
# here over tests/cpp/rccl_standin.cpp -- a TEST DOUBLE loaded through ITM_RCCL_LIBRARY that stages the all-gather through shared
# memory.  It says nothing about RCCL itself; it says the code around the collective is right for more than one rank.
STANDIN_SRC = T.os.path.join(T.ROOT, "tests", "cpp", "rccl_standin.cpp")
STANDIN = T.os.path.join(T.ROOT, "tests", "cpp", "librccl_standin.so")


def build_standin():
    import subprocess
    if not T.os.path.exists(STANDIN) or T.os.path.getmtime(STANDIN) < T.os.path.getmtime(STANDIN_SRC):
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", "-w", "-o", STANDIN, STANDIN_SRC, "-lrt"], check=True, capture_output=True)
    return STANDIN


def test_standin_transport_builds_and_exports_the_five_entry_points():
    import ctypes
    lib = ctypes.CDLL(build_standin())
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclGetErrorString"):
        assert hasattr(lib, name)


def run_world(tmp_path, world, extra_env=None, paces=None):
    import json, subprocess, sys
    env = dict(T.os.environ, ITM_RCCL_LIBRARY=build_standin(), **(extra_env or {}))
    idfile = str(tmp_path / "uid.bin")
    procs = []
    for r in range(world):
        cmd = [sys.executable, T.os.path.join(T.ROOT, "tests", "exchange_world2_worker.py"), str(r), str(world), idfile, str(tmp_path / ("rank%d.npz" % r))]
        if paces:
            cmd.append(str(paces[r]))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [(np.load(str(tmp_path / ("rank%d.npz" % r))), json.load(open(str(tmp_path / ("rank%d.npz.json" % r))))) for r in range(world)]


@pytest.mark.gpu
@pytest.mark.parametrize("world,paces", [(2, None), (2, (0, 15)), (3, (5, 0, 10))])
def test_library_exchange_with_several_ranks_on_one_gpu(hip, tmp_path, world, paces):
    """Every rank runs its own stream of BASELINE configs[3] (the reference's four calls per frame) and publishes through the library's
    exchange; ranks run at different paces, the ring of eight batch buffers wraps.  Every rank must hold the same table, block r of it
    must be what rank r sent (pose, count, ids, -1 padding), and every collective must have passed the own-block check."""
    res = run_world(tmp_path, world, paces=paces)
    for d, meta in res:
        assert meta["error"] is None, meta
        assert meta["self_check"] == [14, 0], meta                      # 3 + 11 batches, each collective checked, no word differed
        # the device-side consumer of every rank (itm_exchange_acquire + a slow checksum kernel while 11 further batches were collected):
        # it read the table of the batch it acquired, and that table was the same on every rank
        assert meta["consumer_first_frame"] == 4 and meta["consumer_checksum"] == res[0][1]["consumer_checksum"], meta
    t0, at = res[0][0]["tables"], res[0][0]["at"]
    assert t0.shape == (4, world, 2, 17 + 1024) and list(at) == [1, 3, 5, 27]
    for d, _ in res[1:]:
        assert np.array_equal(d["tables"], t0) and np.array_equal(d["at"], at)
    for r, (d, _) in enumerate(res):
        for c, k_last in enumerate(at):
            for slot in range(2):
                k = int(k_last) - 1 + slot
                rec = t0[c, r, slot]
                assert np.array_equal(rec[:16].view(np.float32), d["own_M"][k])
                assert rec[16] == d["own_n"][k] and d["own_n"][k] > 100
                assert np.array_equal(rec[17:], d["own_ids"][k])
    assert not np.array_equal(res[0][0]["own_ids"][5], res[1][0]["own_ids"][5])          # the streams differ: the comparison above says something
    # ... and the consumer's checksum is that of the gathered table of batch 2 (frames 4 and 5), which the host copy taken at frame 5 holds
    assert res[0][1]["consumer_checksum"] == expected_checksum(t0[2], 300)


@pytest.mark.gpu
def test_a_table_in_the_wrong_rank_order_is_caught_at_world_two(hip, tmp_path):
    """The transport files every rank's block one place further (what a communicator built with the wrong rank order would deliver):
    both ranks' self-checks fire and the error is loud."""
    res = run_world(tmp_path, 2, extra_env={"ITM_STANDIN_ROTATE_RANKS": "1", "ITM_STANDIN_TIMEOUT_S": "5"})
    for d, meta in res:
        assert meta["error"] and "self-check" in meta["error"], meta
