"""One rank of a world-N run of the library's exchange on ONE GPU (helper of tests/test_native_exchange.py, started as a subprocess with
ITM_RCCL_LIBRARY = the stand-in transport tests/cpp/librccl_standin.so).  usage: exchange_world2_worker.py RANK WORLD IDFILE OUTFILE [PACE_MS]"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import itm_testlib as T                                    # noqa: E402
from infinitam_amd import capi                             # noqa: E402
from infinitam_amd.streams import NativeExchange           # noqa: E402

BATCH, MAX_IDS, CHECKED_BATCHES, FREE_BATCHES = 2, 1024, 3, 11      # 3 batches read back one by one, then 11 without a read: the ring of 8 wraps


def main():
    rank, world, idfile, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    pace = float(sys.argv[5]) / 1e3 if len(sys.argv) > 5 else 0.0
    hip = T.hip_backend()
    if rank == 0:
        uid = NativeExchange.unique_id(hip)
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 60:
                raise SystemExit("no unique id from rank 0")
            time.sleep(0.01)
        uid = open(idfile, "rb").read()
    frames = BATCH * (CHECKED_BATCHES + FREE_BATCHES)
    sc = T.Scenario(name="ex_w%d_r%d" % (world, rank), w=160, h=120, voxelSize=0.01, frames=frames, stream=rank)
    ses = T.Session(hip, sc)
    out = {"rank": rank, "error": None}
    own_M, own_n, own_ids, tables, at = [], [], [], [], []
    import ctypes as C
    stream = C.c_void_p()
    hip.check(hip.fn["stream_create"](C.byref(stream)), "stream_create")
    sums = capi.DevBuffer(hip, 8, np.uint64, (1,))
    try:
        ex = NativeExchange(hip, world, rank, max_ids=MAX_IDS, batch=BATCH, unique_id=uid)
        try:
            for k in range(frames):
                v = ses.frame(k, fused="four")
                ex.step(ses.rs.h, v.M_d, None)
                nv = ses.scene.counters(ses.rs)["noVisibleEntries"]
                ids = np.full(MAX_IDS, -1, np.int32)
                got = ses.scene.download(capi.BUF_VISIBLE_IDS, ses.rs)[:min(nv, MAX_IDS)]
                ids[:len(got)] = got
                own_M.append(np.asarray(v.M_d, np.float32).reshape(16)); own_n.append(nv); own_ids.append(ids)
                batch_no = (k + 1) // BATCH
                if (k + 1) % BATCH == 0 and (batch_no <= CHECKED_BATCHES or k == frames - 1):
                    tables.append(ex.raw_table().copy()); at.append(k)
                if k + 1 == BATCH * CHECKED_BATCHES:
                    # a device-side consumer: the newest table is acquired on a stream of its own and read by a SLOW kernel while the
                    # 11 batches that follow are stepped and collected (the ring of eight wraps around the held slot)
                    table, first = ex.acquire(stream.value)
                    out["consumer_first_frame"] = int(first)
                    hip.check(hip.fn["debug_checksum"](C.c_void_p(table), world * BATCH * (17 + MAX_IDS), 300, C.c_void_p(sums.ptr), stream), "debug_checksum")
                if pace:
                    time.sleep(pace)
            ex.release(stream.value)
            hip.check(hip.fn["stream_synchronize"](stream), "stream_synchronize")
            out["consumer_checksum"] = int(sums.numpy()[0])
            out["self_check"] = list(ex.self_check())
        finally:
            ex.close()
    except capi.ItmError as e:
        out["error"] = str(e)
    finally:
        ses.close()
    np.savez(outfile, own_M=np.array(own_M), own_n=np.array(own_n), own_ids=np.array(own_ids), tables=np.array(tables), at=np.array(at))
    with open(outfile + ".json", "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
