"""Multi-stream sharding (BASELINE configs[3]).

A scene (hash table + voxel pool) is private to one depth stream and never reads another stream's
data, so N streams shard one-per-GPU with no exchange needed for fusion itself.  What is exchanged
every frame -- off the critical path, on a side stream -- is a small fixed-size record per stream,

    { float M_d[16]; int32 noVisibleEntries; int32 visibleEntryIDs[max_ids] (padded with -1) },

all-gathered over the process group (RCCL over xGMI on the GPU box, gloo in the CPU tests) so that
every rank holds the pose and the live block list of every stream (the input of a shared-map
merger / global visibility table).  RCCL has no all-gather-v, hence the fixed record size.

The record is produced by the library on the device (itm_export_visible_record) without a host
round trip; this module only owns the buffers and issues the collective.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np

RECORD_HEADER = 17  # 16 floats of pose + 1 count


def decode_table(words: np.ndarray, world: int, batch: int, max_ids: int) -> List[Tuple[np.ndarray, np.ndarray]]:
    """ONE reader for the gathered table of both implementations (the library's exchange.hip and VisibleListExchange below):
    world x batch records of (17 + max_ids) int32 words, rank-major then frame of the batch; a record is 16 words holding the
    float bits of M_d (column-major, as handed to the frame), the visible count, then the ids padded with -1.
    Returns per stream the NEWEST record of the batch as (M_d[16] float32, ids[min(count, max_ids)] int32)."""
    g = np.asarray(words, np.int32).reshape(world, batch, RECORD_HEADER + max_ids)[:, -1, :]
    out = []
    for r in range(world):
        M = g[r, :16].view(np.float32).copy()
        nv = int(g[r, 16])
        out.append((M, g[r, 17:17 + min(nv, max_ids)].copy()))
    return out


def stream_of_rank(rank: int, world: int) -> int:
    """Stream g <-> rank g: weak scaling, per-GPU work is fixed."""
    return rank


class VisibleListExchange:
    """Owns two batch buffers (ping-pong) of `batch` records each and the gathered table.

    GPU protocol (`step`): each frame's record is written on the FRAME stream right behind the frame's
    kernels (a 3 us copy kernel, no cross-stream wait on the critical path).  Every `batch` frames one
    collective moves the whole batch (fewer, larger collectives: the per-call host cost of the
    collective, ~40 us through torch.distributed, would otherwise bound a 150 us frame) on a side
    stream that waits for the last copy; the frame stream only ever waits for the collective that used
    the same batch buffer two batches earlier, which has long finished."""

    def __init__(self, backend, world: int, rank: int, max_ids: int = 16384, device=None, batch: int = 1):
        import torch
        self.torch = torch
        self.be, self.world, self.rank, self.max_ids, self.batch = backend, world, rank, max_ids, batch
        self.words = RECORD_HEADER + max_ids
        dev = device if device is not None else ("cuda" if backend.on_device else "cpu")
        self.buffers = [torch.full((batch * self.words,), -1, dtype=torch.int32, device=dev) for _ in range(2)]
        self.slots = [[buf[i * self.words:(i + 1) * self.words] for i in range(batch)] for buf in self.buffers]
        self.slot_ptr = [[C.c_void_p(v.data_ptr()) for v in views] for views in self.slots]
        self.record, self.record_ptr = self.slots[0][0], self.slot_ptr[0][0]
        self.gathered = torch.full((world * batch * self.words,), -1, dtype=torch.int32, device=dev)
        self.frame = 0
        self._cuda = str(dev).startswith("cuda")
        if self._cuda:
            self.side = torch.cuda.Stream()
            self.copied = [torch.cuda.Event(), torch.cuda.Event()]
            self.released = [torch.cuda.Event(), torch.cuda.Event()]
            self.in_flight = [False, False]

    @staticmethod
    def pose_array(M_d):
        """ctypes form of a pose; callers on a per-frame path build it once per pose, not once per frame."""
        return (C.c_float * 16)(*[float(x) for x in np.asarray(M_d, np.float32).reshape(16)])

    def publish(self, render_state_handle: int, M_d, stream_ptr=None):
        """Writes this stream's record into self.record on `stream_ptr` (device side, no host sync)."""
        Ma = M_d if isinstance(M_d, C.Array) else self.pose_array(M_d)
        rc = self.be.fn["export_visible_record"](C.c_void_p(render_state_handle), Ma, self.max_ids,
                                                 self.record_ptr, C.c_void_p(stream_ptr))
        self.be.check(rc, "export_visible_record")

    def all_gather(self, group=None, source=None):
        import torch.distributed as dist
        src = source if source is not None else self.buffers[0]
        if self.world == 1 and not dist.is_initialized():
            self.gathered.copy_(src)
        else:
            dist.all_gather_into_tensor(self.gathered, src, group=group)

    def step(self, render_state_handle: int, M_d, frame_stream, group=None):
        """One frame of the GPU protocol described above; `frame_stream` is the torch stream the frame ran on."""
        torch = self.torch
        slot = self.frame % self.batch
        b = (self.frame // self.batch) & 1
        if not self._cuda:                                   # host-memory backends (CPU tests): same schedule, synchronous
            self.record, self.record_ptr = self.slots[b][slot], self.slot_ptr[b][slot]
            self.publish(render_state_handle, M_d, None)
            if slot == self.batch - 1:
                self.all_gather(group, self.buffers[b])
            self.frame += 1
            return
        if slot == 0 and self.in_flight[b]:
            frame_stream.wait_event(self.released[b])      # the collective two batches ago released this buffer
        self.record, self.record_ptr = self.slots[b][slot], self.slot_ptr[b][slot]
        self.publish(render_state_handle, M_d, frame_stream.cuda_stream)
        if slot == self.batch - 1:
            self.copied[b].record(frame_stream)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.copied[b])
                self.all_gather(group, self.buffers[b])
                self.released[b].record(self.side)
            self.in_flight[b] = True
        self.frame += 1

    def table(self) -> List[Tuple[np.ndarray, np.ndarray]]:
        """Host view of the gathered records: per stream (M_d[16] float32, visible ids int32[nv])."""
        if self._cuda:
            self.side.synchronize()      # the gathered table is written by collectives on the side stream
        return decode_table(self.gathered.cpu().numpy(), self.world, self.batch, self.max_ids)

    def raw_table(self) -> np.ndarray:
        """The gathered words as they lie in memory: world x batch x (17 + max_ids) int32, rank-major."""
        if self._cuda:
            self.side.synchronize()
        return self.gathered.cpu().numpy().reshape(self.world, self.batch, self.words).copy()


class NativeExchange:
    """The same exchange issued from the library (exchange.hip): record copy on the frame stream, RCCL all-gather on a side stream
    the library owns.  One C call per frame; no tensor framework on the per-frame path.  `unique_id` is the 128-byte RCCL id of
    rank 0 (`NativeExchange.unique_id(backend)`), distributed by the host; for world == 1 the library makes its own id -- a one-rank
    communicator runs the same ncclAllGather as eight ranks do."""

    def __init__(self, backend, world: int, rank: int, max_ids: int = 16384, batch: int = 1, unique_id: bytes = None):
        self.be, self.world, self.rank, self.max_ids, self.batch = backend, world, rank, max_ids, batch
        self.words = RECORD_HEADER + max_ids
        ida = None
        if unique_id is not None:
            ida = (C.c_ubyte * 128)(*unique_id)
        h = C.c_void_p()
        backend.check(backend.fn["exchange_create"](world, rank, ida, max_ids, batch, C.byref(h)), "exchange_create")
        self.h = h

    @staticmethod
    def unique_id(backend) -> bytes:
        ida = (C.c_ubyte * 128)()
        backend.check(backend.fn["exchange_unique_id"](ida), "exchange_unique_id")
        return bytes(ida)

    def step(self, render_state_handle: int, M_d, frame_stream=None, group=None):
        Ma = M_d if isinstance(M_d, C.Array) else VisibleListExchange.pose_array(M_d)
        sp = C.c_void_p(frame_stream.cuda_stream) if frame_stream is not None else None
        rc = self.be.fn["exchange_step"](self.h, C.c_void_p(render_state_handle), Ma, sp)
        if rc:
            self.be.check(rc, "exchange_step")

    def table(self) -> List[Tuple[np.ndarray, np.ndarray]]:
        return decode_table(self.raw_table(), self.world, self.batch, self.max_ids)

    def raw_table(self) -> np.ndarray:
        g = np.empty(self.world * self.batch * self.words, np.int32)
        self.be.check(self.be.fn["exchange_table"](self.h, g.ctypes.data_as(C.c_void_p), g.size), "exchange_table")
        return g.reshape(self.world, self.batch, self.words)

    def acquire(self, consumer_stream=None):
        """itm_exchange_acquire: (device pointer of the newest table, number of the first frame in it); (0, -1) before the first collective.
        The table is the caller's until release() / the next acquire(); `consumer_stream` (an int handle or None) is made to wait for it."""
        t, f = C.c_void_p(), C.c_longlong()
        self.be.check(self.be.fn["exchange_acquire"](self.h, C.c_void_p(consumer_stream), C.byref(t), C.byref(f)), "exchange_acquire")
        return (t.value or 0), f.value

    def release(self, consumer_stream=None):
        self.be.check(self.be.fn["exchange_release"](self.h, C.c_void_p(consumer_stream)), "exchange_release")

    def self_check(self) -> Tuple[int, int]:
        """(collectives checked, words of this rank's own block that differed from what it sent); -1 checked = self-check off."""
        a, b = C.c_int(), C.c_int()
        self.be.check(self.be.fn["exchange_self_check"](self.h, C.byref(a), C.byref(b)), "exchange_self_check")
        return a.value, b.value

    def close(self):
        if self.h:
            self.be.fn["exchange_destroy"](self.h)
            self.h = None
