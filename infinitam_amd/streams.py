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


def stream_of_rank(rank: int, world: int) -> int:
    """Stream g <-> rank g: weak scaling, per-GPU work is fixed."""
    return rank


class VisibleListExchange:
    def __init__(self, backend, world: int, rank: int, max_ids: int = 16384, device=None):
        import torch
        self.torch = torch
        self.be, self.world, self.rank, self.max_ids = backend, world, rank, max_ids
        self.words = RECORD_HEADER + max_ids
        dev = device if device is not None else ("cuda" if backend.on_device else "cpu")
        self.record = torch.full((self.words,), -1, dtype=torch.int32, device=dev)
        self.gathered = torch.full((world * self.words,), -1, dtype=torch.int32, device=dev)

    def publish(self, render_state_handle: int, M_d, stream_ptr=None):
        """Writes this stream's record into self.record on `stream_ptr` (device side, no host sync)."""
        Ma = (C.c_float * 16)(*[float(x) for x in np.asarray(M_d, np.float32).reshape(16)])
        rc = self.be.fn["export_visible_record"](C.c_void_p(render_state_handle), Ma, self.max_ids,
                                                 C.c_void_p(self.record.data_ptr()), C.c_void_p(stream_ptr))
        self.be.check(rc, "export_visible_record")

    def all_gather(self, group=None):
        import torch.distributed as dist
        if self.world == 1:
            self.gathered.copy_(self.record)
        else:
            dist.all_gather_into_tensor(self.gathered, self.record, group=group)

    def table(self) -> List[Tuple[np.ndarray, np.ndarray]]:
        """Host view of the gathered records: per stream (M_d[16] float32, visible ids int32[nv])."""
        g = self.gathered.cpu().numpy().reshape(self.world, self.words)
        out = []
        for r in range(self.world):
            M = g[r, :16].view(np.float32).copy()
            nv = int(g[r, 16])
            out.append((M, g[r, 17:17 + min(nv, self.max_ids)].copy()))
        return out
