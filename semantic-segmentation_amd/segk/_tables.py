"""Device tables of per-chunk kernel arguments (multi-tensor launches).

A table is a numpy structured array with one row per chunk of at most CHUNK elements of some tensor.  Columns
that never change (parameter addresses, sizes) are filled once; a column that changes every step (gradient
addresses: autograd allocates new gradient tensors each step) is refilled and the table re-uploaded through a small
ring of pinned staging buffers, stream-ordered, without a host synchronisation in the steady state.

While the current stream records a HIP graph, an upload becomes a copy node that reads its pinned source again at
every replay: such an upload gets a staging buffer of its own, kept alive with the table and never rewritten, and
records no event (an event recorded during a capture cannot be waited for by the host).
"""
import numpy as np
import torch

CHUNK = 65536  # elements per workgroup
_RING = 4      # pinned staging buffers in flight


def chunks_of(tensors):
    """-> (tensor index, element offset, element count) per chunk, as numpy arrays."""
    tidx, offs, ns = [], [], []
    for k, t in enumerate(tensors):
        n = t.numel()
        for o in range(0, n, CHUNK):
            tidx.append(k)
            offs.append(o)
            ns.append(min(CHUNK, n - o))
    return np.asarray(tidx, dtype=np.int64), np.asarray(offs, dtype=np.uint64), np.asarray(ns, dtype=np.int32)


def base_ptrs(tensors, tidx):
    return np.asarray([t.data_ptr() for t in tensors], dtype=np.uint64)[tidx]


def flat_storage(t):
    """The elements of a dense tensor in storage order (contiguous and channels_last alike)."""
    dense = t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
    if not dense:
        raise RuntimeError("tensor of shape %s, strides %s is not dense" % (tuple(t.shape), t.stride()))
    return torch.as_strided(t, (t.numel(),), (1,))


class StagedTable:
    def __init__(self, table, device):
        self.table = table
        self.dev = torch.empty(max(table.nbytes, 1), dtype=torch.uint8, device=device)
        self.host = [torch.empty(max(table.nbytes, 1), dtype=torch.uint8).pin_memory() for _ in range(_RING)]
        self.events = [None] * _RING
        self.turn = 0
        # staging buffers for uploads recorded in HIP graphs: allocated here, because pinned memory cannot be
        # allocated while a stream is capturing; one is used up per recorded upload
        self.spares = [torch.empty(max(table.nbytes, 1), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.recorded = []

    def upload(self, rows=None):
        """Copy the table (or the selected rows) to the device; returns the number of rows uploaded."""
        t = self.table if rows is None else self.table[rows]
        n = len(t)
        if n == 0:
            return 0
        raw = torch.from_numpy(np.ascontiguousarray(t).view(np.uint8).reshape(-1))
        if torch.cuda.is_current_stream_capturing():
            if not self.spares:
                raise RuntimeError("this table was already recorded into %d HIP graphs" % len(self.recorded))
            host = self.spares.pop()
            host[:raw.numel()].copy_(raw)
            self.recorded.append(host)
            self.dev[:raw.numel()].copy_(host[:raw.numel()], non_blocking=True)
            return n
        slot = self.turn
        self.turn = (self.turn + 1) % _RING
        if self.events[slot] is not None:
            self.events[slot].synchronize()   # the copy that last used this staging buffer has long finished
        self.host[slot][:raw.numel()].copy_(raw)
        self.dev[:raw.numel()].copy_(self.host[slot][:raw.numel()], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[slot] = ev
        return n

    def ptr(self):
        return self.dev.data_ptr()
