"""One process per GPU: the exchanges that replace nn.DataParallel + the queue-based SyncBN master.

The reference scatters a batch over replica threads, reduces BN statistics through Python queues
(/root/reference/modeling/sync_batchnorm/comm.py:97-125, batchnorm.py:90-111), gathers logits on GPU 0,
computes one loss on the gathered batch (/root/reference/train.py:148-153) and lets DataParallel
reduce-add gradients.  Here every rank owns its shard and three collectives (RCCL through
torch.distributed's "nccl" backend on GPUs, "gloo" in CPU tests) make the result equal to one device
running the whole batch:
  1. SyncBN forward/backward: all-reduce(SUM) of a [2C+1] fp64 vector per layer;
  2. loss: all-reduce(SUM) of [sum w*nll, sum w, batch] so each rank divides by the global normaliser;
  3. gradients: bucketed all-reduce(SUM) (each rank already divided by the global normaliser).
"""
import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend=None):
    """Join the job torchrun (or the bench driver) started: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or (dist.is_available() and dist.is_initialized()):
        return rank(), world_size()
    if backend is None:
        # SEGK_DIST_BACKEND=gloo: rehearse a multi-rank job on a box with fewer GPUs than ranks (RCCL refuses two
        # ranks on one device)
        backend = os.environ.get("SEGK_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def _host_staged(t):
    """gloo moves bytes over sockets; device tensors are staged through host memory here rather than by
    gloo's own device path, which stalls intermittently when several ranks share one GPU."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce_sum_(t):
    """In-place SUM all-reduce on the current stream; identity on one rank."""
    if is_distributed():
        if _host_staged(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class GradientBuckets:
    """Bucketed SUM all-reduce of parameter gradients (replaces DataParallel's reduce-add to GPU 0).

    Parameters are packed, in reverse registration order (the order backward produces them), into flat
    fp32 buckets of about `bucket_mb`; `reduce()` copies grads in, all-reduces each bucket asynchronously
    and copies the sums back.  SUM, not mean: every rank's loss is already divided by the global
    normaliser, so summed gradients equal the single-device gradients of the whole batch.

    `reduce_flat()` is the form for `segk.optim.SGD.step(flat_grads=buckets)`: one HIP launch per bucket packs
    the gradients (fp32 or bf16, in storage order) into the bucket, the all-reduced sums stay there and the
    optimizer reads them through `view_of(p)`; no per-parameter copy kernels in either direction.
    """

    def __init__(self, params, bucket_mb=32.0, overlap=False):
        """overlap=True: a hook on every parameter starts a bucket's pack + all-reduce once its last gradient has been
        accumulated, so the collectives run under the rest of backward; reduce_flat() then only starts the buckets
        still missing (parameters without a gradient) and waits.  Collectives are always issued in bucket order
        (bucket k starts only after buckets 0..k-1), so every rank enqueues the same sequence whatever order its
        hooks fire in.  A further backward() in the same step (gradient accumulation) marks the buckets it touches;
        reduce_flat() packs and reduces those again from the accumulated gradients."""
        self.params = [p for p in params if p.requires_grad]
        limit = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []
        cur, size = [], 0
        for p in reversed(self.params):
            if cur and size + p.numel() > limit:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
        if cur:
            self.buckets.append(cur)
        self._flat = None
        self._gather = None
        self._works = {}
        self._seen = set()
        self._pending = [len(b) for b in self.buckets]
        self._ready = set()    # buckets whose gradients are complete but whose turn has not come
        self._next = 0         # first bucket not started yet (hook path)
        self._dirty = set()    # buckets that received gradients again after their all-reduce was started
        self._bucket_of = {id(p): k for k, b in enumerate(self.buckets) for p in b}
        if overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def _on_grad(self, p):
        k = self._bucket_of[id(p)]
        if id(p) in self._seen:   # a further backward in the same step: p.grad has grown since the bucket was packed
            if k in self._works:
                self._dirty.add(k)
            return
        self._seen.add(id(p))
        self._pending[k] -= 1
        if self._pending[k] == 0:
            self._ready.add(k)
            while self._next in self._ready:
                self._start(self._next)
                self._next += 1

    def view_of(self, p):
        """The slice of a flat bucket that holds parameter p's (summed) gradient after reduce_flat()."""
        self._buffers()
        return self._views[id(p)]

    def _tables(self):
        import numpy as np

        from . import _lib
        from ._tables import StagedTable, base_ptrs, chunks_of
        flats = self._buffers()
        if self._gather is None:
            entry = np.dtype([("src", "<u8"), ("dst", "<u8"), ("n", "<i4"), ("dtype", "<i4")])
            self._gather = []
            for flat, bucket in zip(flats, self.buckets):
                tidx, offs, ns = chunks_of(bucket)
                t = np.zeros(len(tidx), dtype=entry)
                t["dst"] = base_ptrs([self._views[id(p)] for p in bucket], tidx) + offs * np.uint64(4)
                t["n"] = ns
                t["dtype"] = np.asarray([_lib.BF16 if p.dtype == torch.bfloat16 else _lib.F32 for p in bucket],
                                        dtype=np.int32)[tidx]
                esz = np.asarray([p.element_size() for p in bucket], dtype=np.uint64)[tidx]
                self._gather.append((StagedTable(t, flat.device), tidx, offs * esz))
        return flats

    def _start(self, k):
        """Pack bucket k from the gradients as they are now and start its all-reduce."""
        import ctypes

        import numpy as np

        from . import _lib
        flat, bucket = self._tables()[k], self.buckets[k]
        staged, tidx, goff = self._gather[k]
        ptrs = np.zeros(len(bucket), dtype=np.uint64)
        for i, p in enumerate(bucket):
            g = p.grad
            if g is None:
                continue
            if g.dtype != p.dtype or g.stride() != p.stride():
                raise RuntimeError("GradientBuckets: gradient of a %s parameter with strides %s is %s with strides %s"
                                   % (p.dtype, p.stride(), g.dtype, g.stride()))
            ptrs[i] = g.data_ptr()
        per_chunk = ptrs[tidx]
        live = per_chunk != 0
        if not live.all():
            flat.zero_()   # a parameter without a gradient on this rank contributes zeros
        staged.table["src"] = per_chunk + goff
        n = staged.upload(None if live.all() else live)
        if n:
            _lib.call("segk_gather_f32", ctypes.c_void_p(staged.ptr()), n,
                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if not is_distributed():
            self._works[k] = None
        elif _host_staged(flat):
            all_reduce_sum_(flat)
            self._works[k] = None
        else:
            self._works[k] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def reduce_flat(self):
        """Pack every gradient into its bucket (always, also on one rank), all-reduce the buckets and wait."""
        for k in range(len(self.buckets)):
            if k not in self._works:
                self._start(k)
        for k in sorted(self._dirty):   # packed before the last backward of this step: wait, pack and reduce again
            if self._works[k] is not None:
                self._works[k].wait()
            self._start(k)
        for w in self._works.values():
            if w is not None:
                w.wait()
        self._works = {}
        self._seen = set()
        self._pending = [len(b) for b in self.buckets]
        self._ready = set()
        self._next = 0
        self._dirty = set()
        return self

    def _buffers(self):
        if self._flat is None:
            self._flat = [torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device)
                          for b in self.buckets]
            self._views = {}
            for flat, bucket in zip(self._flat, self.buckets):
                off = 0
                for p in bucket:
                    self._views[id(p)] = flat[off:off + p.numel()]
                    off += p.numel()
        return self._flat

    def reduce(self):
        if not is_distributed():
            return
        flats = self._buffers()
        works = []
        for flat, bucket in zip(flats, self.buckets):
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
                off += n
            if _host_staged(flat):
                all_reduce_sum_(flat)
                works.append(None)
            else:
                works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for w, flat, bucket in zip(works, flats, self.buckets):
            if w is not None:
                w.wait()
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[off:off + n].reshape(p.shape).clone()
                else:
                    p.grad.copy_(flat[off:off + n].reshape(p.shape))
                off += n
