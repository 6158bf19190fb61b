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
The small exchanges (1, 2 and the evaluator's matrix) take the latency path when it is available: one-shot
all-reduces through peer-mapped mailboxes (connect_mailboxes, csrc/comm.hip), launches of this library on the
compute stream that a HIP graph can record.  config.sync_exchange: "auto" (default: mailboxes when they connect and
pass their self-test, else torch.distributed), "mailbox" (required), "dist" (never).
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
    from . import config
    mode = config.sync_exchange()
    if mode != "dist" and torch.cuda.is_available():
        # (the device this rank computes on must be current: the mailbox is allocated there)
        if backend != "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        ok = connect_mailboxes()
        if mode == "mailbox" and not ok:
            raise RuntimeError("SEGK_SYNC_EXCHANGE=mailbox, but the mailbox path is not available: %s" % _MAILBOX["why"])
    return dist.get_rank(), dist.get_world_size()


# ---- the latency path: one-shot all-reduce through peer-mapped mailboxes (csrc/comm.hip) -------------------------
_MAILBOX = {"on": False, "grads": False, "why": "not tried"}


def mailbox_active():
    return _MAILBOX["on"]


def exchange_description():
    """What carries the small exchanges of this job (for logs and the benchmark line)."""
    if not is_distributed():
        return None
    if _MAILBOX["on"]:
        return "peer-mapped mailboxes (segk_mailbox_allreduce), gradients: %s" % (
            "mailboxes" if _MAILBOX["grads"] else dist.get_backend())
    return "torch.distributed %s (mailboxes: %s)" % (dist.get_backend(), _MAILBOX["why"])


def connect_mailboxes(timeout_ms=0, self_test=True, gradients=None):
    """Create this rank's mailbox, exchange the IPC handles through torch.distributed, map the peers' mailboxes and
    (self_test) run a few exchanges of known vectors whose sums every rank checks bit for bit.  Returns whether the
    mailbox path is in use afterwards; any failure leaves the job on torch.distributed's collectives and says why
    in exchange_description().  gradients=True also sends the gradient buckets through the mailboxes (in chunks: a
    rehearsal path for ranks that share one device, where RCCL cannot run)."""
    import ctypes

    from . import _lib
    if not is_distributed() or not torch.cuda.is_available():
        _MAILBOX["why"] = "one rank or no GPU"
        return False
    if _MAILBOX["on"]:
        return True
    world, me = dist.get_world_size(), dist.get_rank()
    ok, why = True, ""
    handle = ctypes.create_string_buffer(64)
    try:
        _lib.call("segk_mailbox_create", world, handle)
    except _lib.SegkError as e:
        ok, why = False, str(e)
    gathered = [None] * world
    dist.all_gather_object(gathered, (ok, handle.raw))
    if not all(g[0] for g in gathered):
        _MAILBOX["why"] = why or "another rank could not create its mailbox"
        if ok:
            _lib.call("segk_mailbox_destroy")
        return False
    try:
        _lib.call("segk_mailbox_connect", me, world, ctypes.c_char_p(b"".join(g[1] for g in gathered)), int(timeout_ms))
    except _lib.SegkError as e:
        ok, why = False, str(e)
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    if not all(flags):
        _MAILBOX["why"] = why or "another rank could not map the mailboxes"
        return False
    _MAILBOX["on"] = True
    if self_test:
        # the ranks have just met in all_gather_object: a peer that does not answer within seconds never will
        _lib.call("segk_mailbox_set_timeout", 5000)
        good = _mailbox_self_test(me, world)
        _lib.call("segk_mailbox_set_timeout", int(timeout_ms))
        flags = [None] * world
        dist.all_gather_object(flags, good)
        if not all(flags):
            _MAILBOX["on"] = False
            _MAILBOX["why"] = "self-test failed on rank(s) %s" % [i for i, f in enumerate(flags) if not f]
            return False
    if gradients is None:
        gradients = os.environ.get("SEGK_GRAD_EXCHANGE", "") == "mailbox"
    _MAILBOX["grads"] = bool(gradients)
    _MAILBOX["why"] = "connected"
    return True


def _mailbox_self_test(me, world, rounds=6):
    """Exchanges of vectors every rank can predict (integers, exact in fp64 / fp32), of the lengths the batch norms
    use and one that needs several chunks; False when a sum is wrong or a wait timed out."""
    import ctypes

    from . import _lib
    dev = torch.device("cuda", torch.cuda.current_device())
    try:
        for k, n in enumerate([1, 97, 513, 4097, 3, 10000][:rounds]):
            base = torch.arange(n, dtype=torch.float64, device=dev) % 1021 + k
            mine = base * (me + 1)
            want = base * (world * (world + 1) // 2)
            if k % 2:
                mine, want = mine.float(), want.float()
            all_reduce_sum_(mine)
            if not torch.equal(mine, want):
                return False
        return _lib.call("segk_mailbox_status") >= 0
    except _lib.SegkError:
        return False


def check_exchange():
    """Synchronous: raises when an in-kernel wait of the mailbox path ran out of time (a rank is gone, or the ranks
    issued different sequences of exchanges).  Returns the number of exchanges completed on this rank."""
    from . import _lib
    return _lib.call("segk_mailbox_status") if _MAILBOX["on"] else 0


def step_is_capturable():
    """Whether every exchange of a training step can be recorded in a HIP graph: true when batch-norm statistics,
    loss sums and gradient buckets all travel through the mailboxes (launches of this library on the capturing
    stream); torch.distributed's collectives are left out of graphs here."""
    return not is_distributed() or (_MAILBOX["on"] and _MAILBOX["grads"])


def _mailbox_all_reduce(t):
    import ctypes

    from . import _lib
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if t.dtype in (torch.float64, torch.float32) and t.is_contiguous():
        name = "segk_mailbox_allreduce_f64" if t.dtype == torch.float64 else "segk_mailbox_allreduce_f32"
        _lib.call(name, ctypes.c_void_p(t.data_ptr()), t.numel(), stream)
        return t
    if t.dtype == torch.int64:
        # counts (the confusion matrix): exact in fp64 up to 2^53
        w = t.to(torch.float64).contiguous()
        _lib.call("segk_mailbox_allreduce_f64", ctypes.c_void_p(w.data_ptr()), w.numel(), stream)
        t.copy_(w.to(torch.int64))
        return t
    w = t.to(torch.float64).contiguous()
    _lib.call("segk_mailbox_allreduce_f64", ctypes.c_void_p(w.data_ptr()), w.numel(), stream)
    t.copy_(w)
    return t


def assert_same_on_all_ranks(t, what):
    """Debug check (config.set_debug_sync_stats): gathers the tensor from every rank and compares bit for bit; ranks
    that hold different values raise.  Synchronises with the host."""
    if not is_distributed():
        return
    mine = t.detach().to("cpu", torch.float64).contiguous()
    got = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    if dist.get_backend() == "gloo":
        dist.all_gather(got, mine)
    else:
        dev = [torch.empty_like(t, dtype=torch.float64) for _ in got]
        dist.all_gather(dev, t.detach().to(torch.float64).contiguous())
        got = [d.cpu() for d in dev]
    bad = [r for r, g in enumerate(got) if not torch.equal(g.view(torch.int64), got[0].view(torch.int64))]
    if bad:
        raise RuntimeError("%s differ between ranks: rank(s) %s do not hold rank 0's values (rank %d reporting)"
                           % (what, bad, dist.get_rank()))


def _host_staged(t):
    """gloo moves bytes over sockets; device tensors are staged through host memory here rather than by
    gloo's own device path, which stalls intermittently when several ranks share one GPU."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _host_all_reduce(t):
    h = t.cpu()
    dist.all_reduce(h, op=dist.ReduceOp.SUM)
    t.copy_(h)
    return t


def all_reduce_sum_(t):
    """In-place SUM all-reduce on the current stream; identity on one rank."""
    if is_distributed():
        if _MAILBOX["on"] and t.is_cuda:
            _mailbox_all_reduce(t)
        elif _host_staged(t):
            _host_all_reduce(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


_HOOKED_BUCKETS = [0]   # GradientBuckets(overlap=True) created in this process


def grad_hooks_active():
    """Whether some GradientBuckets start their all-reduces from parameter hooks: weight gradients must then reach
    ``weight.grad`` through autograd's accumulation nodes (segk.functional.weight_grad does not defer them)."""
    return _HOOKED_BUCKETS[0] > 0


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
        self._hooked = bool(overlap)
        if overlap:
            _HOOKED_BUCKETS[0] += 1
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

    def _start(self, k, exchange=True):
        """Pack bucket k from the gradients as they are now and (exchange) start its all-reduce."""
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
        if exchange:
            self._exchange(k)

    def _exchange(self, k):
        """Start the all-reduce of the packed bucket k."""
        flat = self._buffers()[k]
        if not is_distributed():
            self._works[k] = None
        elif _MAILBOX["on"] and _MAILBOX["grads"]:
            _mailbox_all_reduce(flat)
            self._works[k] = None
        elif _host_staged(flat):
            _host_all_reduce(flat)   # (the buckets stay on torch.distributed even when the mailboxes carry the statistics)
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

    def pack_flat(self):
        """First half of reduce_flat(): every gradient into its bucket, nothing exchanged (launches of this library
        only - the half a recorded graph can hold when the buckets travel through torch.distributed)."""
        if self._hooked:
            raise RuntimeError("GradientBuckets(overlap=True) packs and reduces from hooks; use overlap=False")
        for k in range(len(self.buckets)):
            self._start(k, exchange=False)
        return self

    def exchange_flat(self):
        """Second half: all-reduce the packed buckets and wait."""
        for k in range(len(self.buckets)):
            self._exchange(k)
        for w in self._works.values():
            if w is not None:
                w.wait()
        self._works = {}
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
            if _MAILBOX["on"] and _MAILBOX["grads"]:
                _mailbox_all_reduce(flat)
                works.append(None)
            elif _host_staged(flat):
                _host_all_reduce(flat)
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
