"""One process per GPU: the exchanges that replace nn.DataParallel + the queue-based SyncBN master.

The reference scatters a batch over replica threads, reduces BN statistics through Python queues
(/root/reference/modeling/sync_batchnorm/comm.py:97-125, batchnorm.py:90-111), gathers logits on GPU 0,
computes one loss on the gathered batch (/root/reference/train.py:148-153) and lets DataParallel
reduce-add gradients.  Here every rank owns its shard and three collectives (RCCL through
torch.distributed's "nccl" backend on GPUs, "gloo" in CPU tests) make the result equal to one device
running the whole batch:
  1. SyncBN forward/backward: all-reduce(SUM) of a [2C+1] fp32 vector per layer;
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
    """

    def __init__(self, params, bucket_mb=32.0):
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

    def _buffers(self):
        if self._flat is None:
            self._flat = [torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device)
                          for b in self.buckets]
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
