"""SGD with momentum and weight decay on the device: one HIP launch per parameter group.

Same arithmetic and constructor shape as ``torch.optim.SGD(groups, momentum=0.9, weight_decay=5e-4,
nesterov=False)`` as every trainer of the reference builds it (/root/reference/train.py:56-61, groups from
``get_1x_lr_params`` / ``get_10x_lr_params``), so ``LR_Scheduler`` (utils/lr_scheduler.py:50-73) keeps driving
``param_groups[i]['lr']``.

``low_precision`` names parameters that are to live in bf16 (the feeder's convolution weights when the feeder
runs in bf16): the optimizer keeps their fp32 master copy, takes their bf16 gradient as it is and writes the
rounded weight back in the same pass, so no per-step cast kernels remain.  ``full_precision_state_dict`` returns
a checkpoint with the fp32 masters in place of the bf16 copies.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import BF16, F32, call

CHUNK = 65536  # elements per workgroup
_ENTRY = np.dtype([("param", "<u8"), ("grad", "<u8"), ("momentum", "<u8"), ("master", "<u8"), ("n", "<i4"),
                   ("dtype", "<i4")])
assert _ENTRY.itemsize == 40
_RING = 4  # pinned staging tables in flight


class _Group:
    """Chunk table of one parameter group; everything but the gradient addresses is fixed."""

    def __init__(self, params, state):
        self.params = params
        tidx, offs, ns = [], [], []
        for k, p in enumerate(params):
            n = p.numel()
            for o in range(0, n, CHUNK):
                tidx.append(k)
                offs.append(o)
                ns.append(min(CHUNK, n - o))
        self.tidx = np.asarray(tidx, dtype=np.int64)
        self.offs = np.asarray(offs, dtype=np.uint64)
        esz = np.asarray([p.element_size() for p in params], dtype=np.uint64)[self.tidx]
        self.goff = self.offs * esz
        self.table = np.zeros(len(tidx), dtype=_ENTRY)
        base = lambda ts: np.asarray([t.data_ptr() for t in ts], dtype=np.uint64)[self.tidx]  # noqa: E731
        self.table["param"] = base(params) + self.goff
        self.table["momentum"] = base([state[p]["momentum_buffer"] for p in params]) + self.offs * np.uint64(4)
        self.table["master"] = base([state[p]["master"] for p in params]) + self.offs * np.uint64(4)
        self.table["n"] = np.asarray(ns, dtype=np.int32)
        self.table["dtype"] = np.asarray([BF16 if p.dtype == torch.bfloat16 else F32 for p in params],
                                         dtype=np.int32)[self.tidx]
        dev = params[0].device
        self.dev_table = torch.empty(self.table.nbytes, dtype=torch.uint8, device=dev)
        self.host = [torch.empty(self.table.nbytes, dtype=torch.uint8).pin_memory() for _ in range(_RING)]
        self.events = [None] * _RING
        self.turn = 0

    def upload(self):
        """Fill in this step's gradient addresses and copy the table to the device (stream-ordered)."""
        ptrs = np.zeros(len(self.params), dtype=np.uint64)
        for k, p in enumerate(self.params):
            g = p.grad
            if g is None:  # parameter unused this step: torch.optim.SGD skips it as well
                continue
            if g.dtype != p.dtype or g.stride() != p.stride():
                raise RuntimeError("segk.optim.SGD: gradient of a %s parameter with strides %s is %s with strides %s"
                                   % (p.dtype, p.stride(), g.dtype, g.stride()))
            ptrs[k] = g.data_ptr()
        per_chunk = ptrs[self.tidx]
        self.table["grad"] = per_chunk + self.goff
        live = per_chunk != 0
        t = self.table if live.all() else self.table[live]
        n = len(t)
        if n == 0:
            return 0
        slot = self.turn
        self.turn = (self.turn + 1) % _RING
        if self.events[slot] is not None:
            self.events[slot].synchronize()   # the copy that last used this staging buffer has long finished
        raw = torch.from_numpy(np.ascontiguousarray(t).view(np.uint8).reshape(-1))
        self.host[slot][:raw.numel()].copy_(raw)
        self.dev_table[:raw.numel()].copy_(self.host[slot][:raw.numel()], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[slot] = ev
        return n


def _flat_storage(t):
    """The elements of a dense tensor in storage order (contiguous and channels_last weights alike)."""
    dense = t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
    if not dense:
        raise RuntimeError("segk.optim.SGD: parameter of shape %s, strides %s is not dense" % (tuple(t.shape), t.stride()))
    return torch.as_strided(t, (t.numel(),), (1,))


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=0.01, momentum=0.0, weight_decay=0.0, nesterov=False, dampening=0.0,
                 low_precision=()):
        if nesterov or dampening != 0.0:
            raise ValueError("segk.optim.SGD: nesterov / dampening are not implemented (the reference uses neither)")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        _lib.load()
        low = {id(p) for p in low_precision}
        self._groups = None
        for g in self.param_groups:
            g["params"] = [p for p in g["params"] if p.requires_grad]
            for p in g["params"]:
                if not p.is_cuda:
                    raise RuntimeError("segk.optim.SGD: parameters must live on the GPU")
                if p.dtype not in (torch.float32, torch.bfloat16):
                    raise RuntimeError("segk.optim.SGD: unsupported parameter dtype %s" % p.dtype)
                st = self.state[p]
                if id(p) in low and p.dtype == torch.float32:
                    st["master"] = _flat_storage(p.data.detach().clone())   # keeps the parameter's strides
                    p.data = p.data.to(torch.bfloat16)
                elif p.dtype == torch.bfloat16:
                    st["master"] = _flat_storage(p.data.detach().float())
                else:
                    st["master"] = _flat_storage(p.data)                    # fp32: updated in place
                _flat_storage(p.data)
                st["momentum_buffer"] = torch.zeros(p.numel(), dtype=torch.float32, device=p.device)

    def _build(self):
        self._groups = [_Group(g["params"], self.state) for g in self.param_groups if g["params"]]

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._groups is None:
            self._build()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        live_groups = [g for g in self.param_groups if g["params"]]
        for grp, hp in zip(self._groups, live_groups):
            n = grp.upload()
            if n:
                call("segk_sgd_step", ctypes.c_void_p(grp.dev_table.data_ptr()), n, float(hp["lr"]),
                     float(hp["momentum"]), float(hp["weight_decay"]), stream)
        return loss

    def full_precision_state_dict(self, model):
        """model.state_dict() with every bf16-held parameter replaced by its fp32 master."""
        sd = model.state_dict()
        by_ptr = {p.data_ptr(): p for g in self.param_groups for p in g["params"] if p.dtype == torch.bfloat16}
        for k, v in list(sd.items()):
            p = by_ptr.get(v.data_ptr()) if torch.is_tensor(v) else None
            if p is not None and v.dtype == torch.bfloat16:
                sd[k] = torch.as_strided(self.state[p]["master"], p.shape, p.stride()).contiguous().clone()
        return sd
