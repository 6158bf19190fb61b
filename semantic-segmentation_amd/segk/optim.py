"""SGD with momentum and weight decay on the device: one HIP launch per parameter group.

Same arithmetic and constructor shape as ``torch.optim.SGD(groups, momentum=0.9, weight_decay=5e-4,
nesterov=False)`` as every trainer of the reference builds it (/root/reference/train.py:56-61, groups from
``get_1x_lr_params`` / ``get_10x_lr_params``), so ``LR_Scheduler`` (utils/lr_scheduler.py:50-73) keeps driving
``param_groups[i]['lr']``.

``low_precision`` names parameters that are to live in bf16 (the feeder's convolution weights when the feeder
runs in bf16): the optimizer keeps their fp32 master copy, takes their bf16 gradient as it is and writes the
rounded weight back in the same pass, so no per-step cast kernels remain.  ``full_precision_state_dict`` returns
a checkpoint with the fp32 masters in place of the bf16 copies.

A step recorded in a HIP graph (``segk.graph.GraphedTrainStep``) reads lr / momentum / weight decay from device
memory (``segk_sgd_step_dev``); ``sync_hyper()`` writes the current ``param_groups`` values there before a replay,
so the schedule keeps working although a recorded launch keeps its by-value arguments.

``step(flat_grads=buckets)`` reads the gradients from the flat fp32 buckets of
``segk.parallel.GradientBuckets.reduce_flat()`` instead of ``p.grad`` (data-parallel runs: the all-reduced sums
are consumed where they lie, nothing is copied back).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import BF16, F32, call
from ._tables import StagedTable, base_ptrs, chunks_of, flat_storage

_ENTRY = np.dtype([("param", "<u8"), ("grad", "<u8"), ("momentum", "<u8"), ("master", "<u8"), ("n", "<i4"),
                   ("dtype", "<i4")])
assert _ENTRY.itemsize == 40


class _Group:
    """Chunk table of one parameter group; everything but the gradient addresses is fixed."""

    def __init__(self, params, state, flat_grads=None):
        self.params = params
        self.tidx, self.offs, ns = chunks_of(params)
        esz = np.asarray([p.element_size() for p in params], dtype=np.uint64)[self.tidx]
        self.goff = self.offs * esz
        t = np.zeros(len(self.tidx), dtype=_ENTRY)
        t["param"] = base_ptrs(params, self.tidx) + self.goff
        t["momentum"] = base_ptrs([state[p]["momentum_buffer"] for p in params], self.tidx) + self.offs * np.uint64(4)
        t["master"] = base_ptrs([state[p]["master"] for p in params], self.tidx) + self.offs * np.uint64(4)
        t["n"] = ns
        pd = np.asarray([BF16 if p.dtype == torch.bfloat16 else F32 for p in params], dtype=np.int32)[self.tidx]
        self.staged = StagedTable(t, params[0].device)
        self.flat = flat_grads is not None
        if self.flat:   # gradients at fixed addresses inside fp32 buckets: upload once
            t["grad"] = base_ptrs([flat_grads.view_of(p) for p in params], self.tidx) + self.offs * np.uint64(4)
            t["dtype"] = pd | (F32 << 8)
            self.rows = self.staged.upload()
        else:
            t["dtype"] = pd | (pd << 8)

    def refresh(self):
        """-> rows to launch; fills in this step's gradient addresses unless they are fixed."""
        if self.flat:
            return self.rows
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
        self.staged.table["grad"] = per_chunk + self.goff
        live = per_chunk != 0
        return self.staged.upload(None if live.all() else live)


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=0.01, momentum=0.0, weight_decay=0.0, nesterov=False, dampening=0.0,
                 low_precision=()):
        if nesterov or dampening != 0.0:
            raise ValueError("segk.optim.SGD: nesterov / dampening are not implemented (the reference uses neither)")
        # the group keys torch.optim.SGD writes, so that its load_state_dict accepts a checkpoint from here
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0.0, weight_decay=weight_decay,
                                      nesterov=False, maximize=False, foreach=None, differentiable=False,
                                      fused=None))
        _lib.load()
        low = {id(p) for p in low_precision}
        self._groups = None
        self._groups_src = None
        self._recorded_groups = []   # chunk tables a recorded HIP graph replays: never freed while this object lives
        self._hyper_dev = None    # [groups, 4] fp32 on the device, read by recorded steps
        self._hyper_host = None   # pinned ring of the same shape
        self._hyper_turn = 0
        self._hyper_last = None
        for g in self.param_groups:
            g["params"] = [p for p in g["params"] if p.requires_grad]
            for p in g["params"]:
                if not p.is_cuda:
                    raise RuntimeError("segk.optim.SGD: parameters must live on the GPU")
                if p.dtype not in (torch.float32, torch.bfloat16):
                    raise RuntimeError("segk.optim.SGD: unsupported parameter dtype %s" % p.dtype)
                st = self.state[p]
                if id(p) in low and p.dtype == torch.float32:
                    st["master"] = flat_storage(p.data.detach().clone())   # keeps the parameter's strides
                    p.data = p.data.to(torch.bfloat16)
                elif p.dtype == torch.bfloat16:
                    st["master"] = flat_storage(p.data.detach().float())
                else:
                    st["master"] = flat_storage(p.data)                    # fp32: updated in place
                flat_storage(p.data)
                st["momentum_buffer"] = torch.zeros(p.numel(), dtype=torch.float32, device=p.device)
        # convolution weights keep their packed operands between uses; step() renews them in one launch
        from . import ops
        ops.cache_weight_packs([p for g in self.param_groups for p in g["params"]], owner=self)

    @torch.no_grad()
    def step(self, closure=None, flat_grads=None):
        loss = closure() if closure is not None else None
        recording = torch.cuda.is_current_stream_capturing()
        if recording and (self._groups is None or self._groups_src is not flat_grads or self._hyper_dev is None):
            # the chunk tables and their pinned staging buffers cannot be allocated while a stream is capturing
            raise RuntimeError("segk.optim.SGD: run one step eagerly and call sync_hyper() before recording a step "
                               "in a HIP graph")
        if self._groups is None or self._groups_src is not flat_grads:
            self._groups = [_Group(g["params"], self.state, flat_grads) for g in self.param_groups if g["params"]]
            self._groups_src = flat_grads
        if recording and not any(g is self._groups for g in self._recorded_groups):
            self._recorded_groups.append(self._groups)   # the graph holds these tables' device addresses
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        live_groups = [g for g in self.param_groups if g["params"]]
        for k, (grp, hp) in enumerate(zip(self._groups, live_groups)):
            n = grp.refresh()
            if not n:
                continue
            if recording:
                call("segk_sgd_step_dev", ctypes.c_void_p(grp.staged.ptr()), n,
                     ctypes.c_void_p(self._hyper_dev[k].data_ptr()), stream)
            else:
                call("segk_sgd_step", ctypes.c_void_p(grp.staged.ptr()), n, float(hp["lr"]), float(hp["momentum"]),
                     float(hp["weight_decay"]), stream)
        # the kernels above wrote the weights through raw pointers: torch's version counters did not move
        from . import ops
        ops.repack_cached_weights(owner=self)
        return loss

    def after_graph_replay(self):
        """For whoever replays a HIP graph that contains step(): kept weight operands the recorded launch does not
        renew are marked stale."""
        from . import ops
        ops.invalidate_unrecorded_packs(owner=self)

    def sync_hyper(self):
        """Write lr / momentum / weight decay of every (non-empty) parameter group to the device table that recorded
        steps read.  Stream-ordered on the current stream, no host synchronisation; a no-op while nothing changed."""
        live_groups = [g for g in self.param_groups if g["params"]]
        vals = [[float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), 0.0] for g in live_groups]
        if self._hyper_dev is None:
            dev = live_groups[0]["params"][0].device
            self._hyper_dev = torch.zeros((len(vals), 4), dtype=torch.float32, device=dev)
            self._hyper_host = [torch.zeros((len(vals), 4), dtype=torch.float32).pin_memory() for _ in range(8)]
        elif vals == self._hyper_last:
            return
        # a ring of pinned rows: a copy still in flight keeps its own source (eight steps of slack, and the copies
        # are stream-ordered behind one another)
        host = self._hyper_host[self._hyper_turn]
        self._hyper_turn = (self._hyper_turn + 1) % len(self._hyper_host)
        host.copy_(torch.tensor(vals, dtype=torch.float32))
        self._hyper_dev.copy_(host, non_blocking=True)
        self._hyper_last = vals

    # ---- checkpoints (utils/saver.py:33-60 stores optimizer.state_dict(); train.py:124 loads it back) ----------
    @staticmethod
    def _shaped(flat, p):
        """The flat fp32 state of parameter p seen with p's shape and strides."""
        return torch.as_strided(flat, p.shape, p.stride())

    def state_dict(self):
        """The layout torch.optim.SGD writes: per parameter a `momentum_buffer` of the parameter's shape (fp32), so
        either optimizer loads the other's checkpoint.  A bf16-held parameter also stores its fp32 `master` (an
        entry torch.optim.SGD ignores); an fp32 parameter IS its master and stores none."""
        packed = super().state_dict()["param_groups"]
        state = {}
        idx = 0
        for g in self.param_groups:
            for p in g["params"]:
                st = self.state[p]
                entry = {"momentum_buffer": self._shaped(st["momentum_buffer"], p).clone()}
                if p.dtype == torch.bfloat16:
                    entry["master"] = self._shaped(st["master"], p).clone()
                state[idx] = entry
                idx += 1
        return {"state": state, "param_groups": packed}

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Copies the saved values INTO the existing fp32 state tensors (their addresses are in the kernels' tables
        and, for fp32 parameters, the master is the parameter itself), whatever dtype or device they were saved
        with.  A parameter without a saved entry gets a zero momentum buffer (= torch's 'no buffer yet').  A
        bf16-held parameter takes its master from the checkpoint when it has one; without one (a torch.optim.SGD
        checkpoint) the existing master stays when the parameter still equals its bf16 rounding
        (`load_full_precision_state_dict` restored it exactly) and is re-derived from the parameter otherwise (the
        weights were loaded into the model after the master was last set)."""
        for hook in self._optimizer_load_state_dict_pre_hooks.values():
            changed = hook(self, state_dict)
            if changed is not None:
                state_dict = changed
        groups = state_dict["param_groups"]
        saved = state_dict["state"]
        if len(groups) != len(self.param_groups):
            raise ValueError("segk.optim.SGD: checkpoint has %d parameter groups, optimizer has %d"
                             % (len(groups), len(self.param_groups)))
        for g, sg in zip(self.param_groups, groups):
            if len(sg["params"]) != len(g["params"]):
                raise ValueError("segk.optim.SGD: a parameter group holds %d parameters in the checkpoint, %d here"
                                 % (len(sg["params"]), len(g["params"])))
            if sg.get("nesterov") or sg.get("dampening", 0) != 0 or sg.get("maximize"):
                raise ValueError("segk.optim.SGD: the checkpoint was written with nesterov / dampening / maximize")
            for key in ("lr", "momentum", "weight_decay"):
                if key in sg:
                    g[key] = sg[key]
            for p, idx in zip(g["params"], sg["params"]):
                st = self.state[p]
                entry = saved.get(idx, saved.get(str(idx)))
                mom = entry.get("momentum_buffer") if entry else None
                dst = self._shaped(st["momentum_buffer"], p)
                if mom is None:
                    dst.zero_()
                else:
                    if tuple(mom.shape) != tuple(p.shape):
                        mom = mom.reshape(p.shape)   # raises on a size mismatch
                    dst.copy_(mom)
                if p.dtype == torch.bfloat16:
                    master = self._shaped(st["master"], p)
                    if entry and entry.get("master") is not None:
                        master.copy_(entry["master"].reshape(p.shape))
                    elif not torch.equal(master.to(torch.bfloat16), p.data):
                        # no master in the checkpoint (one written by torch.optim.SGD) and the bf16 copy is not the
                        # rounded master any more: the weights were loaded into the parameter, they are what counts.
                        # A master that still rounds to the parameter (load_full_precision_state_dict restored it
                        # exactly) is left alone.
                        master.copy_(p.data)
                    p.data.copy_(master)
        self._groups = None   # rebuild the chunk tables on the next step (tables of a recorded step stay alive)
        self._groups_src = None
        from . import ops
        ops.invalidate_weight_packs()   # p.data.copy_ above moved no version counter
        for hook in self._optimizer_load_state_dict_post_hooks.values():
            hook(self)

    @torch.no_grad()
    def load_full_precision_state_dict(self, model, state_dict, strict=True):
        """model.load_state_dict(state_dict) for a model whose weights this optimizer holds in bf16: the fp32 values
        of the checkpoint go to the masters unrounded, the bf16 copies are re-derived from them."""
        result = model.load_state_dict(state_dict, strict=strict)
        by_ptr = {p.data_ptr(): p for g in self.param_groups for p in g["params"] if p.dtype == torch.bfloat16}
        for k, v in model.state_dict().items():
            p = by_ptr.get(v.data_ptr()) if torch.is_tensor(v) else None
            if p is not None and k in state_dict:
                master = self._shaped(self.state[p]["master"], p)
                master.copy_(state_dict[k].reshape(p.shape))
                p.data.copy_(master)
        from . import ops
        ops.invalidate_weight_packs()
        return result

    def full_precision_state_dict(self, model):
        """model.state_dict() with every bf16-held parameter replaced by its fp32 master."""
        sd = model.state_dict()
        by_ptr = {p.data_ptr(): p for g in self.param_groups for p in g["params"] if p.dtype == torch.bfloat16}
        for k, v in list(sd.items()):
            p = by_ptr.get(v.data_ptr()) if torch.is_tensor(v) else None
            if p is not None and v.dtype == torch.bfloat16:
                sd[k] = torch.as_strided(self.state[p]["master"], p.shape, p.stride()).contiguous().clone()
        return sd
