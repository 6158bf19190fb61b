"""Tensor-level wrappers over the C ABI: argument checks, workspace allocation, one call each.

Pixel-major activations are 2-D torch views ``[M, C]`` with strides ``(ld, 1)`` (so a channel
slice of a wider concat buffer is just ``buf[:, a:b]``).  Everything here requires CUDA/HIP
tensors; nothing computes on the host.
"""
import ctypes
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import BF16, F32, NCHW, PM, ConvArgs, GemmArgs, GemmTnArgs, WgradArgs, WgradGroupArgs, call
from ._tables import StagedTable


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("segk supports float32 and bfloat16 tensors, got %s" % (t.dtype,))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream (the raw getter is ~20x cheaper than building a Stream object)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _row_ptrs(coef):
    """Pointers to the four rows (mean, inv_std, scale, shift) of a contiguous [4, C] float32 tensor."""
    base, step = coef.data_ptr(), coef.shape[1] * 4
    return [ctypes.c_void_p(base + i * step) for i in range(4)]


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("segk kernels need tensors on the GPU (got a %s tensor)" % t.device.type)


def _pm(t):
    """(ld, C) of a pixel-major view."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("pixel-major view must be 2-D with unit channel stride, got shape %s strides %s"
                         % (tuple(t.shape), t.stride()))
    return t.stride(0), t.shape[1]


def round_up(a, b):
    return (a + b - 1) // b * b


class KernelTimer:
    """Optional HIP-event timing of individual launches on the current stream (bench.py's roofline figures).
    Assign an instance to ops.TIMER; the convolution wrappers then bracket each launch with two events and
    record its algorithmic FLOPs.  None (default) = no events are created."""

    def __init__(self):
        self.records = {}   # name -> list of (flops or bytes, start_event, end_event)
        self.hbm = False    # also time the memory-bound batch-norm launches (their record holds algorithmic bytes)
        self.scope = ""     # suffix of the record names while a caller's `scoped()` block is open

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, name, flops, start, inbounds=None):
        """inbounds: the part of `flops` whose kernel taps fall inside the image (convolutions; default: all)."""
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.setdefault(name + self.scope, []).append((flops, start, e, flops if inbounds is None else inbounds))

    def summary(self):
        """name -> dict(launches, flops (sum), flops_inbounds (sum), ms (sum), avg_ms); call after
        torch.cuda.synchronize()."""
        out = {}
        for name, recs in self.records.items():
            ms = sum(r[1].elapsed_time(r[2]) for r in recs)
            fl = sum(r[0] for r in recs)
            out[name] = {"launches": len(recs), "flops": fl, "flops_inbounds": sum(r[3] for r in recs), "ms": ms,
                         "avg_ms": ms / max(len(recs), 1)}
        return out


TIMER = None


class scoped:
    """`with ops.scoped("@feeder"):` files the launches timed inside under `<name>@feeder` (no effect without a
    timer): bench.py keeps the head's convolutions and the feeder's apart."""

    def __init__(self, suffix):
        self.suffix = suffix

    def __enter__(self):
        self.timer = TIMER
        if self.timer is not None:
            self.old = self.timer.scope
            self.timer.scope = self.suffix

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.scope = self.old


def inbounds_fraction(H, W, R, dil):
    """Share of the R x R taps of a stride-1, padding = dilation convolution on an H x W map that read inside the
    image, over all pixels: tap offset (dy, dx) is in bounds for (H - |dy|)(W - |dx|) pixels."""
    if R == 1:
        return 1.0
    rows = H + 2 * max(H - dil, 0)
    cols = W + 2 * max(W - dil, 0)
    return rows * cols / float(R * R * H * W)


def _hbm_start():
    return TIMER.start() if (TIMER is not None and TIMER.hbm) else None


def _hbm_stop(name, nbytes, t0):
    if t0 is not None:
        TIMER.stop(name, float(nbytes), t0)


def _nbytes(*ts):
    """Algorithmic bytes of full-tensor operands (each touched once)."""
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


# ------------------------------------------------------------------ layout ----
def nchw_to_pm(x, out):
    """x: [N,C,H,W] float32 contiguous -> out: pixel-major view [N*H*W, C]."""
    _dev(x, out)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("nchw_to_pm expects a contiguous float32 NCHW tensor")
    N, C, H, W = x.shape
    ld, c = _pm(out)
    if c != C or out.shape[0] != N * H * W:
        raise ValueError("destination shape %s does not match %s" % (tuple(out.shape), tuple(x.shape)))
    call("segk_nchw_to_pm", _ptr(x), _ptr(out), _dt(out), N, C, H, W, ld, _stream())
    return out


def pm_to_nchw(src, N, H, W):
    _dev(src)
    ld, C = _pm(src)
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=src.device)
    call("segk_pm_to_nchw", _ptr(src), _dt(src), ld, _ptr(out), N, C, H, W, _stream())
    return out


def _weight_strides(w):
    """(s_co, s_ci, s_tap) element strides of a dense [Cout, Cin, R, R] weight: contiguous or channels_last."""
    Cout, Cin, R, S = w.shape
    if R != S:
        raise ValueError("square kernels only")
    st = w.stride()
    if R == 1:
        s_tap = 1
    elif st[2] == R * st[3]:
        s_tap = st[3]
    else:
        raise ValueError("weight of strides %s is not dense over its taps" % (st,))
    return st[0], st[1], s_tap


# ---- packed operands of registered weights, kept across a step --------------------------------------------------
# A convolution weight is packed once per use (forward operand, transposed operand of the data gradient): some 110
# launches of ~10 us per training step, each in front of the convolution that waits for it.  Weights registered
# here (segk.optim.SGD registers its parameters) keep their packed operands: pack_weight returns the kept tensor
# while the weight's version counter is the one it was packed at, and repack_cached_weights() - called by the
# optimizer right after it has written the weights through raw pointers, which leaves the counters alone - renews
# all of them in one launch.  A weight changed by a torch operation has a new version and is packed again on use.
_PACK_ITEM = np.dtype([("w", "<u8"), ("dst", "<u8"), ("s_co", "<i8"), ("s_ci", "<i8"), ("s_tap", "<i8"),
                       ("w_dtype", "<i4"), ("dst_dtype", "<i4"), ("Cout", "<i4"), ("Cin", "<i4"), ("R", "<i4"),
                       ("transpose", "<i4"), ("first_block", "<i4"), ("reserved", "<i4")])
assert _PACK_ITEM.itemsize == 72


class _PackEntry:
    __slots__ = ("dst", "version", "layout", "owner")


class _OwnerTable:
    """The device table of one owner's entries (an owner = one optimizer): rebuilt when their set changes."""
    __slots__ = ("table", "blocks", "keys", "recorded")

    def __init__(self):
        self.table, self.blocks, self.keys = None, 0, None
        self.recorded = None   # keys of the table a recorded HIP graph replays, once there is one


class _WeightPackCache:
    def __init__(self):
        self.params = {}     # data_ptr -> (weakref of the registered parameter, owner id)
        self.entries = {}    # (data_ptr, packed dtype, transpose) -> _PackEntry
        self.tables = {}     # owner id -> _OwnerTable

    def register(self, params, owner):
        for p in params:
            if isinstance(p, torch.Tensor) and p.is_cuda and p.dim() == 4 and p.shape[2] == p.shape[3] \
                    and p.shape[2] in (1, 3):
                old = self.params.get(p.data_ptr())
                if old is not None and old[0]() is not p:
                    # the allocator handed a freed weight's address to this one: what was kept for the old weight
                    # (operands, owner) says nothing about the new
                    for k in [k for k in self.entries if k[0] == p.data_ptr()]:
                        del self.entries[k]
                self.params[p.data_ptr()] = (weakref.ref(p), owner)

    def owner(self, w):
        rec = self.params.get(w.data_ptr())
        p = rec[0]() if rec is not None else None
        if p is None or p.data_ptr() != w.data_ptr() or p.shape != w.shape or p.stride() != w.stride() \
                or p.dtype != w.dtype:
            return None, None
        return p, rec[1]

    def param(self, key):
        rec = self.params.get(key[0])
        return rec[0]() if rec is not None else None

    def sweep(self):
        dead = [k for k, (ref, _) in self.params.items() if ref() is None or ref().data_ptr() != k]
        for k in dead:
            del self.params[k]
        gone = [k for k in self.entries if k[0] not in self.params]
        for k in gone:
            del self.entries[k]
        return bool(dead or gone)


_PACKS = _WeightPackCache()
_SHARED_OWNER = 0
# Tables (and the tensors their rows point at) that a recorded HIP graph replays: the graph holds raw device
# addresses, so none of this may ever return to the allocator while the process lives.
_RECORDED_KEEP = []


def cache_weight_packs(params, owner=None):
    """Keep the packed operands of these convolution weights between uses (see repack_cached_weights).  `owner`
    (any object, e.g. the optimizer that updates them) groups the weights that one repack_cached_weights(owner)
    call renews; weights registered without one share a group."""
    _PACKS.register(params, id(owner) if owner is not None else _SHARED_OWNER)


def invalidate_weight_packs():
    """Forget what the kept operands hold: for code that has changed registered weights in place through `.data`
    (which moves no version counter)."""
    for e in _PACKS.entries.values():
        e.version = -1


def invalidate_unrecorded_packs(owner=None):
    """After the replay of a HIP graph that contains repack_cached_weights(owner): kept operands of that owner that
    the recorded table does not renew (created after the recording) no longer match their weights."""
    oid = id(owner) if owner is not None else _SHARED_OWNER
    t = _PACKS.tables.get(oid)
    if t is None or t.recorded is None:
        return
    for k, e in _PACKS.entries.items():
        if e.owner == oid and k not in t.recorded:
            e.version = -1


def repack_cached_weights(owner=None):
    """Renew every kept packed operand of `owner`'s weights in one launch (segk_pack_weights).  For callers that
    change registered weights without torch noticing (segk.optim.SGD's kernels write through raw pointers).
    Recorded into a HIP graph, the launch keeps reading the device table it was recorded with: that table, the
    weights and the operands it names stay allocated for the life of the process, and a table is never built
    during a capture (run one step eagerly first)."""
    c = _PACKS
    oid = id(owner) if owner is not None else _SHARED_OWNER
    capturing = torch.cuda.is_current_stream_capturing()
    if not capturing:
        c.sweep()
    keys = tuple(k for k, e in c.entries.items() if e.owner == oid)
    if not keys:
        return
    t = c.tables.get(oid)
    if t is None:
        t = c.tables[oid] = _OwnerTable()
    if t.keys != keys:
        if capturing:
            raise RuntimeError("segk.ops.repack_cached_weights: the set of kept weight operands changed since the "
                               "last eager step; run one step eagerly before recording a HIP graph")
        rows = np.zeros(len(keys), dtype=_PACK_ITEM)
        first = 0
        for i, k in enumerate(keys):
            e = c.entries[k]
            p = c.param(k)
            Cout, Cin, R, s_co, s_ci, s_tap = e.layout
            rows[i] = (p.data_ptr(), e.dst.data_ptr(), s_co, s_ci, s_tap, _dt(p), _dt(e.dst), Cout, Cin, R,
                       1 if k[2] else 0, first, 0)
            first += call("segk_pack_weight_blocks", Cout, Cin, R, s_tap, 1 if k[2] else 0)
        # (a table that a graph replays is left alone: the new set gets a table of its own)
        t.table = StagedTable(rows, c.entries[keys[0]].dst.device)
        t.table.upload()
        t.blocks, t.keys = first, keys
    if capturing:
        _RECORDED_KEEP.append((t.table, [c.param(k) for k in keys], [c.entries[k].dst for k in keys]))
        t.recorded = frozenset(keys)
    t0 = _hbm_start()
    call("segk_pack_weights", ctypes.c_void_p(t.table.ptr()), len(keys), t.blocks, _stream())
    if t0 is not None:
        _hbm_stop("pack_weight", sum(_nbytes(c.param(k), c.entries[k].dst) for k in keys), t0)
    for k in keys:
        c.entries[k].version = c.param(k)._version


# Debug switch (segk.config.set_debug_packs): every use of a kept operand packs the weight again and compares.
def _check_kept_pack(w, entry, dtype, transpose):
    fresh = torch.empty_like(entry.dst)
    s_co, s_ci, s_tap = _weight_strides(w)
    Cout, Cin, R, _ = w.shape
    call("segk_pack_weight_strided", _ptr(w), _dt(w), s_co, s_ci, s_tap, _ptr(fresh), _dt(fresh), Cout, Cin, R,
         1 if transpose else 0, _stream())
    if not torch.equal(fresh.view(torch.uint8), entry.dst.view(torch.uint8)):
        raise RuntimeError("segk: the kept packed operand of a %s weight no longer matches the weight (it was changed "
                           "in place through .data?); call segk.ops.invalidate_weight_packs() after such a change"
                           % (tuple(w.shape),))


def pack_weight(w, dtype, transpose=False, out=None, keep=False):
    """nn.Conv2d weight [Cout,Cin,R,R] (fp32 or bf16; contiguous or channels_last) -> [taps, rows, Ck] operand
    (see segk_pack_weight); `out` may be a contiguous slice of a larger buffer of that shape.  Without `out`, a
    weight registered with cache_weight_packs gets its kept operand (packed again only if the weight's version
    counter moved).  keep=True with `out`: that slice IS the kept operand of a registered weight (a caller that
    needs several operands side by side owns the buffer and keeps it alive): nothing is packed while the weight's
    version stands, and repack_cached_weights renews the slice in place."""
    _dev(w)
    if w.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("weights are float32 or bfloat16")
    Cout, Cin, R, S = w.shape
    dense = w.is_contiguous() or w.is_contiguous(memory_format=torch.channels_last)
    entry = None
    if (out is None or keep) and dense and _PACKS.params:
        owner, oid = _PACKS.owner(w)
        if owner is not None:
            key = (w.data_ptr(), dtype, bool(transpose))
            entry = _PACKS.entries.get(key)
            if entry is not None and entry.owner != oid:
                del _PACKS.entries[key]        # kept for another weight that once lived at this address
                entry = None
            if entry is not None and out is not None and (entry.dst is None or entry.dst.data_ptr() != out.data_ptr()):
                entry.version = -1             # the caller's buffer moved: pack into the new one
                if torch.cuda.is_current_stream_capturing():
                    entry = None               # (and keep nothing that lives in a graph's memory pool)
            if entry is not None and entry.version == owner._version:
                from . import config
                if config.debug_packs() and not torch.cuda.is_current_stream_capturing():
                    _check_kept_pack(w, entry, dtype, transpose)
                return entry.dst
            # (no new entry while a graph is recorded: a kept operand must not live in the graph's memory pool)
            if entry is None and not torch.cuda.is_current_stream_capturing():
                entry = _PACKS.entries[key] = _PackEntry()
                entry.dst = None
                entry.owner = oid
            if entry is not None:
                entry.version = owner._version
    if not dense:
        w = w.contiguous()
    s_co, s_ci, s_tap = _weight_strides(w)
    rows, k = (Cin, Cout) if transpose else (Cout, Cin)
    shape = (R * R, rows, round_up(k, 32))
    if out is None:
        if entry is not None and entry.dst is not None:
            out = entry.dst
        else:
            out = torch.empty(shape, dtype=dtype, device=w.device)
    elif tuple(out.shape) != shape or out.dtype != dtype or not out.is_contiguous():
        raise ValueError("pack_weight destination must be a contiguous %s tensor of shape %s" % (dtype, shape))
    if entry is not None:
        entry.dst, entry.layout = out, (Cout, Cin, R, s_co, s_ci, s_tap)
    t0 = _hbm_start()
    call("segk_pack_weight_strided", _ptr(w), _dt(w), s_co, s_ci, s_tap, _ptr(out), _dt(out), Cout, Cin, R,
         1 if transpose else 0, _stream())
    _hbm_stop("pack_weight", _nbytes(w, out), t0)
    return out


# -------------------------------------------------------------------- conv ----
def weight_usable_in_place(w, R, Cout):
    """An nn.Conv2d weight the forward convolution can read without packing: [Cout, Cin, R, R], dense channels_last
    (input channels contiguous), Cin a multiple of 32."""
    if w.dim() != 4 or w.shape[0] != Cout or w.shape[2] != R or w.shape[3] != R or w.shape[1] % 32 != 0:
        return False
    Cin = w.shape[1]
    if R == 1:
        return w.stride(1) == 1 and w.stride(0) == Cin
    return w.stride(1) == 1 and w.stride(3) == Cin and w.stride(2) == R * Cin and w.stride(0) == R * R * Cin


class BwdStatRequest:
    """What conv_fwd(..., bwd_stats=...) needs to emit a batch norm's backward statistics from the data-gradient launch
    that produces the layer's output gradient: raw = the layer's input [M, C] (pixel-major bf16), coef = its [4, C]
    coefficient rows, relu = whether a ReLU follows it, bits = the stored ReLU pattern ([M, C/8] uint8) or None (the
    pattern is raw * scale + shift > 0).  conv_fwd leaves the [rows, 2, C] float64 partial sums of g and g * xhat in
    `rows` when the launch can emit them, None otherwise."""
    __slots__ = ("raw", "coef", "relu", "bits", "rows")

    def __init__(self, raw=None, coef=None, relu=True, bits=None):
        self.raw, self.coef, self.relu, self.bits, self.rows = raw, coef, relu, bits, None


def conv_fwd(segs, N, H, W, Cout, out, bias=None, img_bias=None, want_stats=False, true_ck=None, batch=1,
             dil_step=0, out_batch=0, addend=None, addend_bits=None, bwd_stats=None):
    """segs: list of (in_view [M, >=Ck], packed_w [taps, Cout, Ck], R, dil); instead of a packed operand a segment
    may carry the nn.Conv2d weight itself when it is channels_last [Cout, Cin, R, R] of the operand dtype with Cin a
    multiple of 32 (used in place).  Writes `out` ([M, Cout'] view, Cout' >= Cout rounded up to 4).  Returns the [rows, 2, Cout] statistic partials when want_stats.
    true_ck: un-padded contracted channel counts per segment (only used for FLOP accounting).
    addend: pixel-major view [M, >=Cout] of out's dtype: out = round(conv) + addend (see conv_addend_ok);
    addend_bits: uint8 [M, >=Cout/8] beside it, one bit per element: a clear bit adds zero instead.
    bwd_stats (BwdStatRequest): `out` is the output gradient of the batch norm the request describes; its backward
    statistics are written by this launch when the kernels can (request.rows is then set)."""
    M = N * H * W
    timer_ck = true_ck if true_ck is not None else [(w.shape[1] if w.dim() == 4 and batch <= 1 else w.shape[-1])
                                                    for _, w, _, _ in segs]
    a = ConvArgs()
    a.N, a.H, a.W, a.Cout, a.nseg = N, H, W, Cout, len(segs)
    if not 1 <= len(segs) <= _lib.MAX_SEG:
        raise ValueError("1..%d segments" % _lib.MAX_SEG)
    dt = None
    for i, (x, w, R, dil) in enumerate(segs):
        _dev(x, w)
        ld, _ = _pm(x)
        if x.shape[0] != M:
            raise ValueError("segment %d has %d pixels, expected %d" % (i, x.shape[0], M))
        in_place = batch <= 1 and weight_usable_in_place(w, R, Cout)
        if in_place:
            pass
        elif batch > 1:
            if w.dim() != 4 or not w.is_contiguous() or w.shape[0] != batch or w.shape[1] != R * R or w.shape[2] != Cout:
                raise ValueError("segment %d: batched packed weight shape %s does not match batch=%d taps=%d Cout=%d"
                                 % (i, tuple(w.shape), batch, R * R, Cout))
        elif w.dim() != 3 or not w.is_contiguous() or w.shape[0] != R * R or w.shape[1] != Cout:
            raise ValueError("segment %d: packed weight shape %s does not match taps=%d Cout=%d"
                             % (i, tuple(w.shape), R * R, Cout))
        if x.dtype != w.dtype or (dt is not None and x.dtype != dt):
            raise ValueError("all conv operands share one dtype")
        dt = x.dtype
        Ck = w.shape[1] if in_place else w.shape[-1]
        a.seg[i].w_row_stride, a.seg[i].w_tap_stride = (w.stride(0), (w.stride(3) if R > 1 else Ck)) if in_place else (0, 0)
        if x.shape[1] < Ck and ld < Ck:
            raise ValueError("segment %d: activation row has %d elements, weight contracts %d" % (i, ld, Ck))
        a.seg[i].inp, a.seg[i].w = x.data_ptr(), w.data_ptr()
        a.seg[i].in_ld, a.seg[i].Ck, a.seg[i].R, a.seg[i].dil = ld, Ck, R, dil
    _dev(out, bias, img_bias)
    old, oc = _pm(out)
    if out.shape[0] != M:
        raise ValueError("output has %d pixels, expected %d" % (out.shape[0], M))
    a.dtype = _dt(segs[0][0])
    a.out, a.out_ld, a.out_dtype = out.data_ptr(), old, _dt(out)
    for name, v, n in (("bias", bias, Cout), ("img_bias", img_bias, N * Cout)):
        if v is not None and (v.dtype != torch.float32 or not v.is_contiguous() or v.numel() != n):
            raise ValueError("%s must be contiguous float32 with %d elements" % (name, n))
    a.bias = bias.data_ptr() if bias is not None else None
    a.img_bias = img_bias.data_ptr() if img_bias is not None else None
    if addend is not None:
        _dev(addend)
        ald, ac = _pm(addend)
        if addend.dtype != out.dtype or addend.shape[0] != M or ac < Cout:
            raise ValueError("the addend is a pixel-major view of the output's dtype with at least Cout channels")
        a.addend, a.addend_ld = addend.data_ptr(), ald
        if addend_bits is not None:
            _dev(addend_bits)
            if addend_bits.dtype != torch.uint8 or addend_bits.dim() != 2 or not addend_bits.is_contiguous() \
                    or addend_bits.shape[0] != M or addend_bits.shape[1] * 8 < Cout:
                raise ValueError("addend_bits is a contiguous uint8 [M, >=Cout/8] tensor")
            a.addend_bits, a.addend_bits_ld = addend_bits.data_ptr(), addend_bits.shape[1]
    else:
        a.addend, a.addend_ld = None, 0
        if addend_bits is not None:
            raise ValueError("addend_bits without an addend")
    if batch > 1:
        if len(segs) != 1 or bias is not None or img_bias is not None:
            raise ValueError("a batched convolution has one segment and no bias")
        a.batch, a.w_batch, a.out_batch, a.dil_step = batch, segs[0][1][0].numel(), out_batch, dil_step
    stats = None
    if want_stats:
        rows = call("segk_conv_stat_rows", ctypes.byref(a))
        stats = torch.empty((rows, 2, Cout) if batch <= 1 else (batch, rows, 2, Cout), dtype=torch.float64,
                            device=out.device)
        a.stat_partials = stats.data_ptr()
    else:
        a.stat_partials = None
    a.bst_rows = None
    if bwd_stats is not None:
        bwd_stats.rows = None
        q = bwd_stats
        if (addend is None and q.raw is not None and q.raw.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and q.raw.dim() == 2
                and q.raw.shape[0] == M and q.raw.shape[1] == Cout and q.raw.stride(1) == 1 and q.coef.shape == (4, Cout)
                and q.coef.dtype == torch.float32 and q.coef.is_contiguous()
                and (q.bits is None or (q.bits.dtype == torch.uint8 and q.bits.dim() == 2 and q.bits.is_contiguous()
                                        and q.bits.shape[0] == M))):
            _dev(q.raw, q.coef, q.bits)
            r = _row_ptrs(q.coef)
            a.bst_x, a.bst_x_ld = q.raw.data_ptr(), q.raw.stride(0)
            a.bst_mean, a.bst_inv_std, a.bst_scale, a.bst_shift = r[0].value, r[1].value, r[2].value, r[3].value
            a.bst_bits, a.bst_bits_ld = (q.bits.data_ptr(), q.bits.shape[1]) if q.bits is not None else (None, 0)
            a.bst_relu = 1 if q.relu else 0
            rows = call("segk_conv_bstat_rows", ctypes.byref(a))
            if rows > 0:
                q.rows = torch.empty((rows, 2, Cout), dtype=torch.float64, device=out.device)
                a.bst_rows = q.rows.data_ptr()
    t0 = TIMER.start() if TIMER is not None else None
    call("segk_conv_fwd", ctypes.byref(a), _stream())
    if t0 is not None:
        # nominal FLOPs: every tap counted (also the ones that fall outside the image), padded channels not
        fl = inb = 0.0
        for b in range(max(batch, 1)):
            for i in range(len(segs)):
                f = 2.0 * M * Cout * timer_ck[i] * segs[i][2] ** 2
                fl += f
                inb += f * inbounds_fraction(H, W, segs[i][2], segs[i][3] + b * dil_step)
        TIMER.stop("conv_fwd_kernel", fl, t0, inbounds=inb)
    return stats


def unpack_bits(bits, C):
    """[M, C/8] uint8 bit pattern (bit c % 8 of byte c / 8) -> bool [M, C]: the slow, plain-PyTorch reading of what the
    kernels consume directly; for fall-back paths only."""
    shifts = torch.arange(8, device=bits.device, dtype=torch.uint8)
    return ((bits.unsqueeze(-1) >> shifts) & 1).reshape(bits.shape[0], -1)[:, :C].bool()


def conv_addend_ok(segs, Cout, out_dtype, addend):
    """Whether conv_fwd(..., addend=addend) can add `addend` in its epilogue: bf16 operands whose contracted rows are
    multiples of 128 bytes (the ring tiles), Cout a multiple of 128, addend rows aligned for the output's stores."""
    if addend is None or addend.dim() != 2 or addend.stride(1) != 1 or addend.dtype != out_dtype:
        return False
    es = addend.element_size()
    if Cout % 128 or addend.shape[1] < Cout or addend.data_ptr() % (4 * es) or (addend.stride(0) * es) % (4 * es):
        return False
    return all(x.dtype == torch.bfloat16 and (w.shape[1] if w.dim() == 4 else w.shape[-1]) % 64 == 0 for x, w, _, _ in segs)


def _ring_parts(chunks, Cin, Cout, R, cus=256, groups=1, want_time=False):
    """Parts of the pixel range for the ring tile (256 x 128 of one tap per workgroup, one workgroup per CU).  A
    cost model in microseconds fitted to kernel durations on MI355X: workgroups run in rounds of one per CU, a
    64-pixel chunk takes 0.9 us, a workgroup has 8 us of fixed cost (prologue, 128 KB slab store), and every part
    adds a whole fp32 copy of the gradient that the fixed-order sum reads back (0.3 us per MB).  groups: that many
    convolutions of this shape in one launch (conv_wgrad_group).  want_time: -> (parts, modelled microseconds)."""
    tiles = (Cout // 256) * ((Cin + 127) // 128) * R * R * groups
    slab_mb = R * R * Cin * Cout * 4 / 1e6 * groups
    best, best_t = 1, None
    for sp in range(1, max(1, min(SPLIT_CAP, chunks // 8)) + 1):
        rounds = (tiles * sp + cus - 1) // cus
        t = rounds * (0.9 * ((chunks + sp - 1) // sp) + 8.0) + 3.0 + sp * slab_mb * 0.3
        if best_t is None or t < best_t:
            best, best_t = sp, t
    return (best, best_t) if want_time else best


def wgrad_group_pays(M, Cin, Cout, R, dtype):
    """Whether weight gradients of this shape are worth collecting for conv_wgrad_group: the ring tile takes them and a
    single one has to be cut into four or more parts to fill the chip."""
    if dtype != torch.bfloat16 or Cout % 256:
        return False
    return _ring_parts(max(1, (M + 63) // 64), Cin, Cout, R) >= 4


def wgrad_plan(M, Cin, Cout, R, dtype, ring=True):
    """-> (tile, splits) of a weight gradient: parts of the pixel range so that the launch fills the chip.  bf16
    gradients onto a multiple of 256 output channels take the ring tile (tile code 384, parts from _ring_parts);
    ring=False gives the plan of the older tiles for everything (comparisons).  The
    256 x 256-tile kernel (Cout a multiple of 256, Cin at least 256 with a last tile more than half full) runs one
    workgroup per CU, the 128 x 128 one two (fp32) or three (bf16, LDS-DMA).  Every part writes a whole fp32 copy of
    the gradient, so with few pixels per part (the 33 x 33 maps of the feeder's layers 3 and 4) the slabs and the
    pass that sums them outweigh the multiplication: there the small tile is used (four times the workgroups per
    part) with only as many parts as keep the slabs within SLAB_BUDGET, but at least one workgroup per CU."""
    bp = 64 if dtype == torch.bfloat16 else 32
    chunks = max(1, (M + bp - 1) // bp)
    if ring and dtype == torch.bfloat16 and Cout % 256 == 0:
        return 384, _ring_parts(chunks, Cin, Cout, R)
    big = Cin >= 256 and Cout % 256 == 0 and (Cin % 256 == 0 or Cin % 256 > 128)

    # a gradient of at most 256 K elements (a 1 MB slab) may be cut into SPLIT_CAP_SMALL parts: its slabs stay
    # small and segk_wgrad_finish sums them with eight loads in flight per thread
    cap = SPLIT_CAP_SMALL if R * R * Cin * Cout <= (1 << 18) else SPLIT_CAP

    def parts(t, target):
        tiles = ((Cin + t - 1) // t) * ((Cout + t - 1) // t) * R * R
        return max(1, min(target // tiles if tiles <= target else 1, cap, chunks // 8 if chunks >= 8 else 1))
    small_target = 768 if dtype == torch.bfloat16 else 1024
    if big:
        sp = parts(256, 256)
        slab = R * R * Cin * Cout * 4
        if dtype == torch.bfloat16 and sp * slab > SLAB_BUDGET and chunks // sp < 64:
            fill, most = parts(128, 256), parts(128, small_target)
            return 128, max(fill, min(most, SLAB_BUDGET // slab))
        return 256, sp
    return 128, parts(128, small_target)


SPLIT_CAP = 64
WGRAD_GROUP_MAX = _lib.WGRAD_GROUP_MAX
SPLIT_CAP_SMALL = 384
SLAB_BUDGET = 32 << 20


def default_splits(M, Cin, Cout, R, dtype):
    return wgrad_plan(M, Cin, Cout, R, dtype)[1]


def conv_wgrad(x, dout, N, H, W, R, dil, Cout_true, Cin_true, splits=None, like=None, tile=0):
    """Weight gradient [Cout_true, Cin_true, R, R] from the forward input view x [M, Cin_pad] and the output gradient
    view dout [M, Cout_pad] (padded channels must hold zeros): contiguous fp32, or with the dtype and strides of
    the weight `like` (what autograd expects for a bf16 / channels_last parameter)."""
    _dev(x, dout)
    M = N * H * W
    in_ld, Cin = _pm(x)
    do_ld, Cout = _pm(dout)
    if x.shape[0] != M or dout.shape[0] != M or x.dtype != dout.dtype:
        raise ValueError("wgrad operands disagree: %s %s M=%d" % (tuple(x.shape), tuple(dout.shape), M))
    # A 1 x 1 gradient is a plain product of the two operands over the pixels, so it can be computed transposed: when
    # only the INPUT channels fill the ring tile's 256-channel side (512 -> 128, 256 -> 32: the first 1 x 1 of ResNet
    # layer 2's bottlenecks, the classifier), the operands change places and the fixed-order sum writes the result
    # back through swapped strides (the same values: every product and every partial sum is the same).
    swapped = (splits is None and tile == 0 and R == 1 and x.dtype == torch.bfloat16 and Cout % 256 != 0
               and Cin % 256 == 0 and Cout % 32 == 0 and M >= 4096)
    if swapped:
        x, dout, in_ld, do_ld, Cin, Cout = dout, x, do_ld, in_ld, Cout, Cin
        Cout_true, Cin_true = Cin_true, Cout_true
    if splits is None:   # (an explicit `splits` leaves the tile to the library unless `tile` names it: 128 or 256)
        tile, splits = wgrad_plan(M, Cin, Cout, R, x.dtype)
    slabs = torch.empty((splits, R * R, Cin, Cout), dtype=torch.float32, device=x.device)
    a = WgradArgs()
    a.N, a.H, a.W, a.dtype = N, H, W, _dt(x)
    a.inp, a.in_ld, a.Cin = x.data_ptr(), in_ld, Cin
    a.dout, a.dout_ld, a.Cout = dout.data_ptr(), do_ld, Cout
    a.R, a.dil, a.splits, a.slabs, a.tile = R, dil, splits, slabs.data_ptr(), tile
    co_t, ci_t = (Cin_true, Cout_true) if swapped else (Cout_true, Cin_true)   # the gradient's own shape
    if like is None:
        dw = torch.empty((co_t, ci_t, R, R), dtype=torch.float32, device=x.device)
    else:
        if tuple(like.shape) != (co_t, ci_t, R, R):
            raise ValueError("gradient layout model has shape %s" % (tuple(like.shape),))
        dw = torch.empty_strided(like.shape, like.stride(), dtype=like.dtype, device=x.device)
    s_co, s_ci, s_tap = _weight_strides(dw)
    if swapped:
        s_co, s_ci = s_ci, s_co
    # the bracket holds the split-slab kernel AND the fixed-order sum of the slabs: the whole weight-gradient cost
    t0 = TIMER.start() if TIMER is not None else None
    call("segk_conv_wgrad", ctypes.byref(a), _stream())
    call("segk_wgrad_finish_strided", _ptr(slabs), splits, R, Cin, Cout, _ptr(dw), _dt(dw), s_co, s_ci, s_tap,
         Cout_true, Cin_true, _stream())
    if t0 is not None:
        fl = 2.0 * M * Cout_true * Cin_true * R * R
        TIMER.stop("conv_wgrad_kernel", fl, t0, inbounds=fl * inbounds_fraction(H, W, R, dil))
    return dw


def conv_wgrad_group(xs, douts, N, H, W, R, dil, Cout_true, Cin_true, like=None, splits=None):
    """Weight gradients of len(xs) <= WGRAD_GROUP_MAX convolutions of one shape in one launch (plus one launch for the
    fixed-order sums): -> list of gradients as conv_wgrad returns them.  bf16 operands whose padded output channels
    are a multiple of 256; every x (and every dout) has the same shape and row stride."""
    G = len(xs)
    if not 1 <= G <= _lib.WGRAD_GROUP_MAX or len(douts) != G:
        raise ValueError("1..%d convolutions per group, got %d / %d" % (_lib.WGRAD_GROUP_MAX, G, len(douts)))
    _dev(*xs, *douts)
    M = N * H * W
    in_ld, Cin = _pm(xs[0])
    do_ld, Cout = _pm(douts[0])
    for x, dout in zip(xs, douts):
        if _pm(x) != (in_ld, Cin) or _pm(dout) != (do_ld, Cout) or x.shape[0] != M or dout.shape[0] != M \
                or x.dtype != torch.bfloat16 or dout.dtype != torch.bfloat16:
            raise ValueError("grouped weight gradients need bf16 operands of one shape and row stride")
    if splits is None:
        splits = _ring_parts(max(1, (M + 63) // 64), Cin, Cout, R, groups=G)
    slabs = torch.empty((G, splits, R * R, Cin, Cout), dtype=torch.float32, device=xs[0].device)
    a = WgradGroupArgs()
    c = a.conv
    c.N, c.H, c.W, c.dtype = N, H, W, _lib.BF16
    c.in_ld, c.Cin, c.dout_ld, c.Cout = in_ld, Cin, do_ld, Cout
    c.R, c.dil, c.splits, c.slabs, c.tile = R, dil, splits, slabs.data_ptr(), 384
    a.groups = G
    for g in range(G):
        a.inp[g], a.dout[g] = xs[g].data_ptr(), douts[g].data_ptr()
    dws = []
    for g in range(G):
        if like is None:
            dws.append(torch.empty((Cout_true, Cin_true, R, R), dtype=torch.float32, device=xs[0].device))
        else:
            if tuple(like.shape) != (Cout_true, Cin_true, R, R):
                raise ValueError("gradient layout model has shape %s" % (tuple(like.shape),))
            dws.append(torch.empty_strided(like.shape, like.stride(), dtype=like.dtype, device=xs[0].device))
    s_co, s_ci, s_tap = _weight_strides(dws[0])
    ptrs = (ctypes.c_void_p * G)(*[d.data_ptr() for d in dws])
    t0 = TIMER.start() if TIMER is not None else None
    call("segk_conv_wgrad_group", ctypes.byref(a), _stream())
    call("segk_wgrad_finish_group", _ptr(slabs), G, splits, R, Cin, Cout, ptrs, _dt(dws[0]), s_co, s_ci, s_tap,
         Cout_true, Cin_true, _stream())
    if t0 is not None:
        fl = 2.0 * G * M * Cout_true * Cin_true * R * R
        TIMER.stop("conv_wgrad_kernel", fl, t0, inbounds=fl * inbounds_fraction(H, W, R, dil))
    return dws


# ---------------------------------------------------- batched matrix products ----
def _b3(t, name):
    """(batch stride, leading dimension) of a [batch, rows, cols] view with unit column stride."""
    if t.dim() != 3 or t.stride(2) != 1:
        raise ValueError("%s must be a 3-D view [batch, rows, cols] with unit column stride, got shape %s strides %s"
                         % (name, tuple(t.shape), t.stride()))
    return t.stride(0), t.stride(1)


def gemm_nt(terms, out, true_k=None):
    """out[b] = sum_t A_t[b] @ B_t[b]^T.  terms: list of (A [b|1, M, K], B [b|1, N, K]); K a multiple of 32 with
    zeros beyond the true contracted length in both factors; a leading dimension of 1 is shared by the batch.
    out: [b, M, N'] view, N' >= N rounded up to 4 (rounding columns receive zeros)."""
    _dev(out)
    c_batch, ldc = _b3(out, "out")
    batch, M = out.shape[0], out.shape[1]
    a = GemmArgs()
    a.batch, a.M, a.nterms = batch, M, len(terms)
    if not 1 <= len(terms) <= _lib.MAX_SEG:
        raise ValueError("1..%d terms" % _lib.MAX_SEG)
    N, dt, flops = None, None, 0.0
    for i, (A, B) in enumerate(terms):
        _dev(A, B)
        ab, lda = _b3(A, "A")
        bb, ldb = _b3(B, "B")
        if A.dtype != B.dtype or (dt is not None and A.dtype != dt):
            raise ValueError("all gemm operands share one dtype")
        dt = A.dtype
        if A.shape[1] != M or A.shape[2] != B.shape[2] or (N is not None and B.shape[1] != N):
            raise ValueError("term %d shapes %s x %s do not match M=%d N=%s" % (i, tuple(A.shape), tuple(B.shape), M, N))
        N = B.shape[1]
        for t, nm in ((A, "A"), (B, "B")):
            if t.shape[0] not in (1, batch):
                raise ValueError("term %d %s has batch %d, expected 1 or %d" % (i, nm, t.shape[0], batch))
        tm = a.term[i]
        tm.A, tm.a_batch, tm.lda = A.data_ptr(), (ab if A.shape[0] == batch and batch > 1 else 0), lda
        tm.B, tm.b_batch, tm.ldb = B.data_ptr(), (bb if B.shape[0] == batch and batch > 1 else 0), ldb
        tm.K = A.shape[2]
        flops += 2.0 * batch * M * N * (true_k[i] if true_k is not None else A.shape[2])
    if out.shape[2] < round_up(N, 4) and ldc < round_up(N, 4):
        raise ValueError("out has %d columns, needs %d" % (out.shape[2], round_up(N, 4)))
    a.N, a.dtype = N, _dt(terms[0][0])
    a.C, a.c_batch, a.ldc, a.c_dtype = out.data_ptr(), (c_batch if batch > 1 else 0), ldc, _dt(out)
    t0 = TIMER.start() if TIMER is not None else None
    call("segk_gemm_nt", ctypes.byref(a), _stream())
    if t0 is not None:
        TIMER.stop("conv_fwd_kernel", flops, t0)
    return out


def gemm_tn(A, B, out, rows=None, cols=None, splits=1, true_k=None):
    """out[b][m][n] = sum_k A[b][k][m] * B[b][k][n] for m < rows, n < cols.  A: [b, K, Mp], B: [b, K, Np] views
    (Mp a multiple of 32, Np of 16, padding columns zero).  out: [b, rows, >=cols] view of any supported dtype."""
    _dev(A, B, out)
    ab, lda = _b3(A, "A")
    bb, ldb = _b3(B, "B")
    cb, ldc = _b3(out, "out")
    batch, K, Mp = A.shape
    Np = B.shape[2]
    if B.shape[0] != batch or B.shape[1] != K or out.shape[0] != batch or A.dtype != B.dtype:
        raise ValueError("gemm_tn operands disagree: %s %s %s" % (tuple(A.shape), tuple(B.shape), tuple(out.shape)))
    rows = out.shape[1] if rows is None else rows
    cols = out.shape[2] if cols is None else cols
    direct = (splits == 1 and out.dtype == torch.float32 and out.is_contiguous() and out.shape[1] == Mp
              and out.shape[2] == Np and rows == Mp and cols == Np)
    slabs = out if direct else torch.empty((batch, splits, Mp, Np), dtype=torch.float32, device=A.device)
    a = GemmTnArgs()
    a.batch, a.M, a.N, a.K, a.dtype = batch, Mp, Np, K, _dt(A)
    a.A, a.a_batch, a.lda = A.data_ptr(), (ab if batch > 1 else 0), lda
    a.B, a.b_batch, a.ldb = B.data_ptr(), (bb if batch > 1 else 0), ldb
    a.splits, a.slabs = splits, slabs.data_ptr()
    t0 = TIMER.start() if TIMER is not None else None
    call("segk_gemm_tn", ctypes.byref(a), _stream())
    if t0 is not None:
        TIMER.stop("conv_wgrad_kernel", 2.0 * batch * rows * cols * (true_k if true_k is not None else K), t0)
    if not direct:
        call("segk_gemm_tn_finish", _ptr(slabs), batch, splits, Mp, Np, _ptr(out), _dt(out), (cb if batch > 1 else 0),
             ldc, rows, cols, _stream())
    return out


def softmax_rows(e, cols, out, reversed_=False):
    """e: contiguous float32 [..., ld]; softmax over the first `cols` columns of every row into out [..., ld_out]
    (columns cols.. of out are zero-filled).  reversed_: softmax(row max - e)."""
    _dev(e, out)
    if e.dtype != torch.float32 or not e.is_contiguous() or not out.is_contiguous():
        raise ValueError("softmax_rows expects contiguous tensors and float32 energies")
    rows = e.numel() // e.shape[-1]
    if out.numel() // out.shape[-1] != rows:
        raise ValueError("row counts differ")
    call("segk_softmax_rows", _ptr(e), rows, cols, e.shape[-1], 1 if reversed_ else 0, _ptr(out), _dt(out),
         out.shape[-1], out.shape[-1], _stream())
    return out


def softmax_rows_bwd(att, datt, cols, out, sign=1.0):
    _dev(att, datt, out)
    if datt.dtype != torch.float32 or not (att.is_contiguous() and datt.is_contiguous() and out.is_contiguous()):
        raise ValueError("softmax_rows_bwd expects contiguous tensors and a float32 upstream gradient")
    rows = att.numel() // att.shape[-1]
    call("segk_softmax_rows_bwd", _ptr(att), _dt(att), att.shape[-1], _ptr(datt), datt.shape[-1], rows, cols,
         ctypes.c_float(sign), _ptr(out), _dt(out), out.shape[-1], out.shape[-1], _stream())
    return out


def transpose(src, out):
    """out[b][c][r] = src[b][r][c]; src [b, R, C] view, out [b, C, >=R] view whose extra columns get zeros."""
    _dev(src, out)
    sb, sld = _b3(src, "src")
    ob, old = _b3(out, "out")
    batch, R, C = src.shape
    if out.shape[0] != batch or out.shape[1] != C or out.shape[2] < R or src.dtype != out.dtype:
        raise ValueError("transpose shapes disagree: %s -> %s" % (tuple(src.shape), tuple(out.shape)))
    call("segk_transpose", _ptr(src), _dt(src), batch, R, C, (sb if batch > 1 else 0), sld, _ptr(out),
         (ob if batch > 1 else 0), old, out.shape[2], _stream())
    return out


def lincomb(out, terms):
    """out = sum_i s_i * dev_i * x_i over pixel-major views; terms: up to three (x, s, dev) with dev a 1-element
    float32 device tensor or None."""
    _dev(out)
    old, C = _pm(out)
    M = out.shape[0]
    args = []
    dt = terms[0][0].dtype
    for x, s, dev in terms:
        _dev(x, dev)
        ld, c = _pm(x)
        if c != C or x.shape[0] != M or x.dtype != dt:
            raise ValueError("lincomb operands disagree")
        if dev is not None and (dev.dtype != torch.float32 or dev.numel() != 1):
            raise ValueError("device scalars are 1-element float32 tensors")
        args += [_ptr(x), ld, ctypes.c_float(s), _ptr(dev)]
    for _ in range(3 - len(terms)):
        args += [None, 0, ctypes.c_float(0.0), None]
    call("segk_lincomb", _ptr(out), _dt(out), old, M, C, *args, _dt(terms[0][0]), _stream())
    return out


def dot(a, b):
    """sum(a * b) over two pixel-major views of one dtype -> 1-element float32 device tensor."""
    _dev(a, b)
    lda, C = _pm(a)
    ldb, Cb = _pm(b)
    if C != Cb or a.shape[0] != b.shape[0] or a.dtype != b.dtype:
        raise ValueError("dot operands disagree")
    M = a.shape[0]
    rows = call("segk_dot_partial_rows", M)
    partials = torch.empty((rows,), dtype=torch.float64, device=a.device)
    res = torch.empty((1,), dtype=torch.float32, device=a.device)
    call("segk_dot", _ptr(a), lda, _ptr(b), ldb, _dt(a), M, C, _ptr(partials), _ptr(res), _stream())
    return res


# ---------------------------------------------------------------------- BN ----
def _bn_geom(x, layout):
    if layout == PM:
        ld, C = _pm(x)
        return 1, C, x.shape[0], ld  # N folded into the pixel count
    if x.dim() < 2 or not x.is_contiguous():
        raise ValueError("NCHW batch-norm input must be contiguous")
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C) if N * C else 0
    return N, C, HW, 0


def bn_stat_partials(x, layout):
    """([rows, 2, C] float64 partial sums / sums of squares of THIS rank's tensor, element count per channel)."""
    _dev(x)
    N, C, HW, ld = _bn_geom(x, layout)
    rows = call("segk_bn_stat_rows", layout, N, C, HW)
    partials = torch.empty((rows, 2, C), dtype=torch.float64, device=x.device)
    t0 = _hbm_start()
    call("segk_bn_stats", _ptr(x), _dt(x), layout, N, C, HW, ld, _ptr(partials), _stream())
    _hbm_stop("bn_stats", _nbytes(x), t0)
    return partials, float(N * HW)


def bn_local_sums(x, layout):
    """[2C+1] float64: per-channel sum, sum of squares, element count of THIS rank's tensor."""
    partials, count = bn_stat_partials(x, layout)
    return bn_reduce(partials, count)


def bn_reduce_coeffs(partials, count, gamma, beta, eps, momentum, running_mean, running_var, legacy_clamp=False):
    """bn_reduce + bn_coeffs in one launch (statistics that are not exchanged between ranks) -> coef [4, C]."""
    rows, two, C = partials.shape
    if partials.dtype != torch.float64 or not partials.is_contiguous():
        raise ValueError("statistic partials are contiguous float64")
    _dev(partials, gamma, beta, running_mean, running_var)
    coef = torch.empty((4, C), dtype=torch.float32, device=partials.device)
    r = _row_ptrs(coef)
    call("segk_bn_reduce_coeffs", _ptr(partials), rows, C, ctypes.c_double(count), _ptr(gamma), _ptr(beta),
         ctypes.c_float(eps), ctypes.c_float(momentum), 1 if legacy_clamp else 0, _ptr(running_mean),
         _ptr(running_var), r[0], r[1], r[2], r[3], None, _stream())
    return coef


def bn_reduce(partials, count, want_f32=False):
    """[rows, 2, C] float64 partial rows -> [2C+1] float64 (two per-channel sums, then the count); with want_f32
    also the two sums as a [2, C] float32 tensor from the same launch."""
    rows, two, C = partials.shape
    if partials.dtype != torch.float64 or not partials.is_contiguous():
        raise ValueError("statistic partials are contiguous float64")
    sums = torch.empty((2 * C + 1,), dtype=torch.float64, device=partials.device)
    f32 = torch.empty((2, C), dtype=torch.float32, device=partials.device) if want_f32 else None
    call("segk_bn_reduce", _ptr(partials), rows, C, ctypes.c_double(count), _ptr(sums), _ptr(f32), _stream())
    return (sums, f32) if want_f32 else sums


def bn_coeffs(sums, gamma, beta, eps, momentum, running_mean, running_var, legacy_clamp=False):
    """-> coef [4, C] float32 rows: mean, inv_std, scale, shift; running statistics updated in place."""
    C = (sums.numel() - 1) // 2
    if sums.dtype != torch.float64:
        raise ValueError("batch-norm sums are float64")
    coef = torch.empty((4, C), dtype=torch.float32, device=sums.device)
    _dev(sums, gamma, beta, running_mean, running_var)
    r = _row_ptrs(coef)
    call("segk_bn_coeffs", _ptr(sums), _ptr(gamma), _ptr(beta), ctypes.c_float(eps), ctypes.c_float(momentum),
         1 if legacy_clamp else 0, _ptr(running_mean), _ptr(running_var), r[0], r[1], r[2], r[3], C, _stream())
    return coef


def bn_coeffs_eval(running_mean, running_var, gamma, beta, eps):
    C = running_mean.numel()
    _dev(running_mean, running_var, gamma, beta)
    coef = torch.empty((4, C), dtype=torch.float32, device=running_mean.device)
    r = _row_ptrs(coef)
    call("segk_bn_coeffs_eval", _ptr(running_mean), _ptr(running_var), _ptr(gamma), _ptr(beta),
         ctypes.c_float(eps), r[0], r[1], r[2], r[3], C, _stream())
    return coef


def _mask_check(mask, x):
    if mask is None:
        return
    if mask.dtype != torch.uint8 or not mask.is_contiguous() or mask.numel() != x.numel():
        raise ValueError("dropout mask must be a contiguous uint8 tensor with one byte per element")
    _dev(mask)


def _pm_operand(t, like, name, layout):
    """(pointer, ld) of an optional pixel-major operand shaped like `like`."""
    if t is None:
        return None, 0
    if layout != PM:
        raise ValueError("%s needs the pixel-major layout" % name)
    _dev(t)
    ld, c = _pm(t)
    if t.shape != like.shape:
        raise ValueError("%s shape %s != %s" % (name, tuple(t.shape), tuple(like.shape)))
    return _ptr(t), ld


def relu_bits_like(y):
    """An empty [M, C/8] uint8 tensor for the ReLU bit pattern of the pixel-major view y, or None when the kernels'
    8-channel vector access does not apply to it."""
    ld, C = _pm(y)
    if C % 8 or ld % 8 or y.data_ptr() % 16:
        return None
    return torch.empty((y.shape[0], C // 8), dtype=torch.uint8, device=y.device)


def _act_operand(act, x, layout):
    """(pointer, ld, is_bits) of the stored ReLU information of a residual layer: its output (x's shape) or the
    [M, C/8] uint8 bit pattern bn_apply wrote."""
    if act is not None and act.dtype == torch.uint8:
        if layout != PM or act.dim() != 2 or act.shape[0] != x.shape[0] or act.shape[1] * 8 < x.shape[1] or not act.is_contiguous():
            raise ValueError("the ReLU bit pattern is a contiguous uint8 [M, C/8] tensor beside pixel-major operands")
        _dev(act)
        return _ptr(act), act.shape[1], 1
    ap, a_ld = _pm_operand(act, x, "stored activation", layout)
    return ap, a_ld, 0


def bn_apply(x, y, coef, layout, relu, mask=None, mask_scale=1.0, res=None, relu_bits=None):
    """y = relu?(x*scale + shift [+ res]) * mask; res (pixel-major, y's dtype) is the shortcut of a residual block.
    relu_bits (from relu_bits_like(y)) receives one bit per element: y > 0."""
    _dev(x, y, coef)
    N, C, HW, x_ld = _bn_geom(x, layout)
    y_ld = _pm(y)[0] if layout == PM else 0
    if y.shape != x.shape:
        raise ValueError("bn_apply output shape %s != input shape %s" % (tuple(y.shape), tuple(x.shape)))
    _mask_check(mask, x)
    if res is not None and res.dtype != y.dtype:
        raise ValueError("the residual input has the output's dtype")
    rp, r_ld = _pm_operand(res, y, "residual", layout)
    r = _row_ptrs(coef)
    t0 = _hbm_start()
    call("segk_bn_apply", _ptr(x), _dt(x), x_ld, _ptr(y), _dt(y), y_ld, layout, N, C, HW, r[2], r[3],
         1 if relu else 0, _ptr(mask), ctypes.c_float(mask_scale), rp, r_ld, _ptr(relu_bits),
         relu_bits.shape[1] if relu_bits is not None else 0, _stream())
    _hbm_stop("bn_apply", _nbytes(x, y, res, mask, relu_bits), t0)
    return y


def bn_bwd_stat_partials(dz, x, coef, layout, relu, mask=None, mask_scale=1.0, act=None):
    """([rows, 2, C] float64 partial sums of g and g*xhat, count) for this rank (g = dz through dropout and ReLU);
    act = the layer's stored output when a residual was added before the ReLU."""
    _dev(dz, x, coef)
    N, C, HW, x_ld = _bn_geom(x, layout)
    dz_ld = _pm(dz)[0] if layout == PM else 0
    if layout == NCHW and not dz.is_contiguous():
        raise ValueError("NCHW gradient must be contiguous")
    _mask_check(mask, x)
    if act is not None and act.dtype not in (dz.dtype, torch.uint8):
        raise ValueError("the stored activation has the gradient's dtype")
    ap, a_ld, a_bits = _act_operand(act, x, layout)
    rows = call("segk_bn_stat_rows", layout, N, C, HW)
    partials = torch.empty((rows, 2, C), dtype=torch.float64, device=x.device)
    r = _row_ptrs(coef)
    t0 = _hbm_start()
    call("segk_bn_bwd_stats", _ptr(dz), _dt(dz), dz_ld, _ptr(x), _dt(x), x_ld, layout, N, C, HW, r[0], r[1], r[2], r[3],
         1 if relu else 0, _ptr(mask), ctypes.c_float(mask_scale), ap, a_ld, a_bits, _ptr(partials), _stream())
    _hbm_stop("bn_bwd_stats", _nbytes(dz, x, act, mask), t0)
    return partials, float(N * HW)


def bn_bwd_local_sums(dz, x, coef, layout, relu, mask=None, mask_scale=1.0, act=None):
    """[2C+1] float64: sum g, sum g*xhat, count for this rank."""
    partials, count = bn_bwd_stat_partials(dz, x, coef, layout, relu, mask, mask_scale, act)
    return bn_reduce(partials, count)


def bn_bwd_apply(dz, x, dx, coef, gsums, layout, relu, mask=None, mask_scale=1.0, eval_mode=False, act=None,
                 dres=None):
    """dx from the global sums; dres (optional, dz's dtype) receives the gradient of the residual input."""
    _dev(dz, x, dx, coef, gsums)
    N, C, HW, x_ld = _bn_geom(x, layout)
    dz_ld = _pm(dz)[0] if layout == PM else 0
    dx_ld = _pm(dx)[0] if layout == PM else 0
    _mask_check(mask, x)
    for t, nm in ((act, "stored activation"), (dres, "residual gradient")):
        if t is not None and t.dtype != dz.dtype and not (t is act and t.dtype == torch.uint8):
            raise ValueError("the %s has the gradient's dtype" % nm)
    ap, a_ld, a_bits = _act_operand(act, x, layout)
    dp, d_ld = _pm_operand(dres, x, "residual gradient", layout)
    r = _row_ptrs(coef)
    t0 = _hbm_start()
    call("segk_bn_bwd_apply", _ptr(dz), _dt(dz), dz_ld, _ptr(x), _dt(x), x_ld, _ptr(dx), _dt(dx), dx_ld, layout, N,
         C, HW, r[0], r[1], r[2], r[3], 1 if relu else 0, _ptr(mask), ctypes.c_float(mask_scale), _ptr(gsums),
         1 if eval_mode else 0, ap, a_ld, a_bits, dp, d_ld, _stream())
    _hbm_stop("bn_bwd_apply", _nbytes(dz, x, dx, act, dres, mask), t0)
    return dx


def dropout_mask(shape, p, device):
    """uint8 keep-mask of `shape` (1 with probability 1 - p) in one launch.  The Philox key and stream offset are two
    int64 drawn on the device from torch's generator: torch.manual_seed fixes the masks, and inside a recorded
    graph every replay draws new ones."""
    n = 1
    for d in shape:
        n *= int(d)
    mask = torch.empty(tuple(shape), dtype=torch.uint8, device=device)
    if n == 0:
        return mask
    seed = torch.randint(-(1 << 62), 1 << 62, (2,), dtype=torch.int64, device=device)
    t0 = _hbm_start()
    call("segk_dropout_mask", _ptr(mask), n, ctypes.c_float(p), _ptr(seed), _stream())
    _hbm_stop("dropout_mask", n, t0)
    return mask


# ------------------------------------------------------- pooling / resample ----
def maxpool3x3s2_ok(v):
    """Whether maxpool3x3s2 takes the pixel-major view v: channels a multiple of 8, 16-byte aligned rows."""
    return (v is not None and v.is_cuda and v.dim() == 2 and v.stride(1) == 1 and v.dtype in (torch.float32, torch.bfloat16)
            and v.shape[1] % 8 == 0 and v.stride(0) % 8 == 0 and v.data_ptr() % 16 == 0)


def maxpool3x3s2(x, N, H, W):
    """nn.MaxPool2d(3, 2, 1) of the pixel-major view x [N*H*W, C] -> (y [N*Ho*Wo, C], idx uint8 [N*Ho*Wo, C] = position
    of each maximum inside its window), Ho = (H - 1) // 2 + 1."""
    _dev(x)
    ld, C = _pm(x)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N * Ho * Wo, C), dtype=x.dtype, device=x.device)
    idx = torch.empty((N * Ho * Wo, C), dtype=torch.uint8, device=x.device)
    t0 = _hbm_start()
    call("segk_maxpool3x3s2_fwd", _ptr(x), _dt(x), ld, N, C, H, W, _ptr(y), C, _ptr(idx), _stream())
    _hbm_stop("maxpool_fwd", _nbytes(x, y, idx), t0)
    return y, idx


def maxpool3x3s2_bwd(dy, idx, N, C, H, W):
    """Gradient of maxpool3x3s2: dy [N*Ho*Wo, C] (pixel-major view) -> dx [N*H*W, C] of dy's dtype."""
    _dev(dy, idx)
    ld, c = _pm(dy)
    if c != C or idx.shape != (dy.shape[0], C) or not idx.is_contiguous():
        raise ValueError("gradient %s / positions %s do not belong to %d channels" % (tuple(dy.shape), tuple(idx.shape), C))
    dx = torch.empty((N * H * W, C), dtype=dy.dtype, device=dy.device)
    t0 = _hbm_start()
    call("segk_maxpool3x3s2_bwd", _ptr(dy), _dt(dy), ld, _ptr(idx), N, C, H, W, _ptr(dx), C, _stream())
    _hbm_stop("maxpool_bwd", _nbytes(dy, idx, dx), t0)
    return dx


def pool_sum(x, N, HW, scale):
    """x: pixel-major view [N*HW, C] -> [N, C] float32 = scale * sum over the pixels of each image."""
    _dev(x)
    ld, C = _pm(x)
    out = torch.empty((N, C), dtype=torch.float32, device=x.device)
    call("segk_pool_sum", _ptr(x), _dt(x), ld, N, HW, C, ctypes.c_float(scale), _ptr(out), _stream())
    return out


def upsample_fwd(src, N, C, Hi, Wi, dst, Ho, Wo, dst_layout=PM):
    """src: pixel-major [N*Hi*Wi, >=C]; dst: pixel-major view [N*Ho*Wo, C] or NCHW float32 [N,C,Ho,Wo]."""
    _dev(src, dst)
    ld = _pm(src)[0]
    if dst_layout == PM:
        out_ld = _pm(dst)[0]
    else:
        out_ld = 0
        if tuple(dst.shape) != (N, C, Ho, Wo) or not dst.is_contiguous():
            raise ValueError("NCHW destination must be contiguous [N,C,Ho,Wo]")
    t0 = _hbm_start()
    call("segk_upsample_fwd", _ptr(src), _dt(src), ld, N, C, Hi, Wi, _ptr(dst), _dt(dst), dst_layout, out_ld, Ho, Wo,
         _stream())
    # algorithmic bytes: C channels of every source and destination pixel once
    _hbm_stop("upsample_fwd", N * C * (Hi * Wi * src.element_size() + Ho * Wo * dst.element_size()), t0)
    return dst


def upsample_bwd(dout, dout_layout, N, C, Ho, Wo, din, cpad, Hi, Wi):
    """din: pixel-major view [N*Hi*Wi, cpad]; channels C..cpad are zero-filled."""
    _dev(dout, din)
    dout_ld = _pm(dout)[0] if dout_layout == PM else 0
    if dout_layout == NCHW and not dout.is_contiguous():
        raise ValueError("NCHW gradient must be contiguous")
    din_ld = _pm(din)[0]
    n_ws = call("segk_upsample_bwd_workspace", dout_layout, N, C, Ho, Wi)
    ws = torch.empty((n_ws,), dtype=torch.float32, device=dout.device) if n_ws > 0 else None
    t0 = _hbm_start()
    call("segk_upsample_bwd", _ptr(dout), _dt(dout), dout_layout, dout_ld, N, C, Ho, Wo, _ptr(din), _dt(din), din_ld,
         cpad, Hi, Wi, _ptr(ws), _stream())
    _hbm_stop("upsample_bwd", N * (C * Ho * Wo * dout.element_size() + cpad * Hi * Wi * din.element_size()), t0)
    return din


# -------------------------------------------------------------------- loss ----
def ce_fwd(logits, target, weight, ignore_index):
    """-> (sums [2] = {sum w*nll, sum w}, lse [N,H,W]) on device; logits NCHW float32, target float32."""
    _dev(logits, target, weight)
    if logits.dtype != torch.float32 or not logits.is_contiguous():
        raise ValueError("logits must be contiguous float32 NCHW")
    N, K, H, W = logits.shape
    if target.dtype != torch.float32 or not target.is_contiguous() or target.numel() != N * H * W:
        raise ValueError("target must be contiguous float32 [N,H,W]")
    if weight is not None and (weight.dtype != torch.float32 or weight.numel() != K):
        raise ValueError("class weight must be float32 [K]")
    rows = call("segk_ce_partial_rows", N * H * W)
    partials = torch.empty((rows, 2), dtype=torch.float32, device=logits.device)
    lse = torch.empty((N, H, W), dtype=torch.float32, device=logits.device)
    sums = torch.empty((2,), dtype=torch.float32, device=logits.device)
    t0 = _hbm_start()
    call("segk_ce_fwd", _ptr(logits), _ptr(target), _ptr(weight), N, K, H * W, ignore_index, _ptr(lse),
         _ptr(partials), _ptr(sums), _stream())
    _hbm_stop("ce_fwd", _nbytes(logits, target, lse), t0)
    return sums, lse


def ce_bwd(logits, target, weight, lse, ignore_index, coef):
    """coef: 0-dim/1-element float32 device tensor = grad_out / (sum w * batch)."""
    _dev(logits, target, weight, lse, coef)
    N, K, H, W = logits.shape
    dl = torch.empty_like(logits)
    t0 = _hbm_start()
    call("segk_ce_bwd", _ptr(logits), _ptr(target), _ptr(weight), _ptr(lse), N, K, H * W, ignore_index, _ptr(coef),
         _ptr(dl), _stream())
    _hbm_stop("ce_bwd", _nbytes(logits, target, lse, dl), t0)
    return dl


def _check_low(low, N, K, Hi, Wi, target, Ho, Wo, weight):
    if low.dtype != torch.float32 or low.dim() != 2 or low.shape[0] != N * Hi * Wi or low.shape[1] != K:
        raise ValueError("low-resolution logits must be a float32 pixel-major view [N*Hi*Wi, K]")
    if target.dtype != torch.float32 or not target.is_contiguous() or target.numel() != N * Ho * Wo:
        raise ValueError("target must be contiguous float32 [N,Ho,Wo]")
    if weight is not None and (weight.dtype != torch.float32 or weight.numel() != K):
        raise ValueError("class weight must be float32 [K]")


def upsample_ce_fwd(low, N, Hi, Wi, target, weight, Ho, Wo, ignore_index):
    """Loss sums of cross_entropy(interpolate(low, (Ho,Wo), align_corners=True), target) without the
    full-resolution logits.  low: float32 pixel-major view [N*Hi*Wi, K].  -> (sums [2], lse [N,Ho,Wo])."""
    _dev(low, target, weight)
    K = low.shape[1]
    _check_low(low, N, K, Hi, Wi, target, Ho, Wo, weight)
    rows = call("segk_ce_partial_rows", N * Ho * Wo)
    partials = torch.empty((rows, 2), dtype=torch.float32, device=low.device)
    lse = torch.empty((N, Ho, Wo), dtype=torch.float32, device=low.device)
    sums = torch.empty((2,), dtype=torch.float32, device=low.device)
    t0 = _hbm_start()
    call("segk_upsample_ce_fwd", _ptr(low), _pm(low)[0], N, K, Hi, Wi, _ptr(target), _ptr(weight), Ho, Wo,
         ignore_index, _ptr(lse), _ptr(partials), _ptr(sums), _stream())
    _hbm_stop("upsample_ce_fwd", N * Hi * Wi * K * 4 + _nbytes(target, lse), t0)
    return sums, lse


def upsample_ce_bwd(low, N, Hi, Wi, target, weight, lse, Ho, Wo, ignore_index, coef, cpad):
    """-> float32 [N*Hi*Wi, cpad]: gradient of the fused loss with respect to low (channels K..cpad zero)."""
    _dev(low, target, weight, lse, coef)
    K = low.shape[1]
    _check_low(low, N, K, Hi, Wi, target, Ho, Wo, weight)
    n_ws = call("segk_upsample_bwd_workspace", NCHW, N, K, Ho, Wi)
    ws = torch.empty((n_ws,), dtype=torch.float32, device=low.device)
    dlow = torch.empty((N * Hi * Wi, cpad), dtype=torch.float32, device=low.device)
    t0 = _hbm_start()
    call("segk_upsample_ce_bwd", _ptr(low), _pm(low)[0], N, K, Hi, Wi, _ptr(target), _ptr(weight), _ptr(lse), Ho, Wo,
         ignore_index, _ptr(coef), _ptr(ws), _ptr(dlow), cpad, cpad, _stream())
    _hbm_stop("upsample_ce_bwd", N * Hi * Wi * K * 4 + _nbytes(target, lse, dlow), t0)
    return dlow


def upsample_argmax(low, N, Hi, Wi, Ho, Wo):
    """argmax over the classes of the bilinearly (align_corners) upsampled logits; low: float32 pixel-major view
    [N*Hi*Wi, K].  -> int64 [N,Ho,Wo]."""
    _dev(low)
    if low.dtype != torch.float32 or low.dim() != 2 or low.shape[0] != N * Hi * Wi:
        raise ValueError("low-resolution logits must be a float32 pixel-major view [N*Hi*Wi, K]")
    pred = torch.empty((N, Ho, Wo), dtype=torch.int64, device=low.device)
    t0 = _hbm_start()
    call("segk_upsample_argmax", _ptr(low), _pm(low)[0], N, low.shape[1], Hi, Wi, Ho, Wo, _ptr(pred), _stream())
    _hbm_stop("upsample_argmax", N * Hi * Wi * low.shape[1] * 4 + _nbytes(pred), t0)
    return pred


def argmax(logits):
    _dev(logits)
    if logits.dtype != torch.float32 or not logits.is_contiguous() or logits.dim() != 4:
        raise ValueError("logits must be contiguous float32 NCHW")
    N, K, H, W = logits.shape
    pred = torch.empty((N, H, W), dtype=torch.int64, device=logits.device)
    t0 = _hbm_start()
    call("segk_argmax", _ptr(logits), N, K, H * W, _ptr(pred), _stream())
    _hbm_stop("argmax", _nbytes(logits, pred), t0)
    return pred


def confmat_accumulate(gt, pred, K, cm):
    """cm: int64 [K*K+1] device accumulator (last slot: predictions outside [0,K))."""
    _dev(gt, pred, cm)
    if gt.dtype not in (torch.float32, torch.int64) or pred.dtype != torch.int64:
        raise ValueError("gt must be float32 or int64 and pred int64")
    if gt.numel() != pred.numel():
        raise ValueError("gt and pred sizes differ")
    if cm.dtype != torch.int64 or cm.numel() != K * K + 1 or not cm.is_contiguous():
        raise ValueError("cm must be contiguous int64 [K*K+1]")
    gt, pred = gt.contiguous(), pred.contiguous()
    t0 = _hbm_start()
    call("segk_confmat", _ptr(gt), 1 if gt.dtype == torch.int64 else 0, _ptr(pred), gt.numel(), K, _ptr(cm),
         _stream())
    _hbm_stop("confmat", _nbytes(gt, pred, cm), t0)   # the kernel's real input types: fp32 (or int64) labels, int64 predictions
    return cm
