"""Process-wide settings of the accelerated path."""
import contextlib

import torch

import os

_state = {"compute_dtype": torch.float32, "legacy_clamp_eps": False, "feeder_conv": "auto", "overlap_wgrad": False,
          "group_wgrad": os.environ.get("SEGK_GROUP_WGRAD", "1") != "0",
          "wgrad_second_stream": os.environ.get("SEGK_WGRAD_SECOND_STREAM", "0") == "1",
          "fuse_bwd_stats": os.environ.get("SEGK_FUSE_BWD_STATS", "1") != "0",
          "head_raw_fp32": os.environ.get("SEGK_HEAD_RAW_FP32", "0") == "1",
          # debug switches (each costs a host synchronisation or an extra pass where it acts)
          "debug_packs": os.environ.get("SEGK_DEBUG_PACKS", "") == "1",
          "debug_labels": os.environ.get("SEGK_DEBUG_LABELS", "") == "1",
          "debug_sync_stats": os.environ.get("SEGK_DEBUG_SYNC_STATS", "") == "1",
          "sync_exchange": os.environ.get("SEGK_SYNC_EXCHANGE", "auto")}


def set_compute_dtype(dtype):
    """Arithmetic type of the head's activations and MFMA operands: torch.float32 (exact fp32 MFMA,
    parity mode) or torch.bfloat16 (bf16 operands, fp32 accumulation; statistics, loss and master
    weights stay fp32)."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
    _state["compute_dtype"] = dtype


def compute_dtype():
    return _state["compute_dtype"]


@contextlib.contextmanager
def compute_dtype_as(dtype):
    old = _state["compute_dtype"]
    set_compute_dtype(dtype)
    try:
        yield
    finally:
        _state["compute_dtype"] = old


def set_legacy_clamp_eps(flag):
    """True: inv_std = clamp(var, eps)^-1/2, the replicated-training formula of the reference
    (/root/reference/modeling/sync_batchnorm/batchnorm.py:125); False (default): (var + eps)^-1/2, the
    F.batch_norm formula its un-replicated and eval paths use (batchnorm.py:50-53)."""
    _state["legacy_clamp_eps"] = bool(flag)


def legacy_clamp_eps():
    return _state["legacy_clamp_eps"]


def set_feeder_conv(mode):
    """Which kernels run the feeder's stride-1 3x3 bottleneck convolutions with 128 or more channels
    (/root/reference/newmodeling/backbone/resnet101_2.py:19-21; ResNet layers 2 to 4) together with the batch norm
    and ReLU behind them, when that batch norm is a SynchronizedBatchNorm2d:
      "auto"  (default) the HIP implicit-GEMM kernels when the activation already is pixel-major memory
              (a channels_last feeder, e.g. under bf16 autocast), PyTorch-ROCm / MIOpen otherwise;
      "hip"   the HIP kernels always (NCHW activations are transposed on the way in and out);
      "torch" never."""
    if mode not in ("auto", "hip", "torch"):
        raise ValueError("feeder_conv mode must be 'auto', 'hip' or 'torch'")
    _state["feeder_conv"] = mode


def feeder_conv():
    return _state["feeder_conv"]


@contextlib.contextmanager
def feeder_conv_as(mode):
    old = _state["feeder_conv"]
    set_feeder_conv(mode)
    try:
        yield
    finally:
        _state["feeder_conv"] = old


def set_overlap_wgrad(flag):
    """True: inside ``loss.backward()`` the weight gradients of the HIP convolution units are taken off the chain of
    data gradients and batch-norm backward passes (nothing but the optimizer waits for a weight gradient): they are
    queued, launched in groups of one shape (set_group_wgrad) and written to ``weight.grad`` by an end-of-backward
    callback, not returned through autograd: use it with ``.backward()`` loops only (``torch.autograd.grad`` would see
    None for those weights, and hooks on the weights do not fire).  Their launches go to the launching stream by
    default and to a second HIP stream with set_wgrad_second_stream(True).  Ignored while a segk.parallel.GradientBuckets with
    overlap=True exists: its all-reduces are started from parameter hooks.  Default False."""
    _state["overlap_wgrad"] = bool(flag)


def overlap_wgrad():
    return _state["overlap_wgrad"]


@contextlib.contextmanager
def overlap_wgrad_as(flag):
    old = _state["overlap_wgrad"]
    set_overlap_wgrad(flag)
    try:
        yield
    finally:
        _state["overlap_wgrad"] = old


def set_wgrad_second_stream(flag):
    """True (or SEGK_WGRAD_SECOND_STREAM=1): deferred weight gradients (set_overlap_wgrad) run on a second HIP stream
    beside the rest of the backward pass.  Default False: measured on MI355X the two streams only take compute units
    from each other - every ring-tile kernel fills a CU's LDS and registers, so nothing co-resides with it, and the
    memory-bound batch-norm launches run at 60 % of their speed beside a weight gradient - 24.61 against 24.91 ms per
    step on one box once the weight gradients are grouped."""
    _state["wgrad_second_stream"] = bool(flag)


def wgrad_second_stream():
    return _state["wgrad_second_stream"]


@contextlib.contextmanager
def wgrad_second_stream_as(flag):
    old = _state["wgrad_second_stream"]
    set_wgrad_second_stream(flag)
    try:
        yield
    finally:
        _state["wgrad_second_stream"] = old


def set_group_wgrad(flag):
    """True (default; SEGK_GROUP_WGRAD=0 turns it off): with overlap_wgrad, weight gradients of one shape that a single
    launch cannot fill the chip with (the repeated bottlenecks of a ResNet stage on 33 x 33 maps) are collected, up to
    eight at a time, and computed by one launch on the second stream.  A gradient waits at most until eight further
    weight gradients have passed without one of its shape, or until the end of the backward pass.  The values differ in
    the last bits from the ungrouped ones (the pixel range is cut into fewer parts)."""
    _state["group_wgrad"] = bool(flag)


def group_wgrad():
    return _state["group_wgrad"]


@contextlib.contextmanager
def group_wgrad_as(flag):
    old = _state["group_wgrad"]
    set_group_wgrad(flag)
    try:
        yield
    finally:
        _state["group_wgrad"] = old


def set_fuse_bwd_stats(flag):
    """True (default; SEGK_FUSE_BWD_STATS=0 turns it off): where a feeder unit's output reaches exactly one other unit,
    that unit's data-gradient convolution emits the backward statistics of the first unit's batch norm from its
    epilogue (segk.functional.StatLink) and the stand-alone pass over the gradient and the pre-normalisation output is
    skipped.  The sums are formed in another order than the stand-alone pass forms them: gradients differ in the last
    bits between the two settings, each is reproducible run to run."""
    _state["fuse_bwd_stats"] = bool(flag)


def fuse_bwd_stats():
    return _state["fuse_bwd_stats"]


@contextlib.contextmanager
def fuse_bwd_stats_as(flag):
    old = _state["fuse_bwd_stats"]
    set_fuse_bwd_stats(flag)
    try:
        yield
    finally:
        _state["fuse_bwd_stats"] = old


def set_head_raw_fp32(flag):
    """False (default; SEGK_HEAD_RAW_FP32=1 turns it on): in bf16 mode the head's convolutions write their
    pre-normalisation output in bf16, as the feeder's units and PyTorch's autocast do (half the bytes for the three
    batch-norm passes that read it).  True keeps it in fp32, the accumulator's precision: the normalised values then
    refer to exactly the numbers the statistics were taken from, and a layer rounds to bf16 once instead of twice."""
    _state["head_raw_fp32"] = bool(flag)


def head_raw_fp32():
    return _state["head_raw_fp32"]


@contextlib.contextmanager
def head_raw_fp32_as(flag):
    old = _state["head_raw_fp32"]
    set_head_raw_fp32(flag)
    try:
        yield
    finally:
        _state["head_raw_fp32"] = old


# ---- debug switches ------------------------------------------------------------------------------------------
def set_debug_packs(flag):
    """True (or SEGK_DEBUG_PACKS=1): every use of a kept packed weight operand (segk.ops.cache_weight_packs) packs the
    weight again and compares; a stale operand - a registered weight changed in place through ``.data``, which moves
    no version counter - raises instead of giving wrong results silently."""
    _state["debug_packs"] = bool(flag)


def debug_packs():
    return _state["debug_packs"]


def set_debug_labels(flag):
    """True (or SEGK_DEBUG_LABELS=1): the cross-entropy counts labels that are neither the ignore index nor a class
    index and raises, as the reference does inside ATen (/root/reference/utils/loss.py:25-30); without it the
    kernels treat such labels as ignored.  Costs one pass over the labels and a host synchronisation."""
    _state["debug_labels"] = bool(flag)


def debug_labels():
    return _state["debug_labels"]


def set_debug_sync_stats(flag):
    """True (or SEGK_DEBUG_SYNC_STATS=1): after every exchange of batch-norm statistics between ranks the summed
    vectors are gathered and compared bit for bit; ranks that normalise with different statistics raise.  (The
    reference's only guards are the asserts of /root/reference/modeling/sync_batchnorm/comm.py:28,90,115,123.)"""
    _state["debug_sync_stats"] = bool(flag)


def debug_sync_stats():
    return _state["debug_sync_stats"]


def set_sync_exchange(mode):
    """How ranks exchange the small vectors (batch-norm statistics, loss sums): "auto" - through peer-mapped
    mailboxes when they can be connected and pass their self-test, else through torch.distributed; "mailbox" -
    mailboxes or an error; "dist" - torch.distributed always.  Read when the process group is initialised
    (segk.parallel.init_from_env); environment: SEGK_SYNC_EXCHANGE."""
    if mode not in ("auto", "mailbox", "dist"):
        raise ValueError("sync_exchange mode must be 'auto', 'mailbox' or 'dist'")
    _state["sync_exchange"] = mode


def sync_exchange():
    m = _state["sync_exchange"]
    return m if m in ("auto", "mailbox", "dist") else "auto"
