"""Process-wide settings of the accelerated path."""
import contextlib

import torch

_state = {"compute_dtype": torch.float32, "legacy_clamp_eps": False}


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
