"""Portable synthetic-tensor generator (test infrastructure).

Values depend only on (seed, tag, flat index) through 64-bit integer mixing and
exactly-rounded float64 adds/multiplies, so the fixture builder and the GPU box
regenerate identical tensors without shipping them.  No libm calls are used.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    # splitmix64 finaliser on a uint64 array (wrap-around arithmetic is intended)
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _key(seed, tag):
    h = zlib.crc32(tag.encode("utf-8")) & 0xFFFFFFFF
    return np.uint64(((int(seed) & 0xFFFFFFFF) << 32) | h)


def bits(seed, tag, count, stream=0):
    """uint64 hash of (seed, tag, stream, i) for i in range(count)."""
    idx = np.arange(count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = _mix(_key(seed, tag) ^ (np.uint64(stream) * np.uint64(0xD6E8FEB86659FD93)))
        return _mix(idx ^ base)


def uniform(seed, tag, shape, lo=0.0, hi=1.0):
    """float32 uniform in [lo, hi) with 24 random bits per element."""
    n = int(np.prod(shape))
    u = (bits(seed, tag, n) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, tag, shape, std=1.0, mean=0.0):
    """float32, approximately N(mean, std^2): scaled sum of four uniforms (Irwin-Hall)."""
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.float64)
    for s in range(4):
        acc += (bits(seed, tag, n, stream=s + 1) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    z = (acc - 2.0) * 1.7320508075688772  # var(sum of 4 U) = 1/3
    return (mean + std * z).astype(np.float32).reshape(shape)


def relu_normal(seed, tag, shape, std=1.0):
    """Post-ReLU feature statistics: max(N(0, std^2), 0)."""
    return np.maximum(normal(seed, tag, shape, std=std), 0.0).astype(np.float32)


def integers(seed, tag, shape, lo, hi):
    """int64 uniform integers in [lo, hi)."""
    n = int(np.prod(shape))
    r = bits(seed, tag, n) >> np.uint64(33)
    return (lo + (r % np.uint64(hi - lo)).astype(np.int64)).reshape(shape)


def labels(seed, tag, shape, num_class, border=0, ignore_frac=0.02, ignore_index=255):
    """float32 label maps [N,H,W] the way the reference's loaders hand them over:
    integers in [0,num_class) with an ignore border and a random sprinkle of 255."""
    n, h, w = shape
    lab = integers(seed, tag, shape, 0, num_class)
    drop = uniform(seed, tag + "/ign", shape) < ignore_frac
    lab[drop] = ignore_index
    if border > 0:
        lab[:, :border, :] = ignore_index
        lab[:, h - border:, :] = ignore_index
        lab[:, :, :border] = ignore_index
        lab[:, :, w - border:] = ignore_index
    return lab.astype(np.float32)


def fill_state_dict(state_dict, seed, prefix=""):
    """Deterministically overwrite every floating tensor of a torch state_dict.

    conv / linear weights: N(0, sqrt(2/fan_in)); BN weight ~ U(0.5,1.5); biases and
    running_mean ~ N(0, 0.1); running_var ~ U(0.5, 1.5); integer buffers untouched.
    """
    import torch

    out = {}
    for name, t in state_dict.items():
        tag = prefix + name
        if not torch.is_floating_point(t):
            out[name] = t.clone()
            continue
        shp = tuple(t.shape)
        if t.dim() >= 2:
            fan_in = int(np.prod(shp[1:]))
            v = normal(seed, tag, shp, std=float(np.sqrt(2.0 / fan_in)))
        elif name.endswith("running_var"):
            v = uniform(seed, tag, shp, 0.5, 1.5)
        elif name.endswith("running_mean") or name.endswith("bias"):
            v = normal(seed, tag, shp, std=0.1)
        else:  # BN gamma
            v = uniform(seed, tag, shp, 0.5, 1.5)
        out[name] = torch.from_numpy(v.copy())
    return out
