"""Comparison helpers shared by the CPU and GPU parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files if not k.startswith("_")}


def rel_err(a, b):
    """max |a-b| / max|b| — the 'relative' of the parity gates, robust to zeros in b."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        raise AssertionError("shape %s vs %s" % (a.shape, b.shape))
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    if not np.array_equal(nan_a, nan_b):
        return float("inf")
    if nan_b.all():
        return 0.0
    a = a[~nan_b]
    b = b[~nan_b]
    scale = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / scale)


def rel_elementwise(a, b, floor=1e-3):
    """Largest |a-b| / |b| over the elements with |b| > floor * max|b| (the element-wise reading of a relative
    tolerance; elements that are small against the tensor's scale have no meaningful relative error), and that
    element's reference value."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    keep = np.abs(b) > floor * np.abs(b).max()
    if not keep.any():
        return 0.0, 0.0
    r = np.abs(a[keep] - b[keep]) / np.abs(b[keep])
    i = int(np.argmax(r))
    return float(r[i]), float(b[keep][i])


def rel_l2(a, b):
    """||a-b||_2 / ||b||_2."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def assert_dicts_close(got, want, tol, grad_tol=None, exact=(), skip=()):
    """Keys starting with 'train_d'/'eval_d'/'d' (gradients) use grad_tol."""
    bad = []
    for k, w in want.items():
        if k in skip:
            continue
        assert k in got, "missing key %s" % k
        if k in exact:
            if not np.array_equal(np.asarray(got[k]), w, equal_nan=True):
                bad.append((k, "not bit-exact"))
            continue
        is_grad = ("_d" in k) or k.startswith("d") or "/d" in k
        t = grad_tol if (grad_tol is not None and is_grad) else tol
        e = rel_err(got[k], w)
        if not e <= t:
            bad.append((k, e, t))
    assert not bad, "parity failures: %s" % (bad,)
