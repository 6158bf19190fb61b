"""segk_ce_fwd_static (the entry point that allocates nothing: its partial rows live in a fixed device array
inside the library) against segk_ce_fwd with a caller workspace, through the C ABI: bit-identical loss sums and
log-sum-exp maps on two calls in a row, the second on a smaller input (the array keeps the first call's rows),
plus the oracle's loss value, class weights, an all-ignored batch and the size limit."""
import numpy as np
import pytest
import torch

from oracle import gen, ops_ref

pytestmark = pytest.mark.gpu


def _inputs(seed, n, K, h, w, dev):
    logits = torch.from_numpy(gen.normal(seed, "noalloc/logits", (n, K, h, w)) * 3.0).to(dev)
    target = torch.from_numpy(gen.labels(seed, "noalloc/labels", (n, h, w), K, border=2)).to(dev)
    return logits.contiguous(), target.contiguous()


def _both(logits, target, weight, ignore_index=255):
    from segk import ops
    from segk._lib import call
    N, K, H, W = logits.shape
    sums_ref, lse_ref = ops.ce_fwd(logits, target, weight, ignore_index)
    # outputs are allocated by the caller BEFORE the call; the call itself allocates nothing
    lse = torch.full((N, H, W), float("nan"), dtype=torch.float32, device=logits.device)
    sums = torch.full((2,), float("nan"), dtype=torch.float32, device=logits.device)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    call("segk_ce_fwd_static", ops._ptr(logits), ops._ptr(target), ops._ptr(weight), N, K, H * W, ignore_index,
         ops._ptr(lse), ops._ptr(sums), ops._stream())
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    return (sums_ref.cpu().numpy(), lse_ref.cpu().numpy()), (sums.cpu().numpy(), lse.cpu().numpy())


def test_static_scratch_equals_workspace_two_calls_in_a_row():
    dev = "cuda:0"
    torch.cuda.set_device(0)
    # first call: 8192 partial rows (the whole array); second: 3 rows, so stale rows sit right behind the live ones
    for shape in ((8, 21, 513, 513), (1, 21, 27, 25)):
        logits, target = _inputs(11, *shape, dev)
        (s0, l0), (s1, l1) = _both(logits, target, None)
        assert np.array_equal(s0, s1), (shape, s0, s1)
        assert np.array_equal(l0, l1)
        ref = float(ops_ref.ce_loss(logits.cpu(), target.cpu())) * shape[0]  # the oracle divides by the batch
        got = float(s1[0] / s1[1])
        assert abs(got - ref) <= 1e-5 * abs(ref), (shape, got, ref)


def test_static_scratch_weights_and_all_ignored():
    dev = "cuda:0"
    logits, target = _inputs(12, 2, 8, 33, 47, dev)
    weight = torch.linspace(0.5, 2.0, 8, device=dev)
    (s0, l0), (s1, l1) = _both(logits, target, weight)
    assert np.array_equal(s0, s1) and np.array_equal(l0, l1)
    target.fill_(255.0)
    (s0, l0), (s1, l1) = _both(logits, target, None)
    assert np.array_equal(s1, np.zeros(2, np.float32)) and np.array_equal(s0, s1) and np.array_equal(l0, l1)


def test_static_scratch_rejects_oversize():
    from segk import ops
    from segk._lib import SegkError, call, load
    limit = load().segk_ce_fwd_static_max_pixels()
    assert limit == 1 << 25
    dev = "cuda:0"
    t = torch.zeros(4, device=dev)
    # shape check happens on the host before any launch: tiny buffers are never touched
    with pytest.raises(SegkError, match="exceed"):
        call("segk_ce_fwd_static", ops._ptr(t), ops._ptr(t), None, 129, 1, 1 << 18, 255, ops._ptr(t), ops._ptr(t),
             ops._stream())
