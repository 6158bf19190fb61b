"""The feeder's stride-1 3x3 bottleneck convolution + batch norm + ReLU as one unit on the implicit-GEMM kernels
(/root/reference/newmodeling/backbone/resnet101_2.py:19-21 and Bottleneck.forward: conv2 -> bn2 -> relu) against the
same three PyTorch operations in fp32 on the host, evaluated on the activation pattern the kernels produced."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import rel_err
from oracle import gen

pytestmark = pytest.mark.gpu
SEED = 777


def _reference(x, w, gamma, beta, gy, dil, mask):
    x = x.clone().requires_grad_(True)
    w = w.clone().requires_grad_(True)
    gamma = gamma.clone().requires_grad_(True)
    beta = beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(w.shape[0]), torch.ones(w.shape[0])
    z = F.batch_norm(F.conv2d(x, w, None, 1, dil, dil), rm, rv, gamma, beta, True, 0.1, 1e-5)
    y = z * mask
    y.backward(gy)
    return {"y": y.detach(), "dx": x.grad, "dw": w.grad, "dgamma": gamma.grad, "dbeta": beta.grad, "rm": rm, "rv": rv}


def _inputs(C, n, h, w):
    x = torch.from_numpy(gen.relu_normal(SEED, "fc/x", (n, C, h, w)))
    wt = torch.from_numpy(gen.normal(SEED, "fc/w", (C, C, 3, 3), std=(2.0 / (9 * C)) ** 0.5))
    gamma = torch.from_numpy(gen.uniform(SEED, "fc/g", (C,), 0.5, 1.5))
    beta = torch.from_numpy(gen.normal(SEED, "fc/b", (C,), std=0.3))
    gy = torch.from_numpy(gen.normal(SEED, "fc/gy", (n, C, h, w)))
    return x, wt, gamma, beta, gy


def _modules(C, dil, wt, gamma, beta, dev):
    from segk.modeling.sync_batchnorm import SynchronizedBatchNorm2d
    conv = nn.Conv2d(C, C, 3, padding=dil, dilation=dil, bias=False).to(dev)
    bn = SynchronizedBatchNorm2d(C).to(dev)
    with torch.no_grad():
        conv.weight.copy_(wt)
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    return conv, bn


@pytest.mark.parametrize("dil", [1, 2, 4])
def test_unit_fp32_nchw(dil):
    from segk import config
    dev, C = "cuda", 256
    x, wt, gamma, beta, gy = _inputs(C, 3, 9, 11)
    conv, bn = _modules(C, dil, wt, gamma, beta, dev)
    xd = x.to(dev).requires_grad_(True)
    with config.feeder_conv_as("hip"):
        assert bn.can_fuse_conv(conv, xd)
        y = bn.forward_conv_relu(conv, xd)
    with config.feeder_conv_as("auto"):
        assert not bn.can_fuse_conv(conv, xd)             # NCHW activations stay on the PyTorch path by default
        assert bn.can_fuse_conv(conv, xd.detach().contiguous(memory_format=torch.channels_last))
    with config.feeder_conv_as("torch"):
        assert not bn.can_fuse_conv(conv, xd.detach().contiguous(memory_format=torch.channels_last))
    assert y.is_contiguous() and y.dtype == torch.float32
    y.backward(gy.to(dev))
    ref = _reference(x, wt, gamma, beta, gy, dil, (y.detach().cpu() > 0).float())
    assert rel_err(y.detach().cpu().numpy(), ref["y"].numpy()) < 1e-4
    assert rel_err(bn.running_mean.cpu().numpy(), ref["rm"].numpy()) < 1e-4
    assert rel_err(bn.running_var.cpu().numpy(), ref["rv"].numpy()) < 1e-4
    for k, g in (("dx", xd.grad), ("dw", conv.weight.grad), ("dgamma", bn.weight.grad), ("dbeta", bn.bias.grad)):
        assert rel_err(g.cpu().numpy(), ref[k].numpy()) < 1e-3, k


def test_unit_bf16_channels_last_weight_gradient_layout():
    """The bench configuration: bf16 channels_last activations and a bf16 channels_last weight (held so by the
    optimizer); the weight gradient comes back in that dtype and layout."""
    dev, C, dil = "cuda", 512, 2
    x, wt, gamma, beta, gy = _inputs(C, 3, 9, 11)
    conv, bn = _modules(C, dil, wt, gamma, beta, dev)
    conv = conv.to(memory_format=torch.channels_last)
    conv.weight.data = conv.weight.data.to(torch.bfloat16)
    xd = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert bn.can_fuse_conv(conv, xd)                     # mode "auto": pixel-major memory
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        y = bn.forward_conv_relu(conv, xd)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    gw = conv.weight.grad
    assert gw.dtype == torch.bfloat16 and gw.stride() == conv.weight.stride()
    assert xd.grad.dtype == torch.bfloat16 and xd.grad.stride() == xd.stride()
    ref = _reference(x.bfloat16().float(), wt.bfloat16().float(), gamma, beta, gy.bfloat16().float(), dil,
                     (y.detach().float().cpu() > 0).float())
    assert rel_err(y.detach().float().cpu().numpy(), ref["y"].numpy()) < 2e-2
    for k, g in (("dx", xd.grad), ("dw", gw), ("dgamma", bn.weight.grad), ("dbeta", bn.bias.grad)):
        assert rel_err(g.float().cpu().numpy(), ref[k].numpy()) < 2e-2, k


def test_unit_1x1_bf16_channels_last():
    """conv1 -> bn1 -> relu of a layer-3 bottleneck (1x1, 1024 -> 256) as the same unit; narrower 1x1 convolutions
    stay on the PyTorch path."""
    from segk.modeling.sync_batchnorm import SynchronizedBatchNorm2d
    dev, Cin, Cout, n, h, w = "cuda", 1024, 256, 3, 9, 11
    x = torch.from_numpy(gen.relu_normal(SEED, "f1/x", (n, Cin, h, w)))
    wt = torch.from_numpy(gen.normal(SEED, "f1/w", (Cout, Cin, 1, 1), std=(2.0 / Cin) ** 0.5))
    gamma = torch.from_numpy(gen.uniform(SEED, "f1/g", (Cout,), 0.5, 1.5))
    beta = torch.from_numpy(gen.normal(SEED, "f1/b", (Cout,), std=0.3))
    gy = torch.from_numpy(gen.normal(SEED, "f1/gy", (n, Cout, h, w)))
    conv = nn.Conv2d(Cin, Cout, 1, bias=False).to(dev).to(memory_format=torch.channels_last)
    bn = SynchronizedBatchNorm2d(Cout).to(dev)
    with torch.no_grad():
        conv.weight.copy_(wt)
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    conv.weight.data = conv.weight.data.to(torch.bfloat16)
    xd = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert bn.can_fuse_conv(conv, xd)
    assert SynchronizedBatchNorm2d(128).to(dev).can_fuse_conv(nn.Conv2d(512, 128, 1, bias=False).to(dev), xd[:, :512])
    assert not SynchronizedBatchNorm2d(64).to(dev).can_fuse_conv(nn.Conv2d(64, 64, 1, bias=False).to(dev), xd[:, :64])
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        y = bn.forward_conv_relu(conv, xd)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    gw = conv.weight.grad
    assert gw.dtype == torch.bfloat16 and gw.shape == conv.weight.shape
    xr = x.bfloat16().float().requires_grad_(True)
    wr = wt.bfloat16().float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.batch_norm(F.conv2d(xr, wr), torch.zeros(Cout), torch.ones(Cout), gr, br, True, 0.1, 1e-5)
    (z * (y.detach().float().cpu() > 0).float()).backward(gy.bfloat16().float())
    assert rel_err(y.detach().float().cpu().numpy(), torch.relu(z).detach().numpy()) < 2e-2
    for k, g, r in (("dx", xd.grad, xr.grad), ("dw", gw, wr.grad), ("dgamma", bn.weight.grad, gr.grad),
                    ("dbeta", bn.bias.grad, br.grad)):
        assert rel_err(g.float().cpu().numpy(), r.numpy()) < 2e-2, k


def test_unit_1x1_residual_bf16_channels_last():
    """conv3 -> bn3 -> += residual -> relu of a layer-3 bottleneck (1x1, 256 -> 1024) as one unit, gradient of the
    residual included; 64 -> 256 (ResNet layer 1) stays on the PyTorch path."""
    from segk.modeling.sync_batchnorm import SynchronizedBatchNorm2d
    dev, Cin, Cout, n, h, w = "cuda", 256, 1024, 3, 9, 11
    x = torch.from_numpy(gen.relu_normal(SEED, "f3/x", (n, Cin, h, w)))
    res = torch.from_numpy(gen.relu_normal(SEED, "f3/r", (n, Cout, h, w)))
    wt = torch.from_numpy(gen.normal(SEED, "f3/w", (Cout, Cin, 1, 1), std=(2.0 / Cin) ** 0.5))
    gamma = torch.from_numpy(gen.uniform(SEED, "f3/g", (Cout,), 0.5, 1.5))
    beta = torch.from_numpy(gen.normal(SEED, "f3/b", (Cout,), std=0.3))
    gy = torch.from_numpy(gen.normal(SEED, "f3/gy", (n, Cout, h, w)))
    conv = nn.Conv2d(Cin, Cout, 1, bias=False).to(dev).to(memory_format=torch.channels_last)
    bn = SynchronizedBatchNorm2d(Cout).to(dev)
    with torch.no_grad():
        conv.weight.copy_(wt)
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    conv.weight.data = conv.weight.data.to(torch.bfloat16)
    cl = lambda t: t.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)  # noqa: E731
    xd, rd = cl(x), cl(res)
    assert bn.can_fuse_conv_residual(conv, xd, rd)
    assert not bn.can_fuse_conv_residual(conv, xd, rd.detach().float())
    small = nn.Conv2d(64, 256, 1, bias=False).to(dev)
    assert not SynchronizedBatchNorm2d(256).to(dev).can_fuse_conv_residual(small, xd[:, :64], rd[:, :256])
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        y = bn.forward_conv_add_relu(conv, xd, rd)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
    xr = x.bfloat16().float().requires_grad_(True)
    rr = res.bfloat16().float().requires_grad_(True)
    wr = wt.bfloat16().float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    z = F.batch_norm(F.conv2d(xr, wr), rm, rv, gr, br, True, 0.1, 1e-5) + rr
    (z * (y.detach().float().cpu() > 0).float()).backward(gy.bfloat16().float())
    assert rel_err(y.detach().float().cpu().numpy(), torch.relu(z).detach().numpy()) < 2e-2
    assert rel_err(bn.running_mean.cpu().numpy(), rm.numpy()) < 1e-2
    assert rel_err(bn.running_var.cpu().numpy(), rv.numpy()) < 1e-2
    assert conv.weight.grad.dtype == torch.bfloat16 and rd.grad.stride() == rd.stride()
    for k, g, r in (("dx", xd.grad, xr.grad), ("dres", rd.grad, rr.grad), ("dw", conv.weight.grad, wr.grad),
                    ("dgamma", bn.weight.grad, gr.grad), ("dbeta", bn.bias.grad, br.grad)):
        assert rel_err(g.float().cpu().numpy(), r.numpy()) < 2e-2, k


def test_pack_weight_layouts_agree():
    """segk_pack_weight_strided: fp32 OIHW, bf16 OIHW and bf16 channels_last sources give the same operand."""
    from segk import ops
    dev = "cuda"
    w = torch.from_numpy(gen.normal(SEED, "pw/w", (70, 100, 3, 3))).to(dev)
    wb = w.to(torch.bfloat16)
    for transpose in (False, True):
        a = ops.pack_weight(wb.float(), torch.bfloat16, transpose)
        b = ops.pack_weight(wb, torch.bfloat16, transpose)
        c = ops.pack_weight(wb.contiguous(memory_format=torch.channels_last), torch.bfloat16, transpose)
        assert torch.equal(a, b) and torch.equal(a, c)
        rows, k = (100, 70) if transpose else (70, 100)
        assert tuple(a.shape) == (9, rows, ops.round_up(k, 32))
        src = wb.float().flip(2, 3).permute(2, 3, 1, 0) if transpose else wb.float().permute(2, 3, 0, 1)
        want = torch.zeros(a.shape, device=dev)
        want[:, :, :k] = src.reshape(9, rows, k)
        assert torch.equal(a.float(), want)
    w1 = torch.from_numpy(gen.normal(SEED, "pw/w1", (48, 256, 1, 1))).to(dev)
    p1 = ops.pack_weight(w1, torch.float32)
    assert torch.equal(p1[0], w1.reshape(48, 256))


def test_unit_1x1_residual_backward_takes_unaligned_gradients():
    """The residual unit keeps one bit per element for its ReLU, which the backward kernels read eight channels at a
    time; a gradient that arrives as a channel slice of a wider tensor (row stride not the channel count, base not
    16-byte aligned) is copied into an aligned buffer first instead of being refused."""
    from segk.modeling.sync_batchnorm import SynchronizedBatchNorm2d
    dev, Cin, Cout, n, h, w = "cuda", 256, 1024, 2, 5, 7
    torch.manual_seed(2)
    conv = nn.Conv2d(Cin, Cout, 1, bias=False).to(dev).to(memory_format=torch.channels_last)
    conv.weight.data = conv.weight.data.to(torch.bfloat16)
    cl = lambda t: t.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    x, res = cl(torch.randn(n, Cin, h, w).relu()), cl(torch.randn(n, Cout, h, w).relu())
    wide = cl(torch.randn(n, Cout + 4, h, w))
    grads = {}
    for name, gy in (("aligned", wide[:, 4:].contiguous(memory_format=torch.channels_last)), ("slice", wide[:, 4:])):
        bn = SynchronizedBatchNorm2d(Cout).to(dev)
        xd, rd = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        conv.weight.grad = None
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            y = bn.forward_conv_add_relu(conv, xd, rd)
        y.backward(gy)
        grads[name] = [xd.grad.clone(), rd.grad.clone(), conv.weight.grad.clone(), bn.weight.grad.clone()]
    assert wide[:, 4:].data_ptr() % 16 != 0
    for a, b in zip(grads["aligned"], grads["slice"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("planes,hw,proj", [(256, 9, False), (64, 17, False), (256, 9, True), (128, 11, True)])
def test_bottleneck_shortcut_gradient_added_in_the_data_gradient(planes, hw, proj):
    """A bottleneck (newmodeling/backbone/resnet101_2.py:31-51): its input feeds conv1 and the shortcut.  With
    the gradient slot the shortcut's gradient is added in the epilogue of conv1's data-gradient convolution; without
    it autograd adds the two tensors.  Same bits in the input gradient and in every parameter gradient.  (256 planes:
    all three units on the HIP kernels; 64 planes, ResNet layer 1: conv1 on them, the shortcut's gradient comes out of
    the stand-alone batch norm with residual.  proj: the first block of a stage, whose shortcut is a 1 x 1 convolution
    with batch norm - the projection's input gradient goes through the slot.)"""
    from segk import functional as SF
    from segk.modeling.backbone.resnet import Bottleneck
    from segk.modeling.sync_batchnorm import SynchronizedBatchNorm2d
    dev = "cuda"
    torch.manual_seed(planes)
    inplanes = planes * 2 if proj else planes * 4
    down = None
    if proj:
        down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, kernel_size=1, bias=False), SynchronizedBatchNorm2d(planes * 4))
    blk = Bottleneck(inplanes, planes, 1, 1, down, SynchronizedBatchNorm2d).to(dev).to(memory_format=torch.channels_last)
    for m in blk.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.to(torch.bfloat16)
    x0 = torch.randn(2, inplanes, hw, hw + 2, device=dev).relu().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(2, planes * 4, hw, hw + 2, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    puts = []
    real_put = SF.GradSlot.put

    def counting_put(self, g, bits=None):
        puts.append(tuple(g.shape))
        return real_put(self, g, bits)

    def run(use_slot):
        import contextlib
        from unittest import mock
        off = mock.patch.object(SynchronizedBatchNorm2d, "new_grad_slot", staticmethod(lambda: None))
        with (contextlib.nullcontext() if use_slot else off), mock.patch.object(SF.GradSlot, "put", counting_put):
            blk.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                y = blk(x)
            y.backward(gy)
        return [x.grad.clone(), y.detach().clone()] + [p.grad.clone() for p in blk.parameters()]

    # (the 64-plane block's 3x3 and its last 1x1 run on MIOpen, whose default gradients differ from run to run)
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        with_slot, without = run(True), run(False)
    finally:
        torch.use_deterministic_algorithms(False)
    for a, b in zip(with_slot, without):
        assert torch.equal(a, b)
    assert float(with_slot[0].float().abs().max()) > 0
    assert len(puts) == 1, puts       # one gradient went through the slot, in the run that had one


def test_gradient_slot_returns_to_autograd_when_its_consumer_has_run():
    """GradToSlotFunction: should the consumer of the slot already have taken from it when the branch's gradient
    arrives (another execution order, a second pass over a kept graph), the gradient goes back to autograd."""
    from segk import functional as SF
    x = torch.ones(3, device="cuda", requires_grad=True)
    slot = SF.GradSlot()
    y = SF.grad_to_slot(x, slot) * 2.0
    y.sum().backward(retain_graph=True)
    assert x.grad is None or float(x.grad.abs().sum()) == 0.0     # the gradient waits in the slot
    g, bits = slot.take()
    assert bits is None and torch.equal(g, torch.full((3,), 2.0, device="cuda"))
    y.sum().backward()
    assert torch.equal(x.grad, torch.full((3,), 2.0, device="cuda"))   # the slot is spent: autograd gets it


def test_bottleneck_backward_statistics_from_the_data_gradient():
    """Three identical bottlenecks on a 33 x 33 map (newmodeling/backbone/resnet101_2.py:31-51): the data-gradient
    convolution of conv2 emits the backward statistics of bn1 and that of conv3 those of bn2; the stand-alone
    statistics pass then runs for bn3 only.  Gradients equal those of the stand-alone passes up to the summation order,
    and are the same bits run to run."""
    from unittest import mock

    from segk import config, ops
    from segk.modeling.backbone.resnet import Bottleneck
    from segk.modeling.sync_batchnorm import SynchronizedBatchNorm2d
    dev = "cuda"
    torch.manual_seed(5)
    planes = 256
    net = nn.Sequential(*[Bottleneck(planes * 4, planes, 1, 1, None, SynchronizedBatchNorm2d) for _ in range(3)])
    net = net.to(dev).to(memory_format=torch.channels_last)
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.to(torch.bfloat16)
    x0 = torch.randn(4, planes * 4, 33, 33, device=dev).relu().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn_like(x0)
    calls = []
    real = ops.bn_bwd_stat_partials

    def counting(dz, x, *a, **k):
        calls.append(tuple(x.shape))
        return real(dz, x, *a, **k)

    def rel(a, b):
        return float((a.float() - b.float()).abs().max() / b.float().abs().max())

    def run(fuse):
        net.zero_grad(set_to_none=True)
        del calls[:]
        x = x0.clone().requires_grad_(True)
        with config.fuse_bwd_stats_as(fuse), mock.patch.object(ops, "bn_bwd_stat_partials", counting):
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                y = net(x)
            y.backward(gy)
        torch.cuda.synchronize()
        return x.grad.clone(), [p.grad.clone() for p in net.parameters()], list(calls)

    gx0, alone, n_alone = run(False)
    assert len(n_alone) == 9, n_alone
    gx1, fused, n_fused = run(True)
    assert n_fused == [(4 * 33 * 33, planes * 4)] * 3, n_fused     # bn3 of the three blocks
    assert rel(gx1, gx0) < 2e-2
    for p, a, b in zip(net.parameters(), alone, fused):
        assert rel(b, a) < 2e-2, (tuple(p.shape), rel(b, a))
    gx2, again, _ = run(True)
    assert torch.equal(gx1, gx2)
    for a, b in zip(fused, again):
        assert torch.equal(a, b)


def test_repeated_bottlenecks_weight_gradients_in_groups():
    """Five identical bottlenecks on a 33 x 33 map (what ResNet-101's layer 3 repeats 22 times,
    newmodeling/backbone/resnet101_2.py:100-122): with overlap_wgrad the weight gradients of one shape are collected
    and computed by one launch per group.  Every gradient arrives, twice as large after a second backward pass, equal
    to the one-by-one gradients up to the summation order of the pixel parts; data gradients are untouched (same
    bits); a group never waits past the end of the pass, nor longer than GROUP_IDLE further gradients."""
    from unittest import mock

    from segk import config, functional as SF, ops
    from segk.modeling.backbone.resnet import Bottleneck
    from segk.modeling.sync_batchnorm import SynchronizedBatchNorm2d
    dev = "cuda"
    torch.manual_seed(3)
    planes = 256
    net = nn.Sequential(*[Bottleneck(planes * 4, planes, 1, 1, None, SynchronizedBatchNorm2d) for _ in range(5)])
    net = net.to(dev).to(memory_format=torch.channels_last)
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.to(torch.bfloat16)
    x0 = torch.randn(2, planes * 4, 33, 33, device=dev).relu().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn_like(x0)
    sizes = []

    def rel(a, b):
        return float((a.float() - b.float()).abs().max() / b.float().abs().max())

    real_group = ops.conv_wgrad_group

    def counting(xs, *a, **k):
        sizes.append(len(xs))
        return real_group(xs, *a, **k)

    def run(group, passes=1, second=False):
        net.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        with config.overlap_wgrad_as(True), config.group_wgrad_as(group), config.wgrad_second_stream_as(second), \
                mock.patch.object(ops, "conv_wgrad_group", counting):
            for _ in range(passes):
                with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                    y = net(x)
                y.backward(gy)
        torch.cuda.synchronize()
        assert not SF._GROUPS and not SF._PENDING
        return x.grad.clone(), [p.grad.clone() for p in net.parameters()]

    gx1, one_by_one = run(False)
    assert sizes == []
    gxg, grouped = run(True)
    # three shapes (1024->256, 3x3 256->256, 256->1024), five members each
    assert sorted(sizes) == [5, 5, 5], sizes
    assert torch.equal(gx1, gxg)
    for p, a, b in zip(net.parameters(), one_by_one, grouped):
        assert a.shape == b.shape == p.shape and a.dtype == b.dtype and a.stride() == b.stride()
        assert rel(b, a) < 1e-2, (tuple(p.shape), rel(b, a))   # one bf16 rounding step
    # the same launches on a second stream: the same bits
    _, on_second = run(True, second=True)
    for a, b in zip(grouped, on_second):
        assert torch.equal(a, b)
    gx2, twice = run(True, passes=2)
    for a, b in zip(grouped, twice):
        if a.dtype == torch.bfloat16:
            assert rel(b, 2 * a.float()) < 1e-6
    # the idle rule: with a short limit, groups are launched while the pass is still running
    del sizes[:]
    with mock.patch.object(SF, "GROUP_IDLE", 2):
        _, early = run(True)
    assert len(sizes) == 0 or max(sizes) < 5, sizes
    for a, b in zip(one_by_one, early):
        assert rel(b, a) < 1e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 64, 33, 35), (1, 8, 6, 7), (3, 16, 1, 9), (2, 24, 16, 2)])
def test_stem_maxpool_equals_aten(dtype, shape):
    """nn.MaxPool2d(3, 2, 1) of the ResNet stem (newmodeling/backbone/resnet101_2.py:79) on the HIP kernels against
    ATen on the same channels_last tensor: the same values, and the same input gradient bit for bit (the first maximum
    of a window takes its gradient; windows are added in ascending order).  The input is a ReLU output with repeated
    values, so windows of equal values (all zero, or tied maxima) are covered; odd and even sizes, a one-row map."""
    from segk import functional as SF
    dev = "cuda"
    N, C, H, W = shape
    g0 = torch.Generator().manual_seed(H * 100 + W)
    x = (torch.randn(N, C, H, W, generator=g0).relu() * 4).round() / 4      # quarter steps: many ties, many zeros
    x = x.to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    x1 = x.clone().requires_grad_(True)
    x2 = x.clone().requires_grad_(True)
    y1 = SF.maxpool3x3s2(x1)
    assert y1 is not None
    y2 = F.max_pool2d(x2, 3, 2, 1)
    assert y1.shape == y2.shape and torch.equal(y1, y2)
    gy = torch.randn(y2.shape, generator=g0).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    y1.backward(gy)
    y2.backward(gy)
    assert torch.equal(x1.grad, x2.grad)
    # a gradient that is not channels_last is taken as well
    x3 = x.clone().requires_grad_(True)
    SF.maxpool3x3s2(x3).backward(gy.contiguous())
    assert torch.equal(x3.grad, x2.grad)
    # NaN is a maximum
    xn = x.clone()
    xn[0, 0, 0, 0] = float("nan")
    assert torch.equal(torch.isnan(SF.maxpool3x3s2(xn)), torch.isnan(F.max_pool2d(xn, 3, 2, 1)))


def test_stem_maxpool_declines_other_layouts():
    from segk import functional as SF
    x = torch.randn(2, 64, 9, 9, device="cuda")                    # NCHW-contiguous: no pixel-major view
    assert SF.maxpool3x3s2(x) is None
    x = torch.randn(2, 12, 9, 9, device="cuda").contiguous(memory_format=torch.channels_last)   # 12 channels
    assert SF.maxpool3x3s2(x) is None
