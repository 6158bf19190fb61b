"""Checkpoints interchange with the reference's format (/root/reference/utils/saver.py:33-60 writes
{'epoch', 'state_dict', 'optimizer', 'best_pred'} with torch.save; /root/reference/train.py:112-131 resumes from it),
and the import-compatible stand-ins for patch_replication_callback / DataParallelWithCallback behave as documented."""
import io

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import cases, gen, head_ref


def _round_trip(obj):
    buf = io.BytesIO()
    torch.save(obj, buf)
    buf.seek(0)
    return torch.load(buf, weights_only=True)   # nothing but tensors and plain containers in the file


def _same_layout(a, b):
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert tuple(a[k].shape) == tuple(b[k].shape), k
        assert a[k].dtype == b[k].dtype, k


def test_state_dict_interchanges_with_reference_shaped_model_cpu():
    """Keys, order, shapes and dtypes of the drop-in DeepLab equal those of the reference-shaped model; a checkpoint
    in the reference's format written from one loads (strict) into the other and back, values unchanged."""
    from segk.modeling.deeplab import DeepLab
    mine = cases._load(DeepLab(backbone="resnet", output_stride=16, num_classes=21, sync_bn=True), 3, "ckpt", "cpu")
    ref = head_ref.RefDeepLab("resnet", 16, 21, nn.BatchNorm2d)
    _same_layout(mine.state_dict(), ref.state_dict())
    g1, g10 = list(mine.get_1x_lr_params()), list(mine.get_10x_lr_params())
    r1, r10 = ref.lr_groups()
    assert [tuple(p.shape) for p in g1] == [tuple(p.shape) for p in r1]
    assert [tuple(p.shape) for p in g10] == [tuple(p.shape) for p in r10]
    opt = torch.optim.SGD([{"params": g1, "lr": 0.007}, {"params": g10, "lr": 0.07}], momentum=0.9, weight_decay=5e-4)
    ck = _round_trip({"epoch": 3, "state_dict": mine.state_dict(), "optimizer": opt.state_dict(), "best_pred": 0.25})
    assert set(ck) == {"epoch", "state_dict", "optimizer", "best_pred"}
    ref.load_state_dict(ck["state_dict"], strict=True)
    ropt = torch.optim.SGD([{"params": r1, "lr": 0.1}, {"params": r10, "lr": 0.1}], momentum=0.9, weight_decay=5e-4)
    ropt.load_state_dict(ck["optimizer"])
    assert ropt.param_groups[1]["lr"] == 0.07
    back = DeepLab(backbone="resnet", output_stride=16, num_classes=21, sync_bn=False)
    back.load_state_dict(_round_trip({"state_dict": ref.state_dict()})["state_dict"], strict=True)
    for (k, a), b in zip(mine.state_dict().items(), back.state_dict().values()):
        assert torch.equal(a, b), k


def test_replication_shims():
    from segk.modeling.sync_batchnorm import DataParallelWithCallback, patch_replication_callback
    from segk.modeling.sync_batchnorm.replicate import execute_replication_callbacks
    m = nn.Conv2d(3, 3, 1)
    dp = nn.DataParallel(m, device_ids=[0])
    patch_replication_callback(dp)                      # one device: accepted, nothing to patch
    assert isinstance(DataParallelWithCallback(m, device_ids=[0]), nn.DataParallel)
    dp.device_ids = [0, 1]                              # what a 2-GPU wrapper carries
    with pytest.raises(RuntimeError, match="one process per GPU"):
        patch_replication_callback(dp)
    with pytest.raises(AssertionError):
        patch_replication_callback(m)                   # the reference asserts the wrapper type as well
    execute_replication_callbacks([m])
    with pytest.raises(RuntimeError):
        execute_replication_callbacks([m, m])


@pytest.mark.gpu
def test_saver_round_trip_bf16_feeder_resume():
    """The bench configuration (bf16 feeder weights with fp32 masters in segk.optim.SGD): the reference-format
    checkpoint holds fp32 tensors only, loads into the reference-shaped model, and a fresh model + optimizer
    resumed from it holds bit-identical state and continues like the run that was not interrupted."""
    from segk import config
    from segk.modeling.deeplab import DeepLab
    from segk.optim import SGD
    from segk.utils.loss import SegmentationLosses
    dev = "cuda"
    n, size, K = 3, 65, 21
    image = torch.from_numpy(gen.normal(5, "ckpt/img", (n, 3, size, size)) * cases._gain(n)).to(dev)
    target = torch.from_numpy(gen.labels(5, "ckpt/lab", (n, size, size), K, border=2)).to(dev)
    crit = SegmentationLosses(cuda=True).build_loss("ce")

    def build():
        model = cases._load(DeepLab(backbone="resnet", output_stride=16, num_classes=K, sync_bn=True), 5, "ckpt", dev)
        cases._zero_dropout(model)
        model.train()
        model.backbone = model.backbone.to(memory_format=torch.channels_last)
        fwd = model.backbone.forward

        def amp_forward(x):
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                return fwd(x.contiguous(memory_format=torch.channels_last))
        model.backbone.forward = amp_forward
        groups = [{"params": list(model.get_1x_lr_params()), "lr": 0.007},
                  {"params": list(model.get_10x_lr_params()), "lr": 0.07}]
        low = [m.weight for m in model.backbone.modules() if isinstance(m, nn.Conv2d)]
        return model, SGD(groups, momentum=0.9, weight_decay=5e-4, low_precision=low)

    def step(model, opt):
        opt.zero_grad(set_to_none=True)
        loss = crit(model(image), target)
        loss.backward()
        opt.step()
        return float(loss)

    with config.compute_dtype_as(torch.bfloat16):
        model, opt = build()
        step(model, opt)
        ck = _round_trip({"epoch": 1, "state_dict": opt.full_precision_state_dict(model),
                          "optimizer": opt.state_dict(), "best_pred": 0.0})
        assert all(v.dtype in (torch.float32, torch.int64) for v in ck["state_dict"].values())
        ref = head_ref.RefDeepLab("resnet", 16, K, nn.BatchNorm2d)
        _same_layout(ck["state_dict"], ref.state_dict())
        ref.load_state_dict(ck["state_dict"], strict=True)
        # torch.optim.SGD on the reference-shaped model takes the optimizer part (the 'master' entries are ignored)
        r1, r10 = ref.lr_groups()
        ropt = torch.optim.SGD([{"params": r1}, {"params": r10}], lr=0.1, momentum=0.9)
        ropt.load_state_dict(ck["optimizer"])
        mom = ropt.state[r10[0]]["momentum_buffer"]
        assert mom.dtype == torch.float32 and tuple(mom.shape) == tuple(r10[0].shape) and float(mom.abs().sum()) > 0

        model2, opt2 = build()
        for p in model2.parameters():       # resume must not depend on what the fresh objects held
            p.data.mul_(0.5)
        opt2.load_full_precision_state_dict(model2, ck["state_dict"])
        opt2.load_state_dict(ck["optimizer"])
        a, b = opt.full_precision_state_dict(model), opt2.full_precision_state_dict(model2)
        for k in a:
            assert torch.equal(a[k], b[k]), k
        for x, y in zip(opt.state_dict()["state"].values(), opt2.state_dict()["state"].values()):
            assert torch.equal(x["momentum_buffer"], y["momentum_buffer"])
        for p, q in zip(model.parameters(), model2.parameters()):
            assert p.dtype == q.dtype and torch.equal(p.detach(), q.detach())
        # The next step of both runs.  MIOpen's bf16 feeder convolutions are not reproducible from run to run (the
        # SAME model evaluated twice on this input gives losses 3e-3 apart at this 65 x 65 size, where the deepest
        # batch norms see 75 values per channel), so beyond the bit-identical state above only the loss is compared.
        l1, l2 = step(model, opt), step(model2, opt2)
        assert np.isfinite(l1) and abs(l1 - l2) <= 2e-2 * abs(l1)
