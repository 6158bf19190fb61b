"""Host-side contract of the drop-in modules, checked without a GPU: constructor signatures, attribute names,
state_dict keys and shapes equal the reference-shaped oracle modules (so checkpoints interchange), the
learning-rate groups pick the same parameters, and nothing computes on the host."""
import pytest
import torch
import torch.nn as nn

from oracle import damm_ref, head_ref


def _keys(m):
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def test_aspp_and_decoder_state_dicts_match():
    from segk.modeling.aspp import build_aspp
    from segk.modeling.decoder import build_decoder
    for backbone, os_ in (("resnet", 16), ("drn", 8), ("mobilenet", 16)):
        assert _keys(build_aspp(backbone, os_, nn.BatchNorm2d)) == _keys(head_ref.RefASPP(backbone, os_, nn.BatchNorm2d))
    for K in (21, 19, 81):
        assert _keys(build_decoder(K, "resnet", nn.BatchNorm2d)) == _keys(head_ref.RefDecoder(K, "resnet", nn.BatchNorm2d))


@pytest.mark.parametrize("backbone", ["resnet", "drn", "mobilenet", "ghostnet"])
def test_damm_state_dict_matches(backbone):
    from segk.modeling.damm_cam_pam import build_damm
    ours, ref = build_damm(backbone, nn.BatchNorm2d), damm_ref.RefDAMM(backbone, nn.BatchNorm2d)
    assert _keys(ours) == _keys(ref)
    assert float(ours.damm1.gamma.detach()) == 0.0 and float(ours.damm2.gamma.detach()) == 0.0  # reference init: identity branch


def test_deeplab_and_deeplab_damm_wiring():
    from segk.modeling.deeplab import DeepLab
    from segk.modeling.deeplab_damm import DeepLab as DeepLabDAMM
    from segk.modeling.sync_batchnorm import SynchronizedBatchNorm2d
    m = DeepLab("resnet", 16, 21, sync_bn=True)
    assert _keys(m) == _keys(head_ref.RefDeepLab("resnet", 16, 21, nn.BatchNorm2d))
    assert isinstance(m.aspp.bn1, SynchronizedBatchNorm2d) and isinstance(m.aspp.bn1, nn.modules.batchnorm._BatchNorm)
    assert not isinstance(m.aspp.bn1, nn.BatchNorm2d)
    g1, g10 = m.lr_groups() if hasattr(m, "lr_groups") else (list(m.get_1x_lr_params()), list(m.get_10x_lr_params()))
    ids = {id(p) for p in g1} | {id(p) for p in g10}
    assert len(ids) == len(g1) + len(g10) == len(list(m.parameters()))
    d = DeepLabDAMM("resnet", 16, 8, sync_bn=False)
    assert _keys(d) == _keys(damm_ref.RefDeepLabDAMM("resnet", 16, 8, nn.BatchNorm2d))
    # as in the reference the attention branch and damm_conv1 are in neither learning-rate group
    grouped = {id(p) for p in d.get_1x_lr_params()} | {id(p) for p in d.get_10x_lr_params()}
    outside = [n for n, p in d.named_parameters() if id(p) not in grouped]
    assert outside and all(n.startswith(("damm.", "damm_conv1.")) for n in outside)
    with pytest.raises(NotImplementedError):
        DeepLabDAMM("mobilenet", 16, 8, sync_bn=False)


def test_modules_refuse_host_tensors():
    from segk._lib import SegkError
    from segk.modeling.damm_cam_pam import _CAM_Module, _PAM_Module
    x = torch.zeros(1, 64, 3, 3)
    for m in (_PAM_Module(64), _CAM_Module(64)):
        with pytest.raises((ValueError, SegkError, RuntimeError)):
            m(x)


def test_chunk_tables_cover_every_element_once():
    """Host logic of the multi-tensor launches (optimizer, gradient packing): chunking and storage-order views."""
    import numpy as np
    import torch
    from segk import _tables
    ts = [torch.zeros(n) for n in (1, 7, _tables.CHUNK, _tables.CHUNK + 1, 3 * _tables.CHUNK - 5)]
    tidx, offs, ns = _tables.chunks_of(ts)
    assert ns.max() <= _tables.CHUNK and ns.min() >= 1
    for k, t in enumerate(ts):
        sel = tidx == k
        o, n = offs[sel].astype(np.int64), ns[sel].astype(np.int64)
        assert o[0] == 0 and np.array_equal(o[1:], (o + n)[:-1]) and (o + n)[-1] == t.numel()
    w = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    cl = w.contiguous(memory_format=torch.channels_last)
    assert torch.equal(_tables.flat_storage(w), w.reshape(-1))
    flat = _tables.flat_storage(cl)
    assert flat.data_ptr() == cl.data_ptr() and torch.equal(flat, cl.permute(0, 2, 3, 1).reshape(-1))
    import pytest
    with pytest.raises(RuntimeError):
        _tables.flat_storage(w[:, ::2])


def test_gradient_buckets_partition_parameters_in_reverse_order():
    import torch
    from segk.parallel import GradientBuckets
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (10, 300000, 5, 200000, 7)]
    frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
    b = GradientBuckets(ps + [frozen], bucket_mb=1.0)
    order = [p for bucket in b.buckets for p in bucket]
    assert [id(p) for p in order] == [id(p) for p in reversed(ps)]
    assert all(sum(p.numel() for p in bucket) <= 262144 or len(bucket) == 1 for bucket in b.buckets)
    assert all(id(p) != id(frozen) for p in order)


def test_stat_link_only_hands_rows_to_the_tensor_they_were_computed_from():
    """segk.functional.StatLink: backward statistics emitted by a data-gradient launch are used only when the gradient
    the producing unit receives is that launch's output, unchanged - the same memory, shape and strides and no
    in-place write since (a second consumer whose gradient autograd adds in place moves the version counter; one it
    adds out of place gives another tensor).  The link is spent by the first question either way."""
    from segk.functional import StatLink
    raw, coef, rows = torch.zeros(6, 8), torch.zeros(4, 8), torch.zeros(2, 2, 8, dtype=torch.float64)

    def fulfilled():
        link, out = StatLink(), torch.ones(6, 8)
        assert not link.offered()
        link.offer(raw, coef, True)
        assert link.offered()
        req = link.request()
        assert req.raw is raw and req.coef is coef and req.relu and req.bits is None and req.rows is None
        link.fulfil(rows, out)
        return link, out

    link, out = fulfilled()
    assert link.rows_for(out.view(6, 8)) is rows          # a view of the same memory is the same gradient
    assert link.rows_for(out) is None and not link.offered()   # spent
    link, out = fulfilled()
    assert link.rows_for(out + 0) is None                  # another tensor
    link, out = fulfilled()
    out.add_(1.0)                                          # written in place after the launch
    assert link.rows_for(out) is None
    link, out = fulfilled()
    assert link.rows_for(out[:3]) is None                  # another shape
    link, out = fulfilled()
    assert link.rows_for(out.t().contiguous().t()) is None  # other strides (and memory)
    link = StatLink()
    link.offer(raw, coef, True)
    assert link.rows_for(torch.ones(6, 8)) is None         # offered, never fulfilled
