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
