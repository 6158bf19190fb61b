"""ASPP on HIP kernels, same module surface as /root/reference/modeling/aspp.py:7-160,297-298
(= /root/reference/newmodeling/aspp.py).  Sub-modules are ordinary nn.Conv2d / batch-norm instances used
as parameter holders (state_dict keys, init and the optimizer's 10x group keep working); the arithmetic of
forward/backward runs in segk.functional.ASPPFunction.
"""
import torch
import torch.nn as nn

from .. import functional as SF

_INPLANES = {"drn": 512, "mobilenet": 320, "ghostnet": 160}
_RATES = {16: (1, 6, 12, 18), 8: (1, 12, 24, 36)}


def _init(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class _ASPPModule(nn.Module):
    def __init__(self, inplanes, planes, kernel_size, padding, dilation, BatchNorm):
        super().__init__()
        if kernel_size not in (1, 3) or padding != dilation * (kernel_size // 2):
            raise NotImplementedError("size-preserving 1x1 / 3x3 atrous branches only")
        self.atrous_conv = nn.Conv2d(inplanes, planes, kernel_size=kernel_size, stride=1, padding=padding,
                                     dilation=dilation, bias=False)
        self.bn = BatchNorm(planes)
        self.relu = nn.ReLU()
        self._init_weight()

    def forward(self, x):
        return SF.BranchFunction.apply(x, self.atrous_conv.weight, self.bn.weight, self.bn.bias, self)

    def _init_weight(self):
        _init(self)


class ASPP(nn.Module):
    def __init__(self, backbone, output_stride, BatchNorm):
        super().__init__()
        inplanes = _INPLANES.get(backbone, 2048)
        if output_stride not in _RATES:
            raise NotImplementedError
        d = _RATES[output_stride]
        self.aspp1 = _ASPPModule(inplanes, 256, 1, padding=0, dilation=d[0], BatchNorm=BatchNorm)
        self.aspp2 = _ASPPModule(inplanes, 256, 3, padding=d[1], dilation=d[1], BatchNorm=BatchNorm)
        self.aspp3 = _ASPPModule(inplanes, 256, 3, padding=d[2], dilation=d[2], BatchNorm=BatchNorm)
        self.aspp4 = _ASPPModule(inplanes, 256, 3, padding=d[3], dilation=d[3], BatchNorm=BatchNorm)
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)),
                                             nn.Conv2d(inplanes, 256, 1, stride=1, bias=False),
                                             BatchNorm(256),
                                             nn.ReLU())
        self.conv1 = nn.Conv2d(1280, 256, 1, bias=False)
        self.bn1 = BatchNorm(256)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.5)
        self._init_weight()

    def forward(self, x):
        params = []
        for br in (self.aspp1, self.aspp2, self.aspp3, self.aspp4):
            params += [br.atrous_conv.weight, br.bn.weight, br.bn.bias]
        pool_bn = self.global_avg_pool[2]
        params += [self.global_avg_pool[1].weight, pool_bn.weight, pool_bn.bias]
        params += [self.conv1.weight, self.bn1.weight, self.bn1.bias]
        return SF.ASPPFunction.apply(self, x, *params)

    def _init_weight(self):
        _init(self)


def build_aspp(backbone, output_stride, BatchNorm):
    return ASPP(backbone, output_stride, BatchNorm)
