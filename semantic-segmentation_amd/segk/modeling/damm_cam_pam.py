"""Dual attention (position + channel) on HIP kernels, same module surface and state_dict keys as
/root/reference/modeling/damm_cam_pam.py:7-139.  The 1x1 projections and the batch norm are ordinary
nn.Conv2d / nn.BatchNorm2d instances used as parameter holders; the arithmetic runs in
segk.functional.PAMFunction / CAMFunction / BranchFunction."""
import types

import torch
import torch.nn as nn

from .. import functional as SF
from .aspp import _init

_INPLANES = {"drn": 512, "mobilenet": 320, "ghostnet": 160}


class _PAM_Module(nn.Module):
    """Position attention: every pixel attends to every pixel of its image."""

    def __init__(self, in_dim):
        super().__init__()
        self.channel_in = in_dim
        self.query_conv = nn.Conv2d(in_dim, in_dim // 8, kernel_size=1)
        self.key_conv = nn.Conv2d(in_dim, in_dim // 8, kernel_size=1)
        self.value_conv = nn.Conv2d(in_dim, in_dim, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        return SF.PAMFunction.apply(x, self.query_conv.weight, self.query_conv.bias, self.key_conv.weight,
                                    self.key_conv.bias, self.value_conv.weight, self.value_conv.bias, self.gamma)

    def _init_weight(self):
        _init(self)


class _CAM_Module(nn.Module):
    """Channel attention: every channel attends to every channel of its image."""

    def __init__(self, in_dim):
        super().__init__()
        self.channel_in = in_dim
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        return SF.CAMFunction.apply(x, self.gamma)

    def _init_weight(self):
        _init(self)


class DAMM(nn.Module):
    def __init__(self, backbone, BatchNorm):
        super().__init__()
        inplanes = _INPLANES.get(backbone, 2048)
        self.damm1 = _PAM_Module(inplanes)
        self.damm2 = _CAM_Module(inplanes)
        # plain nn.BatchNorm2d whatever `BatchNorm` is, as in the reference (damm_cam_pam.py:124-128)
        self.conv1 = nn.Sequential(nn.Conv2d(inplanes, 256, kernel_size=1, stride=1, bias=False),
                                   nn.BatchNorm2d(256),
                                   nn.ReLU())

    def forward(self, x):
        x1 = self.damm1(x)
        x2 = self.damm2(x)
        x = x1 + x2
        unit = types.SimpleNamespace(atrous_conv=self.conv1[0], bn=self.conv1[1])
        return SF.BranchFunction.apply(x, self.conv1[0].weight, self.conv1[1].weight, self.conv1[1].bias, unit)


def build_damm(backbone, BatchNorm):
    return DAMM(backbone, BatchNorm)
