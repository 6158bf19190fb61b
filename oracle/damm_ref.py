"""Plain-PyTorch restatement of the dual-attention branch (test infrastructure; never imported by the
product path).

  RefPAM      /root/reference/modeling/damm_cam_pam.py:7-48   q/k/v 1x1 projections (biased), energy = q^T k
              per image (:38), softmax over the last index (:39), out = v att^T (:43), gamma*out + x (:46)
  RefCAM      /root/reference/modeling/damm_cam_pam.py:63-94  energy = x x^T over pixels (:83), softmax of
              (row max - energy) (:84-87), out = att x (:90), gamma*out + x (:93)
  RefDAMM     /root/reference/modeling/damm_cam_pam.py:109-136 PAM(x) + CAM(x) -> 1x1 -> nn.BatchNorm2d -> ReLU
  RefDeepLabDAMM  /root/reference/modeling/deeplab_damm.py:10-51  3x3 dilation-2 padding-1 conv + BN + ReLU in front of
              ASPP and the attention branch (the map shrinks by 2), sum of both, decoder, upsample
Attribute names equal the reference's so its state_dict loads unchanged.  gamma is initialised to 0 as in the
reference; tests set it to a non-zero value.
"""
import torch
import torch.nn as nn

from . import head_ref
from .head_ref import _relu, _up

_INPLANES = {"drn": 512, "mobilenet": 320, "ghostnet": 160}


class RefPAM(nn.Module):
    def __init__(self, in_dim):
        super().__init__()
        self.channel_in = in_dim
        self.query_conv = nn.Conv2d(in_dim, in_dim // 8, 1)
        self.key_conv = nn.Conv2d(in_dim, in_dim // 8, 1)
        self.value_conv = nn.Conv2d(in_dim, in_dim, 1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        n, c, h, w = x.shape
        q = self.query_conv(x).view(n, -1, h * w).permute(0, 2, 1)
        k = self.key_conv(x).view(n, -1, h * w)
        att = torch.softmax(torch.bmm(q, k), dim=-1)
        v = self.value_conv(x).view(n, -1, h * w)
        out = torch.bmm(v, att.permute(0, 2, 1)).view(n, c, h, w)
        return self.gamma * out + x


class RefCAM(nn.Module):
    def __init__(self, in_dim):
        super().__init__()
        self.channel_in = in_dim
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        n, c, h, w = x.shape
        f = x.view(n, c, -1)
        energy = torch.bmm(f, f.permute(0, 2, 1))
        energy_new = torch.max(energy, -1, keepdim=True)[0].expand_as(energy) - energy
        att = torch.softmax(energy_new, dim=-1)
        out = torch.bmm(att, f).view(n, c, h, w)
        return self.gamma * out + x


class RefDAMM(nn.Module):
    def __init__(self, backbone, BatchNorm, inplanes=None, planes=256):
        super().__init__()
        if inplanes is None:
            inplanes = _INPLANES.get(backbone, 2048)
        self.damm1 = RefPAM(inplanes)
        self.damm2 = RefCAM(inplanes)
        self.conv1 = nn.Sequential(nn.Conv2d(inplanes, planes, 1, bias=False), nn.BatchNorm2d(planes), head_ref._ReLU())

    def forward(self, x):
        return self.conv1(self.damm1(x) + self.damm2(x))


class RefDeepLabDAMM(nn.Module):
    def __init__(self, backbone="resnet", output_stride=16, num_classes=8, BatchNorm=nn.BatchNorm2d, feeder="resnet101"):
        super().__init__()
        fm = head_ref._load_feeder_module()
        ctor = fm.ResNet101 if feeder == "resnet101" else fm.ResNet50
        damm_ch = 2048
        self.backbone = ctor(output_stride, BatchNorm, pretrained=False)
        self.aspp = head_ref.RefASPP(backbone, output_stride, BatchNorm)
        self.decoder = head_ref.RefDecoder(num_classes, backbone, BatchNorm)
        self.damm_conv1 = nn.Sequential(nn.Conv2d(damm_ch, damm_ch, 3, 1, padding=1, dilation=2, bias=False),
                                        BatchNorm(damm_ch), nn.ReLU())
        self.damm = RefDAMM(backbone, BatchNorm)

    def forward(self, input):
        x, low = self.backbone(input)
        x = self.damm_conv1(x)
        x = self.aspp(x) + self.damm(x)
        return _up(self.decoder(x, low), input.shape[2:])
