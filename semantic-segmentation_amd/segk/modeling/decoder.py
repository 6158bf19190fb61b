"""DeepLabv3+ decoder on HIP kernels, same module surface as /root/reference/modeling/decoder.py:86-257
(= /root/reference/newmodeling/decoder.py): conv1/bn1 on the low-level features, `last_conv` Sequential
with the reference's indices (0,1 | 4,5 | 8), classifier with bias.
"""
import torch
import torch.nn as nn

from .. import functional as SF

_LOW_LEVEL = {"resnet": 256, "drn": 256, "xception": 128, "mobilenet": 24, "ghostnet": 24}


class Decoder(nn.Module):
    def __init__(self, num_classes, backbone, BatchNorm, bilinear=True):
        super().__init__()
        if backbone not in _LOW_LEVEL:
            raise NotImplementedError
        low_level_inplanes = _LOW_LEVEL[backbone]
        self.conv1 = nn.Conv2d(low_level_inplanes, 48, 1, bias=False)
        self.bn1 = BatchNorm(48)
        self.relu = nn.ReLU()
        self.last_conv = nn.Sequential(nn.Conv2d(304, 256, kernel_size=3, stride=1, padding=1, bias=False),
                                       BatchNorm(256),
                                       nn.ReLU(),
                                       nn.Dropout(0.5),
                                       nn.Conv2d(256, 256, kernel_size=3, stride=1, padding=1, bias=False),
                                       BatchNorm(256),
                                       nn.ReLU(),
                                       nn.Dropout(0.1),
                                       nn.Conv2d(256, num_classes, kernel_size=1, stride=1))
        self._init_weight()

    def forward(self, x, low_level_feat):
        lc = self.last_conv
        params = [self.conv1.weight, self.bn1.weight, self.bn1.bias,
                  lc[0].weight, lc[1].weight, lc[1].bias,
                  lc[4].weight, lc[5].weight, lc[5].bias,
                  lc[8].weight, lc[8].bias]
        return SF.DecoderFunction.apply(self, x, low_level_feat, *params)

    def _init_weight(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                torch.nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.modules.batchnorm._BatchNorm):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


def build_decoder(num_classes, backbone, BatchNorm):
    return Decoder(num_classes, backbone, BatchNorm)
