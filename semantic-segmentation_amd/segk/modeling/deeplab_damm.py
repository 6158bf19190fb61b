"""DeepLab with the dual-attention branch beside ASPP: wiring and learning-rate groups as
/root/reference/modeling/deeplab_damm.py:10-90, head on HIP kernels, feeder plain PyTorch."""
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as SF
from .aspp import build_aspp
from .backbone import build_backbone
from .damm_cam_pam import build_damm
from .decoder import build_decoder
from .sync_batchnorm.batchnorm import SynchronizedBatchNorm2d


class DeepLab(nn.Module):
    def __init__(self, backbone="resnet", output_stride=16, num_classes=8, sync_bn=True, freeze_bn=False):
        super().__init__()
        if backbone == "drn":
            output_stride = 8
        if backbone != "resnet":
            # the reference only defines the attention width for this feeder (deeplab_damm.py:16-19)
            raise NotImplementedError("the dual-attention DeepLab is defined for the 'resnet' feeder")
        damm_ch = 2048
        BatchNorm = SynchronizedBatchNorm2d if sync_bn == True else nn.BatchNorm2d  # noqa: E712
        self.backbone = build_backbone(backbone, output_stride, BatchNorm)
        self.aspp = build_aspp(backbone, output_stride, BatchNorm)
        self.decoder = build_decoder(num_classes, backbone, BatchNorm)
        # 3x3, dilation 2 but padding 1: the map shrinks by one pixel on every side (33 -> 31)
        self.damm_conv1 = nn.Sequential(nn.Conv2d(damm_ch, damm_ch, kernel_size=3, stride=1, padding=1, dilation=2,
                                                  bias=False),
                                        BatchNorm(damm_ch),
                                        nn.ReLU())
        self.damm = build_damm(backbone, BatchNorm)
        self.freeze_bn = freeze_bn

    def _damm_conv1(self, x):
        conv, bn = self.damm_conv1[0], self.damm_conv1[1]
        d = conv.dilation[0]
        crop = d * (conv.kernel_size[0] // 2) - conv.padding[0]
        if crop < 0 or conv.stride[0] != 1:
            raise NotImplementedError("damm_conv1 must not pad beyond its dilation")
        # a convolution padded by less than its dilation equals the size-preserving one with its border cut off
        y = SF.conv2d_same(x, conv.weight, conv.bias, d)
        if crop:
            y = y[:, :, crop:y.shape[2] - crop, crop:y.shape[3] - crop]
        return F.relu(bn(y))

    def forward(self, input):
        x, low_level_feat = self.backbone(input)
        x = self._damm_conv1(x)
        x1 = self.aspp(x)
        x2 = self.damm(x)
        x = x1 + x2
        x = self.decoder(x, low_level_feat)
        return SF.interpolate_bilinear_ac(x, input.size()[2:])

    def _group(self, roots):
        for root in roots:
            for _, m in root.named_modules():
                if isinstance(m, nn.Conv2d) or (not self.freeze_bn and isinstance(m, nn.modules.batchnorm._BatchNorm)):
                    for p in m.parameters():
                        if p.requires_grad:
                            yield p

    def get_1x_lr_params(self):
        return self._group([self.backbone])

    def get_10x_lr_params(self):
        # as in the reference (deeplab_damm.py:76-77) the attention branch and damm_conv1 are in neither group
        return self._group([self.aspp, self.decoder])
