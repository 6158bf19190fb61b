"""DeepLab container: wiring and learning-rate groups as /root/reference/newmodeling/deeplab.py:9-74
(= modeling/deeplab_origin.py), with the head on HIP kernels and the feeder as plain PyTorch."""
import torch.nn as nn

from .. import functional as SF
from .aspp import build_aspp
from .backbone import build_backbone
from .decoder import build_decoder
from .sync_batchnorm.batchnorm import SynchronizedBatchNorm2d


class DeepLab(nn.Module):
    def __init__(self, backbone="resnet", output_stride=16, num_classes=8, sync_bn=True, freeze_bn=False):
        super().__init__()
        if backbone == "drn":
            output_stride = 8
        BatchNorm = SynchronizedBatchNorm2d if sync_bn == True else nn.BatchNorm2d  # noqa: E712
        self.backbone = build_backbone(backbone, output_stride, BatchNorm)
        self.aspp = build_aspp(backbone, output_stride, BatchNorm)
        self.decoder = build_decoder(num_classes, backbone, BatchNorm)
        # as in the reference, this attribute shadows the method of the same name, so BN is never
        # actually frozen; the flag only selects which parameters the two LR groups yield
        self.freeze_bn = freeze_bn

    def forward(self, input):
        x, low_level_feat = self.backbone(input)
        x = self.aspp(x)
        x = self.decoder(x, low_level_feat)
        return SF.interpolate_bilinear_ac(x, input.size()[2:])

    def forward_loss(self, input, target, weight=None, ignore_index=255, size_average=True, batch_average=True):
        """criterion(self(input), target) for the cross-entropy criterion of SegmentationLosses
        (/root/reference/utils/loss.py:23-35), with the final upsample fused into the loss: the
        [N, K, H, W] logits are never materialised.  Opt-in; `forward` still returns full-size logits."""
        x, low_level_feat = self.backbone(input)
        x = self.aspp(x)
        x = self.decoder(x, low_level_feat)
        if tuple(target.shape[-2:]) != tuple(input.shape[-2:]):
            raise ValueError("target %s does not match the input size %s" % (tuple(target.shape), tuple(input.shape)))
        return SF.upsample_cross_entropy(x, target, weight, ignore_index, size_average, batch_average)

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
        return self._group([self.aspp, self.decoder])
