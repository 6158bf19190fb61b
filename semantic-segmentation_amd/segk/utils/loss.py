"""SegmentationLosses with the cross-entropy on a fused HIP kernel.

Same surface as /root/reference/utils/loss.py:6-53: `SegmentationLosses(weight, size_average, batch_average,
ignore_index, cuda).build_loss(mode)` returns a callable (logit [N,K,H,W], target [N,H,W] float) -> scalar.
Under torch.distributed the value and gradients are those of the loss on the gathered global batch
(/root/reference/train.py:148-153): the normaliser (sum of weights of non-ignored pixels, batch size) is
all-reduced before the division.
"""
import torch

from .. import functional as SF


class SegmentationLosses(object):
    def __init__(self, weight=None, size_average=True, batch_average=True, ignore_index=255, cuda=False):
        self.ignore_index = ignore_index
        self.weight = weight
        self.size_average = size_average
        self.batch_average = batch_average
        self.cuda = cuda

    def build_loss(self, mode="ce"):
        if mode == "ce":
            return self.CrossEntropyLoss
        elif mode == "focal":
            return self.FocalLoss
        else:
            raise NotImplementedError

    def _weight_on(self, device):
        if self.weight is None:
            return None
        return self.weight.detach().to(device=device, dtype=torch.float32).contiguous()

    def _ce(self, logit, target, batch_average):
        return SF.cross_entropy(logit, target, self._weight_on(logit.device), self.ignore_index,
                                self.size_average, batch_average)

    def CrossEntropyLoss(self, logit, target):
        return self._ce(logit, target, self.batch_average)

    def CrossEntropyLossUpsampled(self, logit_low, target):
        """CrossEntropyLoss(F.interpolate(logit_low, target size, bilinear, align_corners=True), target) without the
        full-size logits (the opt-in fused form; see DeepLab.forward_loss)."""
        return SF.upsample_cross_entropy(logit_low, target, self._weight_on(logit_low.device), self.ignore_index,
                                         self.size_average, self.batch_average)

    def FocalLoss(self, logit, target, gamma=2, alpha=0.5):
        # utils/loss.py:37-53: a scalar transform of the mean cross-entropy
        n = logit.size(0)
        logpt = -self._ce(logit, target, False)
        pt = torch.exp(logpt)
        if alpha is not None:
            logpt = logpt * alpha
        loss = -((1 - pt) ** gamma) * logpt
        if self.batch_average:
            loss = loss / n
        return loss
