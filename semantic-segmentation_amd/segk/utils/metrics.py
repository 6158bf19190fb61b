"""Evaluator with the confusion matrix accumulated on the GPU.

Same surface as /root/reference/utils/metrics.py:4-58.  `add_batch(gt, pred)` takes what the reference's
validation loop passes (numpy arrays: float label maps and int64 predictions, train.py:215-219) or torch
tensors already on the GPU; counts are accumulated bit-exactly in an int64 device histogram by an
LDS-privatised kernel and `confusion_matrix` exposes them as the float64 [K,K] array the metric methods
expect.  `add_logits(gt, logits)` additionally does the class argmax on the device so validation never
copies the [N,K,H,W] logits to the host.  Under torch.distributed call `all_reduce()` once at the end of
validation to sum the per-rank matrices.
"""
import numpy as np
import torch

from .. import ops, parallel


class Evaluator(object):
    def __init__(self, num_class, device=None):
        self.num_class = num_class
        self._device = device
        self._cm = None          # int64 [K*K+1] on the GPU, created on first use
        self._host = np.zeros((self.num_class,) * 2)   # counts assigned from the host side

    # ---- device accumulator --------------------------------------------------------------
    def _acc(self, device=None):
        if self._cm is None:
            dev = device or self._device or (torch.device("cuda", torch.cuda.current_device())
                                             if torch.cuda.is_available() else None)
            if dev is None:
                raise RuntimeError("segk Evaluator needs a GPU: the confusion matrix has no host fallback")
            self._cm = torch.zeros(self.num_class ** 2 + 1, dtype=torch.int64, device=dev)
        return self._cm

    def _to_dev(self, a, dev, kind):
        if isinstance(a, torch.Tensor):
            t = a
        else:
            a = np.asarray(a)
            if kind == "gt" and a.dtype.kind == "f":
                a = a.astype(np.float32, copy=False) if a.dtype != np.float64 else a
            t = torch.from_numpy(np.ascontiguousarray(a))
        if kind == "gt":
            if t.dtype == torch.float64:
                # keep the >=0 / <K tests and the truncation exact for doubles: decide on the host side of the copy
                t = torch.where((t >= 0) & (t < self.num_class), t.trunc(), torch.full_like(t, -1.0)).to(torch.int64)
            elif t.dtype != torch.float32:
                t = t.to(torch.int64) if not t.is_floating_point() else t.to(torch.float32)
        else:
            t = t.to(torch.int64)
        return t.to(dev, non_blocking=True).contiguous()

    @property
    def confusion_matrix(self):
        if self._cm is None:
            return self._host
        host = self._cm.cpu().numpy()
        if host[-1] != 0:
            raise ValueError("%d predictions outside [0, %d) were passed to add_batch" % (host[-1], self.num_class))
        return self._host + host[:-1].reshape(self.num_class, self.num_class).astype(np.float64)

    @confusion_matrix.setter
    def confusion_matrix(self, value):
        self._host = np.array(value, dtype=np.float64).reshape(self.num_class, self.num_class)
        if self._cm is not None:
            self._cm.zero_()

    # ---- reference API ---------------------------------------------------------------------
    def Pixel_Accuracy(self):
        cm = self.confusion_matrix
        return np.diag(cm).sum() / cm.sum()

    def Pixel_Accuracy_Class(self):
        cm = self.confusion_matrix
        return np.nanmean(np.diag(cm) / cm.sum(axis=1))

    def Mean_Intersection_over_Union(self):
        cm = self.confusion_matrix
        return np.nanmean(np.diag(cm) / (np.sum(cm, axis=1) + np.sum(cm, axis=0) - np.diag(cm)))

    def Frequency_Weighted_Intersection_over_Union(self):
        cm = self.confusion_matrix
        freq = np.sum(cm, axis=1) / np.sum(cm)
        iu = np.diag(cm) / (np.sum(cm, axis=1) + np.sum(cm, axis=0) - np.diag(cm))
        return (freq[freq > 0] * iu[freq > 0]).sum()

    def add_batch(self, gt_image, pre_image):
        assert tuple(gt_image.shape) == tuple(pre_image.shape)
        dev = gt_image.device if isinstance(gt_image, torch.Tensor) and gt_image.is_cuda else None
        if dev is None and isinstance(pre_image, torch.Tensor) and pre_image.is_cuda:
            dev = pre_image.device
        cm = self._acc(dev)
        ops.confmat_accumulate(self._to_dev(gt_image, cm.device, "gt"), self._to_dev(pre_image, cm.device, "pred"),
                               self.num_class, cm)

    def add_logits(self, gt_image, logits):
        """argmax over the class axis on the device (first maximal index, like np.argmax), then add_batch."""
        if not (isinstance(logits, torch.Tensor) and logits.is_cuda):
            raise RuntimeError("add_logits expects the network output on the GPU")
        pred = ops.argmax(logits.detach().float().contiguous())
        self.add_batch(gt_image, pred)
        return pred

    def add_low_logits(self, gt_image, low_logits):
        """add_logits(gt, F.interpolate(low_logits, gt size, bilinear, align_corners=True)) without the full-size
        logits (validation twin of DeepLab.forward_loss): upsample + argmax in one kernel, then the histogram."""
        if not (isinstance(low_logits, torch.Tensor) and low_logits.is_cuda):
            raise RuntimeError("add_low_logits expects the decoder output on the GPU")
        from .. import functional as SF
        pred = SF.upsample_argmax(low_logits.detach(), tuple(gt_image.shape[-2:]))
        self.add_batch(gt_image, pred)
        return pred

    def reset(self):
        self._host = np.zeros((self.num_class,) * 2)
        if self._cm is not None:
            self._cm.zero_()

    def all_reduce(self):
        """Sum the per-rank matrices (exact: integer counts)."""
        if parallel.is_distributed():
            parallel.all_reduce_sum_(self._acc())
