"""Restatement of the non-convolution arithmetic on the path (test infrastructure).

  sync-BN statistics   /root/reference/modeling/sync_batchnorm/batchnorm.py:55-78,113-125
  cross-entropy        /root/reference/utils/loss.py:23-35
  confusion matrix     /root/reference/utils/metrics.py:12-58, caller argmax /root/reference/train.py:215-219
  poly LR              /root/reference/utils/lr_scheduler.py:45-73
"""
import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- SyncBN ----
def syncbn_mean_invstd_clamp(sum_, ssum, size, eps):
    """mean and clamp(var, eps)^-1/2 from global sum / square-sum (batchnorm.py:113-125)."""
    assert size > 1
    mean = sum_ / size
    sumvar = ssum - sum_ * mean
    return mean, (sumvar / size).clamp(eps) ** -0.5, sumvar / (size - 1)


def syncbn_running_update(running_mean, running_var, mean, unbias_var, momentum):
    """batchnorm.py:122-123."""
    return ((1 - momentum) * running_mean + momentum * mean,
            (1 - momentum) * running_var + momentum * unbias_var)


def syncbn_forward_replicated(shards, weight, bias, eps, legacy_clamp=True):
    """What N replicas compute together in replicated-training mode (batchnorm.py:55-78):
    per-shard sum/ssum -> global stats -> every shard normalised with the same mean/inv_std.
    `shards` is a list of [n_i, C, *] tensors.  Returns (outputs, mean, inv_std, unbiased_var)."""
    C = shards[0].shape[1]
    tot = 0
    s = torch.zeros(C, dtype=shards[0].dtype)
    q = torch.zeros(C, dtype=shards[0].dtype)
    for x in shards:
        v = x.reshape(x.shape[0], C, -1)
        tot += v.shape[0] * v.shape[2]
        s = s + v.sum(dim=0).sum(dim=-1)
        q = q + (v ** 2).sum(dim=0).sum(dim=-1)
    mean, inv_std, unb = syncbn_mean_invstd_clamp(s, q, tot, eps)
    if not legacy_clamp:
        inv_std = ((q - s * mean) / tot + eps) ** -0.5
    outs = []
    for x in shards:
        v = x.reshape(x.shape[0], C, -1)
        o = (v - mean[None, :, None]) * (inv_std * weight)[None, :, None] + bias[None, :, None]
        outs.append(o.reshape(x.shape))
    return outs, mean, inv_std, unb


# ------------------------------------------------------- cross-entropy ----
def ce_loss(logit, target, weight=None, ignore_index=255, size_average=True, batch_average=True):
    """loss.py:23-35: mean cross-entropy over non-ignored pixels, then divided by the batch size."""
    n = logit.shape[0]
    red = "mean" if size_average else "sum"
    loss = F.cross_entropy(logit, target.long(), weight=weight, ignore_index=ignore_index, reduction=red)
    return loss / n if batch_average else loss


def focal_loss(logit, target, weight=None, ignore_index=255, size_average=True, batch_average=True,
               gamma=2, alpha=0.5):
    """loss.py:37-53."""
    n = logit.shape[0]
    red = "mean" if size_average else "sum"
    logpt = -F.cross_entropy(logit, target.long(), weight=weight, ignore_index=ignore_index, reduction=red)
    pt = torch.exp(logpt)
    if alpha is not None:
        logpt = logpt * alpha
    loss = -((1 - pt) ** gamma) * logpt
    return loss / n if batch_average else loss


# ---------------------------------------------------- confusion matrix ----
def confusion_matrix(gt, pred, num_class):
    """metrics.py:44-49 on numpy arrays (gt float or int, pred int)."""
    keep = (gt >= 0) & (gt < num_class)
    lab = num_class * gt[keep].astype("int") + pred[keep]
    return np.bincount(lab, minlength=num_class ** 2).reshape(num_class, num_class)


class RefEvaluator:
    """metrics.py:4-58."""

    def __init__(self, num_class):
        self.num_class = num_class
        self.confusion_matrix = np.zeros((num_class, num_class))

    def add_batch(self, gt_image, pre_image):
        assert gt_image.shape == pre_image.shape
        self.confusion_matrix += confusion_matrix(gt_image, pre_image, self.num_class)

    def reset(self):
        self.confusion_matrix = np.zeros((self.num_class, self.num_class))

    def Pixel_Accuracy(self):
        cm = self.confusion_matrix
        return np.diag(cm).sum() / cm.sum()

    def Pixel_Accuracy_Class(self):
        cm = self.confusion_matrix
        return np.nanmean(np.diag(cm) / cm.sum(axis=1))

    def _iou(self):
        cm = self.confusion_matrix
        return np.diag(cm) / (cm.sum(axis=1) + cm.sum(axis=0) - np.diag(cm))

    def Mean_Intersection_over_Union(self):
        return np.nanmean(self._iou())

    def Frequency_Weighted_Intersection_over_Union(self):
        cm = self.confusion_matrix
        freq = cm.sum(axis=1) / cm.sum()
        iu = self._iou()
        return (freq[freq > 0] * iu[freq > 0]).sum()


def predict(logits):
    """train.py:215-217: host argmax over the class axis (first maximal index wins)."""
    return np.argmax(logits, axis=1)


# -------------------------------------------------------------- poly LR ----
def poly_lr(base_lr, it, total_iters, power=0.9):
    """lr_scheduler.py:50."""
    return base_lr * pow((1 - 1.0 * it / total_iters), power)


# ------------------------------------------------------------------- SGD ----
def sgd_step(w, g, buf, lr, momentum=0.9, weight_decay=5e-4):
    """One update of torch.optim.SGD as train.py:56-61 configures it (dampening 0, nesterov off), fp32 numpy:
    d = g + wd*w; buf = momentum*buf + d (a zero buffer gives torch's first step, buf = d); w = w - lr*buf.
    Returns (w, buf)."""
    w, g, buf = (np.asarray(a, dtype=np.float32) for a in (w, g, buf))
    d = g + np.float32(weight_decay) * w
    buf = np.float32(momentum) * buf + d
    return w - np.float32(lr) * buf, buf


# ------------------------------------------------- class-balanced weights ----
def class_weights(label_batches, num_classes):
    """utils/calculate_weights.py:8-27 without the file output: bincount of the labels inside [0, num_classes)
    (cast through uint8 as the reference does), weight = 1 / ln(1.02 + frequency / total)."""
    z = np.zeros((num_classes,))
    for y in label_batches:
        y = np.asarray(y)
        mask = (y >= 0) & (y < num_classes)
        z += np.bincount(y[mask].astype(np.uint8), minlength=num_classes)
    total = np.sum(z)
    return np.array([1 / (np.log(1.02 + (f / total))) for f in z])
