"""Class-balanced loss weights, /root/reference/utils/calculate_weights.py:6-29: label histogram over a data loader,
weight_c = 1 / ln(1.02 + frequency_c / total).  The histogram runs on the device (the confusion-matrix kernel with
a constant prediction); labels outside [0, num_classes) fall out exactly as the reference's mask drops them."""
import numpy as np
import torch

from .. import ops


def class_weights_from_counts(counts):
    """calculate_weights.py:21-26 on a vector of label counts (float64, like the reference's accumulator)."""
    z = np.asarray(counts, dtype=np.float64)
    total = np.sum(z)
    return np.array([1 / (np.log(1.02 + (f / total))) for f in z])


def calculate_weigths_labels(dataset, dataloader, num_classes, classes_weights_path=None, device=None):
    """Same name (sic) and leading arguments as the reference.  The reference always saves the result under its
    data-set directory; here that is opt-in through `classes_weights_path` (an .npy path)."""
    device = torch.device(device if device is not None else "cuda")
    cm = torch.zeros(num_classes * num_classes + 1, dtype=torch.int64, device=device)
    for sample in dataloader:
        y = sample["label"]
        y = y.detach().to(device)
        gt = y.float().contiguous() if y.dtype != torch.int64 else y.contiguous()
        pred = torch.zeros(gt.shape, dtype=torch.int64, device=device)
        ops.confmat_accumulate(gt, pred, num_classes, cm)   # column 0 of row c counts label c
    counts = cm[:num_classes * num_classes].reshape(num_classes, num_classes)[:, 0].cpu().numpy()
    ret = class_weights_from_counts(counts)
    if classes_weights_path is not None:
        np.save(classes_weights_path, ret)
    return ret
