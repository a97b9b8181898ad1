"""Drop-in for the reference's `src.loss` import surface (/root/reference/foundation_model/src/loss.py): the same
class names, constructor arguments and `get_loss_function` dispatch, computed by the gfx950 loss kernels."""
from typing import Optional

import torch
import torch.nn as nn

from mia_amd import seg_ops  # `mia_amd` is registered by src/__init__.py
from mia_amd.head_ops import CrossEntropyLoss as _DeviceCrossEntropy


class DiceLoss(nn.Module):
    """loss.py:49-80: 1 - mean over (image, class) of (2I + s) / (P + T + s) on softmax probabilities."""

    def __init__(self, num_classes: int = 2, smooth: float = 1e-5):
        super().__init__()
        self.num_classes, self.smooth = num_classes, smooth

    def forward(self, predictions, targets):
        _check_classes(predictions, self.num_classes)
        return seg_ops.seg_loss(predictions, targets, iou=False, smooth=self.smooth, ce_weight=0.0, overlap_weight=1.0)


class IoULoss(nn.Module):
    """loss.py:105-136: 1 - mean over (image, class) of (I + s) / (P + T - I + s)."""

    def __init__(self, num_classes: int = 2, smooth: float = 1e-5):
        super().__init__()
        self.num_classes, self.smooth = num_classes, smooth

    def forward(self, predictions, targets):
        _check_classes(predictions, self.num_classes)
        return seg_ops.seg_loss(predictions, targets, iou=True, smooth=self.smooth, ce_weight=0.0, overlap_weight=1.0)


class CombinedSegmentationLoss(nn.Module):
    """loss.py:27-46: ce_weight * CrossEntropy(weight=weights) + dice_weight * Dice, one fused pass."""

    def __init__(self, num_classes: int = 2, weights: Optional[torch.Tensor] = None, dice_weight: float = 0.5,
                 ce_weight: float = 0.5):
        super().__init__()
        self.num_classes, self.dice_weight, self.ce_weight = num_classes, dice_weight, ce_weight
        self.weights = weights
        self.smooth = 1e-5

    def forward(self, predictions, targets):
        _check_classes(predictions, self.num_classes)
        w = self.weights.to(predictions.device) if self.weights is not None else None
        return seg_ops.seg_loss(predictions, targets, class_weights=w, iou=False, smooth=self.smooth,
                                ce_weight=self.ce_weight, overlap_weight=self.dice_weight)


class FocalLoss(nn.Module):
    """loss.py:83-102: alpha * (1 - pt)^gamma * ce on class logits [B,n] or pixel logits [B,n,H,W]."""

    def __init__(self, alpha: float = 1.0, gamma: float = 2.0, reduction: str = "mean"):
        super().__init__()
        self.alpha, self.gamma, self.reduction = alpha, gamma, reduction

    def forward(self, predictions, targets):
        # the reference treats every reduction other than 'mean' / 'sum' as "no reduction"
        red = self.reduction if self.reduction in ("mean", "sum") else "none"
        return seg_ops.general_ce(predictions, targets, focal=True, alpha=self.alpha, gamma=self.gamma, reduction=red)


class WeightedCrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss(weight=class_weights) (loss.py:147): sum w[t]*ce / sum w[t]."""

    def __init__(self, weight: torch.Tensor):
        super().__init__()
        self.register_buffer("weight", weight.detach().float().clone())

    def forward(self, predictions, targets):
        return seg_ops.general_ce(predictions, targets, class_weights=self.weight)


def _classification_ce(weights):
    return _DeviceCrossEntropy() if weights is None else WeightedCrossEntropyLoss(weights)


class CombinedLoss(nn.Module):
    """loss.py:8-24."""

    def __init__(self, task: str, num_classes: int = None, weights: Optional[torch.Tensor] = None):
        super().__init__()
        self.task, self.num_classes = task, num_classes
        if task == "classification":
            self.criterion = _classification_ce(weights)
        elif task == "segmentation":
            self.criterion = CombinedSegmentationLoss(num_classes=num_classes, weights=weights)
        else:
            raise ValueError(f"Unknown task: {task}")

    def forward(self, predictions, targets):
        return self.criterion(predictions, targets)


def _check_classes(predictions, num_classes):
    if predictions.dim() != 4 or predictions.shape[1] != num_classes:
        raise ValueError(f"expected logits [B, {num_classes}, H, W], got {tuple(predictions.shape)}")


def get_loss_function(task: str, num_classes: int = None, loss_type: str = "default",
                      class_weights: Optional[torch.Tensor] = None):
    """loss.py:139-160."""
    if task == "classification":
        if loss_type == "focal":
            return FocalLoss()
        return _classification_ce(class_weights)
    if task == "segmentation":
        if loss_type == "focal":
            return FocalLoss()
        if loss_type == "dice":
            return DiceLoss(num_classes=num_classes)
        if loss_type == "iou":
            return IoULoss(num_classes=num_classes)
        return CombinedSegmentationLoss(num_classes=num_classes, weights=class_weights)
    raise ValueError(f"Unknown task: {task}")


def calculate_class_weights(dataset, num_classes: int, task: str = "classification"):
    """loss.py:163-196: inverse-frequency class weights, total / (num_classes * count[c]) as a float32 CPU tensor
    (a class that never occurs gets inf, as in the reference). `dataset` yields dicts with 'label' (classification)
    or 'mask' (segmentation; every pixel counts towards the total, values outside [0, num_classes) towards no class).
    Host code: it walks a Python dataset once before training starts."""
    if task not in ("classification", "segmentation"):
        raise ValueError(f"Unknown task: {task}")
    counts = torch.zeros(num_classes, dtype=torch.int64)
    total = 0
    for sample in dataset:
        if task == "classification":
            counts[int(sample["label"])] += 1
            total += 1
        else:
            mask = torch.as_tensor(sample["mask"]).reshape(-1).long().cpu()
            total += mask.numel()
            inside = mask[(mask >= 0) & (mask < num_classes)]
            counts += torch.bincount(inside, minlength=num_classes)
    if task == "classification":
        total = len(dataset)
    # Python int over a float32 tensor, the reference's expression: torch forms reciprocal(x) * total
    return int(total) / (num_classes * counts.to(torch.float32))
