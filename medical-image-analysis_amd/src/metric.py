"""Drop-in for the reference's `src.metric` surface used by the training loops
(/root/reference/foundation_model/src/metric.py: MetricsCalculator, create_metrics_calculator, AverageMeter,
EarlyStopping). Predictions never leave the device: every update() is one kernel that takes the arg-max and adds
to an n x n confusion matrix in HBM; compute_metrics() copies that matrix (n*n integers) once and derives the same
numbers scikit-learn derives from the full prediction lists.

Not reproduced: the Hausdorff / surface-distance extras of metric.py:116-136 (they need the per-pixel masks on the
host and only trigger when an epoch's pixels happen to form one square image) and plot_confusion_matrix (plotting).
"""
from typing import Dict, List, Optional

import numpy as np
import torch

from mia_amd._lib import lib  # `mia_amd` is registered by src/__init__.py
from mia_amd.ops import _p, _stream


class MetricsCalculator:
    """metric.py:12-160 with device-side accumulation."""

    def __init__(self, task: str, num_classes: int, class_names: Optional[List[str]] = None):
        if task not in ("classification", "segmentation"):
            raise ValueError(f"Unknown task: {task}")
        self.task = task
        self.num_classes = num_classes
        self.class_names = class_names or [f"Class_{i}" for i in range(num_classes)]
        self._conf = None
        self._loss_dev = None
        self.reset()

    def reset(self):
        self.running_loss = 0.0
        self.num_samples = 0
        self._seen = False
        if self._conf is not None:
            self._conf.zero_()
            self._loss_dev.zero_()

    def update(self, predictions: torch.Tensor, targets: torch.Tensor, loss):
        """predictions: fp32 logits [B, n] (classification) or [B, n, H, W] (segmentation) on the device; targets
        int64 [B] / [B, H, W]; loss: the batch's loss as a python float or a 0-dim device tensor (no sync then)."""
        if not predictions.is_cuda:
            raise RuntimeError("mia_amd metrics need tensors on an MI355X device; there is no CPU path")
        n = predictions.shape[1]
        if self._conf is None:
            self._conf = torch.zeros((self.num_classes, self.num_classes), dtype=torch.int64, device=predictions.device)
            self._loss_dev = torch.zeros((1,), dtype=torch.float64, device=predictions.device)
        if n != self.num_classes:
            raise ValueError(f"expected {self.num_classes} classes, got logits with {n}")
        z = predictions.detach().float().contiguous()
        t = targets.detach().long().contiguous()
        B = z.shape[0]
        P = z[0, 0].numel() if z.dim() > 2 else 1
        if t.numel() != B * P:
            raise ValueError(f"targets {tuple(t.shape)} do not match predictions {tuple(z.shape)}")
        loss_t = None
        if isinstance(loss, torch.Tensor):
            loss_t = loss.detach().float().reshape(1).contiguous()
        else:
            self.running_loss += float(loss)
        lib().mia_confusion_update(_p(z), n * P, P, 1, _p(t), B, P, n, _p(self._conf), _p(loss_t),
                                   _p(self._loss_dev) if loss_t is not None else None, _stream())
        self.num_samples += targets.size(0)
        self._seen = True

    def confusion_matrix(self) -> np.ndarray:
        """[true, predicted] counts; the only device-to-host copy."""
        if self._conf is None:
            return np.zeros((self.num_classes, self.num_classes), dtype=np.int64)
        return self._conf.cpu().numpy()

    def compute_metrics(self) -> Dict[str, float]:
        loss_sum = self.running_loss + (float(self._loss_dev.item()) if self._loss_dev is not None else 0.0)
        if not self._seen:
            return {"loss": loss_sum / max(self.num_samples, 1)}
        avg_loss = loss_sum / self.num_samples
        cm = self.confusion_matrix().astype(np.float64)
        if self.task == "classification":
            return self._classification(avg_loss, cm)
        return self._segmentation(avg_loss, cm)

    def _classification(self, avg_loss, cm):
        support, predicted, tp = cm.sum(1), cm.sum(0), np.diag(cm)
        total = cm.sum()
        with np.errstate(divide="ignore", invalid="ignore"):
            prec = np.where(predicted > 0, tp / predicted, 0.0)       # zero_division=0
            rec = np.where(support > 0, tp / support, 0.0)
            f1 = np.where(prec + rec > 0, 2 * prec * rec / (prec + rec), 0.0)
        wsum = support.sum()
        metrics = {"loss": avg_loss, "accuracy": float(tp.sum() / total) if total else 0.0,
                   "precision": float((prec * support).sum() / wsum) if wsum else 0.0,
                   "recall": float((rec * support).sum() / wsum) if wsum else 0.0,
                   "f1_score": float((f1 * support).sum() / wsum) if wsum else 0.0}
        # scikit-learn's average=None arrays cover only the labels PRESENT in targets or predictions, and the
        # reference pairs entry i of those arrays with class_names[i] (metric.py:77-81); kept as is.
        present = np.flatnonzero((support > 0) | (predicted > 0))
        for i, name in enumerate(self.class_names):
            if i < len(present):
                c = present[i]
                metrics[f"precision_{name}"] = float(prec[c])
                metrics[f"recall_{name}"] = float(rec[c])
                metrics[f"f1_{name}"] = float(f1[c])
        return metrics

    def _segmentation(self, avg_loss, cm):
        support, predicted, tp = cm.sum(1), cm.sum(0), np.diag(cm)
        total = cm.sum()
        iou = tp / (support + predicted - tp + 1e-7)
        dice = 2.0 * tp / (support + predicted + 1e-7)
        metrics = {"loss": avg_loss, "accuracy": float(tp.sum() / total) if total else 0.0,
                   "mean_iou": float(iou.mean()), "mean_dice": float(dice.mean())}
        for i, name in enumerate(self.class_names):
            if i < len(iou):
                metrics[f"iou_{name}"] = float(iou[i])
                metrics[f"dice_{name}"] = float(dice[i])
        return metrics


def create_metrics_calculator(task: str, num_classes: int, class_names: Optional[List[str]] = None) -> MetricsCalculator:
    return MetricsCalculator(task=task, num_classes=num_classes, class_names=class_names)


class AverageMeter:
    """metric.py:227-245."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class EarlyStopping:
    """metric.py:248-276: stop after `patience` calls without an improvement larger than min_delta."""

    def __init__(self, patience: int = 10, min_delta: float = 0.0, mode: str = "min"):
        self.patience, self.min_delta, self.mode = patience, min_delta, mode
        self.counter, self.best_score, self.early_stop = 0, None, False

    def _better(self, score, best):
        return score < best - self.min_delta if self.mode == "min" else score > best + self.min_delta

    def __call__(self, score):
        if self.best_score is None:
            self.best_score = score
        elif self._better(score, self.best_score):
            self.best_score, self.counter = score, 0
        else:
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True
        return self.early_stop
