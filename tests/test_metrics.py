"""Evaluation counters: the numbers derived from the confusion matrix against scikit-learn (what the reference's
MetricsCalculator calls, src/metric.py:62-74) on CPU, and the device kernel against numpy counts (bit-exact)."""
import numpy as np
import pytest
import torch

import mia_amd  # noqa: F401
from mia_amd.src.metric import AverageMeter, EarlyStopping, MetricsCalculator, create_metrics_calculator


def _inject(calc, y_true, y_pred, n, loss_sum, samples):
    cm = np.zeros((n, n), dtype=np.int64)
    np.add.at(cm, (y_true, y_pred), 1)
    calc._conf = torch.from_numpy(cm)
    calc._loss_dev = torch.zeros(1, dtype=torch.float64)
    calc.running_loss, calc.num_samples, calc._seen = loss_sum, samples, True


@pytest.mark.parametrize("n,absent", [(4, None), (4, 2), (2, None), (5, 0)])
def test_classification_metrics_equal_sklearn(n, absent):
    from sklearn.metrics import accuracy_score, f1_score, precision_score, recall_score
    rng = np.random.default_rng(n * 7 + (absent or 0))
    y_true, y_pred = rng.integers(0, n, 500), rng.integers(0, n, 500)
    if absent is not None:      # a class that never occurs: scikit-learn drops it from the per-class arrays
        y_true[y_true == absent] = (absent + 1) % n
        y_pred[y_pred == absent] = (absent + 1) % n
    names = [f"c{i}" for i in range(n)]
    calc = create_metrics_calculator("classification", n, names)
    _inject(calc, y_true, y_pred, n, 12.5, 500)
    m = calc.compute_metrics()
    assert m["loss"] == pytest.approx(12.5 / 500)
    assert m["accuracy"] == pytest.approx(accuracy_score(y_true, y_pred), abs=1e-12)
    for key, fn in (("precision", precision_score), ("recall", recall_score), ("f1_score", f1_score)):
        assert m[key] == pytest.approx(fn(y_true, y_pred, average="weighted", zero_division=0), abs=1e-12), key
    per = {"precision": precision_score(y_true, y_pred, average=None, zero_division=0),
           "recall": recall_score(y_true, y_pred, average=None, zero_division=0),
           "f1": f1_score(y_true, y_pred, average=None, zero_division=0)}
    for i, name in enumerate(names):      # the reference pairs array entry i with class_names[i] (metric.py:77-81)
        for key, arr in per.items():
            if i < len(arr):
                assert m[f"{key}_{name}"] == pytest.approx(arr[i], abs=1e-12), (key, name)
            else:
                assert f"{key}_{name}" not in m


def test_segmentation_metrics_equal_reference_formulas():
    rng = np.random.default_rng(3)
    n = 3
    y_true, y_pred = rng.integers(0, n, 4096), rng.integers(0, n, 4096)
    calc = MetricsCalculator("segmentation", n)
    _inject(calc, y_true, y_pred, n, 3.0, 4)
    m = calc.compute_metrics()
    ious, dices = [], []
    for c in range(n):      # metric.py:104-114
        t, p = y_true == c, y_pred == c
        inter, union = np.logical_and(t, p).sum(), np.logical_or(t, p).sum()
        ious.append(inter / (union + 1e-7))
        dices.append(2.0 * inter / (t.sum() + p.sum() + 1e-7))
        assert m[f"iou_Class_{c}"] == pytest.approx(ious[-1], abs=1e-12)
        assert m[f"dice_Class_{c}"] == pytest.approx(dices[-1], abs=1e-12)
    assert m["mean_iou"] == pytest.approx(np.mean(ious), abs=1e-12)
    assert m["mean_dice"] == pytest.approx(np.mean(dices), abs=1e-12)
    assert m["accuracy"] == pytest.approx((y_true == y_pred).mean(), abs=1e-12)
    assert m["loss"] == pytest.approx(0.75)


def test_empty_calculator_and_host_utilities():
    calc = MetricsCalculator("classification", 3)
    assert calc.compute_metrics() == {"loss": 0.0}
    with pytest.raises(ValueError):
        MetricsCalculator("detection", 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        calc.update(torch.zeros(2, 3), torch.zeros(2, dtype=torch.long), 0.1)
    am = AverageMeter()
    am.update(2.0, 2)
    am.update(4.0, 2)
    assert am.avg == 3.0 and am.val == 4.0 and am.count == 4
    es = EarlyStopping(patience=2, min_delta=0.1, mode="min")
    assert [es(s) for s in (1.0, 0.95, 0.8, 0.79, 0.78)] == [False, False, False, False, True]
    es = EarlyStopping(patience=1, mode="max")
    assert [es(s) for s in (0.5, 0.6, 0.6)] == [False, False, True]


@pytest.mark.gpu
@pytest.mark.parametrize("task,shape", [("classification", (257, 4)), ("segmentation", (3, 5, 37, 41)),
                                        ("segmentation", (2, 1, 16, 16)), ("classification", (1, 64))])
def test_confusion_kernel_is_exact(task, shape):
    n = shape[1]
    rng = np.random.default_rng(11)
    calc = MetricsCalculator(task, n)
    cm = np.zeros((n, n), dtype=np.int64)
    loss_sum = 0.0
    for it in range(3):
        z = torch.from_numpy(rng.integers(-3, 4, shape).astype(np.float32))      # small integers: many exact ties
        tshape = (shape[0],) if task == "classification" else (shape[0],) + shape[2:]
        t = torch.from_numpy(rng.integers(0, n, tshape))
        pred = z.argmax(1)
        np.add.at(cm, (t.numpy().ravel(), pred.numpy().ravel()), 1)
        loss = 0.25 * (it + 1)
        calc.update(z.cuda(), t.cuda(), torch.tensor(loss, device="cuda") if it % 2 == 0 else loss)
        loss_sum += loss
    assert np.array_equal(calc.confusion_matrix(), cm)
    m = calc.compute_metrics()
    assert m["loss"] == pytest.approx(loss_sum / (3 * shape[0]))
    assert m["accuracy"] == pytest.approx(np.trace(cm) / cm.sum())
    calc.reset()
    assert calc.confusion_matrix().sum() == 0 and calc.compute_metrics() == {"loss": 0.0}


@pytest.mark.gpu
def test_confusion_kernel_full_size_counts_every_pixel():
    n, B, H, W = 2, 16, 300, 300
    g = torch.Generator(device="cuda").manual_seed(0)
    z = torch.randn(B, n, H, W, device="cuda", generator=g)
    t = torch.randint(0, n, (B, H, W), device="cuda", generator=g)
    calc = MetricsCalculator("segmentation", n)
    calc.update(z, t, 0.0)
    cm = calc.confusion_matrix()
    assert cm.sum() == B * H * W
    assert np.array_equal(cm.sum(1), np.bincount(t.cpu().numpy().ravel(), minlength=n))
    assert np.array_equal(cm.sum(0), np.bincount(z.argmax(1).cpu().numpy().ravel(), minlength=n))
