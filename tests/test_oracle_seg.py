"""CPU tests: the oracle's segmentation restatement (oracle/seg_ref.py) against golden vectors produced by the
reference's own DecoderBlock / UNetDecoder / EnhancedUNetDecoder and loss classes (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from seeding import load_seeded, randn
from oracle import seg_ref as S
from test_oracle import check_module, digest

G = os.path.join(os.path.dirname(__file__), "golden")
CHANS = [24, 32, 48, 136, 384]


def feature_sizes(size):
    out, s = [], size
    for _ in range(5):
        s = (s - 1) // 2 + 1
        out.append(s)
    return out


@pytest.mark.parametrize("train", [True, False])
def test_decoder_block_matches_reference_golden(train):
    gold = np.load(os.path.join(G, f"decblock_24_16_6x7_{'train' if train else 'eval'}.npz"))
    mod = load_seeded(S.DecoderBlock(24, 16, dropout=0.0), 51).train(train)
    x = randn(52, 2, 24, 6, 7).requires_grad_(True)
    out = mod(x)
    np.testing.assert_allclose(out.detach().numpy(), gold["out"], rtol=1e-4, atol=1e-5)
    if not train:
        return
    out.backward(randn(777, *out.shape))
    np.testing.assert_allclose(x.grad.numpy(), gold["dx"], rtol=1e-4, atol=1e-5)
    for k, p in mod.named_parameters():
        # a conv bias feeding a train-mode BatchNorm has an exactly-zero gradient: both sides hold rounding noise
        atol = 2e-5 if k in ("conv1.bias", "conv2.bias") else 1e-5
        np.testing.assert_allclose(p.grad.numpy(), gold["grad." + k], rtol=1e-4, atol=atol, err_msg=k)
    for k, b in mod.named_buffers():
        if "running" in k:
            np.testing.assert_allclose(b.numpy(), gold["buf." + k], rtol=1e-5, atol=1e-7, err_msg=k)


@pytest.mark.parametrize("kind", ["enhanced", "unet"])
@pytest.mark.parametrize("size,ncls", [(64, 3), (45, 1)])
@pytest.mark.parametrize("train", [True, False])
def test_decoder_matches_reference_golden(kind, size, ncls, train):
    gold = np.load(os.path.join(G, f"dec_{kind}_{size}_n{ncls}_{'train' if train else 'eval'}.npz"))
    cls = S.EnhancedUNetDecoder if kind == "enhanced" else S.UNetDecoder
    dec = load_seeded(cls(CHANS, num_classes=ncls, dropout=0.0), 61).train(train)
    feats = [randn(62 + i, 2, c, s, s).requires_grad_(True) for i, (c, s) in enumerate(zip(CHANS, feature_sizes(size)))]
    out = dec(feats, (size, size))
    np.testing.assert_allclose(out.detach().numpy(), gold["out"], rtol=1e-4, atol=1e-5)
    if not train:
        return
    out.backward(randn(778, *out.shape))
    for i, f in enumerate(feats):
        np.testing.assert_allclose(digest(f.grad), gold[f"dfeat{i}"], rtol=1e-3, atol=1e-4, err_msg=f"dfeat{i}")
    for k, p in dec.named_parameters():
        np.testing.assert_allclose(digest(p.grad), gold["grad." + k], rtol=1e-3, atol=2e-4, err_msg=k)
    for k, b in dec.named_buffers():
        if "running" in k:
            np.testing.assert_allclose(b.numpy(), gold["buf." + k], rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("tag", ["n1", "n3", "n4"])
def test_segmentation_losses_match_reference_golden(tag):
    gold = np.load(os.path.join(G, f"segloss_{tag}.npz"))
    logits, target = torch.from_numpy(gold["logits"]), torch.from_numpy(gold["target"])
    weights = torch.from_numpy(gold["weights"])
    n = logits.shape[1]
    cases = {"combined": lambda z: S.combined_segmentation_loss(z, target, n),
             "combined_w": lambda z: S.combined_segmentation_loss(z, target, n, weights=weights),
             "dice": lambda z: S.dice_loss(z, target, n), "iou": lambda z: S.iou_loss(z, target, n)}
    for name, fn in cases.items():
        z = logits.clone().requires_grad_(True)
        loss = fn(z)
        loss.backward()
        np.testing.assert_allclose(loss.item(), gold[name + ".loss"], rtol=1e-6, atol=1e-7, err_msg=name)
        np.testing.assert_allclose(z.grad.numpy(), gold[name + ".dlogits"], rtol=1e-5, atol=1e-9, err_msg=name)


def test_calculate_class_weights_matches_reference():
    """calculate_class_weights against the reference function's outputs (tests/golden/make_class_weights.py)."""
    import os
    import numpy as np
    import torch
    import mia_amd  # noqa: F401
    from mia_amd.src.loss import calculate_class_weights
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "class_weights.npz"))
    cls = [{"label": int(v)} for v in g["labels"]]
    seg = [{"mask": torch.from_numpy(m)} for m in g["masks"]]
    for n, key in ((4, "cls_w4"), (6, "cls_w6")):
        w = calculate_class_weights(cls, n, "classification")
        assert w.dtype == torch.float32
        np.testing.assert_array_equal(w.numpy(), g[key])
    for n, key in ((3, "seg_w3"), (2, "seg_w2")):
        np.testing.assert_array_equal(calculate_class_weights(seg, n, "segmentation").numpy(), g[key])
    import pytest
    with pytest.raises(ValueError, match="Unknown task"):
        calculate_class_weights(cls, 4, "detection")
