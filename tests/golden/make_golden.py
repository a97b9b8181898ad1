"""Generate golden vectors from the REFERENCE's own classes (run once, in the build
container, where /root/reference is mounted). Output: small .npz files next to
this script, committed. Nothing here runs on the GPU box.

The reference module /root/reference/foundation_model/src/model.py imports two
third-party packages that are absent offline (timm, segmentation_models_pytorch);
they are only needed by FoundationModel.__init__, not by the head classes, so
they are replaced by empty module objects for the import.

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from seeding import load_seeded, randn  # noqa: E402

REF = "/root/reference/foundation_model"


def import_reference_model():
    sys.dont_write_bytecode = True
    for name in ("timm", "segmentation_models_pytorch"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    from src import model as ref  # type: ignore
    sys.path.pop(0)
    return ref


def grad_digest(t: torch.Tensor) -> np.ndarray:
    """[sum, l2, first 16 values] — enough to pin a large gradient without storing it."""
    f = t.detach().double().flatten()
    head = f[:16].numpy()
    if head.size < 16:
        head = np.pad(head, (0, 16 - head.size))
    return np.concatenate([[f.sum().item(), f.norm().item()], head]).astype(np.float64)


def run_module(mod, x, train: bool, full_grads: bool):
    mod.train(train)
    x = x.clone().requires_grad_(True)
    out = mod(x)
    res = {"out": out.detach().numpy()}
    if train:
        # fixed, non-trivial upstream gradient
        g = randn(777, *out.shape)
        out.backward(g)
        res["dx"] = x.grad.numpy()
        for k, p in mod.named_parameters():
            if p.grad is None:
                continue
            res["grad." + k] = p.grad.numpy() if full_grads else grad_digest(p.grad)
        for k, b in mod.named_buffers():
            if "running" in k:
                res["buf." + k] = b.detach().numpy().copy()
    return res


def import_reference_loss():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from src import loss as ref_loss  # type: ignore
    sys.path.pop(0)
    return ref_loss


def seg_feature_sizes(size):
    """Spatial sizes of the five trunk features for a size x size input (3x3/5x5 stride-2 convs with 'same' pad)."""
    out, s = [], size
    for _ in range(5):
        s = (s - 1) // 2 + 1
        out.append(s)
    return out


def run_decoder(dec, feats, size, train, full):
    dec.train(train)
    feats = [f.clone().requires_grad_(True) for f in feats]
    out = dec(feats, (size, size))
    res = {"out": out.detach().numpy()}
    if train:
        out.backward(randn(778, *out.shape))
        for i, f in enumerate(feats):
            res[f"dfeat{i}"] = f.grad.numpy() if full else grad_digest(f.grad)
        for k, p in dec.named_parameters():
            res["grad." + k] = p.grad.numpy() if full else grad_digest(p.grad)
        for k, b in dec.named_buffers():
            if "running" in k:
                res["buf." + k] = b.detach().numpy().copy()
    return res


def segmentation_goldens(ref):
    """G9: DecoderBlock, UNetDecoder, EnhancedUNetDecoder (model.py:179-237, :417-494) with dropout 0 so train mode is
    deterministic; G10: the segmentation losses (loss.py:27-80, :105-136)."""
    chans = [24, 32, 48, 136, 384]
    for train in (True, False):
        m = load_seeded(ref.DecoderBlock(24, 16, dropout=0.0), 51)
        r = run_module(m, randn(52, 2, 24, 6, 7), train, True)
        np.savez_compressed(os.path.join(HERE, f"decblock_24_16_6x7_{'train' if train else 'eval'}.npz"), **r)
    for (cls, tag) in ((ref.EnhancedUNetDecoder, "enhanced"), (ref.UNetDecoder, "unet")):
        for (size, ncls) in ((64, 3), (45, 1)):
            sizes = seg_feature_sizes(size)
            for train in (True, False):
                dec = load_seeded(cls(chans, num_classes=ncls, dropout=0.0), 61)
                feats = [randn(62 + i, 2, c, s, s) for i, (c, s) in enumerate(zip(chans, sizes))]
                r = run_decoder(dec, feats, size, train, False)
                np.savez_compressed(os.path.join(HERE, f"dec_{tag}_{size}_n{ncls}_{'train' if train else 'eval'}.npz"), **r)
    ref_loss = import_reference_loss()
    rng = np.random.default_rng(71)
    for (n, h, w, tag) in ((3, 17, 19, "n3"), (1, 8, 8, "n1"), (4, 32, 32, "n4")):
        logits = randn(72 + n, 2, n, h, w, scale=2.0)
        target = torch.from_numpy(rng.integers(0, n, (2, h, w)).astype(np.int64))
        weights = torch.from_numpy((0.5 + rng.random(n)).astype(np.float32))
        res = {"logits": logits.numpy(), "target": target.numpy(), "weights": weights.numpy()}
        for (name, crit) in (("combined", ref_loss.CombinedSegmentationLoss(num_classes=n)),
                             ("combined_w", ref_loss.CombinedSegmentationLoss(num_classes=n, weights=weights)),
                             ("dice", ref_loss.DiceLoss(num_classes=n)), ("iou", ref_loss.IoULoss(num_classes=n))):
            lg = logits.clone().requires_grad_(True)
            loss = crit(lg, target)
            loss.backward()
            res[name + ".loss"] = np.float64(loss.item())
            res[name + ".dlogits"] = lg.grad.numpy()
        np.savez_compressed(os.path.join(HERE, f"segloss_{tag}.npz"), **res)

    # G11: the remaining loss variants of loss.py: FocalLoss (:83-102) on class logits and on pixel logits, and the
    # class-weighted cross-entropy get_loss_function returns for classification (:143-147)
    res = {}
    for (tag, shape) in (("cls", (37, 4)), ("pix", (2, 3, 9, 11))):
        n = shape[1]
        logits = randn(81 + n, *shape, scale=2.0)
        tshape = (shape[0],) if len(shape) == 2 else (shape[0],) + shape[2:]
        target = torch.from_numpy(rng.integers(0, n, tshape).astype(np.int64))
        weights = torch.from_numpy((0.5 + rng.random(n)).astype(np.float32))
        res.update({f"{tag}.logits": logits.numpy(), f"{tag}.target": target.numpy(), f"{tag}.weights": weights.numpy()})
        cases = {"focal": ref_loss.FocalLoss(), "focal_a05_g3_sum": ref_loss.FocalLoss(alpha=0.5, gamma=3.0, reduction="sum"),
                 "wce": ref_loss.get_loss_function("classification", class_weights=weights)}
        for name, crit in cases.items():
            lg = logits.clone().requires_grad_(True)
            loss = crit(lg, target)
            loss.backward()
            res[f"{tag}.{name}.loss"] = np.float64(loss.item())
            res[f"{tag}.{name}.dlogits"] = lg.grad.numpy()
        with torch.no_grad():
            res[f"{tag}.focal_none"] = ref_loss.FocalLoss(reduction="none")(logits, target).numpy()
    np.savez_compressed(os.path.join(HERE, "loss_variants.npz"), **res)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    ref = import_reference_model()

    # G1: CBAM (model.py:96-108)
    for (c, hw, tag, full) in ((384, 10, "c384_10x10", False), (64, 2, "c64_2x2", True), (64, 5, "c64_5x5", True)):
        for train in (True, False):
            m = load_seeded(ref.CBAM(c), 11)
            x = randn(12, 2, c, hw, hw)
            r = run_module(m, x, train, full)
            np.savez_compressed(os.path.join(HERE, f"cbam_{tag}_{'train' if train else 'eval'}.npz"), **r)

    # G2: MultiHeadAttentionClassifier (model.py:112-175), dropout 0 so train mode is deterministic
    for (c, hw, ncls, tag, full) in ((384, 10, 4, "c384_10x10", False), (384, 2, 4, "c384_2x2", False), (64, 3, 2, "c64_3x3", True)):
        for train in (True, False):
            m = load_seeded(ref.MultiHeadAttentionClassifier(c, 8, ncls, dropout=0.0), 21)
            x = randn(22, 2, c, hw, hw)
            r = run_module(m, x, train, full)
            np.savez_compressed(os.path.join(HERE, f"mha_{tag}_{'train' if train else 'eval'}.npz"), **r)

    # G3: simple head (model.py:308-316) — built exactly as the reference builds it
    import torch.nn as nn
    for (c, hw, ncls, tag) in ((384, 10, 4, "c384_10x10"),):
        head = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Dropout(0.0), nn.Linear(c, 512),
                             nn.ReLU(), nn.Dropout(0.0), nn.Linear(512, ncls))
        m = load_seeded(head, 31)
        x = randn(32, 2, c, hw, hw)
        r = run_module(m, x, True, True)
        np.savez_compressed(os.path.join(HERE, f"simple_{tag}_train.npz"), **r)

    # G4: ChannelPool channel order [max, mean] (model.py:77-79)
    x = randn(41, 2, 16, 3, 3)
    np.savez_compressed(os.path.join(HERE, "channelpool.npz"), out=ref.ChannelPool()(x).numpy())

    # G5: load_model_weights_safely key handling (model.py:549-604) on a toy module
    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Linear(2, 2)
            self.b = nn.Linear(2, 2)
    toy = Toy()
    sd = {"a.weight": torch.ones(2, 2), "a.bias": torch.zeros(2), "classifier.weight": torch.ones(1),
          "bn2.running_mean": torch.ones(1), "conv_head.weight": torch.ones(1), "zzz.weight": torch.ones(1)}
    missing, unexpected = ref.load_model_weights_safely(toy, {"model_state_dict": sd}, suppress_warnings=True)
    missing2, unexpected2 = ref.load_model_weights_safely(Toy(), sd, suppress_warnings=True)
    assert (missing, unexpected) == (missing2, unexpected2)
    with open(os.path.join(HERE, "load_weights_keys.txt"), "w") as f:
        f.write("missing=" + ",".join(missing) + "\n")
        f.write("unexpected=" + ",".join(unexpected) + "\n")

    # G6: parameter counts of the reference-owned modules
    counts = {
        "CBAM(384)": sum(p.numel() for p in ref.CBAM(384).parameters()),
        "MHA(384,8,4)": sum(p.numel() for p in ref.MultiHeadAttentionClassifier(384, 8, 4).parameters()),
        "MHA(384,8,2)": sum(p.numel() for p in ref.MultiHeadAttentionClassifier(384, 8, 2).parameters()),
        "EnhancedUNetDecoder(nc=2)": sum(
            p.numel() for p in ref.EnhancedUNetDecoder([24, 32, 48, 136, 384], num_classes=2).parameters()),
    }
    with open(os.path.join(HERE, "param_counts.txt"), "w") as f:
        for k, v in counts.items():
            f.write(f"{k}={v}\n")
    print(counts)
    segmentation_goldens(ref)
    print("state-dict keys of MHA head:", list(ref.MultiHeadAttentionClassifier(384, 8, 4).state_dict().keys()))


if __name__ == "__main__":
    main()
