"""End-to-end training sanity in bf16 on the device: a few AdamW steps on one fixed batch must drive the loss down,
for the classification task (trunk + CBAM/attention head + cross-entropy) and the segmentation task (trunk +
CBAM-UNet decoder + cross-entropy/Dice), with dropout active. Also: the bf16 segmentation gradients agree in
direction with the fp32 ones on the same weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mia_amd  # noqa: E402,F401
from mia_amd.optim import AdamW  # noqa: E402
from mia_amd.src.loss import get_loss_function  # noqa: E402
from mia_amd.src.metric import create_metrics_calculator  # noqa: E402
from mia_amd.src.model import FoundationModel  # noqa: E402


def _blobs(n, size, seed):
    """Images with a bright disc on a dark, noisy background; the mask is the disc, the label its quadrant."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    imgs, masks, labels = [], [], []
    for _ in range(n):
        q = int(rng.integers(0, 4))
        cy = size * (0.3 if q < 2 else 0.7) + rng.normal(0, 2)
        cx = size * (0.3 if q % 2 == 0 else 0.7) + rng.normal(0, 2)
        m = ((yy - cy) ** 2 + (xx - cx) ** 2) <= (size * 0.18) ** 2
        img = rng.normal(0, 0.3, (3, size, size)) + 2.0 * m[None]
        imgs.append(img.astype(np.float32)); masks.append(m.astype(np.int64)); labels.append(q)
    return (torch.from_numpy(np.stack(imgs)).cuda(), torch.from_numpy(np.stack(masks)).cuda(),
            torch.tensor(labels).cuda())


@pytest.mark.parametrize("task", ["classification", "segmentation"])
def test_loss_goes_down_in_bf16(task):
    torch.manual_seed(0)
    n_out = 4 if task == "classification" else 2
    model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2,
                            classification_heads={"d": 4} if task == "classification" else {},
                            segmentation_heads={"d": 2} if task == "segmentation" else {}, use_attention=True,
                            compute_dtype="bf16").cuda().train()
    x, masks, labels = _blobs(16, 96, 1)
    target = labels if task == "classification" else masks
    crit = get_loss_function(task, num_classes=n_out)
    opt = AdamW(model.parameters(), lr=2e-3, weight_decay=1e-4)
    calc = create_metrics_calculator(task, n_out)
    losses = []
    for _ in range(25):
        opt.zero_grad(set_to_none=True)
        out = model(x, task, "d")
        loss = crit(out, target)
        loss.backward()
        opt.step()
        calc.update(out, target, loss.detach())
        losses.append(loss.item())
    assert all(np.isfinite(losses))
    assert np.mean(losses[-5:]) < 0.8 * np.mean(losses[:3]), losses
    assert losses[-1] < losses[0]
    m = calc.compute_metrics()
    assert 0.0 <= m["accuracy"] <= 1.0 and np.isfinite(m["loss"])


def test_bf16_decoder_gradients_point_the_same_way_as_fp32():
    """The CBAM-UNet decoder alone, same weights and the same (bf16-representable) trunk features in both precisions:
    the bf16 backward kernels (im2col / col2im, upsample gradient, BatchNorm + ReLU backward, CBAM on large maps) must
    give gradients nearly parallel to the fp32 ones. (Not closer than ~0.98: BatchNorm's backward subtracts the
    per-channel mean and the x-hat component of the incoming gradient, which at initialisation is most of it, so the
    2^-9 rounding of a bf16 gradient tensor is a few percent of what remains.)"""
    from mia_amd.decoders import EnhancedUNetDecoder
    chans, sizes = [24, 32, 48, 136, 384], [64, 32, 16, 8, 4]
    g = torch.Generator(device="cuda").manual_seed(3)
    feats16 = [torch.randn(4, s, s, c, device="cuda", generator=g).bfloat16() for c, s in zip(chans, sizes)]
    target = torch.randint(0, 2, (4, 128, 128), device="cuda", generator=g)
    torch.manual_seed(2)
    grads, state = {}, None
    for dt in (torch.float32, torch.bfloat16):
        dec = EnhancedUNetDecoder(chans, num_classes=2, dropout=0.0).cuda().train()
        if state is None:
            state = {k: v.clone() for k, v in dec.state_dict().items()}
        else:
            dec.load_state_dict(state)
        feats = [f.to(dt).requires_grad_(True) for f in feats16]
        get_loss_function("segmentation", num_classes=2)(dec(feats, (128, 128)), target).backward()
        grads[dt] = {k: p.grad.detach().float().flatten() for k, p in dec.named_parameters()}
        grads[dt].update({f"feat{i}": f.grad.detach().float().flatten() for i, f in enumerate(feats)})

    def cosine(a, b):
        return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()
    cos = {k: cosine(grads[torch.float32][k], grads[torch.bfloat16][k]) for k in grads[torch.float32]
           if grads[torch.float32][k].numel() >= 256 and not k.endswith("bias")}
    worst = min(cos, key=cos.get)
    assert np.median(list(cos.values())) > 0.97 and cos[worst] > 0.9, (np.median(list(cos.values())), worst, cos[worst])
