"""GPU parity of the segmentation path: kernels against plain PyTorch fp32 formulas on the same seeded inputs, the
decoder modules and the losses against golden vectors produced by the REFERENCE's own classes
(tests/golden/make_golden.py ran DecoderBlock / UNetDecoder / EnhancedUNetDecoder of
/root/reference/foundation_model/src/model.py and the losses of src/loss.py on CPU).
fp32: outputs within 1e-4, gradients within 1e-3, relative to each tensor's largest magnitude; bf16: 3e-2."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import mia_amd  # noqa: E402,F401
from mia_amd import seg_ops  # noqa: E402
from mia_amd.decoders import DecoderBlock, EnhancedUNetDecoder, UNetDecoder  # noqa: E402
import relu_ties  # noqa: E402
from seeding import load_seeded, randn  # noqa: E402
from test_heads_gpu import check, digest, nhwc, rel  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
CHANS = [24, 32, 48, 136, 384]


def feature_sizes(size):
    out, s = [], size
    for _ in range(5):
        s = (s - 1) // 2 + 1
        out.append(s)
    return out


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("shape", [(2, 5, 7, 8), (1, 1, 1, 4), (3, 16, 9, 24)])
def test_im2col_col2im_are_exact_transposes(shape):
    B, H, W, C = shape
    x = randn(1, B, C, H, W).cuda()
    col = seg_ops.im2col3x3(nhwc(x))
    ref = F.unfold(x, 3, padding=1).view(B, C, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 9 * C)
    assert torch.equal(col, ref)
    g = randn(2, B * H * W, 9 * C).cuda()
    dx = seg_ops.col2im3x3(g, (B, H, W, C))
    gref = F.fold(g.view(B, H * W, 9, C).permute(0, 3, 2, 1).reshape(B, C * 9, H * W), (H, W), 3, padding=1)
    assert rel(dx.permute(0, 3, 1, 2), gref) < 1e-6


def test_im2col_prologue_applies_bn_relu_and_mask():
    B, H, W, C = 2, 4, 5, 8
    x = randn(3, B, C, H, W).cuda()
    sc, sh = randn(4, C).cuda(), randn(5, C).cuda()
    gate = (randn(6, B, C).cuda() > 0).float() * 2.0
    col = seg_ops.im2col3x3(nhwc(x), scale=sc, shift=sh, act=seg_ops.ACT_RELU, gate=gate)
    a = F.relu(x * sc[None, :, None, None] + sh[None, :, None, None]) * gate[:, :, None, None]
    ref = F.unfold(a, 3, padding=1).view(B, C, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 9 * C)
    assert rel(col, ref) < 1e-6


@pytest.mark.parametrize("h,w,H,W,C1,C2", [(2, 2, 3, 3, 8, 4), (10, 10, 19, 19, 16, 8), (5, 7, 9, 14, 4, 0),
                                            (19, 19, 38, 38, 8, 8), (6, 6, 6, 6, 4, 4), (8, 8, 3, 5, 4, 0)])
def test_upsample_concat_matches_torch(h, w, H, W, C1, C2):
    B = 2
    x = randn(7, B, C1, h, w).cuda().requires_grad_(True)
    skip = randn(8, B, C2, H, W).cuda().requires_grad_(True) if C2 else None
    xn = nhwc(x.detach()).requires_grad_(True)
    sn = nhwc(skip.detach()).requires_grad_(True) if C2 else None
    out = seg_ops.UpsampleConcatFn.apply(xn, sn, H, W)
    up = F.interpolate(x, size=(H, W), mode="bilinear", align_corners=False)
    ref = torch.cat((up, skip), 1) if C2 else up
    assert rel(out.permute(0, 3, 1, 2), ref) < 1e-6
    g = randn(9, B, C1 + C2, H, W).cuda()
    out.backward(nhwc(g))
    ref.backward(g)
    assert rel(xn.grad.permute(0, 3, 1, 2), x.grad) < 1e-5
    if C2:
        assert torch.equal(sn.grad.permute(0, 3, 1, 2), skip.grad)


@pytest.mark.parametrize("n,C,H,W", [(3, 16, 9, 11), (1, 16, 4, 4), (8, 64, 5, 5), (2, 24, 33, 31), (9, 16, 7, 9),
                                     (21, 16, 12, 10), (32, 64, 6, 5)])
def test_pixel_linear_matches_conv1x1(n, C, H, W):
    B = 2
    x = randn(11, B, C, H, W).cuda().requires_grad_(True)
    conv = torch.nn.Conv2d(C, n, 1).cuda()
    xn = nhwc(x.detach()).requires_grad_(True)
    w, b = conv.weight.detach().clone().requires_grad_(True), conv.bias.detach().clone().requires_grad_(True)
    out = seg_ops.PixelLinearFn.apply(xn, w, b)
    ref = conv(x)
    assert out.shape == ref.shape and rel(out, ref) < 1e-5
    g = randn(12, *ref.shape).cuda()
    out.backward(g)
    ref.backward(g)
    assert rel(xn.grad.permute(0, 3, 1, 2), x.grad) < 1e-5
    assert rel(w.grad, conv.weight.grad) < 1e-4
    assert rel(b.grad, conv.bias.grad) < 1e-4


def test_channel_dropout_mask_keeps_whole_channels():
    m = DecoderBlock(8, 16, dropout=0.5).cuda().train()
    x = nhwc(randn(13, 4, 8, 6, 6)).cuda()
    torch.manual_seed(5)
    mask = torch.empty((64, 256), device="cuda")
    mia_amd.lib().mia_channel_dropout_mask(mask.data_ptr(), 64, 256, 0.5, 1234, None, None)
    vals = set(mask.unique().tolist())
    assert vals == {0.0, 2.0}
    assert abs(mask.eq(0).float().mean().item() - 0.5) < 0.03
    out = m(x)     # runs with a mask; output stays finite and the right shape
    assert out.shape == (4, 6, 6, 16) and torch.isfinite(out).all()
    out.sum().backward()


# ------------------------------------------------------------------------------------------------ losses
@pytest.mark.parametrize("tag", ["n1", "n3", "n4"])
def test_segmentation_losses_match_reference(tag):
    from mia_amd.src.loss import CombinedSegmentationLoss, DiceLoss, IoULoss, get_loss_function
    gold = np.load(os.path.join(G, f"segloss_{tag}.npz"))
    logits, target = torch.from_numpy(gold["logits"]).cuda(), torch.from_numpy(gold["target"]).cuda()
    weights = torch.from_numpy(gold["weights"]).cuda()
    n = logits.shape[1]
    cases = {"combined": get_loss_function("segmentation", num_classes=n),
             "combined_w": CombinedSegmentationLoss(num_classes=n, weights=weights),
             "dice": DiceLoss(num_classes=n), "iou": IoULoss(num_classes=n)}
    assert isinstance(cases["combined"], CombinedSegmentationLoss)
    for name, crit in cases.items():
        z = logits.clone().requires_grad_(True)
        loss = crit(z, target)
        loss.backward()
        assert abs(loss.item() - float(gold[name + ".loss"])) <= 1e-5 * max(1.0, abs(float(gold[name + ".loss"]))), name
        ref = gold[name + ".dlogits"]
        if np.abs(ref).max() == 0:
            assert z.grad.abs().max().item() < 1e-9, name
        else:
            assert rel(z.grad, ref) < 1e-4, name


@pytest.mark.parametrize("n", [9, 21, 32])
def test_segmentation_losses_with_many_classes_match_oracle(n):
    """More than 8 classes take the 32-class build of the loss kernels (the reference accepts any num_classes,
    model.py:214): loss and gradient against the CPU restatement of loss.py:27-80 / :105-136."""
    from mia_amd.src.loss import CombinedSegmentationLoss, DiceLoss, IoULoss
    from oracle import seg_ref as S
    logits = randn(60 + n, 2, n, 13, 11, scale=2.0)
    target = torch.from_numpy(np.random.default_rng(n).integers(0, n, (2, 13, 11)).astype(np.int64))
    weights = torch.from_numpy((0.5 + np.random.default_rng(n + 1).random(n)).astype(np.float32))
    cases = ((CombinedSegmentationLoss(num_classes=n, weights=weights),
              lambda lg: S.combined_segmentation_loss(lg, target, n, weights=weights)),
             (DiceLoss(num_classes=n), lambda lg: S.dice_loss(lg, target, n)),
             (IoULoss(num_classes=n), lambda lg: S.iou_loss(lg, target, n)))
    for crit, ref_fn in cases:
        lr = logits.clone().requires_grad_(True)
        ref = ref_fn(lr)
        ref.backward()
        lm = logits.clone().cuda().requires_grad_(True)
        out = crit(lm, target.cuda())
        out.backward()
        assert abs(out.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item())), type(crit).__name__
        assert rel(lm.grad.cpu(), lr.grad) < 1e-4, type(crit).__name__
    with pytest.raises(Exception, match="classes"):
        DiceLoss(num_classes=33)(torch.zeros(1, 33, 4, 4, device="cuda"), torch.zeros(1, 4, 4, dtype=torch.int64, device="cuda"))


def test_segmentation_loss_scales_with_upstream_gradient():
    from mia_amd.src.loss import CombinedSegmentationLoss
    logits = randn(21, 2, 3, 20, 20).cuda()
    target = torch.from_numpy(np.random.default_rng(3).integers(0, 3, (2, 20, 20))).cuda()
    z1, z2 = logits.clone().requires_grad_(True), logits.clone().requires_grad_(True)
    CombinedSegmentationLoss(3)(z1, target).backward()
    (CombinedSegmentationLoss(3)(z2, target) * 2.5).backward()
    assert rel(z2.grad, z1.grad * 2.5) < 1e-6


# ------------------------------------------------------------------------------------------------ modules
@pytest.mark.parametrize("train", [True, False])
def test_decoder_block_matches_reference(train):
    gold = np.load(os.path.join(G, f"decblock_24_16_6x7_{'train' if train else 'eval'}.npz"))
    # a conv bias feeding a train-mode BatchNorm has an exactly-zero gradient: both sides hold rounding noise
    check(load_seeded(DecoderBlock(24, 16, dropout=0.0), 51), randn(52, 2, 24, 6, 7), gold, train, True,
          zero_grad_params=("conv1.bias", "conv2.bias") if train else ())


def run_decoder(cls, size, ncls, train, dtype):
    dec = load_seeded(cls(CHANS, num_classes=ncls, dropout=0.0), 61).cuda().train(train)
    feats = [nhwc(randn(62 + i, 2, c, s, s)).cuda().to(dtype).requires_grad_(True)
             for i, (c, s) in enumerate(zip(CHANS, feature_sizes(size)))]
    out = dec(feats, (size, size))
    return dec, feats, out


@pytest.mark.parametrize("kind", ["enhanced", "unet"])
@pytest.mark.parametrize("size,ncls", [(64, 3), (45, 1)])
@pytest.mark.parametrize("train", [True, False])
def test_decoder_matches_reference(kind, size, ncls, train):
    gold = np.load(os.path.join(G, f"dec_{kind}_{size}_n{ncls}_{'train' if train else 'eval'}.npz"))
    cls = EnhancedUNetDecoder if kind == "enhanced" else UNetDecoder
    dec, feats, out = run_decoder(cls, size, ncls, train, torch.float32)
    assert out.dtype == torch.float32 and tuple(out.shape) == gold["out"].shape
    assert rel(out, gold["out"]) < 1e-4
    if not train:
        return
    out.backward(randn(778, *out.shape).cuda())

    def close(d, ref, k):
        assert abs(d[1] - ref[1]) <= 2e-3 * abs(ref[1]) + 1e-7, k                                   # l2 norm
        assert abs(d[0] - ref[0]) <= 2e-3 * max(abs(ref[1]), 1e-12) + 1e-5, k                       # sum vs l2 scale
        assert np.abs(d[2:] - ref[2:]).max() <= 2e-3 * max(np.abs(ref[2:]).max(), abs(ref[1]) * 1e-3) + 1e-7, k
    for i, f in enumerate(feats):
        close(digest(f.grad.permute(0, 3, 1, 2).contiguous()), gold[f"dfeat{i}"], f"dfeat{i}")
    for k, p in dec.named_parameters():
        assert p.grad is not None, k
        if k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            assert p.grad.abs().max().item() < 1e-3 * max(1.0, gold["grad." + k.replace("bias", "weight")][1]), k
            continue
        close(digest(p.grad), gold["grad." + k], k)
    for k, b in dec.named_buffers():
        if "running" in k:
            assert rel(b, gold["buf." + k]) < 1e-4, k


@pytest.mark.parametrize("kind", ["enhanced", "unet"])
def test_decoder_bf16_close_to_reference(kind):
    gold = np.load(os.path.join(G, f"dec_{kind}_64_n3_eval.npz"))
    cls = EnhancedUNetDecoder if kind == "enhanced" else UNetDecoder
    _, _, out = run_decoder(cls, 64, 3, False, torch.bfloat16)
    assert rel(out, gold["out"]) < 3e-2
    agree = (out.argmax(1).cpu() == torch.from_numpy(gold["out"]).argmax(1)).float().mean().item()
    assert agree > 0.97


def test_foundation_model_segmentation_path():
    """The drop-in's segmentation route: config-built heads, forward/backward at a non-multiple-of-32 size, new head
    added later, error behaviour of model.py:357-363."""
    from mia_amd.src.model import create_foundation_model
    cfg = {"model": {"backbone": "efficientnet_b3", "pretrained": False, "dropout": 0.2, "use_attention": True},
           "classification_datasets": [{"name": "cls", "classes": ["a", "b"]}],
           "segmentation_datasets": [{"name": "seg", "num_classes": 2}]}
    torch.manual_seed(0)
    model = create_foundation_model(cfg).cuda().train()
    assert model.segmentation_heads == {"seg": 2} and "seg" in model.segmentation_layers
    assert sum(p.numel() for p in model.segmentation_layers["seg"].parameters()) == 2_274_021
    x = torch.randn(2, 3, 72, 72, device="cuda")
    out = model(x, "segmentation", "seg")
    assert out.shape == (2, 2, 72, 72) and out.dtype == torch.float32
    from mia_amd.src.loss import get_loss_function
    target = torch.randint(0, 2, (2, 72, 72), device="cuda")
    loss = get_loss_function("segmentation", num_classes=2)(out, target)
    loss.backward()
    assert torch.isfinite(loss)
    for k, p in model.named_parameters():
        if k.startswith("backbone") or k.startswith("segmentation_layers"):
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
    with pytest.raises(ValueError):
        model(x, "segmentation", "missing")
    model.add_segmentation_head("extra", 3)
    assert model(x, "segmentation", "extra").shape == (2, 3, 72, 72)
    model.eval()
    with torch.no_grad():
        assert model(x, "segmentation", "seg").shape == (2, 2, 72, 72)


@pytest.mark.parametrize("use_attention", [True, False])
def test_segmentation_step_matches_oracle_fp32(use_attention):
    """Whole model, segmentation task: trunk + decoder + combined loss, one forward/backward at 96x96, batch 4,
    train-mode BatchNorm, dropout 0, against the CPU oracle with the same seeded state dict."""
    from oracle import effnet_b3 as O
    from oracle import seg_ref as S
    from seeding import seeded_state_dict
    from mia_amd.src.loss import get_loss_function
    from mia_amd.src.model import FoundationModel, load_model_weights_safely
    ref = O.OracleFoundationModel({}, dropout=0.0, use_attention=use_attention, segmentation_heads={"seg": 3})
    sd = seeded_state_dict(ref, 5)
    ref.load_state_dict(sd)
    mine = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.0, classification_heads={},
                           segmentation_heads={"seg": 3}, use_attention=use_attention)
    missing, unexpected = load_model_weights_safely(mine, {"model_state_dict": sd})
    assert not missing and not unexpected
    mine = mine.cuda().train()
    ref.train()
    x = randn(200, 4, 3, 96, 96)
    t = torch.from_numpy(np.random.default_rng(9).integers(0, 3, (4, 96, 96)))
    with relu_ties.recording(S) as rec:
        lr = ref(x, "segmentation", "seg")
        loss_r = S.combined_segmentation_loss(lr, t, 3)
        loss_r.backward(retain_graph=True)
        flips = relu_ties.responses(rec, ref)
    lm = mine(x.cuda(), "segmentation", "seg")
    loss_m = get_loss_function("segmentation", num_classes=3)(lm, t.cuda())
    loss_m.backward()
    assert rel(lm, lr) < 1e-4
    assert abs(loss_m.item() - loss_r.item()) < 1e-5 * max(1.0, abs(loss_r.item()))
    pr = dict(ref.named_parameters())
    diffs = {}
    for n, p in mine.named_parameters():
        assert p.grad is not None, n
        a, b = p.grad.detach().double().cpu(), pr[n].grad.detach().double()
        diffs[n] = (a - b, b.abs().max().item())
    for n, (d, mag) in list(diffs.items()):
        # parameters whose true gradient is exactly zero (a shift removed by the next train-mode BatchNorm) hold
        # rounding noise on both sides: compare them against the gradient scale of the same layer's weight
        if n.endswith(".bias") and ((".bn" in n and "backbone" in n) or "decoder_blocks" in n and ".conv" in n):
            mag = max(mag, diffs[n[:-4] + "weight"][1])
        diffs[n] = (d, max(mag, 1e-30))
    # equal up to the side taken at ReLU inputs within forward rounding of zero (see relu_ties): the fitted
    # coefficient of every such element must be a clean 0 (same side as the oracle) or 1 (the other side)
    coef, res = relu_ties.explain(diffs, flips)
    assert all(min(abs(c), abs(c - 1.0)) < 0.05 for c in coef), coef
    worst = max((e, n) for n, e in res.items())
    assert worst[0] < 1e-3, (worst, coef)


@pytest.mark.parametrize("c,h,w", [(16, 48, 48), (32, 75, 75), (256, 19, 19), (64, 40, 33), (16, 150, 150)])
@pytest.mark.parametrize("train", [True, False])
def test_cbam_large_map_matches_oracle(c, h, w, train):
    """Decoder-level CBAM shapes take the row-split reduction kernels; the oracle's CBAM (itself pinned to the
    reference's by the small goldens) is the checker. ReLU-like inputs (many exact zeros and ties) included. The
    150x150 case (45,000 pixels in the batch) also takes the two-launch BatchNorm(1) backward and the chunked 7x7
    weight gradient."""
    from oracle import effnet_b3 as O
    from mia_amd.heads import CBAM
    ref = load_seeded(O.CBAM(c), 81).train(train)
    mine = load_seeded(CBAM(c), 81).cuda().train(train)
    x = torch.relu(randn(82, 2, c, h, w))
    xr = x.clone().requires_grad_(True)
    xm = nhwc(x).cuda().requires_grad_(True)
    out_r, out_m = ref(xr), mine(xm)
    assert rel(out_m.permute(0, 3, 1, 2), out_r) < 1e-4
    g = randn(83, 2, c, h, w)
    out_r.backward(g)
    out_m.backward(nhwc(g).cuda())
    assert rel(xm.grad.permute(0, 3, 1, 2), xr.grad) < 1e-3
    pr = dict(ref.named_parameters())
    for k, p in mine.named_parameters():
        assert rel(p.grad, pr[k].grad) < 1e-3, k


def test_decoder_block_eval_mode_gradients_match_oracle():
    """Frozen BatchNorm (eval mode, e.g. fine-tuning with the decoder's statistics fixed): here the conv biases DO
    receive a gradient; every parameter and the input gradient are checked against the oracle block."""
    from oracle import seg_ref as S
    ref = load_seeded(S.DecoderBlock(24, 16, dropout=0.0), 51).eval()
    mine = load_seeded(DecoderBlock(24, 16, dropout=0.0), 51).cuda().eval()
    x = randn(52, 2, 24, 6, 7)
    xr, xm = x.clone().requires_grad_(True), nhwc(x).cuda().requires_grad_(True)
    out_r, out_m = ref(xr), mine(xm)
    assert rel(out_m.permute(0, 3, 1, 2), out_r) < 1e-4
    g = randn(53, *out_r.shape)
    out_r.backward(g)
    out_m.backward(nhwc(g).cuda())
    assert rel(xm.grad.permute(0, 3, 1, 2), xr.grad) < 1e-3
    pr = dict(ref.named_parameters())
    for k, p in mine.named_parameters():
        assert rel(p.grad, pr[k].grad) < 1e-3, k


def test_segmentation_full_size_matches_oracle_fp32():
    """300x300 input (the size of the headline configuration), batch 2, eval mode: whole segmentation forward against
    the CPU oracle. Every decoder level runs at its real spatial size (10, 19, 38, 75, 150, then 300)."""
    from oracle import effnet_b3 as O
    from seeding import seeded_state_dict
    from mia_amd.src.model import FoundationModel, load_model_weights_safely
    ref = O.OracleFoundationModel({}, dropout=0.0, use_attention=True, segmentation_heads={"seg": 2}).eval()
    sd = seeded_state_dict(ref, 6)
    ref.load_state_dict(sd)
    mine = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.0, classification_heads={},
                           segmentation_heads={"seg": 2}, use_attention=True)
    load_model_weights_safely(mine, sd)
    mine = mine.cuda().eval()
    x = randn(201, 2, 3, 300, 300)
    with torch.no_grad():
        lr = ref(x, "segmentation", "seg")
        lm = mine(x.cuda(), "segmentation", "seg")
    assert lm.shape == (2, 2, 300, 300)
    assert rel(lm, lr) < 1e-4
    assert (lm.argmax(1).cpu() == lr.argmax(1)).float().mean().item() > 0.999


@pytest.mark.parametrize("tag", ["cls", "pix"])
def test_focal_and_weighted_ce_match_reference(tag):
    """FocalLoss and the class-weighted cross-entropy of src/loss.py against values the reference's own classes gave
    (tests/golden/loss_variants.npz) on class logits [B,n] and pixel logits [B,n,H,W]."""
    from mia_amd.src.loss import FocalLoss, WeightedCrossEntropyLoss, get_loss_function
    gold = np.load(os.path.join(G, "loss_variants.npz"))
    logits, target = torch.from_numpy(gold[f"{tag}.logits"]).cuda(), torch.from_numpy(gold[f"{tag}.target"]).cuda()
    weights = torch.from_numpy(gold[f"{tag}.weights"]).cuda()
    wce = get_loss_function("classification", class_weights=weights)
    assert isinstance(wce, WeightedCrossEntropyLoss)
    assert isinstance(get_loss_function("segmentation", num_classes=3, loss_type="focal"), FocalLoss)
    cases = {"focal": FocalLoss(), "focal_a05_g3_sum": FocalLoss(alpha=0.5, gamma=3.0, reduction="sum"), "wce": wce}
    for name, crit in cases.items():
        z = logits.clone().requires_grad_(True)
        loss = crit(z, target)
        loss.backward()
        ref = float(gold[f"{tag}.{name}.loss"])
        assert abs(loss.item() - ref) <= 1e-5 * max(1.0, abs(ref)), name
        assert rel(z.grad, gold[f"{tag}.{name}.dlogits"]) < 1e-4, name
    z = logits.clone().requires_grad_(True)
    per = FocalLoss(reduction="none")(z, target)
    assert per.shape == target.shape and rel(per, gold[f"{tag}.focal_none"]) < 1e-5
    up = randn(91, *per.shape).cuda()
    (per * up).sum().backward()
    z2 = logits.clone().cpu().requires_grad_(True)
    ce = F.cross_entropy(z2, target.cpu(), reduction="none")
    ((1 - torch.exp(-ce)) ** 2 * ce * up.cpu()).sum().backward()
    assert rel(z.grad, z2.grad) < 1e-4
