"""CPU tests: the oracle restatements against the golden vectors produced by the
reference's own classes (tests/golden/make_golden.py), plus the structural pins."""
import os

import numpy as np
import pytest
import torch

from seeding import load_seeded, randn
from oracle import effnet_b3 as O
from oracle import preprocess_ref as P

G = os.path.join(os.path.dirname(__file__), "golden")


def digest(t):
    f = t.detach().double().flatten()
    head = f[:16].numpy()
    if head.size < 16:
        head = np.pad(head, (0, 16 - head.size))
    return np.concatenate([[f.sum().item(), f.norm().item()], head])


def check_module(mod, x, gold, train, rtol=1e-5, atol=1e-6):
    mod.train(train)
    x = x.clone().requires_grad_(True)
    out = mod(x)
    np.testing.assert_allclose(out.detach().numpy(), gold["out"], rtol=rtol, atol=atol)
    if not train:
        return
    out.backward(randn(777, *out.shape))
    np.testing.assert_allclose(x.grad.numpy(), gold["dx"], rtol=1e-4, atol=1e-6)
    for k, p in mod.named_parameters():
        g = gold["grad." + k]
        if g.shape == tuple(p.shape):
            np.testing.assert_allclose(p.grad.numpy(), g, rtol=1e-4, atol=1e-6, err_msg=k)
        else:
            np.testing.assert_allclose(digest(p.grad), g, rtol=1e-4, atol=1e-5, err_msg=k)
    for k, b in mod.named_buffers():
        if "running" in k:
            np.testing.assert_allclose(b.numpy(), gold["buf." + k], rtol=1e-5, atol=1e-7, err_msg=k)


@pytest.mark.parametrize("c,hw,tag", [(384, 10, "c384_10x10"), (64, 2, "c64_2x2"), (64, 5, "c64_5x5")])
@pytest.mark.parametrize("train", [True, False])
def test_cbam_matches_reference_golden(c, hw, tag, train):
    gold = np.load(os.path.join(G, f"cbam_{tag}_{'train' if train else 'eval'}.npz"))
    check_module(load_seeded(O.CBAM(c), 11), randn(12, 2, c, hw, hw), gold, train)


@pytest.mark.parametrize("c,hw,n,tag", [(384, 10, 4, "c384_10x10"), (384, 2, 4, "c384_2x2"), (64, 3, 2, "c64_3x3")])
@pytest.mark.parametrize("train", [True, False])
def test_mha_head_matches_reference_golden(c, hw, n, tag, train):
    gold = np.load(os.path.join(G, f"mha_{tag}_{'train' if train else 'eval'}.npz"))
    m = load_seeded(O.MultiHeadAttentionClassifier(c, 8, n, dropout=0.0), 21)
    check_module(m, randn(22, 2, c, hw, hw), gold, train)


def test_simple_head_matches_reference_golden():
    gold = np.load(os.path.join(G, "simple_c384_10x10_train.npz"))
    m = load_seeded(O.simple_head(384, 4, 0.0), 31)
    check_module(m, randn(32, 2, 384, 10, 10), gold, True)


def test_channel_pool_order_is_max_then_mean():
    x = randn(41, 2, 16, 3, 3)
    gold = np.load(os.path.join(G, "channelpool.npz"))["out"]
    mine = torch.cat((x.amax(1, keepdim=True), x.mean(1, keepdim=True)), 1).numpy()
    np.testing.assert_allclose(mine, gold, rtol=1e-6, atol=1e-7)


def test_param_count_known_answers():
    # training.log:5 of the reference: "Model created with 23903135 parameters"
    # = trunk + 3 attention heads (4,4,2 classes) + 5 CBAM-UNet decoders (2,274,021 each, golden file).
    counts = dict(l.strip().rsplit("=", 1) for l in open(os.path.join(G, "param_counts.txt")))
    trunk = sum(p.numel() for p in O.EfficientNetB3Features().parameters())
    assert trunk == 10_103_336
    heads = 2 * int(counts["MHA(384,8,4)"]) + int(counts["MHA(384,8,2)"])
    dec = 5 * int(counts["EnhancedUNetDecoder(nc=2)"])
    assert trunk + heads + dec == 23_903_135
    assert sum(p.numel() for p in O.MultiHeadAttentionClassifier(384, 8, 4).parameters()) == int(counts["MHA(384,8,4)"])
    assert sum(p.numel() for p in O.CBAM(384).parameters()) == int(counts["CBAM(384)"])


def test_trunk_feature_shapes():
    m = O.EfficientNetB3Features().eval()
    with torch.no_grad():
        f = m(torch.zeros(1, 3, 64, 64))
    assert [t.shape[1] for t in f] == [24, 32, 48, 136, 384]
    assert [t.shape[2] for t in f] == [32, 16, 8, 4, 2]
    # 300x300: 150/75/38/19/10 (odd sizes 75 and 19)
    hs = [300]
    for s in (2, 2, 2, 2, 2):
        hs.append((hs[-1] - 1) // s + 1)
    assert hs[1:] == [150, 75, 38, 19, 10]


def test_trunk_state_dict_key_names():
    keys = set(O.EfficientNetB3Features().state_dict().keys())
    for k in ("conv_stem.weight", "bn1.running_var", "blocks.0.0.conv_dw.weight", "blocks.0.0.se.conv_reduce.bias",
              "blocks.0.0.conv_pw.weight", "blocks.0.0.bn2.weight", "blocks.1.0.conv_pw.weight",
              "blocks.1.0.conv_pwl.weight", "blocks.1.0.bn3.num_batches_tracked", "blocks.6.1.se.conv_expand.weight"):
        assert k in keys, k
    assert not any(k.startswith(("conv_head", "bn2.", "classifier")) for k in keys)


# ----------------------------- preprocessing ------------------------------
def test_resize_identity_and_bounds():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    np.testing.assert_array_equal(P.resize_u8_linear(img, 37, 53), img)
    flat = np.full((16, 16, 3), 200, np.uint8)
    assert (P.resize_u8_linear(flat, 45, 23) == 200).all()


@pytest.mark.parametrize("shape,out", [((256, 256, 3), (300, 300)), ((128, 128, 3), (64, 64)), ((97, 131, 3), (300, 300))])
def test_resize_close_to_float_bilinear(shape, out):
    # cross-check of the sampling grid: torch's float bilinear (align_corners=False, no antialias)
    # uses the same half-pixel centres; the fixed-point path may differ by 1 LSB.
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    mine = P.resize_u8_linear(img, *out).astype(np.int32)
    t = torch.from_numpy(img).permute(2, 0, 1)[None].float()
    ref = torch.nn.functional.interpolate(t, size=out, mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    assert np.abs(mine - ref.numpy()).max() <= 1.0 + 1e-3


def test_normalize_formula():
    img = np.arange(2 * 2 * 3, dtype=np.uint8).reshape(2, 2, 3) * 20
    out = P.normalize_chw(img)
    exp = (img.astype(np.float64) / 255.0 - np.array(P.IMAGENET_MEAN)) / np.array(P.IMAGENET_STD)
    np.testing.assert_allclose(out, exp.transpose(2, 0, 1), rtol=1e-5, atol=1e-6)
    assert out.dtype == np.float32 and out.shape == (3, 2, 2)


def test_zero_slice_filter_spec():
    rng = np.random.default_rng(7)
    vol = np.zeros((12, 16, 16), np.uint16)
    vol[3, 0, 0] = 1
    vol[5, 15, 15] = 7
    vol[6, 0, 15] = 65535
    vol[9] = rng.integers(1, 100, (16, 16))
    keep, idx = P.zero_slice_filter(vol)
    assert keep.tolist() == [i in (3, 5, 6, 9) for i in range(12)]
    assert idx.dtype == np.int32 and idx.tolist() == [3, 5, 6, 9]
    k, i = P.zero_slice_filter(np.zeros((4, 8, 8), np.float32))
    assert not k.any() and i.size == 0
    k, i = P.zero_slice_filter(np.ones((4, 8, 8), np.uint8))
    assert k.all() and i.tolist() == [0, 1, 2, 3]
    f = np.zeros((2, 4, 4), np.float32)
    f[1, 2, 2] = -0.0
    assert not P.zero_slice_filter(f)[0].any()   # -0.0 == 0
    f[0, 0, 0] = np.nan
    assert P.zero_slice_filter(f)[0].tolist() == [True, False]   # NaN != 0
