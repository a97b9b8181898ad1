"""GPU parity of the head kernels against golden vectors produced by the REFERENCE's own classes
(tests/golden/make_golden.py ran CBAM / MultiHeadAttentionClassifier / the simple head of
/root/reference/foundation_model/src/model.py on CPU). fp32: outputs within 1e-4, gradients within 1e-3,
relative to each tensor's largest magnitude."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mia_amd  # noqa: E402,F401
from mia_amd import head_ops  # noqa: E402
from mia_amd.heads import CBAM, MultiHeadAttentionClassifier, SimpleHead  # noqa: E402
from seeding import load_seeded, randn  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def digest(t):
    f = t.detach().double().flatten().cpu()
    head = f[:16].numpy()
    if head.size < 16:
        head = np.pad(head, (0, 16 - head.size))
    return np.concatenate([[f.sum().item(), f.norm().item()], head])


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def check(mod, x_nchw, gold, train, out_is_map, zero_grad_params=()):
    mod = mod.cuda().train(train)
    x = nhwc(x_nchw).cuda().requires_grad_(True)
    out = mod(x)
    out_cmp = out.permute(0, 3, 1, 2) if out_is_map else out
    assert rel(out_cmp, gold["out"]) < 1e-4
    if not train:
        return
    g = randn(777, *gold["out"].shape)
    out.backward((nhwc(g) if out_is_map else g).cuda())
    assert rel(x.grad.permute(0, 3, 1, 2), gold["dx"]) < 1e-3
    for k, p in mod.named_parameters():
        ref = gold["grad." + k]
        assert p.grad is not None, k
        if any(z in k for z in zero_grad_params):
            # the true gradient is exactly zero (softmax is invariant to a key bias): both sides hold noise
            assert p.grad.abs().max().item() < 1e-5, k
            continue
        if ref.shape == tuple(p.shape):
            assert rel(p.grad, ref) < 1e-3, k
        else:
            d = digest(p.grad)
            assert abs(d[0] - ref[0]) <= 1e-3 * max(abs(ref[1]), 1e-12) + 1e-6, k      # sum vs l2 scale
            assert abs(d[1] - ref[1]) <= 1e-3 * abs(ref[1]) + 1e-9, k                   # l2 norm
            assert np.abs(d[2:] - ref[2:]).max() <= 1e-3 * max(np.abs(ref[2:]).max(), abs(ref[1]) * 1e-3), k
    for k, b in mod.named_buffers():
        if "running" in k:
            assert rel(b, gold["buf." + k]) < 1e-5, k


@pytest.mark.parametrize("c,hw,tag", [(384, 10, "c384_10x10"), (64, 2, "c64_2x2"), (64, 5, "c64_5x5")])
@pytest.mark.parametrize("train", [True, False])
def test_cbam_matches_reference(c, hw, tag, train):
    gold = np.load(os.path.join(G, f"cbam_{tag}_{'train' if train else 'eval'}.npz"))
    check(load_seeded(CBAM(c), 11), randn(12, 2, c, hw, hw), gold, train, True)


@pytest.mark.parametrize("c,hw,n,tag", [(384, 10, 4, "c384_10x10"), (384, 2, 4, "c384_2x2"), (64, 3, 2, "c64_3x3")])
@pytest.mark.parametrize("train", [True, False])
def test_mha_head_matches_reference(c, hw, n, tag, train):
    gold = np.load(os.path.join(G, f"mha_{tag}_{'train' if train else 'eval'}.npz"))
    m = load_seeded(MultiHeadAttentionClassifier(c, 8, n, dropout=0.0), 21)
    check(m, randn(22, 2, c, hw, hw), gold, train, False, zero_grad_params=("key.bias",))


def test_simple_head_matches_reference():
    gold = np.load(os.path.join(G, "simple_c384_10x10_train.npz"))
    check(load_seeded(SimpleHead(384, 4, 0.0), 31), randn(32, 2, 384, 10, 10), gold, True, False)


def test_cross_entropy_matches_torch():
    logits = randn(1, 37, 4).cuda().requires_grad_(True)
    labels = torch.from_numpy(np.random.default_rng(2).integers(0, 4, 37)).cuda()
    loss = head_ops.CrossEntropyLoss()(logits, labels)
    loss.backward()
    ref_l = logits.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_l, labels)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
    assert rel(logits.grad, ref_l.grad) < 1e-5


def test_cross_entropy_ignored_and_invalid_labels():
    """nn.CrossEntropyLoss semantics for labels that are not classes: -100 rows leave the mean (sum and divisor) and get
    a zero gradient; a label >= n (a device assert in torch) is never used as an index: zero gradient row, NaN loss."""
    logits = randn(4, 300, 4).cuda().requires_grad_(True)
    lab = np.random.default_rng(5).integers(0, 4, 300)
    lab[[0, 7, 130, 299]] = -100
    labels = torch.from_numpy(lab).cuda()
    loss = head_ops.CrossEntropyLoss()(logits, labels)
    loss.backward()
    ref_l = logits.detach().cpu().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_l, labels.cpu())
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
    assert rel(logits.grad.cpu(), ref_l.grad) < 1e-5
    assert float(logits.grad[[0, 7, 130, 299]].abs().sum()) == 0.0
    bad = labels.clone()
    bad[7], bad[9] = 4, -3
    l2 = logits.detach().clone().requires_grad_(True)
    loss2 = head_ops.CrossEntropyLoss()(l2, bad)
    loss2.backward()
    assert torch.isnan(loss2).item()
    assert float(l2.grad[[7, 9]].abs().sum()) == 0.0 and torch.isfinite(l2.grad).all()
    allign = torch.full((5,), -100, dtype=torch.int64, device="cuda")
    assert torch.isnan(head_ops.CrossEntropyLoss()(randn(6, 5, 4).cuda(), allign)).item()   # 0/0, as in torch


def test_dropout_statistics_and_determinism():
    """Train-mode dropout (p=0.2): kept fraction ~0.8, kept values scaled by 1/0.8, same mask in backward."""
    x = torch.ones(1 << 16, device="cuda", requires_grad=True)
    y = head_ops.DropoutFn.apply(x, 0.2, 12345)
    kept = (y > 0).float().mean().item()
    assert abs(kept - 0.8) < 0.01
    assert torch.allclose(y[y > 0], torch.full_like(y[y > 0], 1.25))
    y.sum().backward()
    assert torch.equal(x.grad, y.detach())
    y2 = head_ops.DropoutFn.apply(x, 0.2, 12346)
    assert not torch.equal(y2, y)


def test_mha_head_bf16_close_to_fp32():
    m = load_seeded(MultiHeadAttentionClassifier(384, 8, 4, dropout=0.0), 21).cuda().eval()
    x = nhwc(randn(22, 4, 384, 10, 10)).cuda()
    with torch.no_grad():
        a = m(x)
        b = m(x.bfloat16())
    assert rel(b, a) < 3e-2


@pytest.mark.parametrize("heads", [3, 4, 16])
def test_mha_head_other_head_counts_match_oracle(heads):
    """num_attention_heads other than the default 8 (config.yaml:43; the reference accepts any divisor of the
    feature width, model.py:119): head widths 128, 96 and 24 against the CPU restatement of model.py:112-175."""
    from oracle import effnet_b3 as O
    ref = O.MultiHeadAttentionClassifier(384, heads, 4, 0.0)
    mine = MultiHeadAttentionClassifier(384, heads, 4, 0.0)
    from seeding import seeded_state_dict
    sd = seeded_state_dict(ref, 41)
    ref.load_state_dict(sd)
    mine.load_state_dict(sd)
    ref.train()
    mine = mine.cuda().train()
    x = randn(91, 3, 384, 5, 5)
    xr = x.clone().requires_grad_(True)
    out_r = ref(xr)
    xm = nhwc(x).cuda().requires_grad_(True)
    out_m = mine(xm)
    assert rel(out_m, out_r) < 1e-4
    g = randn(92, *out_r.shape)
    out_r.backward(g)
    out_m.backward(g.cuda())
    assert rel(xm.grad.permute(0, 3, 1, 2), xr.grad) < 1e-3
    pr = dict(ref.named_parameters())
    for k, p in mine.named_parameters():
        if k.endswith("key_projection.bias") or k.endswith("key.bias"):
            continue      # exactly zero in exact arithmetic
        assert rel(p.grad, pr[k].grad) < 1e-3, k


@pytest.mark.parametrize("tokens,heads,hd,pdrop", [(100, 8, 48, 0.0), (100, 8, 48, 0.1), (25, 4, 16, 0.1), (9, 2, 32, 0.0),
                                                   (128, 2, 64, 0.1), (49, 3, 48, 0.0), (16, 2, 16, 0.1), (4, 8, 48, 0.1), (1, 8, 48, 0.0)])
def test_attention_matrix_core_path_matches_fp32_kernels(tokens, heads, hd, pdrop):
    """bf16 attention with up to 128 tokens and head widths 16/32/48/64 runs on the matrix cores; fp32 runs the
    per-thread kernels. Same bf16-rounded q, k, v and output gradient into both, same dropout seed (the masks are a
    function of seed and element index, so they coincide): the results differ only by the bf16 rounding of P and dS
    as MFMA operands and of the outputs. Stated tolerance: 1 % of each tensor's largest magnitude (bf16 has 8 bits)."""
    B, C = 3, heads * hd
    qkv = (randn(51, B * tokens, 3 * C) * 0.7).bfloat16().cuda()
    g = randn(52, B * tokens, C).bfloat16().cuda()
    res = {}
    for dt in (torch.bfloat16, torch.float32):
        x = qkv.detach().to(dt).clone().requires_grad_(True)
        out = head_ops.AttentionFn.apply(x, B, tokens, C, heads, pdrop, 777)
        out.backward(g.to(dt))
        res[dt] = (out.detach().float(), x.grad.float())
    assert rel(res[torch.bfloat16][0], res[torch.float32][0]) < 1e-2
    for i, name in enumerate(("dq", "dk", "dv")):
        a = res[torch.bfloat16][1][:, i * C:(i + 1) * C]
        b = res[torch.float32][1][:, i * C:(i + 1) * C]
        # with one token dq and dk are exactly zero in exact arithmetic (softmax of one score): both sides hold rounding
        # noise there, so the yardstick is never smaller than a thousandth of the incoming gradient
        scale = max(b.abs().max().item(), 1e-3 * g.float().abs().max().item())
        assert (a - b).abs().max().item() < 1e-2 * scale, name
