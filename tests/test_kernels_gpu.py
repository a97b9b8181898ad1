"""GPU parity tests of the individual gfx950 kernels, called through the C ABI (ctypes), against plain
PyTorch fp32 references of the same op. Integer/byte kernels must be bit-exact; floating-point kernels
are held to the tolerances written in each test (relative to the reference tensor's largest magnitude)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import mia_amd  # noqa: E402,F401
from mia_amd import ops  # noqa: E402

DEV = "cuda"


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def tol(dtype, f32=2e-5, bf16=2e-2):
    return f32 if dtype == torch.float32 else bf16


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def test_library_targets_this_device():
    mia_amd.lib().mia_check_device()


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(300, 24, 40), (1000, 144, 24), (777, 32, 192), (513, 48, 288), (260, 136, 816),
                                   (200, 384, 2304), (129, 232, 136), (64, 512, 384), (130, 4, 512)])
def test_gemm_nt_plain(dtype, M, N, K):
    a, b = rnd(M, K, seed=1, dtype=dtype), rnd(N, K, seed=2, scale=K ** -0.5, dtype=dtype)
    out, _ = ops.gemm_nt(a, b)
    ref = a.float() @ b.float().t()
    assert relerr(out.float(), ref) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,hw", [(2 * 361, 136, 816, 361), (3 * 100, 32, 192, 100), (4 * 49, 384, 2304, 49),
                                      # maps of 1024 pixels and more take the prologue form that keeps one gate vector
                                      # per tile: tiles that lie in one image, tiles that straddle two (1500 and 1100
                                      # are no multiples of the 128-row tile), several k stages (K = 192, 288)
                                      (3 * 1500, 24, 40, 1500), (4 * 1100, 32, 192, 1100), (2 * 1444, 48, 288, 1444),
                                      (5 * 1024, 24, 24, 1024)])
def test_gemm_nt_prologue_gate_stats(dtype, M, N, K, hw):
    a, b = rnd(M, K, seed=1, dtype=dtype), rnd(N, K, seed=2, scale=K ** -0.5, dtype=dtype)
    sc, sh = rnd(K, seed=3).abs() + 0.5, rnd(K, seed=4, scale=0.3)
    gate = torch.sigmoid(rnd(M // hw, K, seed=5))
    rows = 7
    out, part = ops.gemm_nt(a, b, pscale=sc, pshift=sh, pro_act=ops.ACT_SILU, gate=gate, rows_per_img=hw,
                            stats_rows=rows)
    ap = F.silu(a.float() * sc + sh) * gate.repeat_interleave(hw, 0)
    if dtype == torch.bfloat16:
        ap = ap.bfloat16().float()
    ref = ap @ b.float().t()
    assert relerr(out.float(), ref) < tol(dtype)
    s = part.sum(0)
    assert relerr(s[0], ref.sum(0)) < tol(dtype, 1e-4)
    assert relerr(s[1], (ref * ref).sum(0)) < tol(dtype, 1e-4)


@pytest.mark.parametrize("M,N,K", [(50001, 816, 136), (40003, 1392, 232), (70001, 200, 64), (33001, 576, 96)])
def test_gemm_nt_plain_tall(M, N, K):
    """Data-gradient shaped launches (no statistics, no bias) tall enough for the 128-row tiles, with ragged last
    row and column tiles, with and without the residual operand."""
    a, b = rnd(M, K, seed=3, dtype=torch.bfloat16), rnd(N, K, seed=4, scale=K ** -0.5, dtype=torch.bfloat16)
    d = rnd(M, N, seed=5, dtype=torch.bfloat16)
    ref = a.float() @ b.float().t()
    out, _ = ops.gemm_nt(a, b)
    assert relerr(out.float(), ref) < tol(torch.bfloat16)
    out, _ = ops.gemm_nt(a, b, add=d)
    assert relerr(out.float(), ref + d.float()) < tol(torch.bfloat16)


# The streaming kernel of the large-map projections (400 000 rows and more, at most 192 -> 32 channels): the shapes of
# EfficientNet-B3's first two stages at a batch that reaches the row threshold; images whose row count is no multiple
# of the 16- or 32-row blocks, a row count that is not a multiple of the image size of the last case's ranges.
STREAM_PRO = [(20, 22500, 24, 40), (20, 22500, 24, 24), (72, 5625, 32, 144), (72, 5625, 32, 192), (400, 1024, 24, 40),
              (401, 1027, 28, 160)]


@pytest.mark.parametrize("B,hw,N,K", STREAM_PRO)
def test_gemm_nt_streaming_prologue_gate_stats(B, hw, N, K):
    M = B * hw
    a, b = rnd(M, K, seed=1, dtype=torch.bfloat16), rnd(N, K, seed=2, scale=K ** -0.5, dtype=torch.bfloat16)
    sc, sh = rnd(K, seed=3).abs() + 0.5, rnd(K, seed=4, scale=0.3)
    gate = torch.sigmoid(rnd(B, K, seed=5))
    out, part = ops.gemm_nt(a, b, pscale=sc, pshift=sh, pro_act=ops.ACT_SILU, gate=gate, rows_per_img=hw,
                            stats_rows=512)
    ap = F.silu(a.float() * sc + sh) * gate.repeat_interleave(hw, 0)
    ref = ap @ b.float().t()
    assert relerr(out.float(), ref) < 2e-2
    s = part.sum(0)
    assert relerr(s[0], ref.sum(0)) < 2e-3
    assert relerr(s[1], (ref * ref).sum(0)) < 2e-3
    # every image gets ITS gate: a wrong image index at a block boundary shows as a per-image error
    err = (out.float() - ref).view(B, hw, N).abs().amax((1, 2)) / ref.abs().max()
    assert err.max().item() < 2e-2
    # without a gate (one "image" of M rows); repeated launches agree bit for bit (fixed summation order)
    outs = [ops.gemm_nt(a, b, pscale=sc, pshift=sh, pro_act=ops.ACT_SILU, stats_rows=384) for _ in range(3)]
    ref2 = F.silu(a.float() * sc + sh) @ b.float().t()
    assert relerr(outs[0][0].float(), ref2) < 2e-2
    assert relerr(outs[0][1].sum(0)[1], (ref2 * ref2).sum(0)) < 2e-3
    assert all(torch.equal(o[1], outs[0][1]) and torch.equal(o[0], outs[0][0]) for o in outs[1:])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_nt_bias_relu_add(dtype):
    M, N, K = 300, 512, 384
    a, b = rnd(M, K, seed=1, dtype=dtype), rnd(N, K, seed=2, scale=K ** -0.5, dtype=dtype)
    bias, add = rnd(N, seed=3), rnd(M, N, seed=4, dtype=dtype)
    out, _ = ops.gemm_nt(a, b, bias=bias, epi_act=ops.ACT_RELU, add=add)
    ref = F.relu(a.float() @ b.float().t() + bias)
    if dtype == torch.bfloat16:
        ref = ref.bfloat16().float()
    ref = ref + add.float()
    assert relerr(out.float(), ref) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(3000, 144, 24), (1000, 32, 192), (500, 136, 816), (300, 2304, 384),
                                   (200, 384, 2304), (128, 512, 384), (5000, 40, 32)])
def test_gemm_tn(dtype, M, N, K):
    g, x = rnd(M, N, seed=1, dtype=dtype), rnd(M, K, seed=2, dtype=dtype)
    out = ops.gemm_tn(g, x)
    ref = g.float().t() @ x.float()
    assert relerr(out, ref) < tol(dtype, 2e-5, 1e-2)


@pytest.mark.parametrize("M,N,K", [(400000, 144, 24), (60000, 816, 136), (12800, 1392, 232), (90000, 40, 32)])
def test_gemm_tn_split_reduction_is_reproducible(M, N, K):
    """Long reductions are split over many workgroups: the partial tiles are added in a fixed order by a second
    kernel, so repeated launches agree bit for bit."""
    g, x = rnd(M, N, seed=1, dtype=torch.bfloat16), rnd(M, K, seed=2, dtype=torch.bfloat16)
    outs = [ops.gemm_tn(g, x) for _ in range(4)]
    ref = (g.double().t() @ x.double()).float()
    assert relerr(outs[0], ref) < 2e-3
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_tn_prologue_gate(dtype):
    hw, B, N, K = 100, 5, 48, 288
    M = B * hw
    g, x = rnd(M, N, seed=1, dtype=dtype), rnd(M, K, seed=2, dtype=dtype)
    sc, sh = rnd(K, seed=3).abs() + 0.5, rnd(K, seed=4, scale=0.3)
    gate = torch.sigmoid(rnd(B, K, seed=5))
    out = ops.gemm_tn(g, x, pscale=sc, pshift=sh, pro_act=ops.ACT_SILU, gate=gate, rows_per_img=hw)
    xp = F.silu(x.float() * sc + sh) * gate.repeat_interleave(hw, 0)
    if dtype == torch.bfloat16:
        xp = xp.bfloat16().float()
    ref = g.float().t() @ xp
    assert relerr(out, ref) < tol(dtype, 2e-5, 1e-2)


# ------------------------------------------------------------------ depthwise
DW_CASES = [(3, 1, 2, 19, 19, 96), (3, 2, 2, 38, 38, 288), (5, 1, 2, 19, 19, 136), (5, 2, 2, 19, 19, 816),
            (5, 2, 1, 75, 75, 192), (3, 1, 1, 150, 150, 40), (3, 2, 1, 150, 150, 144), (3, 1, 3, 10, 10, 2304),
            (5, 1, 3, 10, 10, 1392), (3, 1, 2, 2, 2, 24), (5, 2, 2, 4, 4, 48),
            # bf16 shapes of the matrix-core kernel: several tiles per image, 32- and 64-channel blocks, channel tails
            # of 8, stride 2 onto a 38x38 map
            (5, 1, 2, 38, 38, 288), (3, 1, 2, 38, 38, 72), (5, 1, 1, 19, 19, 816), (3, 2, 1, 75, 75, 144),
            (3, 1, 2, 23, 17, 200), (5, 1, 2, 24, 24, 128), (5, 1, 1, 36, 36, 64)]


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,s,B,H,W,C", DW_CASES)
def test_dwconv_fwd(dtype, k, s, B, H, W, C):
    x = rnd(B, H, W, C, seed=1, dtype=dtype)
    w = rnd(C, 1, k, k, seed=2, scale=1.0 / k)
    sc, sh = rnd(C, seed=3).abs() + 0.5, rnd(C, seed=4, scale=0.3)
    rows = 5
    y, part = ops.dwconv_fwd(x, w.reshape(C, k * k), k, s, pscale=sc, pshift=sh, stats_rows=rows)
    a = F.silu(x.float() * sc + sh)
    if dtype == torch.bfloat16:
        a = a.bfloat16().float()
    ref = nhwc(F.conv2d(nchw(a), w, None, s, (k - 1) // 2, 1, C))
    assert y.shape == ref.shape
    assert relerr(y.float(), ref) < tol(dtype)
    st = part.sum(0)
    assert relerr(st[0], ref.sum((0, 1, 2))) < tol(dtype, 1e-4)
    assert relerr(st[1], (ref * ref).sum((0, 1, 2))) < tol(dtype, 1e-4)
    # no prologue
    y2, _ = ops.dwconv_fwd(x, w.reshape(C, k * k), k, s)
    ref2 = nhwc(F.conv2d(nchw(x.float()), w, None, s, (k - 1) // 2, 1, C))
    assert relerr(y2.float(), ref2) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,s,B,H,W,C", DW_CASES)
def test_dwconv_bwd(dtype, k, s, B, H, W, C):
    pad = (k - 1) // 2
    x = rnd(B, H, W, C, seed=1, dtype=dtype)
    w = rnd(C, 1, k, k, seed=2, scale=1.0 / k)
    sc, sh = rnd(C, seed=3).abs() + 0.5, rnd(C, seed=4, scale=0.3)
    OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    dy = rnd(B, OH, OW, C, seed=5, dtype=dtype)
    # reference through autograd: a = silu(x*sc+sh); y = dw(a)
    xr = x.float().clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    z = xr * sc + sh
    a = F.silu(z)
    a.retain_grad()
    yref = F.conv2d(a.permute(0, 3, 1, 2), wr, None, s, pad, 1, C)
    yref.backward(dy.float().permute(0, 3, 1, 2))
    # plain data gradient
    dx, _ = ops.dwconv_bwd_data(dy, w.reshape(C, k * k), k, s, (H, W))
    assert relerr(dx.float(), a.grad) < tol(dtype)
    # fused epilogue: g = dA * silu'(z) and its partial sums
    mean, rstd = rnd(C, seed=6, scale=0.2), rnd(C, seed=7).abs() + 0.5
    g, part = ops.dwconv_bwd_data(dy, w.reshape(C, k * k), k, s, (H, W), yfwd=x, escale=sc, eshift=sh, emean=mean,
                                  erstd=rstd, stats_rows=4)
    gref = xr.grad / sc  # d/dz
    assert relerr(g.float(), gref) < tol(dtype)
    st = part.sum(0)
    xhat = (x.float() - mean) * rstd
    assert relerr(st[0], gref.sum((0, 1, 2))) < tol(dtype, 1e-4, 3e-2)
    assert relerr(st[1], (gref * xhat).sum((0, 1, 2))) < tol(dtype, 1e-4, 3e-2)
    # weight gradient
    dw = ops.dwconv_bwd_weight(x, dy, k, s, pscale=sc, pshift=sh)
    assert relerr(dw, wr.grad) < tol(dtype, 5e-5, 2e-2)


@pytest.mark.parametrize("k,B,H,W,C,Cin", [(3, 2, 37, 41, 144, 24), (3, 1, 8, 6, 144, 8), (3, 3, 150, 150, 144, 24),
                                            (3, 2, 75, 75, 144, 16), (3, 1, 9, 13, 144, 32), (3, 40, 50, 50, 144, 32)])
def test_dwconv_bwd_data_lin_matches_the_unfused_products(k, B, H, W, C, Cin):
    """The fused stride-2 data gradient never stores g; its two images g.(ca o W) and g^T x and the BatchNorm partial
    sums must equal what the separate kernels compute from the stored (bf16-rounded) g: same products, same rounding
    of g, only the order of the fp32 additions differs. Against an fp32 torch reference as well."""
    dtype = torch.bfloat16
    pad = (k - 1) // 2
    OH, OW = (H + 2 * pad - k) // 2 + 1, (W + 2 * pad - k) // 2 + 1
    M = B * H * W
    y1 = rnd(B, H, W, C, seed=1, dtype=dtype)
    w = rnd(C, 1, k, k, seed=2, scale=1.0 / k).reshape(C, k * k).contiguous()
    sc, sh = rnd(C, seed=3).abs() + 0.5, rnd(C, seed=4, scale=0.3)
    mean, rstd = rnd(C, seed=6, scale=0.2), rnd(C, seed=7).abs() + 0.5
    gamma = rnd(C, seed=8).abs() + 0.5
    dy = rnd(B, OH, OW, C, seed=5, dtype=dtype)
    wexp = rnd(C, Cin, seed=9, scale=0.2, dtype=dtype)
    x = rnd(B, H, W, Cin, seed=10, dtype=dtype)
    assert ops.dwconv_bwd_data_lin_ok(dtype, C, k, 2, Cin)
    st = torch.stack([sc, sh, mean, rstd])
    g, part_ref = ops.dwconv_bwd_data(dy, w, k, 2, (H, W), yfwd=y1, escale=sc, eshift=sh, emean=mean, erstd=rstd,
                                      stats_rows=64)
    wcaT = (wexp.float() * (gamma * rstd)[:, None]).to(dtype).t().contiguous()
    dxp_ref, _ = ops.gemm_nt(g.view(M, C), wcaT)
    t1_ref = ops.gemm_tn(g.view(M, C), x.view(M, Cin))
    dxp, t1, part = ops.dwconv_bwd_data_lin(dy, w, k, 2, x, yfwd=y1, st=st, gamma=gamma, wexp=wexp, stats_rows=512)
    torch.cuda.synchronize()
    assert dxp.shape == dxp_ref.shape and t1.shape == t1_ref.shape
    scale = dxp_ref.float().abs().max().item()
    assert (dxp.float() - dxp_ref.float()).abs().max().item() <= 2.0 ** -7 * scale   # one bf16 step at the largest value
    assert relerr(dxp.float(), dxp_ref.float()) < 2e-3
    assert relerr(t1, t1_ref) < 1e-4
    assert relerr(part.sum(0), part_ref.sum(0)) < 1e-4
    # independent fp32 reference of the two products (g from the library's own stored tensor)
    assert relerr(dxp.float(), g.view(M, C).float() @ wcaT.float().t()) < 5e-3
    assert relerr(t1, g.view(M, C).float().t() @ x.view(M, Cin).float()) < 1e-4


# ------------------------------------------------------------------ BN / SE / elementwise
@pytest.mark.parametrize("training", [True, False])
def test_bn_finalize_fwd_matches_torch(training):
    C, R = 136, 4000
    y = rnd(R, C, seed=1) * 2 + 0.5
    bn = torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(rnd(C, seed=2).abs() + 0.5)
        bn.bias.copy_(rnd(C, seed=3))
        bn.running_mean.copy_(rnd(C, seed=4) * 0.1)
        bn.running_var.copy_(rnd(C, seed=5).abs() + 0.5)
    bn.train(training)
    ref_bn = torch.nn.BatchNorm2d(C).to(DEV)
    ref_bn.load_state_dict(bn.state_dict())
    ref_bn.train(training)
    parts = torch.stack([torch.stack([c.sum(0), (c * c).sum(0)]) for c in y.chunk(8, 0)])  # [8,2,C]
    st = ops.bn_finalize_fwd(parts.contiguous(), R, ops.BNState(bn))
    out = y * st[0] + st[1]
    ref = ref_bn(y.t().reshape(1, C, R, 1)).reshape(C, R).t()
    assert relerr(out, ref) < 1e-5
    assert relerr(bn.running_mean, ref_bn.running_mean) < 1e-5
    assert relerr(bn.running_var, ref_bn.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,hw,B", [(144, 100, 3), (1392, 25, 2), (40, 900, 2)])
def test_bn_silu_gate_backward_chain(dtype, C, hw, B):
    """reduce -> finalize -> apply reproduces autograd of  a = silu(bn(y)) ; loss = sum(dZ * a * gate) + sum(dpool*mean_hw(a))."""
    R = B * hw
    y = rnd(R, C, seed=1, dtype=dtype)
    dz = rnd(R, C, seed=2, dtype=dtype)
    gamma, beta = rnd(C, seed=3).abs() + 0.5, rnd(C, seed=4, scale=0.3)
    gate = torch.sigmoid(rnd(B, C, seed=5))
    dpool = rnd(B, C, seed=6, scale=0.01)
    yr = y.float().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    mean, var = yr.mean(0), yr.var(0, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    a = F.silu((yr - mean) * rstd * gr + br)
    loss = (dz.float() * a * gate.repeat_interleave(hw, 0)).sum() + (dpool.repeat_interleave(hw, 0) * a).sum()
    loss.backward()
    st = torch.stack([gamma * rstd.detach(), beta - mean.detach() * gamma * rstd.detach(), mean.detach(),
                      rstd.detach()]).contiguous()
    dz0 = dz.clone()
    g, part = ops.bn_act_bwd_reduce(dz, y, st, ops.ACT_SILU, gate=gate, dpool=dpool, rows_per_img=hw)
    assert torch.equal(dz, dz0) and g.data_ptr() != dz.data_ptr()      # the incoming gradient is left alone
    coef = ops.bn_finalize_bwd(part, R, gamma, st, True)
    dy = ops.affine2(g, coef[2], coef[4], y=y, b=coef[3])
    assert relerr(dy.float(), yr.grad) < tol(dtype, 5e-5, 3e-2)
    assert relerr(coef[0], gr.grad) < tol(dtype, 5e-5, 3e-2)
    assert relerr(coef[1], br.grad) < tol(dtype, 5e-5, 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,hw,B", [(144, 100, 3), (1392, 25, 2), (40, 900, 2), (2304, 100, 5)])
def test_se_bn_one_pass_backward(dtype, C, hw, B):
    """se_bwd_sums -> se_bn_combine -> finalize -> bn_act_bwd_apply equals autograd of a = silu(bn(y)) with
    loss = sum(dZ*a*gate) + sum(dpool*a), and plane 0 of the sums is the gradient w.r.t. the gate."""
    R = B * hw
    side = int(round(hw ** 0.5))
    assert side * side == hw
    y = rnd(R, C, seed=1, dtype=dtype)
    dz = rnd(R, C, seed=2, dtype=dtype)
    gamma, beta = rnd(C, seed=3).abs() + 0.5, rnd(C, seed=4, scale=0.3)
    gate = torch.sigmoid(rnd(B, C, seed=5))
    dpool = rnd(B, C, seed=6, scale=0.01)
    yr = y.float().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    gater = gate.clone().requires_grad_(True)
    mean, var = yr.mean(0), yr.var(0, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    a = F.silu((yr - mean) * rstd * gr + br)
    loss = (dz.float() * a * gater.repeat_interleave(hw, 0)).sum() + (dpool.repeat_interleave(hw, 0) * a).sum()
    loss.backward()
    st = torch.stack([gamma * rstd.detach(), beta - mean.detach() * gamma * rstd.detach(), mean.detach(),
                      rstd.detach()]).contiguous()
    sums = ops.se_bwd_sums(y.view(B, side, side, C), dz.view(B, side, side, C), st)
    assert sums.shape[0] == 5 and sums.shape[1] == B and sums.shape[3] == C
    assert relerr(sums[0].sum(1), gater.grad) < tol(dtype, 5e-5, 3e-2)
    part = ops.se_bn_combine(sums, gate, dpool)
    coef = ops.bn_finalize_bwd(part, R, gamma, st, True)
    g = dz.clone()
    dy = ops.bn_act_bwd_apply(g, y, st, ops.ACT_SILU, coef, gate=gate, dpool=dpool, rows_per_img=hw)
    assert dy.data_ptr() == g.data_ptr()
    assert relerr(dy.float(), yr.grad) < tol(dtype, 5e-5, 3e-2)
    assert relerr(coef[0], gr.grad) < tol(dtype, 5e-5, 3e-2)
    assert relerr(coef[1], br.grad) < tol(dtype, 5e-5, 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_affine2_and_img_colsum(dtype):
    B, H, W, C = 3, 37, 41, 144
    x, res = rnd(B, H, W, C, seed=1, dtype=dtype), rnd(B, H, W, C, seed=2, dtype=dtype)
    a, d = rnd(C, seed=3), rnd(C, seed=4)
    out = ops.affine2(x, a, d, res=res, act=ops.ACT_SILU)
    ref = F.silu(x.float() * a + d) + res.float()
    assert relerr(out.float(), ref) < tol(dtype)
    ps = ops.img_colsum(x, None, a, d, ops.ACT_SILU)
    assert relerr(ps.sum(1), F.silu(x.float() * a + d).sum((1, 2))) < tol(dtype, 1e-5, 1e-3)
    ps2 = ops.img_colsum(x, res, a, d, ops.ACT_SILU)
    assert relerr(ps2.sum(1), (F.silu(x.float() * a + d) * res.float()).sum((1, 2))) < tol(dtype, 1e-5, 1e-3)


@pytest.mark.parametrize("C,RD,B,hw", [(144, 6, 4, 100), (2304, 96, 3, 49), (40, 10, 2, 400), (816, 34, 21, 361), (1392, 58, 16, 100)])
def test_se_gate_fwd_bwd(C, RD, B, hw):
    pool_part = rnd(B, 3, C, seed=1)
    wr, br = rnd(RD, C, seed=2, scale=C ** -0.5), rnd(RD, seed=3, scale=0.1)
    we, be = rnd(C, RD, seed=4, scale=RD ** -0.5), rnd(C, seed=5, scale=0.1)
    weT = we.t().contiguous()
    pooled, hpre, gate = ops.se_gate_fwd(pool_part, hw, wr, br, weT, be)
    pp = pool_part.clone().requires_grad_(True)
    wr_, br_, we_, be_ = [t.clone().requires_grad_(True) for t in (wr, br, we, be)]
    p_ref = pp.sum(1) / hw
    g_ref = torch.sigmoid(F.linear(F.silu(F.linear(p_ref, wr_, br_)), we_, be_))
    assert relerr(gate, g_ref) < 1e-5 and relerr(pooled, p_ref) < 1e-5
    dgate_part = rnd(B, 2, C, seed=6)
    g_ref.backward(dgate_part.sum(1))
    dpool, dpre2, dpre1 = ops.se_gate_bwd(dgate_part, hw, wr, weT, gate, hpre, pooled)
    dwr, dbr, dwe, dbe = ops.se_param_grad(dpre2, dpre1, pooled, hpre)
    # dpool is the per-pixel gradient: d/d(sum over pixels) / hw  == pp.grad of any split
    assert relerr(dpool, pp.grad[:, 0]) < 1e-4
    for mine, ref in ((dwr, wr_.grad), (dbr, br_.grad), (dwe, we_.grad), (dbe, be_.grad)):
        assert relerr(mine, ref) < 1e-4


@pytest.mark.parametrize("C,RD,B,side", [(576, 24, 9, 19), (1392, 58, 16, 10), (2304, 96, 5, 10), (144, 6, 4, 10),
                                         (40, 10, 3, 20)])
def test_se_single_split_paths_match_general_ones(C, RD, B, side):
    """On maps of up to 400 pixels the per-image reductions use one row per image, the forward gate takes that row as
    it is (the 1/HW of the mean rides in the first FC) and the backward takes plane 0 of se_bwd_sums already multiplied
    by the sigmoid's derivative. Same results as the general path fed the same sums as a 3-way split (1e-6)."""
    hw = side * side
    y = rnd(B, side, side, C, seed=1)
    dz = rnd(B, side, side, C, seed=2)
    st = torch.stack([rnd(C, seed=3).abs() + 0.5, rnd(C, seed=4), rnd(C, seed=5), rnd(C, seed=6).abs() + 0.5]).contiguous()
    wr, br = rnd(RD, C, seed=7, scale=C ** -0.5), rnd(RD, seed=8, scale=0.1)
    we, be = rnd(C, RD, seed=9, scale=RD ** -0.5), rnd(C, seed=10, scale=0.1)
    weT = we.t().contiguous()
    pool1 = ops.img_colsum(y, None, st[0], st[1], ops.ACT_SILU)
    assert pool1.shape[1] == 1
    # the general path: the same sums presented as three splits
    pool3 = torch.stack([pool1[:, 0] * 0.5, pool1[:, 0] * 0.25, pool1[:, 0] * 0.25], 1).contiguous()
    pooled1, hpre1, gate1 = ops.se_gate_fwd(pool1, hw, wr, br, weT, be)
    pooled3, hpre3, gate3 = ops.se_gate_fwd(pool3, hw, wr, br, weT, be)
    for a, b in ((pooled1, pooled3), (hpre1, hpre3), (gate1, gate3)):
        assert relerr(a, b) < 1e-6
    sums = ops.se_bwd_sums(y, dz, st, gate=gate1)
    plain = ops.se_bwd_sums(y, dz, st)
    assert sums.pre_scaled and not plain.pre_scaled
    assert torch.equal(sums[1:], plain[1:])
    assert relerr(sums[0, :, 0], plain[0, :, 0] * gate1 * (1 - gate1)) < 1e-6
    dpool1, dpre2_1, dpre1_1 = ops.se_gate_bwd(sums[0], hw, wr, weT, gate1, hpre1, pooled1, pre_scaled=True)
    dpool0, dpre2_0, dpre1_0 = ops.se_gate_bwd(plain[0], hw, wr, weT, gate1, hpre1, pooled1)
    assert dpre2_1.data_ptr() == sums[0].data_ptr()
    for a, b in ((dpool1, dpool0), (dpre2_1, dpre2_0), (dpre1_1, dpre1_0)):
        assert relerr(a, b) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,W", [(300, 300), (64, 64), (33, 47)])
def test_stem_im2col_gemm_equals_conv(dtype, H, W):
    x = rnd(2, 3, H, W, seed=1)
    w = rnd(40, 3, 3, 3, seed=2, scale=0.2)
    col, OH, OW = ops.stem_im2col(x, dtype)
    wp, _ = ops.cast_weight(w.reshape(40, 27), dtype, pad_to=32)
    y, _ = ops.gemm_nt(col, wp)
    xin = x if dtype == torch.float32 else x.bfloat16().float()
    win = w if dtype == torch.float32 else w.bfloat16().float()
    ref = F.conv2d(xin, win, None, 2, 1).permute(0, 2, 3, 1).reshape(-1, 40)
    assert (OH, OW) == ((H - 1) // 2 + 1, (W - 1) // 2 + 1)
    assert relerr(y.float(), ref) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cast_pad_transpose(dtype):
    w = rnd(144, 24, seed=1)
    d, dT = ops.cast_weight(w, dtype, pad_to=32, want=True, want_t=True)
    assert torch.equal(d[:, :24].float(), w.to(dtype).float()) and float(d[:, 24:].abs().max()) == 0.0
    assert torch.equal(dT.float(), w.to(dtype).float().t())


def test_adamw_matches_torch():
    """mia_adamw_step against torch.optim.AdamW over 5 steps on tensors of awkward sizes (vector tails, a scalar,
    more tensors than one launch carries)."""
    from mia_amd.optim import AdamW
    shapes = [(40, 3, 3, 3), (7,), (1,), (1392, 232), (5, 13), (384,)] + [(17 + i,) for i in range(60)]
    mine_p = [torch.nn.Parameter(rnd(*sh, seed=100 + i)) for i, sh in enumerate(shapes)]
    ref_p = [torch.nn.Parameter(p.detach().clone()) for p in mine_p]
    mine = AdamW(mine_p, lr=1e-3, weight_decay=1e-4)
    ref = torch.optim.AdamW(ref_p, lr=1e-3, weight_decay=1e-4)
    for step in range(5):
        for i, (a, b) in enumerate(zip(mine_p, ref_p)):
            g = rnd(*a.shape, seed=1000 * step + i)
            a.grad = g.clone()
            b.grad = g.clone()
        mine.step()
        ref.step()
    for a, b in zip(mine_p, ref_p):
        assert relerr(a, b) < 2e-6
    for a, b in zip(mine_p, ref_p):
        assert relerr(mine.state[a]["exp_avg"], ref.state[b]["exp_avg"]) < 1e-6
        assert relerr(mine.state[a]["exp_avg_sq"], ref.state[b]["exp_avg_sq"]) < 1e-6
        assert int(mine.state[a]["step"]) == 5
    # a parameter that starts receiving gradients later (unfrozen for fine-tuning) keeps its own step count, and the
    # step tensors of a state dict are right; a reloaded optimizer continues identically
    late_m, late_r = torch.nn.Parameter(rnd(33, seed=7)), None
    late_r = torch.nn.Parameter(late_m.detach().clone())
    mine.add_param_group({"params": [late_m]})
    ref.add_param_group({"params": [late_r]})
    for step in range(3):
        for i, (a, b) in enumerate(zip(mine_p + [late_m], ref_p + [late_r])):
            g = rnd(*a.shape, seed=5000 * (step + 1) + i)
            a.grad, b.grad = g.clone(), g.clone()
        mine.step()
        ref.step()
        if step == 0:
            sd = mine.state_dict()
            assert {int(v["step"]) for v in sd["state"].values()} == {6, 1}
            mine2 = AdamW([{"params": mine_p}, {"params": [late_m]}], lr=1e-3, weight_decay=1e-4)
            mine2.load_state_dict(sd)
            mine = mine2
    for a, b in zip(mine_p + [late_m], ref_p + [late_r]):
        assert relerr(a, b) < 2e-6
    assert int(mine.state_dict()["state"][len(mine_p)]["step"]) == 3 and int(mine.state_dict()["state"][0]["step"]) == 8
    # scheduler compatibility: lr lives in param_groups
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(mine, mode="min", factor=0.5, patience=0)
    sch.step(1.0); sch.step(2.0)
    assert abs(mine.param_groups[0]["lr"] - 5e-4) < 1e-12
