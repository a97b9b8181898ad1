"""Micro-benchmark of single kernels at the shapes of the B3 trunk (per-GPU batch 128), for rocprofv3 / tuning.
usage: python scripts/kbench.py <case> [iters]   cases: dwfwd_s1 dwfwd_s11 dwbwdw_s4 gemm_exp1 gemm_proj1 ..."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mia_amd
from mia_amd import ops

def rnd(*s, dtype=torch.bfloat16):
    return torch.randn(*s, device="cuda").to(dtype)

B = 128
CASES = {}
def case(name):
    def deco(f):
        CASES[name] = f
        return f
    return deco

def dw_case(H, C, k, s, mode):
    x = rnd(B, H, H, C)
    w = torch.randn(C, k * k, device="cuda")
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    pad = (k - 1) // 2
    OH = (H + 2 * pad - k) // s + 1
    dy = rnd(B, OH, OH, C)
    rows_out = ops.part_rows(B * ((OH * OH + 127) // 128))
    rows_in = ops.part_rows(B * ((H * H + 127) // 128))
    if mode == "fwd":
        nbytes = (x.numel() + dy.numel()) * 2
        return (lambda: ops.dwconv_fwd(x, w, k, s, pscale=sc, pshift=sh, stats_rows=rows_out)), nbytes
    if mode == "bwdd":
        nbytes = (dy.numel() + 2 * x.numel()) * 2
        return (lambda: ops.dwconv_bwd_data(dy, w, k, s, (H, H), yfwd=x, escale=sc, eshift=sh, emean=sh, erstd=sc, stats_rows=rows_in)), nbytes
    nbytes = (x.numel() + dy.numel()) * 2
    return (lambda: ops.dwconv_bwd_weight(x, dy, k, s, pscale=sc, pshift=sh)), nbytes

for mode in ("fwd", "bwdd", "bwdw"):
    CASES[f"dw{mode}_s1"] = (lambda m=mode: dw_case(75, 192, 3, 1, m))     # stage 1.1
    CASES[f"dw{mode}_s10"] = (lambda m=mode: dw_case(150, 144, 3, 2, m))   # stage 1.0
    CASES[f"dw{mode}_s4"] = (lambda m=mode: dw_case(19, 816, 5, 1, m))     # stage 4.x
    CASES[f"dw{mode}_s5"] = (lambda m=mode: dw_case(10, 1392, 5, 1, m))    # stage 5.x
    CASES[f"dw{mode}_s2"] = (lambda m=mode: dw_case(38, 288, 5, 1, m))     # stage 2.x
    CASES[f"dw{mode}_s20"] = (lambda m=mode: dw_case(75, 192, 5, 2, m))    # stage 2.0
    CASES[f"dw{mode}_s30"] = (lambda m=mode: dw_case(38, 288, 3, 2, m))    # stage 3.0
    CASES[f"dw{mode}_s3"] = (lambda m=mode: dw_case(19, 576, 3, 1, m))     # stage 3.x
    CASES[f"dw{mode}_s6"] = (lambda m=mode: dw_case(10, 2304, 3, 1, m))    # stage 6.x
    CASES[f"dw{mode}_s0"] = (lambda m=mode: dw_case(150, 40, 3, 1, m))     # stage 0.0

def dwlin_case(H, C, k, Cin, fused):
    """Stride-2 data gradient + the two products of the expansion's backward: one fused kernel, or the three
    separate ones it replaces."""
    y1 = rnd(B, H, H, C)
    xin = rnd(B, H, H, Cin)
    w = torch.randn(C, k * k, device="cuda")
    wexp = rnd(C, Cin)
    st = torch.stack([torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1,
                      torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.5])
    gamma = torch.rand(C, device="cuda") + 0.5
    OH = (H + 2 * ((k - 1) // 2) - k) // 2 + 1
    dy = rnd(B, OH, OH, C)
    rows_in = ops.part_rows(B * ((H * H + 127) // 128))
    M = B * H * H
    nbytes = (dy.numel() + y1.numel() + 2 * xin.numel()) * 2
    if fused:
        return (lambda: ops.dwconv_bwd_data_lin(dy, w, k, 2, xin, yfwd=y1, st=st, gamma=gamma, wexp=wexp,
                                                stats_rows=rows_in)), nbytes
    wcaT = wexp.t().contiguous()
    def sep():
        g, _ = ops.dwconv_bwd_data(dy, w, k, 2, (H, H), yfwd=y1, escale=st[0], eshift=st[1], emean=st[2], erstd=st[3],
                                   stats_rows=rows_in)
        ops.gemm_nt(g.view(M, C), wcaT)
        ops.gemm_tn(g.view(M, C), xin.view(M, Cin))
    return sep, nbytes

CASES["dwlin_s10"] = lambda: dwlin_case(150, 144, 3, 24, True)
CASES["dwlin_s10_sep"] = lambda: dwlin_case(150, 144, 3, 24, False)

def dw_ablate(H, C, k, s, pro, stats):
    x = rnd(B, H, H, C)
    w = torch.randn(C, k * k, device="cuda")
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    pad = (k - 1) // 2
    OH = (H + 2 * pad - k) // s + 1
    rows_out = ops.part_rows(B * ((OH * OH + 127) // 128))
    nbytes = (x.numel() + B * OH * OH * C) * 2
    return (lambda: ops.dwconv_fwd(x, w, k, s, pscale=sc if pro else None, pshift=sh if pro else None,
                                   stats_rows=rows_out if stats else 0)), nbytes


CASES["dwfwd_s1_nopro"] = lambda: dw_ablate(75, 192, 3, 1, False, True)
CASES["dwfwd_s1_nostats"] = lambda: dw_ablate(75, 192, 3, 1, True, False)
CASES["dwfwd_s1_bare"] = lambda: dw_ablate(75, 192, 3, 1, False, False)


def gemm_case(M, N, K, pro, stats, act=None, use_gate=True):
    a, b = rnd(M, K), rnd(N, K)
    sc, sh = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda") * 0.1
    hw = M // B
    gate = torch.rand(B, K, device="cuda")
    nbytes = (M * K + N * K + M * N) * 2
    kw = dict(stats_rows=ops.part_rows((M + 63) // 64) if stats else 0)
    if pro:
        kw.update(pscale=sc, pshift=sh, pro_act=ops.ACT_SILU if act is None else act, rows_per_img=hw)
        if use_gate:
            kw.update(gate=gate)
    return (lambda: ops.gemm_nt(a, b, **kw)), nbytes

def gemm_add_case(M, N, K, add):
    """Data gradient of an expansion convolution: [M,K] x [N,K]^T (+ the residual's gradient)."""
    a, b = rnd(M, K), rnd(N, K)
    d = rnd(M, N) if add else None
    nbytes = (M * K + N * K + M * N * (2 if add else 1)) * 2
    return (lambda: ops.gemm_nt(a, b, add=d)), nbytes

for nm, (hw_, n_, k_) in {"1": (75 * 75, 32, 192), "2": (38 * 38, 48, 288), "3": (19 * 19, 96, 576),
                          "4": (19 * 19, 136, 816), "5": (100, 232, 1392), "6": (100, 384, 2304)}.items():
    CASES[f"gemm_dx{nm}"] = (lambda h=hw_, n=n_, k=k_: gemm_add_case(B * h, n, k, True))
    CASES[f"gemm_dx{nm}_noadd"] = (lambda h=hw_, n=n_, k=k_: gemm_add_case(B * h, n, k, False))

CASES["gemm_proj00"] = lambda: gemm_case(B * 150 * 150, 24, 40, True, True)
CASES["gemm_proj01"] = lambda: gemm_case(B * 150 * 150, 24, 24, True, True)
CASES["gemm_proj21"] = lambda: gemm_case(B * 38 * 38, 48, 288, True, True)
CASES["gemm_exp10"] = lambda: gemm_case(B * 150 * 150, 144, 24, False, True)
CASES["gemm_exp11"] = lambda: gemm_case(B * 75 * 75, 192, 32, False, True)
CASES["gemm_proj11"] = lambda: gemm_case(B * 75 * 75, 32, 192, True, True)
CASES["gemm_exp4"] = lambda: gemm_case(B * 19 * 19, 816, 136, False, True)
CASES["gemm_proj4"] = lambda: gemm_case(B * 19 * 19, 136, 816, True, True)
CASES["gemm_exp6"] = lambda: gemm_case(B * 100, 2304, 384, False, True)
CASES["gemm_proj6"] = lambda: gemm_case(B * 100, 384, 2304, True, True)
CASES["gemm_proj11_noact"] = lambda: gemm_case(B * 75 * 75, 32, 192, True, True, act=ops.ACT_NONE)
CASES["gemm_proj11_nogate"] = lambda: gemm_case(B * 75 * 75, 32, 192, True, True, use_gate=False)
CASES["gemm_proj11_plain"] = lambda: gemm_case(B * 75 * 75, 32, 192, False, True)
CASES["gemm_exp5_nostats"] = lambda: gemm_case(B * 100, 1392, 232, False, False)
CASES["gemm_exp5_k32"] = lambda: gemm_case(B * 100, 1392, 32, False, True)
CASES["gemm_exp5_k64"] = lambda: gemm_case(B * 100, 1392, 64, False, True)
CASES["gemm_exp5_k128"] = lambda: gemm_case(B * 100, 1392, 128, False, True)
CASES["gemm_exp5_k512"] = lambda: gemm_case(B * 100, 1392, 512, False, True)
CASES["gemm_exp5_n144"] = lambda: gemm_case(B * 100, 144, 232, False, True)
CASES["gemm_exp5_m4x"] = lambda: gemm_case(4 * B * 100, 1392, 232, False, True)
CASES["gemm_exp3"] = lambda: gemm_case(B * 19 * 19, 576, 96, False, True)
CASES["gemm_exp2"] = lambda: gemm_case(B * 38 * 38, 288, 48, False, True)
CASES["gemm_dz4"] = lambda: gemm_case(B * 19 * 19, 816, 136, False, False)
CASES["gemm_dz10"] = lambda: gemm_case(B * 150 * 150, 144, 24, False, False)
CASES["gemm_dz11"] = lambda: gemm_case(B * 75 * 75, 192, 32, False, False)

def tn_case(M, N, K, pro):
    g, x = rnd(M, N), rnd(M, K)
    sc, sh = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda") * 0.1
    gate = torch.rand(B, K, device="cuda")
    nbytes = (M * N + M * K) * 2
    kw = {}
    if pro:
        kw.update(pscale=sc, pshift=sh, pro_act=ops.ACT_SILU, gate=gate, rows_per_img=M // B)
    return (lambda: ops.gemm_tn(g, x, **kw)), nbytes

CASES["tn_exp11"] = lambda: tn_case(B * 75 * 75, 192, 32, False)
CASES["tn_proj11"] = lambda: tn_case(B * 75 * 75, 32, 192, True)
CASES["tn_exp4"] = lambda: tn_case(B * 19 * 19, 816, 136, False)
CASES["tn_proj6"] = lambda: tn_case(B * 100, 384, 2304, True)
CASES["tn_exp5"] = lambda: tn_case(B * 100, 1392, 232, False)
CASES["tn_proj5"] = lambda: tn_case(B * 100, 232, 1392, True)
CASES["tn_proj4"] = lambda: tn_case(B * 19 * 19, 136, 816, True)
CASES["tn_exp6"] = lambda: tn_case(B * 100, 2304, 384, False)
CASES["tn_proj0"] = lambda: tn_case(B * 150 * 150, 24, 40, True)
CASES["tn_exp10"] = lambda: tn_case(B * 150 * 150, 144, 24, False)
CASES["gemm_proj5"] = lambda: gemm_case(B * 100, 232, 1392, True, True)
CASES["gemm_exp5"] = lambda: gemm_case(B * 100, 1392, 232, False, True)
CASES["gemm_proj10"] = lambda: gemm_case(B * 75 * 75, 32, 144, True, True)

def se_case(C, RD, hw, mode):
    part = torch.randn(B, 2, C, device="cuda")
    wr, br = torch.randn(RD, C, device="cuda") * C ** -0.5, torch.randn(RD, device="cuda") * 0.1
    weT, be = torch.randn(RD, C, device="cuda") * RD ** -0.5, torch.randn(C, device="cuda") * 0.1
    pooled, hpre, gate = ops.se_gate_fwd(part, hw, wr, br, weT, be)
    nbytes = 2 * C * RD * 4 + 3 * B * C * 4
    if mode == "fwd":
        return (lambda: ops.se_gate_fwd(part, hw, wr, br, weT, be)), nbytes
    return (lambda: ops.se_gate_bwd(part, hw, wr, weT, gate, hpre, pooled)), nbytes

for mode in ("fwd", "bwd"):
    CASES[f"se{mode}_s6"] = (lambda m=mode: se_case(2304, 96, 100, m))
    CASES[f"se{mode}_s5"] = (lambda m=mode: se_case(1392, 58, 100, m))
    CASES[f"se{mode}_s4"] = (lambda m=mode: se_case(816, 34, 361, m))
    CASES[f"se{mode}_s3"] = (lambda m=mode: se_case(576, 24, 361, m))
    CASES[f"se{mode}_s2"] = (lambda m=mode: se_case(288, 12, 1444, m))

def sesums_case(H, C):
    y, dz = rnd(B, H, H, C), rnd(B, H, H, C)
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    st = ops.bn_finalize_fwd(torch.randn(64, 2, C, device="cuda").abs(), 1000.0, ops.BNState(bn))
    gate = torch.rand(B, C, device="cuda")
    nbytes = (y.numel() + dz.numel()) * 2
    return (lambda: ops.se_bwd_sums(y, dz, st, gate=gate)), nbytes

CASES["sesums_s1"] = lambda: sesums_case(75, 192)
CASES["sesums_s0"] = lambda: sesums_case(150, 40)
CASES["sesums_s2"] = lambda: sesums_case(38, 288)
CASES["sesums_s4"] = lambda: sesums_case(19, 816)
CASES["sesums_s5"] = lambda: sesums_case(10, 1392)


def bnfin_case(C, P, mode):
    part = torch.randn(P, 2, C, device="cuda").abs()
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    st = ops.bn_finalize_fwd(part, 1000.0, ops.BNState(bn))
    nbytes = part.numel() * 4
    if mode == "fwd":
        return (lambda: ops.bn_finalize_fwd(part, 1000.0, ops.BNState(bn))), nbytes
    return (lambda: ops.bn_finalize_bwd(part, 1000, bn.weight, st, True)), nbytes

for mode in ("fwd", "bwd"):
    for C_, P_ in ((32, 512), (192, 512), (816, 512), (2304, 400)):
        CASES[f"bnfin{mode}_{C_}"] = (lambda c=C_, p_=P_, m=mode: bnfin_case(c, p_, m))

def _fill_case():
    x = torch.empty(512 * 1024 * 1024, dtype=torch.bfloat16, device="cuda")
    return (lambda: x.zero_()), x.numel() * 2
def _copy_case():
    x = torch.empty(256 * 1024 * 1024, dtype=torch.bfloat16, device="cuda")
    y = torch.empty_like(x)
    return (lambda: y.copy_(x)), x.numel() * 4
def _read_case():
    x = torch.ones(512 * 1024 * 1024, dtype=torch.bfloat16, device="cuda")
    return (lambda: x.sum()), x.numel() * 2
CASES["raw_fill"] = _fill_case
CASES["raw_copy"] = _copy_case
CASES["raw_read"] = _read_case

def main():
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(CASES)
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    if names == ["all"]:
        names = list(CASES)
    for n in names:
        fn, nbytes = CASES[n]()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / iters
        print(f"{n:14s} {us:9.1f} us  {nbytes / 1e6:9.1f} MB  {nbytes / us / 1e6:8.2f} TB/s", flush=True)

if __name__ == "__main__":
    main()
