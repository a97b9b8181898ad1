"""Every mia_gemm_nt shape of one EfficientNet-B3 training step at batch 128, 300x300, bf16, launched REPS times each,
in a fixed order, so that a rocprofv3 kernel trace / counter pass over this script can be cut per shape
(scripts/gemm_table.py does that). Prints one line per case: name M N K prologue stats add.
usage: python scripts/gemm_shapes.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import mia_amd  # noqa: F401
from mia_amd import ops
from mia_amd.trunk import B3_STAGES, B3_STEM

B = 128
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6


def cases():
    """(name, M, N, K, prologue, stats, add) in network order: expansion, projection, their data gradients."""
    out = []
    h = 150
    cin = B3_STEM
    for si, (kind, reps, k, s, e, cout) in enumerate(B3_STAGES):
        for bi in range(reps):
            stride = s if bi == 0 else 1
            mid = cin * e if kind == "ir" else cin
            ho = (h + 2 * ((k - 1) // 2) - k) // stride + 1
            M, M2 = B * h * h, B * ho * ho
            tag = f"{si}.{bi}"
            if kind == "ir":
                out.append((f"exp{tag}", M, mid, cin, False, True, False))
            out.append((f"proj{tag}", M2, cout, mid, True, True, False))
            out.append((f"dz{tag}", M2, mid, cout, False, False, False))
            if kind == "ir":
                out.append((f"dx{tag}", M, cin, mid, False, False, cin == cout and stride == 1))
            h, cin = ho, cout
    return out


def main():
    torch.manual_seed(0)
    seen = {}
    for name, M, N, K, pro, stats, add in cases():
        key = (M, N, K, pro, stats, add)
        if key in seen:     # identical launches (repeated blocks of a stage) are measured once
            print(f"{name} {M} {N} {K} {int(pro)} {int(stats)} {int(add)} same_as {seen[key]}", flush=True)
            continue
        seen[key] = name
        a = torch.randn(M, K, device="cuda").bfloat16()
        b = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        kw = dict(stats_rows=ops.part_rows((M + 63) // 64) if stats else 0)
        if pro:
            kw.update(pscale=torch.rand(K, device="cuda") + 0.5, pshift=torch.randn(K, device="cuda") * 0.1,
                      pro_act=ops.ACT_SILU, gate=torch.rand(B, K, device="cuda"), rows_per_img=M // B)
        if add:
            kw.update(add=torch.randn(M, N, device="cuda").bfloat16())
        for _ in range(REPS):
            ops.gemm_nt(a, b, **kw)
        torch.cuda.synchronize()
        print(f"{name} {M} {N} {K} {int(pro)} {int(stats)} {int(add)} measured", flush=True)
        del a, b, kw


if __name__ == "__main__":
    main()
