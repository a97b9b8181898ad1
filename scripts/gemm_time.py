"""HIP-event timing of every mia_gemm_nt shape of one B3 training step (batch 128, 300x300, bf16), each launched alone:
one line per distinct shape with the microseconds per launch (median of 3 groups of REPS launches).
usage: python scripts/gemm_time.py [reps] [name-prefix filter]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from gemm_shapes import B, cases
from mia_amd import ops

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
FILT = sys.argv[2] if len(sys.argv) > 2 else ""
TILE = int(os.environ.get("STATS_TILE", "64"))   # rows per BatchNorm partial row (the library default assumes 64-row tiles)


def main():
    torch.manual_seed(0)
    seen, total = set(), 0.0
    counts = {}
    for name, M, N, K, pro, stats, add in cases():
        counts[(M, N, K, pro, stats, add)] = counts.get((M, N, K, pro, stats, add), 0) + 1
    for name, M, N, K, pro, stats, add in cases():
        key = (M, N, K, pro, stats, add)
        if key in seen or not name.startswith(FILT):
            continue
        seen.add(key)
        a = torch.randn(M, K, device="cuda").bfloat16()
        b = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        kw = dict(stats_rows=ops.part_rows((M + TILE - 1) // TILE) if stats else 0)
        if pro:
            kw.update(pscale=torch.rand(K, device="cuda") + 0.5, pshift=torch.randn(K, device="cuda") * 0.1,
                      pro_act=ops.ACT_SILU, gate=torch.rand(B, K, device="cuda"), rows_per_img=M // B)
        if add:
            kw.update(add=torch.randn(M, N, device="cuda").bfloat16())
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm_nt(a, b, out, **kw)
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                ops.gemm_nt(a, b, out, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / REPS)
        us = sorted(ts)[1]
        total += us * counts[key]
        print(f"{name:9s} M={M:8d} N={N:5d} K={K:5d} pro={int(pro)} x{counts[key]}  {us:8.1f} us", flush=True)
    print(f"sum over the step's launches: {total / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
