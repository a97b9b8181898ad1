"""HIP-event timing of the row-streaming kernels (per-image channel sums, squeeze-excite / BatchNorm backward sums and
apply, BatchNorm apply) at the expanded-tensor shapes of every B3 block at batch 128, 300x300, bf16, each launched
alone: microseconds and GB/s (algorithmic bytes) per layer, to see where a family's average hides a slow layer.
usage: python scripts/stream_time.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import mia_amd  # noqa: F401
from mia_amd import ops
from mia_amd.trunk import B3_STAGES, B3_STEM

B = 128
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def layers():
    out, h, cin = [], 150, B3_STEM
    for si, (kind, reps, k, s, e, cout) in enumerate(B3_STAGES):
        for bi in range(reps):
            stride = s if bi == 0 else 1
            mid = cin * e if kind == "ir" else cin
            ho = (h + 2 * ((k - 1) // 2) - k) // stride + 1
            out.append((f"{si}.{bi}", ho, mid, cout))
            h, cin = ho, cout
    return out


def timeit(fn):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / REPS)
    return sorted(ts)[1]


def main():
    torch.manual_seed(0)
    seen = {}
    tot = {}
    for tag, ho, C, cout in layers():
        key = (ho, C)
        n = sum(1 for t in layers() if (t[1], t[2]) == key)
        if key in seen:
            continue
        seen[key] = tag
        y = torch.randn(B, ho, ho, C, device="cuda").bfloat16()
        dz = torch.randn(B, ho, ho, C, device="cuda").bfloat16()
        st = [torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1,
              torch.rand(C, device="cuda") + 0.5]
        gate, dpool = torch.rand(B, C, device="cuda"), torch.randn(B, C, device="cuda") * 0.01
        coef = [torch.randn(C, device="cuda") for _ in range(6)]
        nbytes = y.numel() * 2
        res = {
            "img_colsum": (timeit(lambda: ops.img_colsum(y, None, st[0], st[1], ops.ACT_SILU)), 1),
            "se_bwd_sums": (timeit(lambda: ops.se_bwd_sums(y, dz, st)), 2),
            "bwd_apply": (timeit(lambda: ops.bn_act_bwd_apply(dz, y, st, ops.ACT_SILU, coef, gate=gate, dpool=dpool,
                                                               rows_per_img=ho * ho)), 3),
            "affine2_3": (timeit(lambda: ops.affine2(dz, coef[2], coef[4], y=y, b=coef[3], out=dz)), 3),
        }
        line = f"{tag:5s} {ho:3d}x{ho:<3d} C={C:5d} x{n} {nbytes / 1e6:7.1f} MB |"
        for k, (us, passes) in res.items():
            line += f" {k} {us:7.1f} us {nbytes * passes / us / 1e3:5.0f} GB/s |"
            tot[k] = tot.get(k, 0.0) + us * n
        print(line, flush=True)
    print("per step (ms):", {k: round(v / 1e3, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
