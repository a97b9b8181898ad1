"""Before/after of the fused BatchNorm finalize on single layer shapes of the small-map stages (batch 128): the 1x1
convolution + statistics launch followed by the finalize launch, against the one launch that does both. 200 calls
back to back on one stream, HIP-event timed; prints microseconds per call."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mia_amd  # noqa: F401
from mia_amd import ops

dev = torch.device("cuda", 0)


class BN:
    def __init__(self, C):
        self.weight, self.bias = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        self.running_mean, self.running_var = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        self.nbt = torch.zeros((), dtype=torch.long, device=dev)
        self.momentum, self.eps, self.training = 0.1, 1e-5, True


def timed(fn, n=200):
    for _ in range(20):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


SHAPES = [("exp 5.x  10x10 232->1392", 12800, 1392, 232, False), ("proj 5.x 10x10 1392->232", 12800, 232, 1392, True),
          ("exp 6.1  10x10 384->2304", 12800, 2304, 384, False), ("proj 6.1 10x10 2304->384", 12800, 384, 2304, True),
          ("exp 4.x  19x19 136->816", 46208, 816, 136, False), ("proj 4.x 19x19 816->136", 46208, 136, 816, True),
          ("exp 3.x  19x19 96->576", 46208, 576, 96, False), ("proj 3.x 19x19 576->96", 46208, 96, 576, True),
          ("exp 1.1  75x75 32->192", 720000, 192, 32, False)]
print(f"{'layer':28s} {'two launches':>13s} {'fused':>9s}   us per call")
for name, M, N, K, pro in SHAPES:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    hw = 100 if M == 12800 else 361 if M == 46208 else 5625
    kw = {}
    if pro:
        kw = dict(pscale=torch.rand(K, device=dev) + 0.5, pshift=torch.randn(K, device=dev), pro_act=ops.ACT_SILU,
                  gate=torch.rand(M // hw, K, device=dev), rows_per_img=hw)
    bn = BN(N)
    rows = ops.part_rows((M + 63) // 64)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)

    def run():
        ops.gemm_nt(a, b, out, stats_rows=rows, bn=bn, **kw)
    ops.Tickets.enabled = False
    t2 = timed(run)
    ops.Tickets.enabled = True
    t1 = timed(run)
    print(f"{name:28s} {t2:13.1f} {t1:9.1f}")
