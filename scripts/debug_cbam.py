import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import torch
import mia_amd
from mia_amd.heads import CBAM
from oracle import effnet_b3 as O
from seeding import load_seeded, randn

def nhwc(x): return x.permute(0, 2, 3, 1).contiguous()
for (c, h, w, relu) in ((16, 48, 48, True), (16, 48, 48, False), (16, 32, 32, True), (16, 8, 8, True)):
    ref = load_seeded(O.CBAM(c), 81).train(True)
    mine = load_seeded(CBAM(c), 81).cuda().train(True)
    x = randn(82, 2, c, h, w)
    if relu: x = torch.relu(x)
    xr = x.clone().requires_grad_(True); xm = nhwc(x).cuda().requires_grad_(True)
    out_r, out_m = ref(xr), mine(xm)
    g = randn(83, 2, c, h, w)
    out_r.backward(g); out_m.backward(nhwc(g).cuda())
    d = (xm.grad.permute(0, 3, 1, 2).cpu() - xr.grad).abs()
    bad = (d > 1e-3 * xr.grad.abs().max())
    print(c, h, w, relu, "bad frac", bad.float().mean().item(), "max", d.max().item(), "bad where x==0:", (bad & (x == 0)).sum().item(), "of", bad.sum().item())
    idx = bad.nonzero()[:5]
    print(idx.tolist())
