"""Device time of every block of the trunk inside one training step (bench.py's step at batch 128): forward time per
block from events recorded between the blocks, backward time per block from events recorded when the gradient
arrives at each block's input. With --no-fork the weight gradients stay on the caller's stream, so a block's backward
time is the sum of its kernels; by default it is the data-gradient chain with the weight gradients running beside it.
usage: python scripts/block_times.py [batch] [--no-fork]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import mia_amd
from mia_amd import ops, trunk as T
from mia_amd.head_ops import CrossEntropyLoss
from mia_amd.optim import AdamW
from mia_amd.src.model import FoundationModel

batch = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
if "--no-fork" in sys.argv:
    ops.Side.enabled = False
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2, classification_heads={"s": 4},
                        use_attention=True, compute_dtype="bf16").to(dev).train()
opt = AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
crit = CrossEntropyLoss()
x = torch.randn(batch, 3, 300, 300, device=dev)
y = torch.randint(0, 4, (batch,), device=dev)
names = [f"{si}.{bi}" for si, st in enumerate(model.backbone.blocks) for bi, _ in enumerate(st)]
blocks = [b for st in model.backbone.blocks for b in st]
rec = [False]
fwd_ev, bwd_ev = {}, {}


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


orig_fwd = T.MBConv.forward


def fwd(self, h, pre=None):
    i = blocks.index(self)
    if rec[0]:
        fwd_ev[("in", i)] = ev()
        if h.requires_grad:
            h.register_hook(lambda g, i=i: (bwd_ev.__setitem__(("in", i), ev()), g)[1])
    out = orig_fwd(self, h, pre)
    if rec[0]:
        fwd_ev[("out", i)] = ev()
        out.register_hook(lambda g, i=i: (bwd_ev.__setitem__(("out", i), ev()), g)[1])
    return out


T.MBConv.forward = fwd


def step(r):
    rec[0] = r
    opt.zero_grad(set_to_none=True)
    s0 = ev()
    loss = crit(model(x, "classification", "s"), y)
    s1 = ev()
    loss.backward()
    s2 = ev()
    opt.step()
    s3 = ev()
    return s0, s1, s2, s3


for _ in range(6):
    step(False)
torch.cuda.synchronize()
F, Bk, tot = [], [], []
for _ in range(5):
    s = step(True)
    torch.cuda.synchronize()
    F.append([fwd_ev[("in", i)].elapsed_time(fwd_ev[("out", i)]) for i in range(len(blocks))])
    # backward: the gradient at a block's output exists before the block's backward runs, the one at its input after
    Bk.append([bwd_ev[("out", i)].elapsed_time(bwd_ev[("in", i)]) if ("in", i) in bwd_ev else float("nan")
               for i in range(len(blocks))])
    tot.append((s[0].elapsed_time(s[1]), s[1].elapsed_time(s[2]), s[2].elapsed_time(s[3])))
F, Bk, tot = np.median(F, 0), np.median(Bk, 0), np.median(tot, 0)
print(f"batch {batch}, side stream {'on' if ops.Side.enabled else 'off'}: forward {tot[0]:.2f} ms, backward {tot[1]:.2f} ms, "
      f"optimizer {tot[2]:.2f} ms (device time, median of 5 steps)")
print(f"{'block':6s} {'fwd ms':>8s} {'bwd ms':>8s}")
for n, f, b in zip(names, F, Bk):
    print(f"{n:6s} {f:8.3f} {b:8.3f}")
print(f"{'sum':6s} {np.nansum(F):8.3f} {np.nansum(Bk):8.3f}   (stem, head and loss make up the rest)")
