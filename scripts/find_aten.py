"""Which torch (ATen) ops of one bench.py-shaped training step launch device kernels of their own (copies, fills,
elementwise, reductions) and from which line of this package: they are launches the library does not account for.
usage: python scripts/find_aten.py [batch]"""
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import ProfilerActivity, profile

import mia_amd  # noqa: F401
from mia_amd.ddp import GradReducer
from mia_amd.head_ops import CrossEntropyLoss
from mia_amd.optim import AdamW
from mia_amd.src.model import FoundationModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2, classification_heads={"s": 4},
                        use_attention=True, compute_dtype="bf16").to(dev).train()
opt = AdamW(model.parameters(), lr=1e-3)
reducer = GradReducer(model.parameters())
crit = CrossEntropyLoss()
x = torch.randn(B, 3, 300, 300, device=dev)
y = torch.randint(0, 4, (B,), device=dev)


def step():
    reducer.zero_grad()
    crit(model(x, "classification", "s"), y).backward()
    reducer.finish()
    opt.step()


for _ in range(3):
    step()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if not e.name.startswith("aten::") or not e.kernels:
        continue
    where = "?"
    for fr in e.stack:
        if "medical-image-analysis_amd" in fr or "mia_amd" in fr or "bench.py" in fr or "find_aten" in fr:
            where = fr.split("/")[-1]
            break
    c[(e.name, where, e.kernels[0].name[:50])] += len(e.kernels)
tot = 0
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print(v, *k)
    tot += v
print("total device launches from ATen ops:", tot)
