"""Which torch ops of one training step launch device copies / fills (they are not kernels of this library)."""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import ProfilerActivity, profile
import mia_amd
from mia_amd.head_ops import CrossEntropyLoss
from mia_amd.optim import AdamW
from mia_amd.src.model import FoundationModel

dev = torch.device("cuda", 0)
model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2, classification_heads={"s": 4},
                        use_attention=True, compute_dtype="bf16").to(dev).train()
opt = AdamW(model.parameters(), lr=1e-3)
crit = CrossEntropyLoss()
x = torch.randn(32, 3, 300, 300, device=dev)
y = torch.randint(0, 4, (32,), device=dev)
def step():
    for p in model.parameters(): p.grad = None
    crit(model(x, "classification", "s"), y).backward()
    opt.step()
for _ in range(3): step()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::zeros", "aten::to", "aten::_to_copy", "aten::cat", "aten::mul", "aten::add_", "aten::add", "aten::div_", "aten::pad", "aten::ones", "aten::zeros_like"):
        c[(e.name, str(e.input_shapes)[:80])] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:40]:
    print(v, k)
