"""Attribute small torch kernels (fills, copies) of one training step to Python call sites (torch profiler)."""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import mia_amd
from mia_amd.ddp import GradReducer
from mia_amd.src.model import FoundationModel
from mia_amd.head_ops import CrossEntropyLoss

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2, classification_heads={"synthetic": 4},
                        use_attention=True, compute_dtype="bf16").to(dev).train()
reducer = GradReducer(model.parameters())
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
x = torch.randn(16, 3, 300, 300, device=dev)
y = torch.randint(0, 4, (16,), device=dev)
crit = CrossEntropyLoss()

def step():
    reducer.zero_grad()
    loss = crit(model(x, "classification", "synthetic"), y)
    loss.backward()
    reducer.finish()
    opt.step()

for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::ones", "aten::clone", "aten::contiguous", "aten::to"):
        st = [s for s in (e.stack or []) if "mia" in s or "medical" in s or "bench" in s or "find_fills" in s]
        cnt[(e.name, st[0] if st else (e.stack[0] if e.stack else "?"))] += 1
for k, v in cnt.most_common(40):
    print(v, k)
