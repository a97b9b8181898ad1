"""How long does the host take to enqueue one training step, against the device's time to run it?"""
import sys, os, time
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
from mia_amd.optim import AdamW
opt = AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.randn(B, 3, 300, 300, device=dev)
y = torch.randint(0, 4, (B,), device=dev)
crit = CrossEntropyLoss()

def step():
    reducer.zero_grad()
    loss = crit(model(x, "classification", "synthetic"), y)
    loss.backward()
    reducer.finish()
    opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print(f"batch {B}: host enqueue {np.mean(host):.2f} ms, step {np.mean(total):.2f} ms")
