"""Host enqueue time against device completion time at marked points of one training step (bench.py's step at batch
128): where the device finishes a phase only just after the host has queued it, the host is the limiter there.
Markers: end of forward, gradient arriving at each feature map handed out by the trunk (stage ends 6, 4, 2, 1, 0 in
backward order), end of backward, end of the optimizer step.
usage: python scripts/timeline.py [batch] [--no-fork]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import mia_amd
from mia_amd import ops
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
marks = []
recording = [False]


def mark(name):
    if not recording[0]:
        return
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, time.perf_counter(), e))


trunk = model.backbone
orig = trunk.forward_nhwc


def forward_nhwc(inp):
    feats = orig(inp)
    for i, f in enumerate(feats):
        if f.requires_grad:
            f.register_hook(lambda g, i=i: (mark(f"grad at feature {i}"), g)[1])
    return feats


trunk.forward_nhwc = forward_nhwc


def step(rec):
    global marks
    recording[0] = rec
    if rec:
        marks = []
        mark("step start")
    opt.zero_grad(set_to_none=True)
    loss = crit(model(x, "classification", "s"), y)
    mark("forward + loss queued")
    loss.backward()
    mark("backward queued")
    opt.step()
    mark("optimizer queued")


for _ in range(8):
    step(False)
torch.cuda.synchronize()
rows = []
for _ in range(6):
    step(True)
    step(False)      # the next step's launches follow at once, as in a training loop
    torch.cuda.synchronize()
    t0, e0 = marks[0][1], marks[0][2]
    rows.append([(n, (t - t0) * 1e3, e0.elapsed_time(e)) for n, t, e in marks])
print(f"batch {batch}, side stream {'on' if ops.Side.enabled else 'off'}: ms after the step's first launch (median of 6 steps)")
print(f"{'marker':28s} {'queued by host':>15s} {'done on device':>15s} {'device lag':>11s}")
for k in range(len(rows[0])):
    h = float(np.median([r[k][1] for r in rows])); d = float(np.median([r[k][2] for r in rows]))
    print(f"{rows[0][k][0]:28s} {h:15.2f} {d:15.2f} {d - h:11.2f}")
