"""Latency of the eval forward, eager launches against HIP-graph replay, at small batches."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mia_amd  # noqa: F401
from mia_amd.graphs import GraphedForward
from mia_amd.src.model import FoundationModel

torch.manual_seed(0)
model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2, classification_heads={"d": 4},
                        use_attention=True, compute_dtype="bf16").cuda().eval()
for batch in ((1, 8, 32, 128) if len(sys.argv) == 1 else ()):
    x = torch.randn(batch, 3, 300, 300, device="cuda")
    g = GraphedForward(model, "classification", "d", x)
    res = {}
    for name, fn in (("eager", lambda: model(x, "classification", "d")), ("graph", lambda: g(x, copy_out=False))):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 20 * 1e3
    print(f"batch {batch:4d} 300x300 bf16 eval forward: eager {res['eager']:7.2f} ms  graph {res['graph']:7.2f} ms  "
          f"({batch / res['graph'] * 1e3:8.0f} images/s)", flush=True)

# ---- whole training step (forward, cross-entropy, backward, AdamW), eager launches against one graph replay
from mia_amd.graphs import GraphedTrainStep
from mia_amd.head_ops import CrossEntropyLoss
from mia_amd.optim import AdamW

model.train()
crit = CrossEntropyLoss()
opt = AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
for batch in ([int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]):
    x = torch.randn(batch, 3, 300, 300, device="cuda")
    y = torch.randint(0, 4, (batch,), device="cuda")

    def eager():
        opt.zero_grad(set_to_none=True)
        loss = crit(model(x, "classification", "d"), y)
        loss.backward()
        opt.step()

    step = GraphedTrainStep(model, "classification", "d", crit, opt, x, y)
    res = {}
    for name, fn in (("eager", eager), ("graph", lambda: step(x, y, copy_out=False)), ("eager2", eager)):
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 20 * 1e3
    print(f"batch {batch:4d} 300x300 bf16 train step: eager {res['eager']:7.2f} / {res['eager2']:7.2f} ms  graph {res['graph']:7.2f} ms  "
          f"({batch / res['graph'] * 1e3:8.0f} images/s)", flush=True)
    del step
