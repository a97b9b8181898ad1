"""Bisect a replayed training step: forward only, forward+backward, forward+backward+AdamW(lr 0)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mia_amd  # noqa: F401
from mia_amd import ops
from mia_amd.head_ops import CrossEntropyLoss
from mia_amd.optim import AdamW
from mia_amd.src.model import FoundationModel

variant = sys.argv[1]
torch.manual_seed(0)
model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.0, classification_heads={"d": 4},
                        use_attention=True, compute_dtype="fp32").cuda().train()
for m in model.modules():
    if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
        m.p = 0.0
crit = CrossEntropyLoss()
opt = AdamW(model.parameters(), lr=0.0, weight_decay=0.0)
x = torch.randn(4, 3, 96, 96, device="cuda")
y = torch.randint(0, 4, (4,), device="cuda")
words = torch.zeros(64, dtype=torch.int32, device="cuda")
words[4:8] = torch.tensor([1.0, 0.0, 1.0, 0.0]).view(torch.int32).cuda()


def one():
    if variant == "fwd":
        with torch.no_grad():
            out = model(x, "classification", "d")
        return out, None
    opt.zero_grad(set_to_none=True)
    out = model(x, "classification", "d")
    loss = crit(out, y)
    loss.backward()
    if variant == "full":
        opt.step()
    return out, loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        ref_out, _ = one()
torch.cuda.current_stream().wait_stream(side)
ref_grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
print("eager logits", ref_out[0].tolist())
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
if variant == "full":
    opt.begin_capture(words.data_ptr() + 16, 15)
with torch.cuda.graph(g):
    out, loss = one()
if variant == "full":
    opt.end_capture()
for r in range(3):
    g.replay()
    torch.cuda.synchronize()
    bad_b = [n for n, b in model.named_buffers() if not torch.isfinite(b.float()).all()]
    line = f"replay {r}: logits {out[0].tolist()} non-finite buffers {len(bad_b)} {bad_b[:2]}"
    if ref_grads:
        worst = sorted(((float((p.grad - ref_grads[n]).abs().max() / (ref_grads[n].abs().max() + 1e-20)), n)
                        for n, p in model.named_parameters() if p.grad is not None), reverse=True)[:3]
        nf = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        line += f"\n   worst grad diff vs eager {worst}\n   non-finite grads {len(nf)} {nf[:3]}"
    print(line, flush=True)
