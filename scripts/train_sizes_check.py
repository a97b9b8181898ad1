"""One training step at several input sizes (bf16): finite loss and gradients, and bf16 gradients close to the fp32
path's (cosine over all parameters), so every tile-picking branch of the depthwise kernels is exercised."""
import os, sys, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mia_amd  # noqa: F401
from mia_amd.head_ops import CrossEntropyLoss
from mia_amd.src.model import FoundationModel

for size in (128, 160, 224, 256, 384):
    torch.manual_seed(0)
    m32 = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.0, classification_heads={"d": 4},
                          use_attention=True, compute_dtype="fp32").cuda().train()
    for mod in m32.modules():
        if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout2d)):
            mod.p = 0.0
    m16 = copy.deepcopy(m32)
    m16.compute_dtype = "bf16" if hasattr(m16, "compute_dtype") else None
    m16 = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.0, classification_heads={"d": 4},
                          use_attention=True, compute_dtype="bf16").cuda().train()
    m16.load_state_dict(m32.state_dict())
    for mod in m16.modules():
        if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout2d)):
            mod.p = 0.0
    x = torch.randn(6, 3, size, size, device="cuda")
    y = torch.randint(0, 4, (6,), device="cuda")
    crit = CrossEntropyLoss()
    losses, grads = [], []
    for m in (m32, m16):
        loss = crit(m(x, "classification", "d"), y)
        loss.backward()
        losses.append(float(loss.detach()))
        grads.append(torch.cat([p.grad.flatten().float() for p in m.parameters() if p.grad is not None]))
    ok = all(torch.isfinite(g).all() for g in grads)
    cos = float(torch.nn.functional.cosine_similarity(grads[0], grads[1], dim=0))
    print(f"{size}x{size}: loss fp32 {losses[0]:.5f} bf16 {losses[1]:.5f} finite {ok} grad cosine {cos:.4f}", flush=True)
    assert ok and abs(losses[0] - losses[1]) < 0.05 * abs(losses[0]) + 0.02 and cos > 0.5
print("ok")
