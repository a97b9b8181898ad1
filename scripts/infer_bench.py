"""Inference throughput of the finetune/inference path (eval mode, no_grad): images/s at batch 512."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mia_amd
from mia_amd.src.model import FoundationModel

dev = torch.device("cuda", 0)
for dtype in ("fp32", "bf16"):
    for size in (300, 224):
        torch.manual_seed(0)
        model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2, classification_heads={},
                                use_attention=True, compute_dtype=dtype).to(dev)
        model.add_classification_head("alzheimer", 4)
        model.eval()
        x = torch.randn(512, 3, size, size, device=dev)
        with torch.no_grad():
            for _ in range(3):
                model(x, "classification", "alzheimer")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                out = model(x, "classification", "alzheimer")
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        print(f"{dtype} {size}x{size} batch 512: {dt * 1e3:8.2f} ms/batch  {512 / dt:9.0f} images/s", flush=True)
        del model, x
        torch.cuda.empty_cache()
