"""Segmentation training step (trunk + CBAM-UNet decoder + CE/Dice loss + AdamW): images/s and a per-kernel table."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mia_amd  # noqa: F401
from mia_amd.optim import AdamW
from mia_amd.src.loss import get_loss_function
from mia_amd.src.model import FoundationModel

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=300)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--plain", action="store_true", help="UNetDecoder without CBAM")
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2, classification_heads={},
                        segmentation_heads={"seg": 2}, use_attention=not args.plain, compute_dtype=args.dtype).to(dev).train()
opt = AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
crit = get_loss_function("segmentation", num_classes=2)
x = torch.randn(args.batch, 3, args.size, args.size, device=dev)
t = torch.randint(0, 2, (args.batch, args.size, args.size), device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(model(x, "segmentation", "seg"), t)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print(f"segmentation train step: batch {args.batch} {args.size}x{args.size} {args.dtype}: {dt * 1e3:.2f} ms/step "
      f"{args.batch / dt:.1f} images/s  loss {loss.item():.4f}  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB",
      flush=True)
if args.profile:
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
