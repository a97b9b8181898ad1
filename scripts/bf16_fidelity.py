"""Where do bf16 training gradients differ from fp32 ones? One train step (batch B at SxS, train-mode BatchNorm,
dropout 0) of the same weights in three arithmetic modes:
  fp32   this build's kernels in fp32 (pinned to the CPU oracle within 1e-3 by tests/test_model_gpu.py)
  bf16   this build's kernels with bf16 storage / fp32 accumulation
  amp    (optional, --amp) the plain-PyTorch oracle on the GPU under torch.autocast(bfloat16): what stock mixed
         precision gives for the same network, as a yardstick for how much of the difference is bf16 itself
Prints the cosine between the bf16 (and amp) gradient and the fp32 one per parameter group and for the worst
individual parameters, the loss of each mode, and the whole-model cosine."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import mia_amd  # noqa: E402,F401
from mia_amd.head_ops import CrossEntropyLoss  # noqa: E402
from mia_amd.src.model import FoundationModel  # noqa: E402


def group_of(name):
    if "backbone" not in name:
        return "head"
    for key in ("conv_stem", "conv_dw", "conv_pwl", "conv_pw", "se.", "bn"):
        if key in name:
            stage = name.split("blocks.")[1].split(".")[0] if "blocks." in name else "stem"
            return f"s{stage}.{key.strip('.')}"
    return "other"


def cosine(a, b):
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=300)
    ap.add_argument("--plain", action="store_true")
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--top", type=int, default=12)
    args = ap.parse_args()
    torch.manual_seed(0)
    kw = dict(backbone="efficientnet_b3", pretrained=False, dropout=0.0, classification_heads={"d": 4},
              use_attention=not args.plain)
    m32 = FoundationModel(compute_dtype="fp32", **kw).cuda().train()
    m16 = FoundationModel(compute_dtype="bf16", **kw).cuda().train()
    m16.load_state_dict(m32.state_dict())
    g = torch.Generator().manual_seed(1)
    x = torch.randn(args.batch, 3, args.size, args.size, generator=g).cuda()
    y = torch.randint(0, 4, (args.batch,), generator=g).cuda()
    crit = CrossEntropyLoss()
    grads, losses = {}, {}
    for tag, m in (("fp32", m32), ("bf16", m16)):
        loss = crit(m(x, "classification", "d"), y)
        loss.backward()
        losses[tag] = float(loss)
        grads[tag] = {n: p.grad.detach().float().flatten() for n, p in m.named_parameters() if p.grad is not None}
    if args.amp:
        from oracle.effnet_b3 import OracleFoundationModel
        ref = OracleFoundationModel({"d": 4}, dropout=0.0, use_attention=not args.plain)
        ref.load_state_dict(m32.state_dict())
        ref = ref.cuda().train()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(ref(x, "classification", "d").float(), y)
        loss.backward()
        losses["amp"] = float(loss)
        grads["amp"] = {n: p.grad.detach().float().flatten() for n, p in ref.named_parameters() if p.grad is not None}
    print(f"batch {args.batch} at {args.size}x{args.size}, {'plain' if args.plain else 'CBAM/MHA'} head; losses {losses}")
    modes = [t for t in ("bf16", "amp") if t in grads]
    names = [n for n in grads["fp32"] if not (n.endswith(".bias") and ".bn" in n) and not n.endswith("key.bias")]
    for t in modes:
        a = torch.cat([grads[t][n] for n in names])
        b = torch.cat([grads["fp32"][n] for n in names])
        print(f"  whole model: cosine({t}, fp32) = {cosine(a, b):.4f}   |g| ratio {float(a.norm() / b.norm()):.4f}")
    groups = {}
    for n in names:
        groups.setdefault(group_of(n), []).append(n)
    print(f"  {'group':18s} {'elements':>10s} {'|g32|':>10s} " + " ".join(f"{'cos ' + t:>10s}" for t in modes))
    for gname in sorted(groups):
        ns = groups[gname]
        b = torch.cat([grads["fp32"][n] for n in ns])
        row = [cosine(torch.cat([grads[t][n] for n in ns]), b) for t in modes]
        print(f"  {gname:18s} {b.numel():10d} {float(b.norm()):10.3e} " + " ".join(f"{c:10.4f}" for c in row))
    per = sorted(((cosine(grads["bf16"][n], grads["fp32"][n]), n) for n in names if grads["fp32"][n].numel() >= 64))
    print("  worst parameters (bf16 vs fp32):")
    for c, n in per[:args.top]:
        extra = f"  amp {cosine(grads['amp'][n], grads['fp32'][n]):.4f}" if "amp" in grads else ""
        print(f"    {c:8.4f}  {n}  ({grads['fp32'][n].numel()} elements){extra}")


if __name__ == "__main__":
    main()
