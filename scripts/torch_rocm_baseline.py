"""The bench.py workload in STOCK PyTorch-ROCm on the same GPU: the yardstick for "what the reference's own stack does
on an MI355X". The reference trains a timm EfficientNet-B3 trunk + its CBAM / multi-head-attention head through plain
torch.nn modules (model.py:240-363, train.py:344-371); timm is absent here, so the same network is written out below
with nn.Conv2d / nn.BatchNorm2d / nn.SiLU (MIOpen and hipBLASLt underneath), trained the way bench.py trains this
build's model: batch 128 of 300x300 inputs resident on the device, cross-entropy, AdamW, bf16 through torch.autocast,
channels_last memory format. Nothing of this repository's library is loaded.

usage: python scripts/torch_rocm_baseline.py [--batch 128] [--steps 10] [--warmup 3] [--fp32] [--nchw] [--plain-head]
prints one JSON line: {"ms_per_step": .., "images_per_sec": .., ...}"""
import argparse
import json
import math
import threading
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

STAGES = [("ds", 2, 3, 1, 1, 24), ("ir", 3, 3, 2, 6, 32), ("ir", 3, 5, 2, 6, 48), ("ir", 5, 3, 2, 6, 96),
          ("ir", 5, 5, 1, 6, 136), ("ir", 6, 5, 2, 6, 232), ("ir", 2, 3, 1, 6, 384)]
STEM = 40


def conv_bn_act(cin, cout, k, s, groups=1, act=True):
    layers = [nn.Conv2d(cin, cout, k, s, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout)]
    if act:
        layers.append(nn.SiLU(inplace=True))
    return nn.Sequential(*layers)


class SqueezeExcite(nn.Module):
    def __init__(self, ch, rd):
        super().__init__()
        self.reduce, self.expand = nn.Conv2d(ch, rd, 1), nn.Conv2d(rd, ch, 1)

    def forward(self, x):
        g = self.expand(F.silu(self.reduce(x.mean((2, 3), keepdim=True))))
        return x * torch.sigmoid(g)


class Block(nn.Module):
    def __init__(self, kind, cin, cout, k, s, e):
        super().__init__()
        mid = cin * e if kind == "ir" else cin
        self.expand = conv_bn_act(cin, mid, 1, 1) if kind == "ir" else nn.Identity()
        self.dw = conv_bn_act(mid, mid, k, s, groups=mid)
        self.se = SqueezeExcite(mid, max(1, round(cin * 0.25)))
        self.project = conv_bn_act(mid, cout, 1, 1, act=False)
        self.skip = s == 1 and cin == cout

    def forward(self, x):
        y = self.project(self.se(self.dw(self.expand(x))))
        return x + y if self.skip else y


class Head(nn.Module):
    """CBAM -> 8-head self-attention over the pixels (+ residual) -> pool -> LayerNorm -> MLP (model.py:112-175)."""

    def __init__(self, c, classes, heads=8, p=0.2):
        super().__init__()
        self.heads = heads
        self.mlp = nn.Sequential(nn.Linear(c, c // 16), nn.ReLU(), nn.Linear(c // 16, c))
        self.spatial = nn.Sequential(nn.Conv2d(2, 1, 7, 1, 3, bias=False), nn.BatchNorm2d(1, eps=1e-5, momentum=0.01))
        self.q, self.k, self.v, self.o = (nn.Linear(c, c) for _ in range(4))
        self.drop = nn.Dropout(p)
        self.cls = nn.Sequential(nn.LayerNorm(c), nn.Dropout(p), nn.Linear(c, 512), nn.ReLU(), nn.Dropout(p),
                                 nn.Linear(512, classes))

    def forward(self, x):
        B, C, H, W = x.shape
        x = x * torch.sigmoid(self.mlp(x.mean((2, 3))) + self.mlp(x.amax((2, 3))))[:, :, None, None]
        x = x * torch.sigmoid(self.spatial(torch.cat((x.amax(1, keepdim=True), x.mean(1, keepdim=True)), 1)))
        t = x.flatten(2).transpose(1, 2)
        sh = (B, H * W, self.heads, C // self.heads)
        q, k, v = (m(t).view(sh).transpose(1, 2) for m in (self.q, self.k, self.v))
        a = self.drop(torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(C // self.heads), -1)) @ v
        t = t + self.o(a.transpose(1, 2).reshape(B, H * W, C))
        return self.cls(t.mean(1))


class Net(nn.Module):
    def __init__(self, classes=4, attention=True):
        super().__init__()
        self.stem = conv_bn_act(3, STEM, 3, 2)
        blocks, cin = [], STEM
        for kind, reps, k, s, e, cout in STAGES:
            for i in range(reps):
                blocks.append(Block(kind, cin, cout, k, s if i == 0 else 1, e))
                cin = cout
        self.blocks = nn.Sequential(*blocks)
        self.attention = attention
        self.head = Head(cin, classes) if attention else nn.Sequential(
            nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Dropout(0.2), nn.Linear(cin, 512), nn.ReLU(), nn.Dropout(0.2),
            nn.Linear(512, classes))

    def forward(self, x):
        return self.head(self.blocks(self.stem(x)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--size", type=int, default=300)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--nchw", action="store_true")
    ap.add_argument("--plain-head", action="store_true")
    ap.add_argument("--benchmark", action="store_true", help="let MIOpen time its solvers per layer (minutes of warm-up)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = a.benchmark   # on: MIOpen times its solvers per layer during the warm-up steps

    def heartbeat():     # the solver search of the first step prints nothing for minutes
        while True:
            time.sleep(30)
            print(f"... {time.time() - t_start:.0f} s", flush=True)
    t_start = time.time()
    threading.Thread(target=heartbeat, daemon=True).start()
    model = Net(4, not a.plain_head).to(dev).train()
    fmt = torch.contiguous_format if a.nchw else torch.channels_last
    model = model.to(memory_format=fmt)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
    xs = [torch.randn(a.batch, 3, a.size, a.size, device=dev).contiguous(memory_format=fmt) for _ in range(4)]
    ys = [torch.randint(0, 4, (a.batch,), device=dev) for _ in range(4)]

    def step(i):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not a.fp32):
            loss = F.cross_entropy(model(xs[i % 4]).float(), ys[i % 4])
        loss.backward()
        opt.step()
        return loss

    t0 = time.time()
    for i in range(a.warmup):
        step(i)
        torch.cuda.synchronize()
        print(f"warm-up step {i}: {time.time() - t0:.1f} s since start", flush=True)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(a.steps):
        loss = step(i)
    torch.cuda.synchronize()
    ms = (time.time() - t0) * 1e3 / a.steps
    print(json.dumps({"what": "stock PyTorch-ROCm, same network and step as bench.py", "torch": torch.__version__,
                      "dtype": "fp32" if a.fp32 else "bf16 autocast", "memory_format": "NCHW" if a.nchw else "NHWC",
                      "head": "plain" if a.plain_head else "CBAM/MHA", "miopen_benchmark": a.benchmark, "batch": a.batch, "size": a.size,
                      "steps": a.steps, "ms_per_step": round(ms, 2), "images_per_sec": round(a.batch * 1e3 / ms, 1),
                      "final_loss": round(float(loss.detach()), 4), "params": sum(p.numel() for p in model.parameters())}))


if __name__ == "__main__":
    main()
