"""Headline benchmark: train images/sec of EfficientNet-B3 + CBAM/attention head, 300x300, bf16, on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward + cross-entropy + backward (+ bucketed RCCL all-reduce) + AdamW on one synthetic batch that is
already resident in HBM (128 MRI-shaped images per GPU: weak scaling, global batch 1024 at 8 GPUs). Prints ONE
JSON line with the whole-job images/sec, the roofline of the dominant kernel family (HIP-event timed inside the
timed region), the library's kernel launches per step and, at N=1, the CPU oracle timed on the host cores as a
baseline. --graph (one GPU only) replays the step as one HIP graph with the weight-gradient fork captured inside
(graphs.GraphedTrainStep) instead of launching it eagerly; the default, and the headline, is the eager step that a
train.py-style loop gets.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (spec)
FAMILIES = ["gemm_nt", "gemm_tn", "dwconv_fwd", "dwconv_bwd_data", "dwconv_bwd_weight", "bn_act_bwd_reduce",
            "affine2", "img_colsum", "se_bwd_sums", "bn_act_bwd_apply"]
# HBM bytes per launch from the PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, scripts/pmc_traffic.sh), by kernel
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "hbm_traffic.json")
TRAIN_GFLOP_PER_IMG = {"b3_cbam": 11.00, "b3_plain": 10.60}   # 3 x forward FLOPs at 300x300
NBATCH = 4                 # resident synthetic batches the steps rotate over


def phantom_batch(n, seed):
    """MRI-shaped uint8 grayscale 256x256: centred ellipse of N(110,40^2) clipped to [0,255], zero background."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:256, 0:256]
    mask = ((yy - 128) / 107.0) ** 2 + ((xx - 128) / 85.0) ** 2 <= 1.0
    img = np.clip(rng.normal(110, 40, (n, 256, 256)), 0, 255).astype(np.uint8)
    return img * mask[None]


def cpu_baseline(batch=8, steps=5, warmup=2, use_attention=True):
    """The CPU oracle (plain PyTorch fp32 restatement) timed on the host cores: same step, bounded sample
    (median of `steps` steps after `warmup` untimed ones)."""
    from oracle.effnet_b3 import OracleFoundationModel
    torch.manual_seed(0)
    m = OracleFoundationModel({"synthetic": 4}, dropout=0.2, use_attention=use_attention).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
    x = torch.randn(batch, 3, 300, 300)
    y = torch.randint(0, 4, (batch,))

    def step():
        opt.zero_grad()
        torch.nn.functional.cross_entropy(m(x, "classification", "synthetic"), y).backward()
        opt.step()
    for _ in range(warmup):
        step()
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(batch / med, 2), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"median of {steps} fp32 train steps of batch {batch} at 300x300 after {warmup} "
                                      f"warm-ups (oracle/effnet_b3.py, torch {torch.__version__} CPU)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--workload", default="b3_cbam", choices=["b3_cbam", "b3_plain"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-linbn", action="store_true",
                    help="A/B switch: apply BatchNorm backward to the expanded tensor instead of carrying it through "
                         "the expansion convolution")
    ap.add_argument("--linbn-min", type=int, default=0, help="A/B: smallest rows x expanded channels for that path")
    ap.add_argument("--no-fork", action="store_true", help="A/B switch: weight gradients on the main stream instead of the side stream")
    ap.add_argument("--fork-min", type=int, default=0,
                    help="A/B: smallest block (rows x expanded channels) whose weight gradients go to the side stream")
    ap.add_argument("--kernel-table", action="store_true", help="also print a per-family timing table to stderr")
    ap.add_argument("--nchw-input", action="store_true",
                    help="A/B switch: feed float32 NCHW images (the reference loaders' form) instead of the "
                         "channels-last compute-type images the preprocessing kernel can write directly")
    ap.add_argument("--graph", action="store_true",
                    help="time HIP-graph replays of the step (single GPU)")
    ap.add_argument("--graph-fork", action="store_true", help="with --graph: capture the weight-gradient fork/join")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    ndev = torch.cuda.device_count()
    # MIA_BENCH_REHEARSAL=1: the ranks share the GPUs there are and talk over gloo (RCCL refuses two ranks on one
    # device). It exists to run the N > 1 code path of this file on a one-GPU box; its numbers mean nothing and the
    # line says so ("rehearsal": true).
    rehearsal = world > 1 and os.environ.get("MIA_BENCH_REHEARSAL") == "1"
    if world > ndev and not rehearsal:
        raise SystemExit(f"{world} ranks need {world} GPUs (one process per GPU over RCCL), this node has {ndev}")
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def reduce_scalar(t, op):
        """all-reduce of a one-element tensor; on host memory under gloo (ddp.GradReducer stages its buckets likewise)"""
        h = t.cpu() if rehearsal else t
        dist.all_reduce(h, op=op)
        return h

    import mia_amd
    from mia_amd import ops
    from mia_amd import preprocess as PP
    from mia_amd.ddp import GradReducer
    from mia_amd.head_ops import CrossEntropyLoss
    from mia_amd.optim import AdamW
    from mia_amd.src.model import FoundationModel
    mia_amd.lib().mia_check_device()
    if args.no_fork:
        ops.Side.enabled = False
    if args.no_linbn:
        ops.LinBN.enabled = False
    if args.fork_min:
        ops.Side.MIN_WORK = args.fork_min
    if args.linbn_min:
        ops.LinBN.MIN_ELEMS = args.linbn_min

    torch.manual_seed(1234)
    use_attention = args.workload == "b3_cbam"
    model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2,
                            classification_heads={"synthetic": 4}, use_attention=use_attention,
                            compute_dtype=args.dtype).to(dev).train()
    reducer = GradReducer(model.parameters())
    reducer.broadcast_parameters(model)
    opt = AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    # synthetic MRI-shaped batches, preprocessed on the device by the resize+normalise kernel, resident in HBM. The
    # steps rotate over NBATCH different batches so that the timed steps keep back-propagating a real gradient (one
    # fixed batch is memorised within the warm-up and the loss collapses to 0)
    xs, ys = [], []
    for i in range(NBATCH):
        raw = torch.from_numpy(phantom_batch(args.batch, 1234 + 1000 * i + rank)).to(dev)
        # in the compute type and channels-last order: the form the stem reads without a conversion pass
        xs.append(PP.resize_normalize(raw, (300, 300), out_dtype=None if args.nchw_input else cdt))
        ys.append(torch.from_numpy(np.random.default_rng(99 + 1000 * i + rank).integers(0, 4, args.batch)).to(dev))
    del raw

    criterion = CrossEntropyLoss()
    counter = [0]

    def step():
        i = counter[0] % NBATCH
        counter[0] += 1
        reducer.zero_grad()
        logits = model(xs[i], "classification", "synthetic")
        loss = criterion(logits, ys[i])
        loss.backward()
        reducer.finish()
        opt.step()
        return loss

    graphed = None
    if args.graph:
        if world > 1:
            raise SystemExit("--graph is single-GPU: the gradient all-reduce is not captured")
        from mia_amd.graphs import GraphedTrainStep
        graphed = GraphedTrainStep(model, "classification", "synthetic", criterion, opt, xs[0], ys[0],
                                   fork=args.graph_fork)

        def step():     # noqa: F811  (the same step, replayed)
            i = counter[0] % NBATCH
            counter[0] += 1
            return graphed(xs[i], ys[i], copy_out=False)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up; the last warm-up step is event-timed per family to find the dominant one
    first_loss = None
    for i in range(max(args.warmup, 1)):
        if i == max(args.warmup, 1) - 1:
            ops.KernelTimer.start(FAMILIES)
        l0 = step()
        if first_loss is None:
            first_loss = l0
    table = ops.KernelTimer.stop()
    # the dominant family of the critical chain: families queued on the weight-gradient stream run beside it, their
    # event-timed durations are those of kernels sharing the chip (and HBM) with the main chain
    main_fams = {f: v for f, v in table.items() if v[3] == 0}
    dominant = max(main_fams, key=lambda f: main_fams[f][1]) if main_fams else "gemm_nt"

    barrier()
    ops.KernelTimer.start([dominant])
    launches0 = mia_amd.lib().mia_launch_count()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    lib_launches = (mia_amd.lib().mia_launch_count() - launches0) / max(args.steps, 1)
    rec = ops.KernelTimer.stop().get(dominant, (0, 0.0, 0, 0))[:3]
    if graphed is not None:
        # a replay makes no library calls: the events and the launch count come from one eager step instead
        ops.KernelTimer.start([dominant])
        launches0 = mia_amd.lib().mia_launch_count()
        opt.zero_grad(set_to_none=True)
        criterion(model(xs[0], "classification", "synthetic"), ys[0]).backward()
        lib_launches = mia_amd.lib().mia_launch_count() - launches0 + 1     # + the AdamW launch
        rec = ops.KernelTimer.stop().get(dominant, (0, 0.0, 0, 0))[:3]
        rec = (rec[0] * args.steps, rec[1] * args.steps, rec[2] * args.steps)
        opt.zero_grad(set_to_none=True)
    # after the timed region: one more step with the weight-gradient stream off, every family event-timed. Each kernel
    # then has the chip to itself, so bytes / duration is the kernel's own rate (in the timed region the two streams
    # share the HBM and a kernel's duration depends on what runs beside it)
    def eager_step():
        opt.zero_grad(set_to_none=True)
        reducer.zero_grad()
        criterion(model(xs[0], "classification", "synthetic"), ys[0]).backward()
        reducer.finish()
        opt.step()
    side_was = ops.Side.enabled
    ops.Side.enabled = False
    eager_step()
    ops.KernelTimer.start(FAMILIES)
    eager_step()
    solo = ops.KernelTimer.stop()
    ops.Side.enabled = side_was
    if world > 1:
        dt = float(reduce_scalar(torch.tensor([dt], dtype=torch.float64, device=dev), dist.ReduceOp.MAX).item())
    loss_v = float(loss.item())
    first_loss_v = float(first_loss.item())
    in_sync = None
    if world > 1:
        # replicas started identical and applied identical averaged gradients: their parameters must still agree
        cs = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
        lo, hi = reduce_scalar(cs.clone(), dist.ReduceOp.MIN), reduce_scalar(cs.clone(), dist.ReduceOp.MAX)
        in_sync = bool((lo == hi).item())

    if rank == 0:
        ips = world * args.batch * args.steps / dt
        launches, ms, nbytes = rec
        achieved = (nbytes / (ms * 1e-3)) / 1e9 if ms > 0 else 0.0
        # HBM bytes per launch from the counter passes on file: only when they were taken on THIS tree (the file
        # carries the digest of the kernel sources it was measured on) and on this workload; otherwise null
        traffic = None
        try:
            from mia_amd.build import kernel_tree_hash
            with open(TRAFFIC_FILE) as f:
                on_file = json.load(f)
            fam = on_file["families"].get(dominant)
            if (fam and on_file.get("tree") == kernel_tree_hash() and args.batch == 128 and args.dtype == "bf16"
                    and args.workload == "b3_cbam"):
                traffic = int(fam["hbm_bytes_per_launch"])
        except (OSError, KeyError, ValueError, ImportError):
            pass
        out = {
            "metric": "train images/sec (whole node) EffNet-B3+CBAM 300x300 bf16",
            "value": round(ips, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"EfficientNet-B3 {'+ CBAM/MHA head' if use_attention else '(plain head)'}, 4 classes, "
                                   f"300x300, per-GPU batch {args.batch} (global {world * args.batch}), train step = "
                                   f"fwd + CE + bwd + gradient all-reduce{' (no-op at 1 GPU)' if world == 1 else ''} + "
                                   f"AdamW{', replayed as one HIP graph' if args.graph else ''}; random-init weights, "
                                   f"{NBATCH} synthetic MRI-shaped batches resident in HBM, used in rotation",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "kernel_launches_per_step": round(lib_launches, 1),
                       "model_tflops_per_s": round(ips * TRAIN_GFLOP_PER_IMG[args.workload] / 1e3, 1)},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "launches_per_step": launches / max(args.steps, 1),
                         "avg_launch_us": round(ms * 1e3 / max(launches, 1), 2),
                         "algorithmic_bytes_per_launch": int(nbytes / max(launches, 1))},
            "first_loss": round(first_loss_v, 4), "final_loss": round(loss_v, 4),
        }
        # every timed family of the last warm-up step against the HBM roof: the line checks itself
        def fam_table(t):
            return {fam: {"launches": n, "ms": round(fms, 3), "GBps": round(fb / fms / 1e6, 1) if fms else 0.0,
                          "frac": round(fb / fms / 1e6 / HBM_PEAK_GBS, 4) if fms else 0.0, "side_stream": bool(ns)}
                    for fam, (n, fms, fb, ns) in sorted(t.items(), key=lambda kv: -kv[1][1])}
        out["roofline"]["families"] = fam_table(table)
        out["roofline"]["families_alone"] = fam_table(solo)
        out["roofline"]["note"] = ("achieved/frac: the dominant main-stream family, HIP events over the timed region, "
                                   "where weight-gradient kernels run beside it on a second stream; families: last "
                                   "warm-up step, same conditions; families_alone: one extra step after the timed "
                                   "region with that stream off (each kernel alone on the chip)")
        if in_sync is not None:
            out["replicas_in_sync"] = in_sync
        if rehearsal:
            out["rehearsal"] = True
        if args.kernel_table:
            tot = sum(v[1] for v in table.values())
            for fam, (n, fms, fb, _) in sorted(table.items(), key=lambda kv: -kv[1][1]):
                print(f"  {fam:20s} launches {n:4d}  {fms:8.3f} ms  {fb / 1e9:8.2f} GB  "
                      f"{fb / fms / 1e6 if fms else 0:8.1f} GB/s  {100 * fms / tot:5.1f} %", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(use_attention=use_attention)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
