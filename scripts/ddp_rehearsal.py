"""Multi-rank rehearsal of the data-parallel training step on whatever GPUs the box has (ranks share GPU 0 when there
are fewer GPUs than ranks, which RCCL refuses, so the backend defaults to gloo then). Prints a time-stamped line per
phase and dumps every thread's Python stack if a phase takes longer than --stall seconds, so that a hang names its
call site in one run.

    MIA_DDP_TRACE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29533 scripts/ddp_rehearsal.py --steps 3
"""
import argparse
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default=None)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--stall", type=int, default=45)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--per-tensor-broadcast", action="store_true",
                    help="broadcast every parameter and buffer with a collective of its own (the first version of "
                         "broadcast_parameters) instead of one flat buffer per dtype")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    def phase(msg):
        print(f"[rank {rank}] {time.strftime('%H:%M:%S')} {msg}", flush=True)
        faulthandler.cancel_dump_traceback_later()
        faulthandler.dump_traceback_later(args.stall, exit=True)

    phase("start")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local if local < ndev else 0)
    torch.cuda.set_device(dev)
    backend = args.backend or ("nccl" if world <= ndev else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend, rank=rank, world_size=world)
    phase(f"group up: backend {backend}, {world} ranks, device {dev} of {ndev}")

    import mia_amd  # noqa: F401
    from mia_amd.ddp import GradReducer
    from mia_amd.head_ops import CrossEntropyLoss
    from mia_amd.optim import AdamW
    from mia_amd.src.model import FoundationModel
    torch.manual_seed(100 + rank)      # replicas start DIFFERENT: the broadcast must make them equal
    model = FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.2,
                            classification_heads={"synthetic": 4}, use_attention=True,
                            compute_dtype=args.dtype).to(dev).train()
    phase("model built")
    reducer = GradReducer(model.parameters(), always_exchange=True)
    phase(f"{len(reducer.buckets)} buckets")
    if args.per_tensor_broadcast:
        tensors = list(model.parameters()) + list(model.buffers())
        for i, t in enumerate(tensors):
            dist.broadcast(t.data, 0)
            if i % 100 == 0:
                phase(f"broadcast tensor {i} of {len(tensors)} ({tuple(t.shape)}, {t.dtype})")
    else:
        reducer.broadcast_parameters(model)
    torch.cuda.synchronize()
    phase("broadcast done")
    opt = AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    g = torch.Generator(device="cpu").manual_seed(7 + rank)
    x = torch.randn(args.batch, 3, args.size, args.size, generator=g).to(dev)
    y = torch.randint(0, 4, (args.batch,), generator=g).to(dev)
    crit = CrossEntropyLoss()
    for step in range(args.steps):
        reducer.zero_grad()
        loss = crit(model(x, "classification", "synthetic"), y)
        torch.cuda.synchronize()
        phase(f"step {step}: forward done")
        loss.backward()
        phase(f"step {step}: backward returned")
        reducer.finish()
        torch.cuda.synchronize()
        phase(f"step {step}: finish done")
        opt.step()
        torch.cuda.synchronize()
        phase(f"step {step}: optimizer done, loss {loss.item():.4f}")
    cs = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
    if backend != "nccl":
        cs = cs.cpu()      # gloo: host tensors, so that the check does not depend on c10d's device-tensor path
    lo, hi = cs.clone(), cs.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    ok = bool((lo.cpu() == hi.cpu()).item())
    phase(f"replicas in sync: {ok}")
    faulthandler.cancel_dump_traceback_later()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)
    print(f"[rank {rank}] REHEARSAL OK", flush=True)


if __name__ == "__main__":
    main()
