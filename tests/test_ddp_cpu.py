"""The bucketed gradient exchange on 2 ranks with the gloo backend on CPU: after finish() every rank holds the
mean gradient; buckets are launched in reverse parameter order; unused parameters do not hang the step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mia_amd  # noqa: F401
from mia_amd.ddp import GradReducer


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        unused = torch.nn.Linear(4, 4)           # never used in forward: gets no gradient
        params = list(net.parameters()) + list(unused.parameters())
        if rank == 1:
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(1.0)                   # replicas start different; the broadcast must fix that
        red = GradReducer(params, bucket_bytes=64)   # tiny buckets -> several all-reduces
        red.broadcast_parameters(torch.nn.ModuleList([net, unused]))
        assert len(red.buckets) >= 3
        # first bucket holds the LAST parameters (backward order)
        assert id(params[-1]) in red.buckets[0]["views"]
        for step in range(2):
            red.zero_grad()
            x = torch.full((4, 6), float(rank + 1 + step))
            net(x).pow(2).sum().backward()
            red.finish()
            local = [p.grad.clone() for p in net.parameters()]
            gathered = [[torch.zeros_like(g) for _ in range(world)] for g in local]
            for g, outs in zip(local, gathered):
                dist.all_gather(outs, g)
                for o in outs:
                    assert torch.allclose(o, g)          # identical on every rank
        # reference: mean over ranks of single-process gradients (hooks off: no collectives here)
        red.remove()
        ref = []
        for r in range(world):
            net.zero_grad()
            for p in net.parameters():
                p.grad = None
            x = torch.full((4, 6), float(r + 1 + 1))
            net(x).pow(2).sum().backward()
            ref.append([p.grad.clone() for p in net.parameters()])
        mean = [sum(gs) / world for gs in zip(*ref)]
        for g, m in zip(local, mean):
            assert torch.allclose(g, m, rtol=1e-5, atol=1e-6)
        assert all(p.grad is None or float(p.grad.abs().sum()) == 0.0 for p in unused.parameters())
        # a parameter set that differs between ranks: only rank 0 runs `extra`. Every rank must still launch the
        # same collectives (rank 1 contributes zeros), and both end with the mean = half of rank 0's gradient
        extra = torch.nn.Linear(6, 5)
        with torch.no_grad():
            for p in extra.parameters():
                p.fill_(0.25)
        red2 = GradReducer(list(net.parameters()) + list(extra.parameters()), bucket_bytes=64)
        red2.zero_grad()
        x = torch.full((4, 6), 2.0)
        h = net[0](x) + (extra(x) if rank == 0 else 0.0)
        net[2](torch.relu(h)).pow(2).sum().backward()
        red2.finish()
        gw = extra.weight.grad
        assert gw is not None or rank == 1
        flat_w = red2._param_bucket[id(extra.weight)]["views"][id(extra.weight)]
        outs = [torch.zeros_like(flat_w) for _ in range(world)]
        dist.all_gather(outs, flat_w)
        assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().sum()) > 0.0
        # a second backward before finish() would accumulate into a bucket that is already in flight: refused
        red2.zero_grad()
        net(x).pow(2).sum().backward()
        with pytest.raises(RuntimeError, match="second gradient before finish"):
            net(x).pow(2).sum().backward()
        red2.finish()
        red2.remove()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()[-600:]))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_is_a_no_op():
    net = torch.nn.Linear(3, 2)
    red = GradReducer(net.parameters())
    red.zero_grad()
    net(torch.ones(1, 3)).sum().backward()
    g = net.weight.grad.clone()
    red.finish()
    assert torch.equal(net.weight.grad, g)


def _solo_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        x = torch.randn(4, 6)
        net(x).pow(2).sum().backward()
        ref = [p.grad.clone() for p in net.parameters()]
        red = GradReducer(net.parameters(), bucket_bytes=64, always_exchange=True)
        red.broadcast_parameters(net)
        red.zero_grad()
        net(x).pow(2).sum().backward()
        assert len(red._works) == len(red.buckets) >= 2       # the exchange path really ran
        red.finish()
        for p, g in zip(net.parameters(), ref):
            assert torch.equal(p.grad, g)
            assert p.grad.data_ptr() == red._param_bucket[id(p)]["views"][id(p)].data_ptr()
        q.put("ok")
    except Exception:  # pragma: no cover
        import traceback
        q.put(traceback.format_exc()[-600:])
    finally:
        dist.destroy_process_group()


def test_forced_exchange_on_one_rank_is_the_identity():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_solo_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=120)
    p.join(timeout=60)
    assert res == "ok", res
