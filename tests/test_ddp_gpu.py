"""The gradient exchange on the GPU over RCCL (backend "nccl") with a single rank and the exchange forced on: the real
model's backward runs through the hooks, the fused copy into the bucket views, the asynchronous all-reduce and the
stream wait of finish(). With one rank the all-reduce is the identity, so the gradients the optimizer sees must be,
bit for bit, the ones autograd produced. Runs in a child process: a failure inside RCCL cannot touch the other tests."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q):
    try:
        import faulthandler
        faulthandler.dump_traceback_later(150, exit=True)
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import mia_amd  # noqa: F401
        from mia_amd import ops
        from mia_amd.ddp import GradReducer
        ops.Side.MIN_WORK = 0       # every block forks its weight gradients: the two-stream path is what runs here
        from mia_amd.head_ops import CrossEntropyLoss
        from mia_amd.optim import AdamW
        from mia_amd.src.model import FoundationModel

        def make():
            torch.manual_seed(5)
            return FoundationModel(backbone="efficientnet_b3", pretrained=False, dropout=0.0,
                                   classification_heads={"synthetic": 4}, use_attention=True,
                                   compute_dtype="bf16").cuda().train()
        g = torch.Generator().manual_seed(1)
        x = torch.randn(4, 3, 96, 96, generator=g).cuda()
        y = torch.randint(0, 4, (4,), generator=g).cuda()
        crit = CrossEntropyLoss()

        model = make()
        seen = {}
        names = {id(p): n for n, p in model.named_parameters()}
        # registered BEFORE the reducer's hooks: sees the gradient exactly as autograd left it. The gradient may still
        # be in flight on the weight-gradient stream, so the tap joins the streams before it reads (and says so:
        # parameters with hooks that do not are kept off the second stream)
        def tap(p):
            ops.Side.sync()
            seen[names[id(p)]] = p.grad.clone()
        tap._mia_side_aware = True
        taps = [p.register_post_accumulate_grad_hook(tap) for p in model.parameters()]
        red = GradReducer(model.parameters(), bucket_bytes=4 << 20, always_exchange=True)
        red.broadcast_parameters(model)
        assert red.exchange and len(red.buckets) >= 8
        red.zero_grad()
        crit(model(x, "classification", "synthetic"), y).backward()
        launched_in_backward = len(red._works)
        red.finish()
        torch.cuda.synchronize()
        assert launched_in_backward == len(red.buckets), "every bucket goes out during backward when all are used"
        n = 0
        for name, p in model.named_parameters():
            assert p.grad is not None, name
            assert torch.equal(p.grad, seen[name]), f"{name}: exchanged gradient differs from autograd's"
            assert p.grad.data_ptr() == red._param_bucket[id(p)]["views"][id(p)].data_ptr()
            n += p.numel()
        for t in taps:
            t.remove()
        red.remove()

        # ---- the exchange does not stall the data-gradient chain: during the backward pass the caller's stream is
        # never made to wait for the weight-gradient stream because of a bucket. Every stream wait issued from Python
        # while backward runs is recorded with the number of buckets launched so far; the caller's stream may wait
        # for the side stream only once every bucket is out (the join at the end of the pass). (A timing check with a
        # spin kernel parked on the side stream was tried first: two torch streams can share a hardware queue, where
        # the spin kernel then holds up the caller's kernels as well, whatever the events say.)
        m2 = make()
        red2 = GradReducer(m2.parameters(), bucket_bytes=4 << 20, always_exchange=True)
        red2.zero_grad()
        loss2 = crit(m2(x, "classification", "synthetic"), y)
        main_id = torch.cuda.current_stream().cuda_stream
        waits = []
        orig_wait_stream, orig_wait_event = torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event

        def wait_stream(self, other):
            waits.append((self.cuda_stream, other.cuda_stream, len(red2._works)))
            return orig_wait_stream(self, other)

        def wait_event(self, ev):
            waits.append((self.cuda_stream, None, len(red2._works)))
            return orig_wait_event(self, ev)
        torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event = wait_stream, wait_event
        try:
            loss2.backward()
        finally:
            torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event = orig_wait_stream, orig_wait_event
        assert len(red2._works) == len(red2.buckets)
        forks = [w for w in waits if w[0] != main_id]
        joins = [w for w in waits if w[0] == main_id]
        assert len(forks) > 50, "the weight gradients did not run on the second stream"
        assert joins and all(w[2] == len(red2.buckets) for w in joins), \
            f"the caller's stream waited for the side stream before every bucket was out: {joins}"
        red2.finish()
        torch.cuda.synchronize()
        for name, p in m2.named_parameters():
            assert torch.equal(p.grad, seen[name]), f"{name}: gradient changed when the exchange ran on the side stream"
        red2.remove()

        # three optimizer steps with and without the exchange follow the same trajectory
        def run(with_reducer):
            m = make()
            r = GradReducer(m.parameters(), always_exchange=with_reducer)
            opt = AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
            out = []
            for _ in range(3):
                r.zero_grad()
                loss = crit(m(x, "classification", "synthetic"), y)
                loss.backward()
                r.finish()
                opt.step()
                out.append(loss.item())
            r.remove()
            return out
        a, b = run(True), run(False)
        assert all(abs(u - v) <= 2e-2 * max(1.0, abs(v)) for u, v in zip(a, b)), (a, b)
        dist.destroy_process_group()
        q.put(("ok", n, len(red.buckets)))
    except Exception:  # pragma: no cover
        import traceback
        q.put(("error", traceback.format_exc()[-1500:], 0))


def test_real_model_through_rccl_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    try:
        res = q.get(timeout=170)
    finally:
        p.join(timeout=20)
        if p.is_alive():
            p.kill()
    assert res[0] == "ok", res[1]
    assert res[1] > 10_000_000      # the whole B3 + CBAM/attention model went through the buckets


def test_two_ranks_real_model_replicas_stay_in_sync():
    """Two processes sharing GPU 0 (RCCL refuses two ranks on one device, so the collectives run over gloo on host
    copies of the buckets; everything else — hooks, bucket packing on the device, the HIP step — is the multi-GPU code):
    replicas that start from different seeds are made equal by the broadcast and hold identical parameters after
    four optimizer steps on different data."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "scripts", "ddp_rehearsal.py"),
           "--steps", "4", "--stall", "60"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=170, cwd=root)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert out.count("REHEARSAL OK") == 2 and out.count("replicas in sync: True") == 2, out[-3000:]


def test_bench_two_rank_path_prints_the_contract_line():
    """bench.py's N > 1 branch (rendezvous, barrier + MAX over ranks, exchange inside the step, replica check), run as
    the driver launches it but with MIA_BENCH_REHEARSAL=1: two ranks share GPU 0 over gloo. The numbers mean nothing;
    the line's fields and the replicas' agreement are what is checked."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps",
           "3", "--warmup", "2", "--batch", "8"]
    env = dict(os.environ, MIA_BENCH_REHEARSAL="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=170, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]        # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["replicas_in_sync"] is True and d["rehearsal"] is True
    assert d["value"] > 0 and abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    assert "cpu_baseline" not in d and d["vs_baseline"] is None
