"""Data-parallel gradient exchange: bucketed all-reduce over RCCL (torch.distributed backend "nccl" on
ROCm) overlapped with backward. One process per GPU. The reference has no distributed code at all
(single process, train.py:43); this is the multi-GPU form of its training step (train.py:369-371).

Buckets are filled in reverse parameter order (the order backward produces gradients: head and the
parameter-heavy late stages first), each bucket's all-reduce is launched asynchronously the moment its
last gradient lands, and `finish()` makes the compute stream wait for all of them before the optimizer
step. xGMI is point-to-point, so a few large buckets (default 16 MiB of the ~44 MB of fp32 gradients)
keep every link busy without paying per-collective latency many times.

Collectives are matched across ranks by construction: every step, every rank all-reduces EVERY bucket, in
bucket order. A bucket goes out during backward as soon as its last gradient has landed and all earlier
buckets are out; `finish()` launches the rest, with zeros in the slots of parameters that got no gradient
on this rank. A rank-dependent set of used parameters therefore cannot leave one rank waiting in a
collective the others never start. The price: a bucket holding a parameter that gets no gradient only goes
out at `finish()` together with every later bucket (no overlap for those), so pass only the parameters the
step trains (the trunk and the active head), not every head of a multi-head model.
"""
import os
import sys
import time
from typing import Iterable, List

import torch
import torch.distributed as dist


def _trace(msg):
    if os.environ.get("MIA_DDP_TRACE"):
        rank = dist.get_rank() if dist.is_initialized() else 0
        print(f"[ddp rank {rank}] {time.strftime('%H:%M:%S')} {msg}", file=sys.stderr, flush=True)


class GradReducer:
    """params: the parameters being trained. always_exchange: run the pack / all-reduce / wait path even with a single
    rank (where the exchange is the identity), so that one GPU exercises the code the multi-GPU job runs."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 16 << 20, group=None,
                 always_exchange: bool = False, tail_bytes: int = 1 << 20):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if always_exchange and not dist.is_initialized():
            raise RuntimeError("always_exchange needs an initialised process group")
        self.exchange = self.world > 1 or always_exchange
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.buckets = []       # list of dicts: flat, views{param_id: view}, pending, count, ready, launched
        self._param_bucket = {}
        self._works = []
        self._avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        # gloo (CPU rehearsals of the multi-rank step, never production): the collectives run on host copies of the
        # buckets. c10d's own handling of device tensors under gloo hung intermittently on this ROCm build with two
        # processes sharing a GPU (inside dist.broadcast of a small tensor, and inside work.wait() of an all-reduce)
        self._host_staged = dist.is_initialized() and dist.get_backend(group) == "gloo"
        self._in_backward = False
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            # The last bucket's all-reduce can only start when the LAST gradient of the backward pass has landed (the
            # stem's), so nothing is left to hide it behind. In this network the early layers are tiny (stem and
            # stages 0-2: 0.16 M of 10.9 M parameters) but a size-filled last bucket also holds megabytes of stage 3-4
            # gradients that were ready milliseconds earlier: cut it so that only the trailing `tail_bytes` wait for
            # the end of the pass and the rest goes out as soon as its own last gradient lands.
            tail, tail_n = [], 0
            while len(cur) > 1 and tail_n + cur[-1].numel() * 4 <= tail_bytes:
                tail_n += cur[-1].numel() * 4
                tail.insert(0, cur.pop())
            self._close(cur)
            if tail:
                self._close(tail)
        self._next = 0          # buckets [0, _next) are out this step
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def _close(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, dtype=torch.float32, device=plist[0].device)
        views, off = {}, 0
        for p in plist:
            views[id(p)] = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        b = {"flat": flat, "views": views, "count": len(plist), "pending": len(plist), "ready": [], "launched": False,
             "index": len(self.buckets)}
        for p in plist:
            self._param_bucket[id(p)] = b
        self.buckets.append(b)

    def broadcast_parameters(self, module: torch.nn.Module, src: int = 0):
        """One-time broadcast of parameters and buffers from rank `src` (replicas start identical). Tensors are
        packed per dtype into one flat buffer each, so the ~1,000 tensors of the model cost a handful of
        collectives instead of one each."""
        if not self.exchange:
            return
        by_dtype = {}
        for t in list(module.parameters()) + list(module.buffers()):
            by_dtype.setdefault((t.dtype, t.device), []).append(t.data)
        for (dtype, device), tensors in by_dtype.items():
            flat = torch.cat([t.reshape(-1) for t in tensors])
            if self._host_staged and flat.is_cuda:
                host = flat.cpu()
                dist.broadcast(host, src, group=self.group)
                flat.copy_(host)
            else:
                dist.broadcast(flat, src, group=self.group)
            off = 0
            for t in tensors:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
            _trace(f"broadcast {dtype} x{len(tensors)} ({flat.numel()} elements) done")

    def _hook(self, p: torch.nn.Parameter):
        if not self.exchange:
            return  # nothing to exchange: gradients stay where autograd put them
        b = self._param_bucket[id(p)]
        if b["launched"] or b["pending"] == 0:
            raise RuntimeError(
                "GradReducer: a parameter received a second gradient before finish() — its bucket's all-reduce is "
                "already in flight. Call finish() after every backward (gradient accumulation across several "
                "backward passes is not supported by the bucketed exchange).")
        if not self._in_backward:
            self._in_backward = True
            _trace("first hook of this backward")
        b["ready"].append(p)
        b["pending"] -= 1
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._launch(self.buckets[self._next])

    # read by ops.Side.usable: this hook never reads a gradient on the caller's stream before the streams are joined
    _hook._mia_side_aware = True

    def _launch(self, b):
        """Pack the bucket with one fused multi-tensor copy and start its all-reduce while backward continues.

        On the GPU the weight gradients are computed on ops.Side's second stream, so the pack and the collective are
        queued THERE (the side stream first waits for what the caller's stream has queued so far, which covers the
        gradients computed on it): the caller's stream -- the data-gradient chain of the backward pass -- never
        waits for a weight-gradient kernel because of a bucket. It meets the exchange again in finish(), where it
        waits for the collectives, and at the end of the backward pass, where it joins the side stream anyway."""
        ready = b["ready"]
        cuda = b["flat"].is_cuda
        if cuda:
            from .ops import Side, _Inline
            scope = Side(b["flat"].device) if Side.enabled else _Inline()
        else:
            import contextlib
            scope = contextlib.nullcontext()
        b["launched"] = True
        assert b["index"] == self._next, "buckets go out in index order"
        self._next += 1
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        _trace(f"launch bucket {b['index']} ({b['flat'].numel()} elements, {len(ready)} gradients packed)")
        with scope:
            if len(ready) < b["count"]:
                # parameters without a gradient on this rank contribute zeros; their slots still hold the last
                # exchange's result. Every other slot is overwritten by the pack, so nothing else is cleared
                got = {id(p) for p in ready}
                torch._foreach_zero_([v for k, v in b["views"].items() if k not in got])
            if ready:
                views = [b["views"][id(p)] for p in ready]
                grads = [p.grad for p in ready]
                torch._foreach_copy_(views, grads)
                for p, v in zip(ready, views):
                    p.grad = v
                # the unpacked gradient tensors stay referenced until finish(): their memory (some of it from the
                # caller's stream's pool) must not be handed out again before the pack, queued elsewhere, has read it
                b["held"] = grads
            b["ready"] = []
            buf = b["flat"]
            if self._host_staged and buf.is_cuda:
                if "host" not in b:
                    b["host"] = torch.empty(buf.shape, dtype=buf.dtype, pin_memory=True)
                buf = b["host"]
                buf.copy_(b["flat"])    # blocking copy: the gradients packed above have been computed when it returns
            # c10d orders the collective after the work queued on the CURRENT stream, here the one that packed
            self._works.append((dist.all_reduce(buf, op=op, group=self.group, async_op=True), b))

    def finish(self):
        """Launch the buckets that are not out yet, wait for all of them."""
        if self.exchange:
            _trace("finish enter")
            while self._next < len(self.buckets):
                self._launch(self.buckets[self._next])
            for w, b in self._works:
                w.wait()        # the caller's stream waits for the collective (and with it for the pack before it)
                if self._host_staged and b["flat"].is_cuda:
                    b["flat"].copy_(b["host"])
                if not self._avg:
                    b["flat"].div_(self.world)
                b.pop("held", None)
            _trace(f"finish exit ({len(self._works)} buckets exchanged)")
        self._works.clear()
        self._in_backward = False
        self._next = 0
        for b in self.buckets:
            b["pending"] = b["count"]
            b["launched"] = False
            b["ready"] = []

    def zero_grad(self):
        """Drop last step's gradients. The flat buckets are not cleared: a bucket's pack overwrites the slot of every
        parameter that received a gradient, and _launch zeroes the slots of those that did not."""
        for p in self.params:
            p.grad = None

    def remove(self):
        for h in self._handles:
            h.remove()
