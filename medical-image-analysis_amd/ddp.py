"""Data-parallel gradient exchange: bucketed all-reduce over RCCL (torch.distributed backend "nccl" on
ROCm) overlapped with backward. One process per GPU. The reference has no distributed code at all
(single process, train.py:43); this is the multi-GPU form of its training step (train.py:369-371).

Buckets are filled in reverse parameter order (the order backward produces gradients: head and the
parameter-heavy late stages first), each bucket's all-reduce is launched asynchronously the moment its
last gradient lands, and `finish()` makes the compute stream wait for all of them before the optimizer
step. xGMI is point-to-point, so a few large buckets (default 16 MiB of the ~44 MB of fp32 gradients)
keep every link busy without paying per-collective latency many times.
"""
from typing import Iterable, List

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 16 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.buckets = []       # list of dicts: flat, views{param_id: view}, pending, count
        self._param_bucket = {}
        self._works = []
        self._avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def _close(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, dtype=torch.float32, device=plist[0].device)
        views, off = {}, 0
        for p in plist:
            views[id(p)] = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        b = {"flat": flat, "views": views, "count": len(plist), "pending": len(plist), "ready": []}
        for p in plist:
            self._param_bucket[id(p)] = b
        self.buckets.append(b)

    def broadcast_parameters(self, module: torch.nn.Module, src: int = 0):
        """One-time broadcast of parameters and buffers from rank `src` (replicas start identical)."""
        if self.world == 1:
            return
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src, group=self.group)

    def _hook(self, p: torch.nn.Parameter):
        if self.world == 1:
            return  # nothing to exchange: gradients stay where autograd put them
        b = self._param_bucket[id(p)]
        b["ready"].append(p)
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _launch(self, b):
        # one fused multi-tensor copy packs the bucket, then its all-reduce starts while backward continues
        ready = b["ready"]
        views = [b["views"][id(p)] for p in ready]
        torch._foreach_copy_(views, [p.grad for p in ready])
        for p, v in zip(ready, views):
            p.grad = v
        b["ready"] = []
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._works.append((dist.all_reduce(b["flat"], op=op, group=self.group, async_op=True), b))

    def finish(self):
        """Wait for every outstanding bucket, flush buckets whose parameters got no gradient this step."""
        if self.world > 1:
            for b in self.buckets:
                if 0 < b["pending"] < b["count"]:
                    # some parameters of this bucket were unused: their slots hold zeros
                    self._launch(b)
            for w, b in self._works:
                w.wait()
                if not self._avg:
                    b["flat"].div_(self.world)
        self._works.clear()
        for b in self.buckets:
            b["pending"] = b["count"]

    def zero_grad(self):
        """Drop last step's gradients; with several ranks also clear the flat buckets (slots of parameters that
        receive no gradient must contribute zeros to the reduction)."""
        if self.world > 1:
            for b in self.buckets:
                b["flat"].zero_()
        for p in self.params:
            p.grad = None

    def remove(self):
        for h in self._handles:
            h.remove()
