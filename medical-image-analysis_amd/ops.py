"""Thin tensor -> pointer wrappers over the C ABI (include/mia_hip.h) and the block-level autograd
Functions built from them. Tensors are torch CUDA tensors used purely as device memory; every
arithmetic step is a kernel of libmia_hip.so launched on torch's current HIP stream.
"""
import ctypes
import math
from typing import Optional

import torch

from ._lib import lib

F32, BF16 = 0, 1
ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16}
MAX_PART = 512   # persistent workgroups (= BatchNorm partial rows) per launch: two per CU measured best


def _p(t: Optional[torch.Tensor]):
    """Device pointer of a tensor for the C ABI. Every pointer handed to a kernel goes through here, so a host
    tensor can never reach the GPU as an address (that would be a memory fault, not an exception)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("mia_amd kernels need tensors on an MI355X device; there is no CPU path "
                           f"(got a {t.device} tensor of shape {tuple(t.shape)})")
    return t.data_ptr()     # a plain int: the prototypes declare void*, ctypes converts without a wrapper object


def _pr(t, i: int):
    """Device pointer of row i of a 2-D tensor whose rows are contiguous (the BatchNorm state and coefficient tables):
    the address is computed, no view tensor is built. A row view costs about a microsecond of host time, and these
    rows are handed to kernels some 1300 times a step."""
    if not isinstance(t, torch.Tensor):
        return _p(t[i])           # a tuple / list of row tensors
    if not t.is_cuda:
        raise RuntimeError("mia_amd kernels need tensors on an MI355X device; there is no CPU path "
                           f"(got a {t.device} tensor of shape {tuple(t.shape)})")
    return t.data_ptr() + i * t.stride(0) * t.element_size()


# torch.cuda.current_stream() builds a Stream object through several Python layers (~8 us a call, once per kernel
# launch: half of a step's host time); the raw handle is one C call away
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


class Replay:
    """State of a HIP-graph capture of a training step. A replayed graph repeats its kernel arguments, so the two
    things that change from step to step come from device memory instead: the dropout seeds get a per-replay offset
    (`seed_off`, the address of a uint64 counter, None outside a capture) and AdamW reads its step-dependent factors
    from a float table. graphs.GraphedTrainStep sets these around its capture and rewrites the memory per replay."""
    seed_off = None


def seed_offset():
    return Replay.seed_off


class WeightEpoch:
    """Counts the events that change parameters or BatchNorm running statistics through raw device pointers, which
    torch's tensor version counters never see: a train-mode BatchNorm finalize (running_mean / running_var), the HIP
    AdamW step, a replay of a captured training step. Caches derived from the weights (the BatchNorm-folded inference
    copies of the trunk) carry the count they were built at and are rebuilt when it has moved."""
    n = 0

    @classmethod
    def bump(cls):
        cls.n += 1


def _require_cuda(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("mia_amd kernels need tensors on an MI355X device; there is no CPU path")


def part_rows(work_items: int) -> int:
    return max(1, min(MAX_PART, int(work_items)))


class Side:
    """The weight-gradient stream. In a backward pass the weight gradients (mia_gemm_tn, mia_dwconv_bwd_weight) are
    needed by nobody until the optimizer (or the gradient exchange) runs, while the data gradients form the critical
    chain; the kernels of the late, small stages each leave much of the chip idle while they ramp up and drain. So the
    weight-gradient kernels are queued on a second stream and run next to the data-gradient chain (measured on the
    B3 step at batch 128: 28.0 -> 25.4 ms).

        with Side(device, reads=(dy, x, stats)):    # the side stream first waits for everything queued on the caller's
            dw = gemm_tn(dy, x, ...)                # stream; these launches and allocations belong to the side stream

    Ordering and memory rules that make this safe:
      * every block starts with side.wait_stream(main): side kernels see their inputs complete, and memory that went
        back to the side stream's pool after being read on the main stream is not rewritten early;
      * `reads` (tensors allocated on the main stream, read by the side kernels) get record_stream(side): the allocator
        does not hand their memory out again before the side kernels have run, whenever Python lets go of them;
      * the main stream waits for the side stream (Side.sync) at the end of the backward pass (an autograd engine
        callback queued by every block), before a gradient bucket is packed (ddp.GradReducer) and at the top of
        AdamW.step(); code that reads .grad from inside a backward hook must call Side.sync() itself;
      * Side.usable(params) is False, and the caller then stays on its own stream, when Side.enabled is off, when one
        of the parameters already holds a gradient (autograd would add to it on the main stream at once), when a
        parameter carries a tensor hook or a post-accumulate hook that is not marked `_mia_side_aware` (such a hook
        would read the gradient on the main stream before the join), or when a parameter has more than one use in the
        graphs alive at this moment (autograd would add the uses' gradients together on the main stream). A hook
        that waits for the side stream itself (ddp.GradReducer's) sets `_mia_side_aware = True` on its function.
      * Under HIP-graph capture the fork/join events are captured like any others: the side stream joins the capture
        at its first wait and must be joined back (Side.sync) before the capture ends; graphs.GraphedTrainStep does."""
    enabled = True
    MIN_WORK = 8_000_000   # elements of the block's expanded tensor below which its weight gradients stay on the main stream
    depth = 0          # > 0 while launches go to a side stream (inside a `with Side(...)` block)
    _streams = {}      # (device index, main stream id) -> (side stream, fork event)
    _pending = {}      # (device index, main stream id) -> (main, side) with side work the main stream has not waited for
    _callback_task = -1        # autograd graph task that already has the join queued (one callback per backward pass)

    capture_ok = False  # set by graphs.GraphedTrainStep around its capture: the fork is recorded into the graph
    _uses = {}          # id(parameter) -> [uses in live graphs, "had more than one use at a time" flag]

    @classmethod
    def usable(cls, params=()) -> bool:
        if not cls.enabled or (not cls.capture_ok and torch.cuda.is_current_stream_capturing()):
            return False
        for p in params:
            if p is None:
                continue
            if p.grad is not None or p._backward_hooks:
                return False
            post = getattr(p, "_post_accumulate_grad_hooks", None)
            if post and not all(getattr(fn, "_mia_side_aware", False) for fn in post.values()):
                return False
            u = cls._uses.get(id(p))
            if u is not None and u[1]:
                return False
        return True

    class Use:
        """Held by an autograd context for as long as its graph lives: counts the live uses of each parameter whose
        gradient the context's backward would compute on the side stream."""
        __slots__ = ("keys",)

        def __init__(self, params):
            self.keys = [id(p) for p in params if p is not None]
            for k in self.keys:
                u = Side._uses.setdefault(k, [0, False])
                u[0] += 1
                if u[0] > 1:
                    u[1] = True

        def release(self):
            """Called at the end of the context's backward (the graph node itself may live on for as long as the
            caller holds the loss tensor), and again, as a no-op, when the context is collected."""
            for k in self.keys:
                u = Side._uses.get(k)
                if u is not None:
                    u[0] -= 1
                    if u[0] <= 0:
                        del Side._uses[k]
            self.keys = ()

        __del__ = release

    def __init__(self, device, reads=()):
        self.main = torch.cuda.current_stream(device)
        self.key = (device.index, self.main.cuda_stream)
        ent = Side._streams.get(self.key)
        if ent is None:
            # one event per stream pair, re-recorded at every fork (Stream.wait_stream would create and destroy one
            # each time: a backward pass forks about a hundred times)
            ent = (torch.cuda.Stream(device=device), torch.cuda.Event())
            Side._streams[self.key] = ent
        self.side, self.fork = ent
        self.reads = reads

    def __enter__(self):
        self.fork.record(self.main)
        self.side.wait_event(self.fork)
        for t in self.reads:
            if t is not None:
                t.record_stream(self.side)
        Side._pending[self.key] = (self.main, self.side)
        # set_stream directly: cheaper on the host than the torch.cuda.stream() context manager
        torch.cuda.set_stream(self.side)
        Side.depth += 1
        return self

    def __exit__(self, *exc):
        Side.depth -= 1
        torch.cuda.set_stream(self.main)
        task = torch._C._current_graph_task_id()
        if task < 0:
            Side.sync()         # not inside a backward pass: nothing will call back, so join here
        elif task != Side._callback_task:
            torch.autograd.Variable._execution_engine.queue_callback(Side.sync)
            Side._callback_task = task
        return False

    @classmethod
    def sync(cls):
        """The main streams wait for their side streams (no host synchronisation)."""
        if cls._pending:
            for main, side in cls._pending.values():
                main.wait_stream(side)
            cls._pending.clear()


class _Inline:
    """Stand-in for Side when the launches stay on the caller's stream."""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def side_scope(device, params, reads, work=None):
    """work: elements the side kernels stream (None = unknown: fork). Below Side.MIN_WORK the fork is not worth its
    hand-over: an event record + wait each way costs the host ~10 us and the device a queue switch, which kernels of
    a few microseconds cannot hide (eager step at batch 8: 20.0 ms forking everywhere, 15.6 ms on one stream)."""
    if work is not None and work < Side.MIN_WORK:
        return _Inline()
    ok = params if isinstance(params, bool) else Side.usable(params)   # a block's backward asks once and passes the answer
    return Side(device, reads) if ok else _Inline()


class KernelTimer:
    """Optional per-family HIP-event timing of the kernel launches (events are recorded on the stream the
    kernels are launched on). Off by default; bench.py turns it on for the family it prices against the
    roofline. `bytes` is the algorithmic HBM traffic of the launch (operands read once, results written once)."""
    families = None   # None = off, else a set of family names
    records = {}      # family -> list of (start_event, end_event, bytes, launched on the side stream)

    @classmethod
    def start(cls, families):
        cls.families, cls.records = set(families), {}

    @classmethod
    def stop(cls):
        """-> {family: (launches, total_ms, total_bytes, launches on the side stream)}; synchronises."""
        torch.cuda.synchronize()
        out = {}
        for fam, recs in cls.records.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            out[fam] = (len(recs), ms, sum(r[2] for r in recs), sum(1 for r in recs if r[3]))
        cls.families, cls.records = None, {}
        return out


def _launch(family, nbytes, fn, *args):
    if KernelTimer.families is None or family not in KernelTimer.families:
        return fn(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = fn(*args)
    b.record()
    KernelTimer.records.setdefault(family, []).append((a, b, nbytes, Side.depth > 0))
    return r


# ------------------------------------------------------------------------------------------------
# primitive wrappers
# ------------------------------------------------------------------------------------------------
def gemm_nt(a2d, b, out2d=None, *, pscale=None, pshift=None, pro_act=ACT_NONE, gate=None, rows_per_img=0,
            bias=None, epi_act=ACT_NONE, add=None, stats_rows=0, ldc=None):
    """out[M,N] = pro(a)[M,K] @ b[N,K]^T ...; returns (out, part or None)."""
    _require_cuda(a2d)
    M, K = a2d.shape
    N = b.shape[0]
    assert b.shape[1] == K and a2d.dtype == b.dtype, (a2d.shape, b.shape, a2d.dtype, b.dtype)
    if out2d is None:
        out2d = torch.empty((M, N), dtype=a2d.dtype, device=a2d.device)
    part = None
    if stats_rows:
        part = torch.empty((stats_rows, 2, N), dtype=torch.float32, device=a2d.device)
    esz = a2d.element_size()
    _launch("gemm_nt", (M * K + N * K + M * N * (2 if add is not None else 1)) * esz, lib().mia_gemm_nt, _DT[a2d.dtype], _p(a2d), a2d.stride(0), _p(b), b.stride(0), _p(out2d),
                      ldc if ldc is not None else out2d.stride(0), _p(add), M, N, K, _p(pscale), _p(pshift), pro_act,
                      _p(gate), rows_per_img, _p(bias), epi_act, _p(part), stats_rows, _stream())
    return out2d, part


def gemm_tn(g2d, x2d, *, pscale=None, pshift=None, pro_act=ACT_NONE, gate=None, rows_per_img=0):
    """out[N,K] (fp32) = g[M,N]^T @ pro(x)[M,K]; the split reduction over M is added in a fixed order."""
    M, N = g2d.shape
    K = x2d.shape[1]
    assert x2d.shape[0] == M and g2d.dtype == x2d.dtype
    _require_cuda(g2d)
    dt = _DT[g2d.dtype]
    pro = 1 if (pscale is not None or gate is not None) else 0
    out = torch.empty((N, K), dtype=torch.float32, device=g2d.device)
    nws = lib().mia_gemm_tn_ws_floats(dt, M, N, K, pro)
    ws = torch.empty((nws,), dtype=torch.float32, device=g2d.device) if nws > 0 else None
    _launch("gemm_tn", (M * N + M * K) * g2d.element_size() + N * K * 4, lib().mia_gemm_tn, dt, _p(g2d), g2d.stride(0), _p(x2d), x2d.stride(0), _p(out), out.stride(0), M, N, K,
                      _p(pscale), _p(pshift), pro_act, _p(gate), rows_per_img, _p(ws), nws if nws > 0 else -1, _stream())
    return out


def dwconv_fwd(x, w, k, s, *, pscale=None, pshift=None, stats_rows=0):
    B, H, W, C = x.shape
    pad = (k - 1) // 2
    OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    y = torch.empty((B, OH, OW, C), dtype=x.dtype, device=x.device)
    if stats_rows:   # exactly the rows the launch fills: no zero-fill launch for the rest
        stats_rows = lib().mia_dwconv_part_rows(_DT[x.dtype], 0, B, H, W, C, k, s, stats_rows)
    part = torch.empty((stats_rows, 2, C), dtype=torch.float32, device=x.device) if stats_rows else None
    _launch("dwconv_fwd", (x.numel() + y.numel()) * x.element_size(), lib().mia_dwconv_fwd, _DT[x.dtype], _p(x), B, H, W, C, _p(pscale), _p(pshift), _p(w), k, s, _p(y), _p(part),
                         stats_rows, _stream())
    return y, part


def dwconv_bwd_data(dy, w, k, s, in_hw, *, yfwd=None, escale=None, eshift=None, emean=None, erstd=None, stats_rows=0):
    B, OH, OW, C = dy.shape
    H, W = in_hw
    dx = torch.empty((B, H, W, C), dtype=dy.dtype, device=dy.device)
    if stats_rows:
        stats_rows = lib().mia_dwconv_part_rows(_DT[dy.dtype], 1, B, H, W, C, k, s, stats_rows)
    part = torch.empty((stats_rows, 2, C), dtype=torch.float32, device=dy.device) if stats_rows else None
    _launch("dwconv_bwd_data", (dy.numel() + dx.numel() * (2 if yfwd is not None else 1)) * dy.element_size(),
            lib().mia_dwconv_bwd_data, _DT[dy.dtype], _p(dy), B, H, W, C, _p(w), k, s, _p(dx), _p(yfwd), _p(escale),
                              _p(eshift), _p(emean), _p(erstd), _p(part), stats_rows, _stream())
    return dx, part


def dwconv_bwd_data_lin_ok(dtype, C, k, s, Cin):
    return bool(lib().mia_dwconv_bwd_data_lin_supported(_DT[dtype], C, k, s, Cin))


def dwconv_bwd_data_lin(dy, w, k, s, x, *, yfwd, st, gamma, wexp, stats_rows):
    """The stride-2 data gradient g of mia_dwconv_bwd_data (fused silu' epilogue, st = the expansion BatchNorm's
    (scale, shift, mean, rstd)) consumed in the same kernel, never stored:
    -> (dxp [M,Cin] = g . ((gamma*rstd) o wexp), t1 [C,Cin] fp32 = g^T x, partial sums [rows,2,C])."""
    B, OH, OW, C = dy.shape
    _, H, W, Cin = x.shape
    dev = dy.device
    rows = lib().mia_dwconv_bwd_data_lin_rows(B, H, W, k, C, stats_rows)
    part = torch.empty((rows, 2, C), dtype=torch.float32, device=dev)
    ws = torch.empty((rows * C * Cin,), dtype=torch.float32, device=dev)
    dxp = torch.empty((B * H * W, Cin), dtype=dy.dtype, device=dev)
    t1 = torch.empty((C, Cin), dtype=torch.float32, device=dev)
    _launch("dwconv_bwd_data_lin", (dy.numel() + yfwd.numel() + 2 * x.numel()) * dy.element_size(),
            lib().mia_dwconv_bwd_data_lin, _DT[dy.dtype], _p(dy), B, H, W, C, _p(w), k, s, _p(yfwd), _pr(st, 0), _pr(st, 1),
            _pr(st, 2), _pr(st, 3), _p(gamma), _p(wexp), _p(x), Cin, _p(dxp), _p(t1), _p(part), rows, _p(ws), ws.numel(),
            _stream())
    return dxp, t1, part


def dwconv_bwd_weight(x, dy, k, s, *, pscale=None, pshift=None):
    B, H, W, C = x.shape
    dw = torch.empty((C, 1, k, k), dtype=torch.float32, device=x.device)
    ncb = -(-C // (64 if C % 64 == 0 else 48 if C % 48 == 0 else min(C, 64)))
    ws = torch.empty((max(8, min(256, 1024 // ncb)) * C * k * k,), dtype=torch.float32, device=x.device)
    _launch("dwconv_bwd_weight", (x.numel() + dy.numel()) * x.element_size(), lib().mia_dwconv_bwd_weight,
            _DT[x.dtype], _p(x), B, H, W, C, _p(pscale), _p(pshift), _p(dy), k, s, _p(dw), _p(ws), ws.numel(),
            _stream())
    return dw


class BNState:
    """The parameters/buffers of one BatchNorm layer plus mode; plain holder."""
    __slots__ = ("weight", "bias", "running_mean", "running_var", "nbt", "momentum", "eps", "training")

    def __init__(self, bn: torch.nn.Module):
        self.weight, self.bias = bn.weight, bn.bias
        self.running_mean, self.running_var, self.nbt = bn.running_mean, bn.running_var, bn.num_batches_tracked
        self.momentum, self.eps = bn.momentum, bn.eps
        self.training = bn.training or bn.running_mean is None


def bn_finalize_fwd(part, count, bn: BNState, conv_bias=None):
    C = bn.weight.shape[0]
    dev = bn.weight.device
    out = torch.empty((4, C), dtype=torch.float32, device=dev)
    P = part.shape[0] if part is not None else 0
    if bn.training and bn.running_mean is not None:
        WeightEpoch.bump()      # the kernel updates the running statistics in place
    lib().mia_bn_finalize_fwd(_p(part), P, C, float(count), _p(bn.weight), _p(bn.bias), _p(bn.running_mean),
                              _p(bn.running_var), _p(bn.nbt), float(bn.momentum), float(bn.eps),
                              1 if bn.training else 0, _p(conv_bias), _pr(out, 0), _pr(out, 1), _pr(out, 2), _pr(out, 3),
                              _stream())
    return out  # rows: scale, shift, mean, rstd


def bn_finalize_bwd(part, count, gamma, stats, training):
    C = gamma.shape[0]
    coef = torch.empty((6, C), dtype=torch.float32, device=gamma.device)  # dgamma, dbeta, ca, cb, cd, dconv_bias
    lib().mia_bn_finalize_bwd(_p(part), part.shape[0], C, float(count), _p(gamma), _pr(stats, 2), _pr(stats, 3),
                              1 if training else 0, _pr(coef, 0), _pr(coef, 1), _pr(coef, 2), _pr(coef, 3), _pr(coef, 4),
                              _pr(coef, 5), _stream())
    return coef


def bn_act_bwd_reduce(g, y, stats, act, *, gate=None, dpool=None, rows_per_img=0, out=None):
    """-> (g', partials) with g' = (g*gate + dpool) * act'(bn(y)). g' goes to `out` (default: a fresh tensor; pass
    out=g to update in place when the caller owns g -- an incoming autograd gradient is NOT the caller's to change).
    With neither activation nor gate nor dpool g' is g itself and nothing is written."""
    C = g.shape[-1]
    R = g.numel() // C
    rows = part_rows((R + 63) // 64)
    part = torch.empty((rows, 2, C), dtype=torch.float32, device=g.device)
    writes = act != ACT_NONE or gate is not None or dpool is not None
    if not writes:
        out = g
    elif out is None:
        out = torch.empty_like(g)
    _launch("bn_act_bwd_reduce", g.numel() * g.element_size() * (3 if writes else 2), lib().mia_bn_act_bwd_reduce,
            _DT[g.dtype], _p(g), _p(out) if writes else None, _p(y), R, C, _pr(stats, 0), _pr(stats, 1), _pr(stats, 2),
            _pr(stats, 3), act, _p(gate), _p(dpool), rows_per_img, _p(part), rows, _stream())
    return out, part


def affine2(x, a, d, *, y=None, b=None, res=None, act=ACT_NONE, out=None):
    C = x.shape[-1]
    R = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    _launch("affine2", x.numel() * x.element_size() * (2 + (y is not None) + (res is not None)), lib().mia_affine2,
            _DT[x.dtype], _p(x), _p(y), _p(res), _p(out), R, C, _p(a), _p(b), _p(d), act, _stream())
    return out


def _img_splits(B, HW, C):
    """Row splits per image of the per-image reductions: ~2048 workgroups of >= 64 rows each on the large maps; ONE
    on maps of up to 400 pixels (19x19, 10x10), where the squeeze-excite gate kernels then take the per-image row as
    it is and need no launch to add splits up."""
    if HW <= 400:
        return 1
    ncb = -(-C // 256)
    return max(1, min(HW // 64, 64, -(-2048 // (B * ncb))))


def img_colsum(y, wt, scale, shift, act):
    B, H, W, C = y.shape
    HW = H * W
    splits = _img_splits(B, HW, C)
    out = torch.empty((B, splits, C), dtype=torch.float32, device=y.device)
    _launch("img_colsum", y.numel() * y.element_size() * (2 if wt is not None else 1), lib().mia_img_colsum,
            _DT[y.dtype], _p(y), _p(wt), B, HW, C, _p(scale), _p(shift), act, splits, _p(out), _stream())
    return out


def se_bwd_sums(y, dz, stats, gate=None):
    """One pass over (y, dz): [5, B, splits, C] per-image sums for the squeeze-excite and BatchNorm backward.
    gate [B, C]: on maps that take one split per image, plane 0 comes out multiplied by gate*(1-gate), i.e. as the
    dpre2 of se_gate_bwd (which recognises it by out.pre_scaled); on larger maps the argument is ignored."""
    B, H, W, C = y.shape
    HW = H * W
    splits = _img_splits(B, HW, C)
    out = torch.empty((5, B, splits, C), dtype=torch.float32, device=y.device)
    pre = gate is not None and splits == 1
    _launch("se_bwd_sums", y.numel() * y.element_size() * 2, lib().mia_se_bwd_sums, _DT[y.dtype], _p(y), _p(dz), B, HW,
            C, _pr(stats, 0), _pr(stats, 1), _pr(stats, 2), _pr(stats, 3), splits, _p(out), _p(gate) if pre else None,
            _stream())
    out.pre_scaled = pre
    return out


def se_bn_combine(sums, gate, dpool):
    _, B, splits, C = sums.shape
    part = torch.empty((B, 2, C), dtype=torch.float32, device=sums.device)
    lib().mia_se_bn_combine(_p(sums), B, splits, C, _p(gate), _p(dpool), _p(part), _stream())
    return part


def bn_act_bwd_apply(g, y, stats, act, coef, *, gate=None, dpool=None, rows_per_img=0):
    """In place on g: dy = ca*[act'(z)*(g*gate + dpool)] + cb*y + cd."""
    C = g.shape[-1]
    R = g.numel() // C
    _launch("bn_act_bwd_apply", g.numel() * g.element_size() * 3, lib().mia_bn_act_bwd_apply, _DT[g.dtype], _p(g), _p(y),
            R, C, _pr(stats, 0), _pr(stats, 1), act, _p(gate), _p(dpool), rows_per_img, _pr(coef, 2), _pr(coef, 3),
            _pr(coef, 4), _stream())
    return g


def se_gate_fwd(pool_part, hw, wr, br, weT, be):
    B, splits, C = pool_part.shape
    RD = wr.shape[0]
    dev = pool_part.device
    pooled = torch.empty((B, C), dtype=torch.float32, device=dev)
    hpre = torch.empty((B, RD), dtype=torch.float32, device=dev)
    gate = torch.empty((B, C), dtype=torch.float32, device=dev)
    lib().mia_se_gate_fwd(_p(pool_part), B, splits, hw, C, RD, _p(wr), _p(br), _p(weT), _p(be), _p(pooled), _p(hpre),
                          _p(gate), _stream())
    return pooled, hpre, gate


def se_gate_bwd(dgate_part, hw, wr, weT, gate, hpre, pooled, pre_scaled=False):
    """Backward through the gate: -> (dpool, dpre2, dpre1); the parameter gradients come from se_param_grad.
    pre_scaled: dgate_part [B,1,C] already is dpre2 (se_bwd_sums with its gate argument)."""
    B, splits, C = dgate_part.shape
    RD = wr.shape[0]
    dev = dgate_part.device
    dpre2 = dgate_part.view(B, C) if pre_scaled else torch.empty((B, C), dtype=torch.float32, device=dev)
    dpre1 = torch.empty((B, RD), dtype=torch.float32, device=dev)
    dpool = torch.empty((B, C), dtype=torch.float32, device=dev)
    lib().mia_se_gate_bwd(_p(dgate_part), B, splits, hw, C, RD, _p(wr), _p(weT), _p(gate), _p(hpre), _p(pooled),
                          _p(dpre2), _p(dpre1), _p(dpool), None, None, None, None, _stream())
    return dpool, dpre2, dpre1


def se_param_grad(dpre2, dpre1, pooled, hpre):
    B, C = dpre2.shape
    RD = dpre1.shape[1]
    dev = dpre2.device
    dwr = torch.empty((RD, C), dtype=torch.float32, device=dev)
    dbr = torch.empty((RD,), dtype=torch.float32, device=dev)
    dwe = torch.empty((C, RD), dtype=torch.float32, device=dev)
    dbe = torch.empty((C,), dtype=torch.float32, device=dev)
    lib().mia_se_param_grad(B, C, RD, _p(dpre2), _p(dpre1), _p(pooled), _p(hpre), _p(dwr), _p(dbr), _p(dwe), _p(dbe),
                            _stream())
    return dwr, dbr, dwe, dbe


def bn_lin_prep(w, coef):
    """w [Cm,Cin] (compute dtype), coef rows (.., .., ca, cb, cd, ..) -> (WcaT [Cin,Cm], S [Cin,Cin], r [Cin] fp32)."""
    Cm, Cin = w.shape
    wcaT = torch.empty((Cin, Cm), dtype=w.dtype, device=w.device)
    S = torch.empty((Cin, Cin), dtype=w.dtype, device=w.device)
    r = torch.empty((Cin,), dtype=torch.float32, device=w.device)
    lib().mia_bn_lin_prep(_DT[w.dtype], _p(w), Cm, Cin, _pr(coef, 2), _pr(coef, 3), _pr(coef, 4), _p(wcaT), _p(S), _p(r),
                          _stream())
    return wcaT, S, r


def bn_lin_fix(t1, w, coef, G, cs):
    """In place on t1 = g^T x [Cm,Cin] fp32 -> the weight gradient of the convolution in front of the BatchNorm."""
    Cm, Cin = w.shape
    lib().mia_bn_lin_fix(_DT[w.dtype], _p(w), Cm, Cin, _pr(coef, 2), _pr(coef, 3), _pr(coef, 4), _p(G), _p(cs), _p(t1),
                         _stream())
    return t1


class LinBN:
    """BatchNorm backward carried through the expansion convolution (mia_bn_lin_prep / mia_bn_lin_fix) where the
    expanded tensor is large and the block input narrow: there the three passes over the expanded tensor that
    applying dy = ca*g + cb*y + cd costs are the most expensive thing in the block's backward."""
    enabled = True
    # rows x expanded channels. Measured on the B3 step at batch 128 (ms per step): off 24.9; from 300 M (block 1.0
    # only) 24.56; from 100 M (blocks 1.0-2.0) 24.23; from 40 M (also the 38x38 blocks) 25.1 - below ~100 M elements the
    # extra small launches cost more than the three passes they save
    MIN_ELEMS = 100_000_000
    MAX_CIN = 64
    # stride-2 blocks: compute the two products inside the depthwise data-gradient kernel (mia_dwconv_bwd_data_lin)
    fused = True

    @classmethod
    def use(cls, M, Cm, Cin):
        return cls.enabled and Cin <= cls.MAX_CIN and M * Cm >= cls.MIN_ELEMS


def stem_im2col(x, dtype):
    """x [B,3,H,W]: float32 in NCHW order (what the reference's loaders hand over), or already in the compute type in
    channels-last order (preprocess.resize_normalize(..., out_dtype=...)), which is read without any conversion."""
    B, _, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    col = torch.empty((B * OH * OW, 32), dtype=dtype, device=x.device)
    if x.dtype == dtype and x.stride() == (3 * H * W, 1, 3 * W, 3):
        lib().mia_stem_im2col_nhwc(_DT[dtype], _p(x), B, H, W, _p(col), _stream())
        return col, OH, OW
    x = x.contiguous().float()
    lib().mia_stem_im2col(_DT[dtype], _p(x), B, H, W, _p(col), _stream())
    return col, OH, OW


def cast_weight(w2d, dtype, *, pad_to=None, want=True, want_t=False):
    """fp32 [R,Cc] -> (copy [R,ld] or None, transpose [Cc,R] or None) in the compute dtype."""
    R, Cc = w2d.shape
    ld = pad_to or Cc
    dst = dstT = None
    if want:
        if dtype == torch.float32 and ld == Cc:
            dst = w2d
        else:
            dst = torch.empty((R, ld), dtype=dtype, device=w2d.device)
    if want_t:
        dstT = torch.empty((Cc, R), dtype=dtype, device=w2d.device)
    if (dst is not None and dst is not w2d) or dstT is not None:
        lib().mia_cast_pad_transpose(_DT[dtype], _p(w2d), R, Cc, _p(dst if dst is not w2d else None), ld, _p(dstT), R,
                                     _stream())
    return dst, dstT


class WeightCaster:
    """All weight casts of one forward in one or two launches (mia_cast_pad_transpose_multi) instead of one per
    matrix. Built once for a fixed list of fp32 master matrices; every call writes the compute-dtype copies (and,
    for the backward, their transposes) into one fresh flat buffer and returns per-matrix views.
    specs: list of (w2d fp32 [R, Cc] contiguous view of a parameter, want copy, want transpose, pad_to or None)."""

    def __init__(self, specs, dtype):
        self.dtype = dtype
        self.specs = specs
        self.key = tuple((w.data_ptr(), tuple(w.shape), want, want_t, pad) for (w, want, want_t, pad) in specs)
        esz_align = 64
        self.items = []   # (index into specs, R, Cc, ld or 0, offset of copy or -1, offset of transpose or -1)
        off = 0
        for i, (w, want, want_t, pad) in enumerate(specs):
            R, Cc = w.shape
            ld = pad or Cc
            alias = want and dtype == torch.float32 and ld == Cc      # the fp32 master itself serves as the copy
            o1 = o2 = -1
            if want and not alias:
                o1 = off
                off += (R * ld + esz_align - 1) // esz_align * esz_align
            if want_t:
                o2 = off
                off += (Cc * R + esz_align - 1) // esz_align * esz_align
            self.items.append((i, R, Cc, ld, o1, o2, alias))
        self.total = off
        self.work = [it for it in self.items if it[4] >= 0 or it[5] >= 0]
        n = len(self.work)
        self.n = n
        if n:
            vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
            self.src = vp(*[specs[it[0]][0].data_ptr() for it in self.work])
            self.rows = ip(*[it[1] for it in self.work])
            self.cols = ip(*[it[2] for it in self.work])
            self.ldd = ip(*[it[3] for it in self.work])
            self.ldt = ip(*[it[1] for it in self.work])
            self.dst, self.dstT = vp(), vp()

    def matches(self, specs):
        return len(specs) == len(self.specs) and self.key == tuple(
            (w.data_ptr(), tuple(w.shape), want, want_t, pad) for (w, want, want_t, pad) in specs)

    def __call__(self):
        """-> list of (copy or None, transpose or None), one per spec."""
        out = [None] * len(self.specs)
        if self.n == 0:
            return [(w if want else None, None) for (w, want, _, _) in self.specs]
        dev = self.specs[0][0].device
        flat = torch.empty((self.total,), dtype=self.dtype, device=dev)
        base, esz = flat.data_ptr(), flat.element_size()
        for k, (i, R, Cc, ld, o1, o2, alias) in enumerate(self.work):
            self.dst[k] = base + o1 * esz if o1 >= 0 else None
            self.dstT[k] = base + o2 * esz if o2 >= 0 else None
        lib().mia_cast_pad_transpose_multi(_DT[self.dtype], self.n, self.src, self.rows, self.cols, self.dst, self.ldd,
                                           self.dstT, self.ldt, _stream())
        for (i, R, Cc, ld, o1, o2, alias) in self.items:
            w, want, want_t, _ = self.specs[i]
            c = w if alias else (flat[o1:o1 + R * ld].view(R, ld) if o1 >= 0 else None)
            t = flat[o2:o2 + Cc * R].view(Cc, R) if o2 >= 0 else None
            out[i] = (c, t)
        return out


# ------------------------------------------------------------------------------------------------
# block-level autograd Functions. Inputs/outputs are NHWC tensors in the compute dtype.
# `cfg` is a small python object carried outside autograd (BN buffers, geometry, flags).
# ------------------------------------------------------------------------------------------------
def _w2d(w):
    return w.detach().reshape(w.shape[0], -1)


class StemFn(torch.autograd.Function):
    """conv_stem (3x3 s2) -> bn1 -> SiLU, from an NCHW fp32 image to an NHWC activation."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, cfg):
        B = x.shape[0]
        col, OH, OW = stem_im2col(x, cfg.dtype)
        wp, _ = cast_weight(_w2d(w), cfg.dtype, pad_to=32)
        M = B * OH * OW
        bn = cfg.bn
        rows = part_rows((M + 255) // 256) if bn.training else 0
        y0, part = gemm_nt(col, wp, stats_rows=rows)
        st = bn_finalize_fwd(part, M, bn)
        out = affine2(y0, st[0], st[1], act=ACT_SILU).view(B, OH, OW, -1)
        ctx.cfg, ctx.training = cfg, bn.training
        ctx.save_for_backward(col, y0, st, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        col, y0, st, gamma = ctx.saved_tensors
        M = y0.shape[0]
        # into a fresh tensor: the incoming gradient belongs to autograd (a hook or retain_grad() may hold it)
        g, part = bn_act_bwd_reduce(dout.contiguous().view(y0.shape), y0, st, ACT_SILU)
        coef = bn_finalize_bwd(part, M, gamma, st, ctx.training)
        affine2(g, coef[2], coef[4], y=y0, b=coef[3], out=g)
        dw = gemm_tn(g, col)  # [40, 32]
        return None, dw[:, :27].reshape(-1, 3, 3, 3), coef[0], coef[1], None


class MBConvFn(torch.autograd.Function):
    """One EfficientNet block. cfg.kind 'ir': conv_pw-bn1-SiLU-conv_dw-bn2-SiLU-SE-conv_pwl-bn3(+x);
    'ds': conv_dw-bn1-SiLU-SE-conv_pw-bn2(+x)."""

    @staticmethod
    def forward(ctx, x, w_exp, g1, b1, w_dw, g2, b2, w_ser, b_ser, w_see, b_see, w_proj, g3, b3, cfg):
        x = x.contiguous()
        B, H, W, Cin = x.shape
        dt = x.dtype
        k, s = cfg.k, cfg.s
        training = cfg.bn_dw.training
        need_bwd = any(ctx.needs_input_grad)   # inference: skip the transposed weight copies
        # ---- expansion (ir only) ----
        if cfg.kind == "ir":
            M = B * H * W
            # [Cm, Cin] and, for the backward, [Cin, Cm]; cast by the trunk in one launch for all blocks when it can
            wexp, wexpT = cfg.pre["exp"] if cfg.pre else cast_weight(_w2d(w_exp), dt, want_t=need_bwd)
            y1, part = gemm_nt(x.view(M, Cin), wexp, stats_rows=part_rows((M + 63) // 64) if training else 0)
            st1 = bn_finalize_fwd(part, M, cfg.bn_exp)
            y1 = y1.view(B, H, W, -1)
            dw_in, dw_sc, dw_sh = y1, st1[0], st1[1]
        else:
            y1, st1, wexpT = None, None, None
            dw_in, dw_sc, dw_sh = x, None, None
        Cm = dw_in.shape[-1]
        # ---- depthwise ----
        pad = (k - 1) // 2
        OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        M2 = B * OH * OW
        wdw = w_dw.detach().reshape(Cm, k * k)
        y2, part = dwconv_fwd(dw_in, wdw, k, s, pscale=dw_sc, pshift=dw_sh,
                              stats_rows=part_rows(B * ((OH * OW + 127) // 128)) if training else 0)
        st2 = bn_finalize_fwd(part, M2, cfg.bn_dw)
        # ---- squeeze-excite ----
        pool_part = img_colsum(y2, None, st2[0], st2[1], ACT_SILU)
        wr = _w2d(w_ser)
        _, weT = cfg.pre["se"] if cfg.pre else cast_weight(_w2d(w_see), torch.float32, want=False, want_t=True)  # [RD, Cm]
        pooled, hpre, gate = se_gate_fwd(pool_part, OH * OW, wr, b_ser.detach(), weT, b_see.detach())
        # ---- projection ----
        wproj, wprojT = cfg.pre["proj"] if cfg.pre else cast_weight(_w2d(w_proj), dt, want_t=need_bwd)   # [Co, Cm], [Cm, Co]
        y3, part = gemm_nt(y2.view(M2, Cm), wproj, pscale=st2[0], pshift=st2[1], pro_act=ACT_SILU, gate=gate,
                           rows_per_img=OH * OW, stats_rows=part_rows((M2 + 63) // 64) if training else 0)
        st3 = bn_finalize_fwd(part, M2, cfg.bn_proj)
        Co = y3.shape[1]
        y3 = y3.view(B, OH, OW, Co)
        out = affine2(y3, st3[0], st3[1], res=x if cfg.has_skip else None)
        ctx.cfg, ctx.training = cfg, training
        # casts made once per step: the transposed copies serve the data-gradient GEMMs of the backward
        ctx.wexpT, ctx.wprojT, ctx.weT = wexpT, wprojT, weT
        ctx.wexp = wexp if (cfg.kind == "ir" and need_bwd) else None
        ctx.side_use = Side.Use((w_exp, w_dw, w_proj, w_ser, w_see)) if need_bwd else None
        ctx.save_for_backward(x, y1, st1, y2, st2, y3, st3, pooled, hpre, gate, w_exp, g1, w_dw, g2, w_ser, w_see,
                              w_proj, g3)
        return out

    @staticmethod
    def backward(ctx, dout):
        try:
            return MBConvFn._backward(ctx, dout)
        finally:
            if ctx.side_use is not None:
                ctx.side_use.release()

    @staticmethod
    def _backward(ctx, dout):
        (x, y1, st1, y2, st2, y3, st3, pooled, hpre, gate, w_exp, g1, w_dw, g2, w_ser, w_see, w_proj,
         g3) = ctx.saved_tensors
        cfg, training = ctx.cfg, ctx.training
        dout = dout.contiguous()
        B, H, W, Cin = x.shape
        _, OH, OW, Cm = y2.shape
        Co = y3.shape[-1]
        M, M2 = B * H * W, B * OH * OW
        dt = x.dtype
        k, s = cfg.k, cfg.s
        # ---- bn3 (no activation) ----
        _, part = bn_act_bwd_reduce(dout, y3, st3, ACT_NONE)
        c3 = bn_finalize_bwd(part, M2, g3, st3, training)
        dy3 = affine2(dout, c3[2], c3[4], y=y3, b=c3[3]).view(M2, Co)
        # ---- projection: weight and data gradients ----
        work = M2 * Cm
        # one answer for the whole block: none of these parameters changes state while this backward runs
        wparams = work >= Side.MIN_WORK and Side.usable((w_exp, w_dw, w_proj, w_ser, w_see))
        with side_scope(x.device, wparams, (dy3, y2, st2, gate), work):
            dwproj = gemm_tn(dy3, y2.view(M2, Cm), pscale=st2[0], pshift=st2[1], pro_act=ACT_SILU, gate=gate,
                             rows_per_img=OH * OW)
        dz, _ = gemm_nt(dy3, ctx.wprojT)
        dz = dz.view(B, OH, OW, Cm)
        # ---- squeeze-excite backward ----
        sums = se_bwd_sums(y2, dz, st2, gate=gate)
        wr, weT = _w2d(w_ser), ctx.weT
        dpool, dpre2, dpre1 = se_gate_bwd(sums[0], OH * OW, wr, weT, gate, hpre, pooled, pre_scaled=sums.pre_scaled)
        with side_scope(x.device, wparams, (dpre2, dpre1, pooled, hpre), work):
            dwr, dbr, dwe, dbe = se_param_grad(dpre2, dpre1, pooled, hpre)
        # ---- bn after depthwise (+SiLU): its reductions come from the same per-image sums ----
        c2 = bn_finalize_bwd(se_bn_combine(sums, gate, dpool), M2, g2, st2, training)
        dy2 = bn_act_bwd_apply(dz, y2, st2, ACT_SILU, c2, gate=gate, dpool=dpool, rows_per_img=OH * OW)
        wdw = w_dw.detach().reshape(Cm, k * k)
        rows_in = part_rows(B * ((H * W + 127) // 128))
        if cfg.kind == "ir":
            # depthwise data gradient with fused silu' and bn1 partials
            with side_scope(x.device, wparams, (y1, dy2, st1), work):
                ddw = dwconv_bwd_weight(y1, dy2, k, s, pscale=st1[0], pshift=st1[1])
            x2d = x.view(M, Cin)
            if (LinBN.fused and not cfg.has_skip and LinBN.use(M, Cm, Cin)
                    and dwconv_bwd_data_lin_ok(dt, Cm, k, s, Cin)):
                # the expanded gradient g is consumed inside the kernel that computes it: only its images under the
                # two linear maps of the expansion's backward (and the BatchNorm sums) reach memory
                dxp, t1, part = dwconv_bwd_data_lin(dy2, wdw, k, s, x, yfwd=y1, st=st1, gamma=g1, wexp=ctx.wexp,
                                                    stats_rows=rows_in)
                c1 = bn_finalize_bwd(part, M, g1, st1, training)
                _, S, r = bn_lin_prep(ctx.wexp, c1)
                with side_scope(x.device, wparams, (t1, x, c1, ctx.wexp), work):
                    G = gemm_tn(x2d, x2d)
                    cs = img_colsum(x, None, None, None, ACT_NONE).sum((0, 1))
                    dwexp = bn_lin_fix(t1, ctx.wexp, c1, G, cs)
                dx, _ = gemm_nt(x2d, S, bias=r, add=dxp)
                dx = dx.view(B, H, W, Cin)
                return (dx, dwexp.view_as(w_exp), c1[0], c1[1], ddw, c2[0], c2[1], dwr.view_as(w_ser), dbr,
                        dwe.view_as(w_see), dbe, dwproj.view_as(w_proj), c3[0], c3[1], None)
            g1t, part = dwconv_bwd_data(dy2, wdw, k, s, (H, W), yfwd=y1, escale=st1[0], eshift=st1[1], emean=st1[2],
                                        erstd=st1[3], stats_rows=rows_in)
            c1 = bn_finalize_bwd(part, M, g1, st1, training)
            if LinBN.use(M, Cm, Cin):
                # dy1 = ca*g + cb*y1 + cd is never formed: both products that consume it are linear in it
                g2d = g1t.view(M, Cm)
                wcaT, S, r = bn_lin_prep(ctx.wexp, c1)
                with side_scope(x.device, wparams, (g1t, x, c1, ctx.wexp), work):
                    t1 = gemm_tn(g2d, x2d)
                    G = gemm_tn(x2d, x2d)
                    cs = img_colsum(x, None, None, None, ACT_NONE).sum((0, 1))
                    dwexp = bn_lin_fix(t1, ctx.wexp, c1, G, cs)
                xs, _ = gemm_nt(x2d, S, bias=r, add=dout.view(M, Cin) if cfg.has_skip else None)
                dx, _ = gemm_nt(g2d, wcaT, add=xs)
            else:
                dy1 = affine2(g1t, c1[2], c1[4], y=y1, b=c1[3], out=g1t).view(M, Cm)
                with side_scope(x.device, wparams, (dy1, x), work):
                    dwexp = gemm_tn(dy1, x2d)
                dx, _ = gemm_nt(dy1, ctx.wexpT, add=dout.view(M, Cin) if cfg.has_skip else None)
            dx = dx.view(B, H, W, Cin)
            return (dx, dwexp.view_as(w_exp), c1[0], c1[1], ddw, c2[0], c2[1], dwr.view_as(w_ser), dbr,
                    dwe.view_as(w_see), dbe, dwproj.view_as(w_proj), c3[0], c3[1], None)
        # ds block: the depthwise conv reads the block input directly
        with side_scope(x.device, wparams, (x, dy2), work):
            ddw = dwconv_bwd_weight(x, dy2, k, s)
        dx, _ = dwconv_bwd_data(dy2, wdw, k, s, (H, W))
        if cfg.has_skip:
            one = torch.ones((Cin,), dtype=torch.float32, device=dx.device)
            affine2(dx, one, torch.zeros_like(one), res=dout, out=dx)
        return (dx, None, None, None, ddw, c2[0], c2[1], dwr.view_as(w_ser), dbr, dwe.view_as(w_see), dbe,
                dwproj.view_as(w_proj), c3[0], c3[1], None)
