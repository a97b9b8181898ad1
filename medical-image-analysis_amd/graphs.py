"""HIP-graph replay of the eval-mode forward and of the whole training step. One B3 forward is ~400 kernel launches
and a training step ~1700; at small batches the host needs longer to enqueue them than the device to run them. The
launch sequence for a fixed input shape is captured once (torch.cuda.CUDAGraph records the library's launches because
they go to torch's current stream) and replayed with a single call.

In eval mode the forward has no random numbers and updates no buffers. The training step has two things that change
from step to step although a replay repeats every kernel argument: the dropout masks and AdamW's step-dependent
factors (bias corrections, learning rate). Both are read from a small block of device words that is rewritten before
every replay (mia_set_words): a uint64 added to every captured dropout seed, and one row of three floats per AdamW
launch."""
import ctypes
import struct

import torch

from . import ops
from ._lib import lib


class GraphedForward:
    """g = GraphedForward(model, task, dataset_name, example_input); logits = g(x) for x of the same shape/dtype.
    The model must be in eval mode and its weights must not be re-allocated afterwards (in-place updates such as
    load_state_dict are fine: the graph reads the parameters' storage at replay time, but the compute-dtype weight
    copies are part of the captured sequence, so they are refreshed on every replay)."""

    def __init__(self, model, task: str, dataset_name: str, example_input: torch.Tensor, warmup: int = 2):
        if model.training:
            raise RuntimeError("GraphedForward captures the eval-mode forward: call model.eval() first")
        if not example_input.is_cuda:
            raise RuntimeError("GraphedForward needs an input on an MI355X device; there is no CPU path")
        self.model, self.task, self.dataset_name = model, task, dataset_name
        self.static_in = example_input.detach().clone()
        side = torch.cuda.Stream(device=example_input.device)
        side.wait_stream(torch.cuda.current_stream(example_input.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):      # sizes the allocator's pools and loads every kernel before capture
                model(self.static_in, task, dataset_name)
        torch.cuda.current_stream(example_input.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = model(self.static_in, task, dataset_name)

    def __call__(self, x: torch.Tensor, copy_out: bool = True) -> torch.Tensor:
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype:
            raise ValueError(f"captured for input {tuple(self.static_in.shape)} {self.static_in.dtype}, "
                             f"got {tuple(x.shape)} {x.dtype}")
        if self.model.training:
            raise RuntimeError("the model was switched to train mode after capture")
        refresh = getattr(self.model, "refresh_inference_weights", None)
        if refresh is not None:
            refresh()      # BatchNorm-folded weights are kept outside the captured launches: rewritten in place when stale
        self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out.clone() if copy_out else self.static_out


class GraphedTrainStep:
    """One training step -- forward, loss, backward, AdamW update -- as a single graph replay:

        step = GraphedTrainStep(model, task, dataset_name, criterion, optimizer, example_x, example_y)
        for x, y in loader:
            loss = step(x, y)          # same shapes/dtypes as the examples; step.outputs holds the logits

    It replaces the loop body of the reference's train_epoch (train.py:241-262: zero_grad, forward, criterion,
    backward, optimizer.step) for a fixed batch shape; a ragged last batch goes through the eager path. The optimizer
    must be mia_amd.optim.AdamW; learning-rate changes made by a scheduler between steps are honoured (the rate is
    read from param_groups at every replay), as are the per-parameter step counts in the optimizer's state.
    Construction runs `warmup` eager steps on the example batch and then puts the model, its BatchNorm buffers and
    the optimizer state back as they were, so a graphed run starts from the same point as an eager one. Dropout masks
    are drawn from torch's CPU generator state at every replay (reproducible under torch.manual_seed) but are not
    the same masks an eager step would draw. Single process only: gradients are not all-reduced inside the graph.

    fork=True captures the weight-gradient kernels on ops.Side's second stream, as the eager step runs them: the fork
    and join events become edges of the graph, so a replay may run the two branches side by side. The capture stream
    is a stream of this object's own, used for the warm-up steps as well, so the stream pair exists before the
    capture begins. fork=False (default) records everything on one stream. Measured on an MI355X at batch 128
    (bench.py --graph): one stream 25.7 ms per step, forked 26.1 ms, eager two-stream launches 24.6 ms -- the graph
    executor of this ROCm release gains nothing from the forked branches and pays for their join nodes, so the fork is
    kept as a tested option (bit-identical results either way) and not made the default."""

    WORDS = 64           # device words: [0:2] seed offset (uint64), [4:] AdamW rows of 4 floats
    MAX_ROWS = (WORDS - 4) // 4

    def __init__(self, model, task: str, dataset_name: str, criterion, optimizer, example_input: torch.Tensor,
                 example_target: torch.Tensor, warmup: int = 2, fork: bool = False):
        from .optim import AdamW
        if not model.training:
            raise RuntimeError("GraphedTrainStep captures the train-mode step: call model.train() first")
        if not example_input.is_cuda or not example_target.is_cuda:
            raise RuntimeError("GraphedTrainStep needs its batch on an MI355X device; there is no CPU path")
        if not isinstance(optimizer, AdamW):
            raise TypeError("GraphedTrainStep needs mia_amd.optim.AdamW (its update reads device-side step factors)")
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size() > 1:
            raise RuntimeError("GraphedTrainStep is single-process: the gradient all-reduce is not captured")
        self.model, self.task, self.dataset_name = model, task, dataset_name
        self.criterion, self.optimizer = criterion, optimizer
        dev = example_input.device
        self.static_in = example_input.detach().clone()
        self.static_target = example_target.detach().clone()
        self.words = torch.zeros((self.WORDS,), dtype=torch.int32, device=dev)

        # ---- remember where training stands
        tensors = [p for p in model.parameters()] + [b for b in model.buffers()]
        saved = [t.detach().clone() for t in tensors]
        opt_params = [p for g in optimizer.param_groups for p in g["params"]]
        opt_saved = {p: {k: v.clone() for k, v in optimizer.state[p].items()} for p in opt_params
                     if len(optimizer.state.get(p, {}))}

        def one_step():
            optimizer.zero_grad(set_to_none=True)
            out = model(self.static_in, task, dataset_name)
            loss = criterion(out, self.static_target)
            loss.backward()
            optimizer.step()
            return out, loss

        self.fork = bool(fork) and ops.Side.enabled
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream(dev))
        side_was = ops.Side.enabled
        ops.Side.enabled = self.fork
        try:
            with torch.cuda.stream(cap):
                for _ in range(max(1, warmup)):  # allocates the optimizer state, sizes the pools, loads every kernel
                    one_step()
            torch.cuda.current_stream(dev).wait_stream(cap)
            torch.cuda.synchronize(dev)

            self.graph = torch.cuda.CUDAGraph()
            optimizer.zero_grad(set_to_none=True)
            ops.Replay.seed_off = self.words.data_ptr()
            optimizer.begin_capture(self.words.data_ptr() + 16, self.MAX_ROWS)
            ops.Side.capture_ok = self.fork
            try:
                with torch.cuda.graph(self.graph, stream=cap):
                    self.outputs, self.loss = one_step()
                    ops.Side.sync()     # a capture may only end with every forked branch joined (a no-op after AdamW.step)
            finally:
                ops.Side.capture_ok = False
                ops.Replay.seed_off = None
                self.rows = optimizer.end_capture()
        finally:
            ops.Side.enabled = side_was

        # ---- put everything back: neither the warm-up steps nor the capture count as training
        with torch.no_grad():
            for t, s0 in zip(tensors, saved):
                t.copy_(s0)
            for p in opt_params:
                st = optimizer.state.get(p)
                if not st:
                    continue
                if p in opt_saved:
                    st["exp_avg"].copy_(opt_saved[p]["exp_avg"])
                    st["exp_avg_sq"].copy_(opt_saved[p]["exp_avg_sq"])
                    st["step"] = opt_saved[p]["step"]
                else:
                    st["exp_avg"].zero_()
                    st["exp_avg_sq"].zero_()
                    st["step"] = torch.tensor(0.0)
                optimizer._pstep[p] = int(float(st["step"]))
        self._buf = (ctypes.c_uint * self.WORDS)()

    def __call__(self, x: torch.Tensor, target: torch.Tensor, copy_out: bool = True) -> torch.Tensor:
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype or \
                target.shape != self.static_target.shape or target.dtype != self.static_target.dtype:
            raise ValueError(f"captured for a batch {tuple(self.static_in.shape)} {self.static_in.dtype} / "
                             f"{tuple(self.static_target.shape)} {self.static_target.dtype}")
        if not self.model.training:
            raise RuntimeError("the model was switched to eval mode after capture")
        self.static_in.copy_(x, non_blocking=True)
        self.static_target.copy_(target, non_blocking=True)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        factors = self.optimizer.advance_captured(self.rows)
        n = 4 + len(factors)
        buf = self._buf
        buf[0], buf[1] = seed & 0xFFFFFFFF, seed >> 32
        packed = struct.unpack(f"{len(factors)}I", struct.pack(f"{len(factors)}f", *factors))
        for i, w in enumerate(packed):
            buf[4 + i] = w
        lib().mia_set_words(self.words.data_ptr(), n, buf, ops._stream())
        ops.WeightEpoch.bump()      # the replayed kernels rewrite parameters and BatchNorm buffers in place
        self.graph.replay()
        return self.loss.clone() if copy_out else self.loss
