"""AdamW on the device through the multi-tensor HIP kernel (mia_adamw_step).

Same constructor arguments, param_groups and state-dict layout (step / exp_avg / exp_avg_sq) as torch.optim.AdamW,
which the reference builds at train.py:193-197 and finetune.py:232-248, so LR schedulers
(ReduceLROnPlateau, train.py:199-201) and optimizer checkpoints (train.py:440) keep working."""
import ctypes
import math

import torch

from ._lib import lib
from .ops import Side, WeightEpoch, _stream


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}
        # per-parameter step counts as Python ints. torch.optim.AdamW increments one CPU tensor per parameter and
        # step (several hundred tiny host ops); here the parameters that share a count share ONE state["step"] tensor,
        # made once per step
        self._pstep = {}
        # set while graphs.GraphedTrainStep captures step(): (address of a device float table, list of launches);
        # each captured launch reads its step-dependent factors from one 4-float row of the table
        self._capture = None

    @staticmethod
    def step_factors(lr, b1, b2, weight_decay, k):
        """The three step-dependent factors of update k, formed in double as mia_adamw_step forms them."""
        return 1.0 - lr * weight_decay, lr / (1.0 - b1 ** k), 1.0 / math.sqrt(1.0 - b2 ** k)

    def begin_capture(self, table_ptr: int, max_rows: int):
        self._capture = (table_ptr, max_rows, [])

    def end_capture(self):
        """-> [(group index, [params])], one entry per table row, in the order the launches were captured."""
        rows = self._capture[2]
        self._capture = None
        return rows

    def advance_captured(self, rows):
        """Host-side bookkeeping of one replayed step: bumps the step counts of the captured launches and returns the
        table contents (4 floats per row) for that step at the groups' CURRENT learning rates."""
        out = []
        for gi, plist in rows:
            group = self.param_groups[gi]
            k = self._pstep[plist[0]] + 1
            shared = torch.tensor(float(k))
            for p in plist:
                self._pstep[p] = k
                self.state[p]["step"] = shared
            b1, b2 = group["betas"]
            out.extend(self.step_factors(float(group["lr"]), float(b1), float(b2), float(group["weight_decay"]), k))
            out.append(0.0)
        return out

    def _table(self, gi, ps):
        """ctypes pointer tables of one group, rebuilt only when the set of (param, grad) storages changes."""
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached[1]
        n = len(ps)
        arr = ctypes.c_void_p * n
        st = [self.state[p] for p in ps]
        tab = (n, arr(*[p.data_ptr() for p in ps]), arr(*[p.grad.data_ptr() for p in ps]),
               arr(*[s["exp_avg"].data_ptr() for s in st]), arr(*[s["exp_avg_sq"].data_ptr() for s in st]),
               (ctypes.c_longlong * n)(*[p.numel() for p in ps]))
        self._tables[gi] = (key, tab)
        return tab

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        Side.sync()     # weight gradients still running on the side stream are ordered before the update
        WeightEpoch.bump()      # the kernel rewrites the parameters through their raw pointers
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if (not p.is_cuda or not p.grad.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32
                        or not p.is_contiguous() or p.grad.is_sparse):
                    raise RuntimeError("mia_amd.optim.AdamW needs contiguous fp32 parameters on an MI355X device")
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
            by_step = {}
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                k = self._pstep.get(p)
                if k is None:
                    k = int(float(st["step"]))      # first step, or the first one after load_state_dict()
                k += 1
                self._pstep[p] = k
                by_step.setdefault(k, []).append(p)
            b1, b2 = group["betas"]
            # parameters that joined later (unfrozen for fine-tuning) carry their own step count, as in torch: one
            # launch per distinct count (a single launch in the usual case)
            for si, (k, plist) in enumerate(sorted(by_step.items())):
                shared = torch.tensor(float(k))
                for p in plist:
                    self.state[p]["step"] = shared
                n, tp, tg, tm, tv, sizes = self._table((gi, si), plist)
                dev_hyper = None
                if self._capture is not None:
                    ptr, max_rows, rows = self._capture
                    if len(rows) >= max_rows:
                        raise RuntimeError("too many distinct (group, step count) launches for one captured step")
                    if grad_scale != 1.0:
                        raise RuntimeError("grad_scale cannot be captured")
                    dev_hyper = ptr + 16 * len(rows)
                    rows.append((gi, plist))
                lib().mia_adamw_step(n, tp, tg, tm, tv, sizes, float(group["lr"]), float(b1), float(b2),
                                     float(group["eps"]), float(group["weight_decay"]), k, float(grad_scale), dev_hyper,
                                     _stream())
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._pstep.clear()     # the loaded step tensors are read back at the next step()
        self._tables.clear()
