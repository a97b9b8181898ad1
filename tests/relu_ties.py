"""ReLU ties in a gradient comparison between two fp32 implementations of one network.

Two correct forward passes differ by rounding (here ~5e-6 on a decoder pre-activation after 26 trunk blocks). A ReLU
input closer to zero than that lands on different sides in the two, so one side passes the incoming gradient at that
element and the other does not. The forward value hardly moves (the input is ~0 either way), but the gradient of every
parameter upstream shifts by that element's incoming gradient carried back through the network, and a decoder map has
pixels that carry several per cent of a layer's gradient. About one such element per step is expected (half a million
ReLU inputs of unit scale against a noise of a few 1e-6), so a per-tensor tolerance alone passes or fails by luck.

The comparison is made exact instead: the checker records every F.relu of the oracle's decoder, finds the inputs within
`tau` of zero whose incoming gradient is not negligible, and computes for each the gradient every parameter WOULD gain
(masked input) or lose (passed input) if that element's mask were flipped: one extra backward from that pre-activation.
The tested gradients must then equal the oracle's plus a 0/1 combination of those responses, the combination found by
least squares over all parameters and required to come out at 0 or 1."""
import contextlib

import numpy as np
import torch
import torch.nn.functional as TF


class _RecordingF:
    """Stands in for a module's `F`: F.relu also records its input (with graph) and the gradient of its output."""

    def __init__(self):
        self.rec = []

    def __getattr__(self, k):
        return getattr(TF, k)

    def relu(self, z, inplace=False):
        out = torch.relu(z)
        r = {"z": z}
        self.rec.append(r)
        if out.requires_grad:
            out.register_hook(lambda g, r=r: r.__setitem__("g", g.detach()))
        return out


@contextlib.contextmanager
def recording(module):
    """Replace `module.F` (a `torch.nn.functional` alias) by the recorder for the duration of the block."""
    saved, f = module.F, _RecordingF()
    module.F = f
    try:
        yield f
    finally:
        module.F = saved


def responses(rec, model, tau=2e-5, min_share=1e-4):
    """For every recorded ReLU input with |z| < tau and |incoming gradient| >= min_share of that layer's largest:
    (sign, {parameter name: gradient change if the element's mask flips}). Needs the graph: call after
    `loss.backward(retain_graph=True)`. sign = +1 for an element the oracle masks (a flip adds its gradient)."""
    names = [n for n, _ in model.named_parameters()]
    params = [p for _, p in model.named_parameters()]
    out = []
    for r in rec.rec:
        if "g" not in r:
            continue
        z, g = r["z"], r["g"]
        cand = ((z.detach().abs() < tau) & (g.abs() >= min_share * g.abs().max())).nonzero()
        for idx in cand:
            idx = tuple(idx.tolist())
            e = torch.zeros_like(g)
            sign = 1.0 if z[idx].item() <= 0 else -1.0
            e[idx] = sign * g[idx]
            gs = torch.autograd.grad(z, params, grad_outputs=e, retain_graph=True, allow_unused=True)
            out.append({n: q.detach().double() for n, q in zip(names, gs) if q is not None})
    return out


def explain(diffs, resp):
    """diffs: {name: (tested - oracle, scale)}. Least-squares coefficients of the flip responses over all parameters
    (each tensor weighted by 1/scale), rounded to 0/1; returns (coefficients as fitted, {name: residual / scale})."""
    k = len(resp)
    s = np.zeros(0)
    if k:
        gram, rhs = np.zeros((k, k)), np.zeros(k)
        for n, (d, mag) in diffs.items():
            cols = [r[n].reshape(-1) / mag if n in r else None for r in resp]
            dv = d.reshape(-1) / mag
            for i, ci in enumerate(cols):
                if ci is None:
                    continue
                rhs[i] += float(ci @ dv)
                for j in range(i, k):
                    if cols[j] is not None:
                        gram[i, j] += float(ci @ cols[j])
                        gram[j, i] = gram[i, j]
        s = np.linalg.lstsq(gram, rhs, rcond=None)[0]
    flip = np.round(s).clip(0, 1)
    res = {}
    for n, (d, mag) in diffs.items():
        e = d.clone()
        for f, r in zip(flip, resp):
            if f and n in r:
                e = e - r[n]
        res[n] = e.abs().max().item() / mag
    return s, res
