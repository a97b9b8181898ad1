"""Deterministic, platform-independent tensors for parity tests.

Everything derives from numpy's PCG64 Generator (stable across numpy versions
and machines), never from torch's RNG, so the golden generator (run once in the
build container against the reference's classes) and the tests (run anywhere)
reconstruct bit-identical weights and inputs from a seed alone.
"""
import numpy as np
import torch


def randn(seed: int, *shape, scale: float = 1.0) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def seeded_state_dict(module: torch.nn.Module, seed: int) -> dict:
    """A full state dict for `module` whose values depend only on (key order, shapes, seed)."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, v in module.state_dict().items():
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros_like(v)
            continue
        a = rng.standard_normal(tuple(v.shape)).astype(np.float32)
        if k.endswith("running_var"):
            a = np.abs(a) * 0.5 + 0.5
        elif k.endswith("running_mean"):
            a = a * 0.1
        elif v.dim() <= 1:
            # BN / LayerNorm affine weights near 1; every bias near 0
            a = (1.0 + 0.1 * a) if k.endswith("weight") else 0.1 * a
        else:
            fan_in = int(np.prod(v.shape[1:])) if v.dim() > 1 else int(v.shape[0])
            a = a * np.float32(1.0 / np.sqrt(max(fan_in, 1)))
        out[k] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return out


def load_seeded(module: torch.nn.Module, seed: int) -> torch.nn.Module:
    module.load_state_dict(seeded_state_dict(module, seed), strict=True)
    return module
