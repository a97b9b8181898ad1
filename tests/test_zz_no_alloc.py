"""The allocation-free entry point of the zero-intensity slice filter (flag words in a device array built into the
library) against the workspace-taking entry point, through the C ABI: bit-exact, on two calls in a row, the second
on a smaller volume (the built-in array still holds the first call's flags)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mia_amd  # noqa: E402
from mia_amd.ops import _p, _stream  # noqa: E402
from oracle import preprocess_ref as R  # noqa: E402


def _volume(n, seed, h=256, w=256):
    rng = np.random.default_rng(seed)
    vol = rng.integers(1, 4000, (n, h, w)).astype(np.int16)
    vol[rng.random(n) < 0.35] = 0
    return vol


def _run(entry, tv, with_ws):
    n = tv.shape[0]
    dev = tv.device
    keep = torch.full((n,), 7, dtype=torch.uint8, device=dev)
    idx = torch.full((n,), -1, dtype=torch.int32, device=dev)
    count = torch.full((1,), -1, dtype=torch.int32, device=dev)
    args = [_p(tv), tv.element_size(), 0, n, tv.numel() // n]
    if with_ws:
        ws = torch.empty((n,), dtype=torch.int32, device=dev)
        args.append(_p(ws))
    entry(*args, _p(keep), _p(idx), _p(count), _stream())
    torch.cuda.synchronize()
    k = int(count.item())
    return keep.cpu().numpy(), idx.cpu().numpy()[:k], k


def test_static_scratch_matches_workspace_entry_point():
    lib = mia_amd.lib()
    for n, seed in ((700, 21), (93, 22)):      # second call smaller: stale flags of the first must not leak in
        vol = _volume(n, seed)
        tv = torch.from_numpy(vol).cuda()
        keep_a, idx_a, k_a = _run(lib.mia_zero_slice_filter, tv, True)
        keep_b, idx_b, k_b = _run(lib.mia_zero_slice_filter_static, tv, False)
        assert k_a == k_b
        np.testing.assert_array_equal(keep_b, keep_a)
        np.testing.assert_array_equal(idx_b, idx_a)
        keep_ref, idx_ref = R.zero_slice_filter(vol)
        np.testing.assert_array_equal(keep_b.astype(bool), keep_ref)
        np.testing.assert_array_equal(idx_b, idx_ref)


def test_static_scratch_large_slices_second_call_smaller():
    """Slices above 1 MiB take the path whose flag words are OR-ed into the scratch array by several workgroups per
    slice (and cleared first): again two calls in a row, the second on fewer slices, against the workspace entry."""
    lib = mia_amd.lib()
    for n, seed in ((11, 31), (4, 32)):
        vol = _volume(n, seed, h=1024, w=640)
        tv = torch.from_numpy(vol).cuda()
        keep_a, idx_a, k_a = _run(lib.mia_zero_slice_filter, tv, True)
        keep_b, idx_b, k_b = _run(lib.mia_zero_slice_filter_static, tv, False)
        assert k_a == k_b
        np.testing.assert_array_equal(keep_b, keep_a)
        np.testing.assert_array_equal(idx_b, idx_a)
        keep_ref, idx_ref = R.zero_slice_filter(vol)
        np.testing.assert_array_equal(keep_b.astype(bool), keep_ref)
        np.testing.assert_array_equal(idx_b, idx_ref)


def test_static_scratch_rejects_oversize():
    lib = mia_amd.lib()
    tv = torch.zeros((4, 16), dtype=torch.uint8, device="cuda")
    out = torch.empty((4,), dtype=torch.int32, device="cuda")
    keep = torch.empty((4,), dtype=torch.uint8, device="cuda")
    with pytest.raises(mia_amd._lib.MiaError, match="exceed the built-in scratch"):
        lib.mia_zero_slice_filter_static(_p(tv), 1, 0, 65536, 1, _p(keep), _p(out), _p(out), _stream())
