"""GPU parity of the preprocessing kernels against the numpy oracle: bit-exact (integer / byte work,
and two fp32 operations with no contraction)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mia_amd  # noqa: E402,F401
from mia_amd import preprocess as PP  # noqa: E402
from oracle import preprocess_ref as R  # noqa: E402


def phantom(n, h, w, seed):
    """MRI-shaped slices: centred ellipse of N(110, 40^2) clipped to [0,255] on a zero background."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    mask = ((yy - h / 2) / (0.42 * h)) ** 2 + ((xx - w / 2) / (0.33 * w)) ** 2 <= 1.0
    img = np.clip(rng.normal(110, 40, (n, h, w)), 0, 255).astype(np.uint8)
    return img * mask[None]


@pytest.mark.parametrize("shape,out", [((4, 256, 256, 3), (300, 300)), ((3, 128, 128, 3), (64, 64)),
                                       ((2, 97, 131, 3), (300, 300)), ((2, 300, 300, 3), (300, 300)),
                                       ((1, 1, 1, 3), (5, 7)), ((2, 512, 384, 3), (224, 224)),
                                       ((1, 1200, 1000, 3), (300, 300)), ((5, 256, 256, 3), (301, 299))])
@pytest.mark.parametrize("normalize", [True, False])
def test_resize_normalize_bit_exact(shape, out, normalize):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    got = PP.resize_normalize(torch.from_numpy(img).cuda(), out, normalize=normalize).cpu().numpy()
    fn = R.resize_normalize if normalize else R.resize_plain
    exp = np.stack([fn(im, *out) for im in img])
    assert got.dtype == np.float32 and got.shape == exp.shape
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("shape,out", [((4, 256, 256, 3), (300, 300)), ((2, 97, 131, 3), (300, 300)),
                                       ((1, 1200, 1000, 3), (300, 300)), ((5, 256, 256, 1), (301, 299)),
                                       ((3, 64, 64, 3), (9, 6))])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("normalize", [True, False])
def test_resize_normalize_channels_last_compute_dtype(shape, out, dtype, normalize):
    """The channels-last output in the trunk's compute type holds the float32 results of the default form rounded
    once to that type (bit-exact after the same rounding), has the shape of the default form and NHWC storage."""
    rng = np.random.default_rng(4)
    img = torch.from_numpy(rng.integers(0, 256, shape, dtype=np.uint8)).cuda()
    ref = PP.resize_normalize(img, out, normalize=normalize)
    got = PP.resize_normalize(img, out, normalize=normalize, out_dtype=dtype)
    assert got.dtype == dtype and tuple(got.shape) == tuple(ref.shape)
    assert got.permute(0, 2, 3, 1).is_contiguous()
    assert torch.equal(got, ref.to(dtype))


def test_resize_gray_replicates_channels():
    g = phantom(3, 256, 256, 1234)
    got = PP.resize_normalize(torch.from_numpy(g).cuda(), (300, 300)).cpu().numpy()
    exp = np.stack([R.resize_normalize(np.repeat(s[..., None], 3, -1), 300, 300) for s in g])
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32])
@pytest.mark.parametrize("n", [1, 32, 128, 513])
def test_zero_slice_filter_bit_exact(dtype, n):
    rng = np.random.default_rng(7)
    vol = phantom(n, 256, 256, 7).astype(dtype)
    dead = rng.random(n) < 0.3
    vol[dead] = 0
    keep_ref, idx_ref = R.zero_slice_filter(vol)
    tv = torch.from_numpy(vol.view(np.int16) if dtype == np.uint16 else vol).cuda()
    keep, idx, count = PP.zero_slice_filter(tv)
    k = int(count.item())
    np.testing.assert_array_equal(keep.cpu().numpy(), keep_ref)
    assert k == idx_ref.size
    np.testing.assert_array_equal(idx.cpu().numpy()[:k], idx_ref)
    out = PP.gather_slices(tv, idx, count).cpu().numpy()[:k]
    np.testing.assert_array_equal(out.view(vol.dtype) if dtype == np.uint16 else out, vol[idx_ref])


def test_zero_slice_filter_adversarial():
    H = W = 256
    cases = []
    for (y, x) in ((0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (H // 2, W // 2)):
        v = np.zeros((6, H, W), np.uint16)
        v[2, y, x] = 1
        v[4, y, x] = 0x8000
        cases.append(v)
    cases.append(np.zeros((5, H, W), np.uint16))
    cases.append(np.ones((5, H, W), np.uint16))
    odd = np.zeros((7, 31, 33), np.uint8)   # slice size not a multiple of 16 bytes, unaligned slice starts
    odd[1, 30, 32] = 9
    odd[3, 0, 0] = 1
    odd[6, 15, 7] = 255
    cases.append(odd)
    f = np.zeros((6, 64, 64), np.float32)
    f[1, 3, 3] = -0.0
    f[2, 5, 5] = np.nan
    f[4, 63, 63] = 1e-45  # denormal is non-zero
    cases.append(f)
    for v in cases:
        keep_ref, idx_ref = R.zero_slice_filter(v)
        tv = torch.from_numpy(v.view(np.int16) if v.dtype == np.uint16 else v).cuda()
        keep, idx, count = PP.zero_slice_filter(tv)
        np.testing.assert_array_equal(keep.cpu().numpy(), keep_ref)
        k = int(count.item())
        assert k == idx_ref.size
        np.testing.assert_array_equal(idx.cpu().numpy()[:k], idx_ref)


def test_idempotent_and_sorted_at_full_size():
    """Size-independent properties at the largest configured volume (512 slices)."""
    vol = phantom(512, 256, 256, 11).astype(np.uint16)
    vol[np.random.default_rng(11).random(512) < 0.3] = 0
    tv = torch.from_numpy(vol.view(np.int16)).cuda()
    keep, idx, count = PP.zero_slice_filter(tv)
    k = int(count.item())
    ids = idx.cpu().numpy()[:k]
    assert (np.diff(ids) > 0).all()
    packed = PP.gather_slices(tv, idx, count)[:k]
    keep2, idx2, count2 = PP.zero_slice_filter(packed.contiguous())
    assert int(count2.item()) == k and bool(keep2.all())          # filtering a filtered volume keeps everything
    assert int(keep.sum().item()) == k


def test_zero_slice_filter_many_calls_and_large_slices():
    """Slices of up to 1 MiB take the one-launch path (a workgroup per slice, the last one to finish compacts; its
    arrival counters come from a pool of 64 that each call leaves at zero): 150 calls in a row with changing slice
    counts cycle through every counter more than twice. Larger slices take the split path (several workgroups per
    slice). Both bit-exact against the numpy specification."""
    rng = np.random.default_rng(5)
    for call in range(150):
        n = int(rng.integers(1, 40))
        vol = rng.integers(0, 2, (n, 64, 64)).astype(np.uint8) * (rng.random((n, 1, 1)) < 0.6)
        keep_ref, idx_ref = R.zero_slice_filter(vol)
        keep, idx, count = PP.zero_slice_filter(torch.from_numpy(vol).cuda())
        k = int(count.item())
        assert k == idx_ref.size, call
        np.testing.assert_array_equal(keep.cpu().numpy(), keep_ref)
        np.testing.assert_array_equal(idx.cpu().numpy()[:k], idx_ref)
    big = np.zeros((7, 1024, 640), np.int16)        # 1.25 MiB per slice
    big[1, 1023, 639] = 1
    big[2, 0, 0] = -1
    big[5, 512, 17] = 3
    keep_ref, idx_ref = R.zero_slice_filter(big)
    keep, idx, count = PP.zero_slice_filter(torch.from_numpy(big).cuda())
    assert int(count.item()) == 3
    np.testing.assert_array_equal(keep.cpu().numpy(), keep_ref)
    np.testing.assert_array_equal(idx.cpu().numpy()[:3], idx_ref)
