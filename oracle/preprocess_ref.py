"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the preprocessing kernels, in numpy.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.

resize_u8_linear / resize_normalize restate what the reference's eval transform
does (/root/reference/foundation_model/src/dataset.py:247-251 and
inference.py:116-140: A.Resize -> A.Normalize(ImageNet) -> ToTensorV2), whose
arithmetic lives in third-party OpenCV + albumentations (both absent here,
versions unpinned by the reference). The restatement follows OpenCV's published
8-bit INTER_LINEAR algorithm: half-pixel centres, coefficients rounded to 11-bit
fixed point (2048), horizontal pass in int32, vertical pass
((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2, result uint8; then
albumentations' normalize: (float32(x) - 255*mean) * (1/(255*std)) in float32.
PARITY UNPINNED against cv2 itself (no cv2 build and no reference fixture here);
the HIP kernels are held bit-exact to THIS restatement.

resize_plain restates the no-transform fallback (dataset.py:180-182):
cv2.resize -> CHW -> /255.0.

zero_slice_filter is the build's own specification: the reference has no
zero-intensity slice filter (zi_remover.py:3-19 deletes Zone.Identifier files),
so there is nothing to pin against: keep[z] = any(vol[z] != 0), idx = ascending
indices of kept slices (int32). PARITY UNPINNED vs reference by construction.
"""
import numpy as np

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def linear_coeffs(src: int, dst: int):
    """Per-destination source index pair and 11-bit fixed-point weights."""
    scale = 1.0 / (float(dst) / float(src))  # double, as OpenCV computes it
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0.0
    s[hi] = src - 1
    s1 = np.minimum(s + 1, src - 1).astype(np.int32)
    # cvRound = round-half-to-even on the float product
    w1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int32)
    w0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_SCALE)).astype(np.int32)
    return s, s1, w0, w1


def resize_u8_linear(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img uint8 [h, w, c] -> uint8 [out_h, out_w, c]."""
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, _ = img.shape
    x0, x1, a0, a1 = linear_coeffs(w, out_w)
    y0, y1, b0, b1 = linear_coeffs(h, out_h)
    src = img.astype(np.int32)
    hor = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]   # [h, out_w, c] * 2048
    s0 = hor[y0]  # [out_h, out_w, c]
    s1 = hor[y1]
    v = ((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16)
    v = (v + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def normalize_chw(img_u8: np.ndarray, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> np.ndarray:
    """uint8 [H, W, 3] -> float32 [3, H, W]."""
    m = (np.asarray(mean, dtype=np.float32) * np.float32(255.0)).astype(np.float32)
    r = (np.float32(1.0) / (np.asarray(std, dtype=np.float32) * np.float32(255.0))).astype(np.float32)
    x = img_u8.astype(np.float32)
    x = (x - m[None, None, :]).astype(np.float32)
    x = (x * r[None, None, :]).astype(np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def resize_normalize(img_u8: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    return normalize_chw(resize_u8_linear(img_u8, out_h, out_w))


def resize_plain(img_u8: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    r = resize_u8_linear(img_u8, out_h, out_w).astype(np.float32)
    return np.ascontiguousarray((r / np.float32(255.0)).astype(np.float32).transpose(2, 0, 1))


def zero_slice_filter(vol: np.ndarray):
    """vol [N, H, W] (any numeric dtype) -> (keep bool[N], idx int32[K])."""
    n = vol.shape[0]
    keep = (vol.reshape(n, -1) != 0).any(axis=1)
    idx = np.flatnonzero(keep).astype(np.int32)
    return keep, idx


def zero_slice_compact(vol: np.ndarray):
    keep, idx = zero_slice_filter(vol)
    return keep, idx, np.ascontiguousarray(vol[idx])
