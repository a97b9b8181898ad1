"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the training-time augmentation kernels, in numpy. Only tests/ may
import this file.

Restates the transform list of /root/reference/foundation_model/src/dataset.py:229-239
(A.Resize -> HorizontalFlip -> VerticalFlip -> Rotate(limit=15) -> RandomBrightnessContrast(0.2, 0.2) ->
GaussianBlur(blur_limit=3) -> Normalize -> ToTensorV2) as a deterministic function of the image and the sampled
parameters. The arithmetic of those transforms lives in albumentations + OpenCV (both absent here, versions
unpinned by the reference), so this is PARITY UNPINNED against them; the restated rules are:
  * resize: preprocess_ref.resize_u8_linear (OpenCV's 8-bit INTER_LINEAR);
  * rotation by +angle (counter-clockwise on screen) about (W/2-0.5, H/2-0.5): destination pixel (x,y) samples the
    flipped image at  sx = a*(x-cx) - b*(y-cy) + cx,  sy = b*(x-cx) + a*(y-cy) + cy  (a=cos, b=sin, float64), the
    coordinates rounded to 1/32 pixel, bilinear with integer weights (32-fx)(32-fy).. summing to 1024, result
    (sum + 512) >> 10, border reflect-101 (OpenCV's warpAffine works in the same 1/32-pixel grid);
  * brightness/contrast: uint8(clip(float32(v)*alpha + beta*255, 0, 255)) with truncation (a 256-entry look-up
    table built with astype, as albumentations does for uint8 images);
  * blur: [1 2 1] x [1 2 1] / 16 with rounding ((sum + 8) >> 4), border reflect-101 (the 3x3 sigma-0 kernel);
  * masks: flips and the rotation with the nearest source pixel ((S + 16) >> 5), nothing else.
The HIP kernels are held bit-exact to THIS file.
"""
import numpy as np

from .preprocess_ref import normalize_chw, resize_u8_linear


def reflect101(i, n):
    i = np.asarray(i).copy()
    if n == 1:
        return np.zeros_like(i)
    for _ in range(64):
        bad = (i < 0) | (i >= n)
        if not bad.any():
            break
        i = np.where(i < 0, -i, i)
        i = np.where(i >= n, 2 * (n - 1) - i, i)
    return i


def _rot_coords(H, W, a, b):
    cx, cy = W * 0.5 - 0.5, H * 0.5 - 0.5
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    dx, dy = x - cx, y - cy
    sx = a * dx - b * dy + cx
    sy = b * dx + a * dy + cy
    return np.rint(sx * 32.0).astype(np.int64), np.rint(sy * 32.0).astype(np.int64)


def augment_u8(img, p):
    """img uint8 [H, W, 3] (already resized); p = dict(hflip, vflip, angle or None, alpha/beta or None, blur)."""
    H, W, _ = img.shape
    f = img
    if p["hflip"]:
        f = f[:, ::-1]
    if p["vflip"]:
        f = f[::-1]
    f = f.astype(np.int64)
    if p.get("angle") is not None:
        a, b = np.cos(np.deg2rad(p["angle"])), np.sin(np.deg2rad(p["angle"]))
        SX, SY = _rot_coords(H, W, a, b)
        ix, iy, fx, fy = SX >> 5, SY >> 5, SX & 31, SY & 31
        x0, x1, y0, y1 = reflect101(ix, W), reflect101(ix + 1, W), reflect101(iy, H), reflect101(iy + 1, H)
        w00, w01, w10, w11 = (32 - fx) * (32 - fy), fx * (32 - fy), (32 - fx) * fy, fx * fy
        f = (w00[..., None] * f[y0, x0] + w01[..., None] * f[y0, x1] + w10[..., None] * f[y1, x0]
             + w11[..., None] * f[y1, x1] + 512) >> 10
    if p.get("alpha") is not None:
        v = f.astype(np.float32) * np.float32(p["alpha"])
        v = (v + np.float32(p["beta255"])).astype(np.float32)
        f = np.clip(v, 0, 255).astype(np.uint8).astype(np.int64)
    if p["blur"]:
        yy, xx = np.arange(H), np.arange(W)
        acc = np.zeros_like(f)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                wgt = (2 - abs(dx)) * (2 - abs(dy))
                acc = acc + wgt * f[reflect101(yy + dy, H)][:, reflect101(xx + dx, W)]
        f = (acc + 8) >> 4
    return f.astype(np.uint8)


def train_transform(img_u8, out_h, out_w, p):
    """Full chain on one raw image: uint8 [h, w, 3] -> float32 [3, out_h, out_w]."""
    return normalize_chw(augment_u8(resize_u8_linear(img_u8, out_h, out_w), p))


def augment_mask(mask, p):
    """mask uint8 [H, W] (already resized, nearest) -> int64 [H, W]."""
    H, W = mask.shape
    f = mask
    if p["hflip"]:
        f = f[:, ::-1]
    if p["vflip"]:
        f = f[::-1]
    if p.get("angle") is not None:
        a, b = np.cos(np.deg2rad(p["angle"])), np.sin(np.deg2rad(p["angle"]))
        SX, SY = _rot_coords(H, W, a, b)
        f = f[reflect101((SY + 16) >> 5, H), reflect101((SX + 16) >> 5, W)]
    return f.astype(np.int64)
