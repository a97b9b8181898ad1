"""Device-side training transform: the augmentation list the reference builds with albumentations in
get_transforms(augment=True) (src/dataset.py:229-239), applied to whole uint8 batches on the GPU and fused with the
normalise step. Random parameters are drawn on the host (one small table per batch, same probabilities and limits
as the reference's transform list); the pixels never leave the device."""
import ctypes
import math
from typing import Optional, Tuple

import numpy as np
import torch

from ._lib import lib
from .ops import _p, _stream
from .preprocess import IMAGENET_MEAN, IMAGENET_STD, _f3

N_PARAMS = 9   # hflip, vflip, rotate?, cos, sin, brightness/contrast?, alpha, beta*255, blur?


def params_table(plist) -> np.ndarray:
    """list of dicts (hflip, vflip, angle|None, alpha|None, beta255, blur) -> float64 [N, 9] for the kernels."""
    t = np.zeros((len(plist), N_PARAMS), dtype=np.float64)
    for i, p in enumerate(plist):
        t[i, 0], t[i, 1] = float(bool(p["hflip"])), float(bool(p["vflip"]))
        if p.get("angle") is not None:
            t[i, 2], t[i, 3], t[i, 4] = 1.0, math.cos(math.radians(p["angle"])), math.sin(math.radians(p["angle"]))
        if p.get("alpha") is not None:
            t[i, 5], t[i, 6], t[i, 7] = 1.0, float(np.float32(p["alpha"])), float(np.float32(p["beta255"]))
        t[i, 8] = float(bool(p["blur"]))
    return t


class TrainTransform:
    """HorizontalFlip(p=.5), VerticalFlip(p=.3), Rotate(limit=15, p=.5), RandomBrightnessContrast(.2, .2, p=.5),
    GaussianBlur(3, p=.3) after A.Resize, then Normalize + ToTensorV2 (dataset.py:229-239)."""

    def __init__(self, image_size: Tuple[int, int], seed: Optional[int] = None, mean=IMAGENET_MEAN, std=IMAGENET_STD,
                 p_hflip=0.5, p_vflip=0.3, p_rotate=0.5, rotate_limit=15.0, p_bc=0.5, brightness_limit=0.2,
                 contrast_limit=0.2, p_blur=0.3):
        self.size = (int(image_size[0]), int(image_size[1]))
        self.rng = np.random.default_rng(seed)
        self.mean, self.std = mean, std
        self.p = dict(hflip=p_hflip, vflip=p_vflip, rotate=p_rotate, bc=p_bc, blur=p_blur)
        self.rotate_limit, self.brightness_limit, self.contrast_limit = rotate_limit, brightness_limit, contrast_limit

    def sample(self, n: int):
        out, r = [], self.rng
        for _ in range(n):
            d = {"hflip": r.random() < self.p["hflip"], "vflip": r.random() < self.p["vflip"], "angle": None,
                 "alpha": None, "beta255": 0.0, "blur": False}
            if r.random() < self.p["rotate"]:
                d["angle"] = float(r.uniform(-self.rotate_limit, self.rotate_limit))
            if r.random() < self.p["bc"]:
                d["alpha"] = 1.0 + float(r.uniform(-self.contrast_limit, self.contrast_limit))
                d["beta255"] = float(r.uniform(-self.brightness_limit, self.brightness_limit)) * 255.0
            d["blur"] = bool(r.random() < self.p["blur"])
            out.append(d)
        return out

    def __call__(self, images_u8: torch.Tensor, masks_u8: Optional[torch.Tensor] = None, params=None):
        """images_u8: uint8 [N,h,w,3] or [N,h,w] on the device; masks_u8: uint8 [N,H,W] at the target size (the
        reference resizes masks with nearest interpolation before the transform, dataset.py:206). Returns
        float32 [N,3,H,W] (and int64 [N,H,W] masks)."""
        if not images_u8.is_cuda or images_u8.dtype != torch.uint8:
            raise RuntimeError("TrainTransform expects a uint8 tensor on an MI355X device; there is no CPU path")
        if images_u8.dim() == 3:
            images_u8 = images_u8.unsqueeze(-1)
        images_u8 = images_u8.contiguous()
        N, h, w, ch = images_u8.shape
        H, W = self.size
        dev = images_u8.device
        plist = params if params is not None else self.sample(N)
        table = torch.from_numpy(params_table(plist)).to(dev, non_blocking=True)
        resized = torch.empty((N, H, W, 3), dtype=torch.uint8, device=dev)
        lib().mia_resize_u8(_p(images_u8), N, h, w, ch, _p(resized), H, W, _stream())
        out = torch.empty((N, 3, H, W), dtype=torch.float32, device=dev)
        m, s = _f3(self.mean), _f3(self.std)
        lib().mia_augment_normalize_u8(_p(resized), N, H, W, _p(table), _p(out), ctypes.cast(m, ctypes.c_void_p),
                                       ctypes.cast(s, ctypes.c_void_p), _stream())
        if masks_u8 is None:
            return out
        if not masks_u8.is_cuda or masks_u8.dtype != torch.uint8 or tuple(masks_u8.shape) != (N, H, W):
            raise RuntimeError(f"masks must be uint8 [{N},{H},{W}] on the device")
        mout = torch.empty((N, H, W), dtype=torch.int64, device=dev)
        lib().mia_augment_mask_u8(_p(masks_u8.contiguous()), N, H, W, _p(table), _p(mout), _stream())
        return out, mout
