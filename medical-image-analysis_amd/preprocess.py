"""Device-side preprocessing: the eval transform of the reference (A.Resize -> A.Normalize(ImageNet) ->
ToTensorV2, src/dataset.py:247-251 / inference.py:116-140), its no-transform fallback
(cv2.resize -> /255, dataset.py:180-182), and the zero-intensity slice filter (build-specified: the
reference has none)."""
import ctypes
from typing import Tuple

import torch

from ._lib import lib
from .ops import _p, _stream

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _f3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


def resize_normalize(images_u8: torch.Tensor, size: Tuple[int, int], mean=IMAGENET_MEAN, std=IMAGENET_STD,
                     normalize: bool = True, out_dtype=None) -> torch.Tensor:
    """uint8 [N,h,w,3] (RGB, HWC) or [N,h,w] / [N,h,w,1] (gray, replicated) on the device -> float32 [N,3,H,W].

    out_dtype (torch.bfloat16 or torch.float32): the result in that type and in channels-last memory order instead
    -- still a tensor of shape [N,3,H,W], so it is passed to the model like the default form, but stored [N,H,W,3].
    That is the form the trunk's first convolution reads directly when out_dtype is the model's compute type: the
    values are the float32 results rounded once to out_dtype, which is the rounding the trunk applies to a float32
    input anyway, so the model's outputs are bit-identical either way."""
    if not images_u8.is_cuda or images_u8.dtype != torch.uint8:
        raise RuntimeError("resize_normalize expects a uint8 CUDA tensor")
    if images_u8.dim() == 3:
        images_u8 = images_u8.unsqueeze(-1)
    images_u8 = images_u8.contiguous()
    N, h, w, ch = images_u8.shape
    oh, ow = int(size[0]), int(size[1])
    m, s = _f3(mean), _f3(std)
    if out_dtype is not None:
        if out_dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f"resize_normalize: out_dtype must be float32 or bfloat16, got {out_dtype}")
        out = torch.empty((N, oh, ow, 3), dtype=out_dtype, device=images_u8.device)
        lib().mia_resize_normalize_u8_nhwc(_p(images_u8), N, h, w, ch, _p(out), 1 if out_dtype == torch.bfloat16 else 0,
                                           oh, ow, ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p),
                                           0 if normalize else 1, _stream())
        return out.permute(0, 3, 1, 2)
    out = torch.empty((N, 3, oh, ow), dtype=torch.float32, device=images_u8.device)
    lib().mia_resize_normalize_u8(_p(images_u8), N, h, w, ch, _p(out), oh, ow, ctypes.cast(m, ctypes.c_void_p),
                                  ctypes.cast(s, ctypes.c_void_p), 0 if normalize else 1, _stream())
    return out


def zero_slice_filter(vol: torch.Tensor):
    """vol [N, ...] (uint8 / int16 / uint16 / float32) -> (keep bool[N], idx int32[N] (first count valid), count int32[1]).
    Everything stays on the device; call .item() on count only when the host needs it."""
    if not vol.is_cuda:
        raise RuntimeError("zero_slice_filter expects a CUDA tensor")
    vol = vol.contiguous()
    N = vol.shape[0]
    elems = vol.numel() // N
    esz = vol.element_size()
    is_float = vol.dtype == torch.float32
    if vol.dtype not in (torch.uint8, torch.int8, torch.int16, torch.uint16, torch.int32, torch.float32):
        raise TypeError(f"unsupported volume dtype {vol.dtype}")
    dev = vol.device
    # one allocation for the four small buffers (each allocator call costs the host as much as the kernel runs)
    words = torch.empty((2 * N + 1 + (N + 3) // 4,), dtype=torch.int32, device=dev)
    flags, idx, count = words[:N], words[N:2 * N], words[2 * N:2 * N + 1]
    keep = words[2 * N + 1:].view(torch.bool)[:N]               # one byte per slice, written as 0 / 1
    lib().mia_zero_slice_filter(_p(vol), esz, 1 if is_float else 0, N, elems, _p(flags), _p(keep), _p(idx), _p(count),
                                _stream())
    return keep, idx, count


def gather_slices(vol: torch.Tensor, idx: torch.Tensor, count: torch.Tensor) -> torch.Tensor:
    """Stable compaction: out[k] = vol[idx[k]] for k < count; rows beyond count are left untouched."""
    vol = vol.contiguous()
    N = vol.shape[0]
    out = torch.empty_like(vol)
    lib().mia_gather_slices(_p(vol), vol.element_size(), N, vol.numel() // N, _p(idx), _p(count), _p(out), _stream())
    return out
