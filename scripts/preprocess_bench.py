"""Throughput of the preprocessing kernels on the device against HBM bandwidth, with the numpy specification timed
beside them on the host: zero-intensity slice filter (+ ordered compaction, gather) on [N,256,256] uint16 volumes and
resize + normalise of 256x256 uint8 slices to 300x300. Prints one JSON line per case."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import mia_amd  # noqa: F401
from mia_amd import preprocess as PP
from oracle import preprocess_ref as R     # host baseline only (the specification, timed on a bounded sample)

HBM_PEAK = 8000.0
dev = torch.device("cuda", 0)


def phantom_volume(n, seed=7):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:256, 0:256]
    mask = ((yy - 128) / 107.0) ** 2 + ((xx - 128) / 85.0) ** 2 <= 1.0
    vol = (np.clip(rng.normal(1100, 400, (n, 256, 256)), 0, 4095).astype(np.uint16)) * mask[None]
    vol[rng.random(n) < 0.3] = 0
    return vol


def gpu_time(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def cpu_time(fn, min_s=1.0):
    fn()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < min_s:
        fn()
        n += 1
    return (time.perf_counter() - t0) / n


for n in (32, 128, 512, 8192):
    vol_h = phantom_volume(min(n, 512))
    if n > 512:
        vol_h = np.tile(vol_h, (n // 512, 1, 1))
    vol = torch.from_numpy(vol_h.view(np.int16)).to(dev).view(torch.uint16)
    keep, idx, count = PP.zero_slice_filter(vol)
    ref_keep, ref_idx = R.zero_slice_filter(vol_h)
    assert np.array_equal(keep.cpu().numpy(), ref_keep) and np.array_equal(idx.cpu().numpy()[:int(count.item())], ref_idx)
    t_f = gpu_time(lambda: PP.zero_slice_filter(vol))
    t_g = gpu_time(lambda: PP.gather_slices(vol, idx, count))
    kept = int(count.item())
    sample = vol_h[:min(n, 512)]
    t_c = cpu_time(lambda: R.zero_slice_filter(sample)) * (n / sample.shape[0])
    print(json.dumps({"case": "zero_slice_filter", "slices": n, "dtype": "u16", "kept": kept,
                      "gpu_us": round(t_f * 1e6, 1), "slices_per_s": round(n / t_f), "GB_per_s": round(vol.numel() * 2 / t_f / 1e9, 1),
                      "frac_hbm": round(vol.numel() * 2 / t_f / 1e9 / HBM_PEAK, 3),
                      "gather_us": round(t_g * 1e6, 1), "gather_GB_per_s": round(kept * 65536 * 2 * 2 / t_g / 1e9, 1),
                      "cpu_numpy_slices_per_s": round(n / t_c), "cpu_cores": 1}), flush=True)

rng = np.random.default_rng(1)
for batch in (128, 512, 2048):
    img_h = rng.integers(0, 256, (batch, 256, 256, 3), dtype=np.uint8)
    img = torch.from_numpy(img_h).to(dev)
    out = PP.resize_normalize(img, (300, 300))
    ref = R.resize_normalize(img_h[0], 300, 300)
    assert np.array_equal(out[0].cpu().numpy(), ref), "resize+normalise differs from the specification"
    t = gpu_time(lambda: PP.resize_normalize(img, (300, 300)))
    nbytes = img.numel() + out.numel() * 4
    t_c = cpu_time(lambda: R.resize_normalize(img_h[0], 300, 300))
    print(json.dumps({"case": "resize_normalize", "images": batch, "in": "256x256x3 u8", "out": "3x300x300 f32",
                      "gpu_us": round(t * 1e6, 1), "images_per_s": round(batch / t), "GB_per_s": round(nbytes / t / 1e9, 1),
                      "frac_hbm": round(nbytes / t / 1e9 / HBM_PEAK, 3), "cpu_numpy_images_per_s": round(1 / t_c, 1),
                      "cpu_cores": 1}), flush=True)

    # the channels-last bf16 form the trunk's stem reads directly: half the write bytes
    out_cl = PP.resize_normalize(img, (300, 300), out_dtype=torch.bfloat16)
    assert torch.equal(out_cl, out.to(torch.bfloat16)), "channels-last bf16 output differs from the rounded float32 one"
    t = gpu_time(lambda: PP.resize_normalize(img, (300, 300), out_dtype=torch.bfloat16))
    nbytes = img.numel() + out_cl.numel() * 2
    print(json.dumps({"case": "resize_normalize", "images": batch, "in": "256x256x3 u8", "out": "300x300x3 bf16 (NHWC)",
                      "gpu_us": round(t * 1e6, 1), "images_per_s": round(batch / t), "GB_per_s": round(nbytes / t / 1e9, 1),
                      "frac_hbm": round(nbytes / t / 1e9 / HBM_PEAK, 3)}), flush=True)

from mia_amd.augment import TrainTransform
from oracle import augment_ref as A
for batch in (128, 512):
    img_h = rng.integers(0, 256, (batch, 256, 256, 3), dtype=np.uint8)
    img = torch.from_numpy(img_h).to(dev)
    tf = TrainTransform((300, 300), seed=0)
    plist = tf.sample(batch)
    out = tf(img, params=plist)
    assert np.array_equal(out[0].cpu().numpy(), A.train_transform(img_h[0], 300, 300, plist[0]))
    t = gpu_time(lambda: tf(img, params=plist), iters=10)
    t_c = cpu_time(lambda: A.train_transform(img_h[0], 300, 300, plist[0]))
    print(json.dumps({"case": "train_transform (resize + flips + rotate + brightness/contrast + blur + normalise)",
                      "images": batch, "gpu_us": round(t * 1e6, 1), "images_per_s": round(batch / t),
                      "note": "includes the host-side parameter table upload", "cpu_numpy_images_per_s": round(1 / t_c, 1),
                      "cpu_cores": 1}), flush=True)
