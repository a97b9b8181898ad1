"""GPU parity of the training-time augmentation kernels against the numpy restatement (oracle/augment_ref.py) on the
same seeded images and parameters: bit-exact (every stage is integer arithmetic; the final normalise is two fp32
operations). The restatement itself is unpinned against albumentations/OpenCV (absent here); its properties are
checked instead: identity parameters reproduce the eval transform, flips are involutions, a rotation by 0 degrees
changes nothing, masks follow the image geometry."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import mia_amd  # noqa: E402,F401
from mia_amd import preprocess as PP  # noqa: E402
from mia_amd.augment import TrainTransform  # noqa: E402
from oracle import augment_ref as A  # noqa: E402
from oracle import preprocess_ref as R  # noqa: E402

IDENT = {"hflip": False, "vflip": False, "angle": None, "alpha": None, "beta255": 0.0, "blur": False}


def cases():
    return [IDENT,
            dict(IDENT, hflip=True), dict(IDENT, vflip=True), dict(IDENT, hflip=True, vflip=True),
            dict(IDENT, angle=15.0), dict(IDENT, angle=-7.3, hflip=True), dict(IDENT, angle=0.0),
            dict(IDENT, alpha=1.2, beta255=-0.2 * 255), dict(IDENT, alpha=0.8, beta255=0.15 * 255),
            dict(IDENT, blur=True), dict(IDENT, angle=11.1, alpha=1.1, beta255=12.0, blur=True, vflip=True),
            dict(IDENT, angle=-15.0, blur=True)]


@pytest.mark.parametrize("shape,size", [((256, 256), (300, 300)), ((97, 131), (64, 80)), ((40, 40), (33, 31))])
def test_train_transform_bit_exact_to_restatement(shape, size):
    plist = cases()
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, (len(plist),) + shape + (3,), dtype=np.uint8)
    tf = TrainTransform(size, seed=0)
    got = tf(torch.from_numpy(imgs).cuda(), params=plist).cpu().numpy()
    for i, p in enumerate(plist):
        exp = A.train_transform(imgs[i], size[0], size[1], p)
        np.testing.assert_array_equal(got[i], exp, err_msg=str(p))


def test_masks_follow_the_image_geometry():
    plist = cases()
    rng = np.random.default_rng(6)
    H, W = 48, 56
    imgs = rng.integers(0, 256, (len(plist), H, W, 3), dtype=np.uint8)
    masks = (rng.random((len(plist), H, W)) < 0.4).astype(np.uint8)
    tf = TrainTransform((H, W), seed=0)
    out, mout = tf(torch.from_numpy(imgs).cuda(), torch.from_numpy(masks).cuda(), params=plist)
    assert mout.dtype == torch.int64
    for i, p in enumerate(plist):
        np.testing.assert_array_equal(mout[i].cpu().numpy(), A.augment_mask(masks[i], p), err_msg=str(p))
    # a blob image and its own mask stay aligned under flips + rotation: the rotated image is bright where the mask is 1
    blob = np.zeros((1, 64, 64), dtype=np.uint8)
    blob[0, 10:30, 35:60] = 1
    img = np.repeat((blob * 255)[..., None], 3, -1).astype(np.uint8)
    p = [dict(IDENT, hflip=True, angle=12.0)]
    o, m = TrainTransform((64, 64))(torch.from_numpy(img).cuda(), torch.from_numpy(blob).cuda(), params=p)
    bright = (o[0, 0] > 0).cpu().numpy()
    agree = (bright == (m[0].cpu().numpy() == 1)).mean()
    assert agree > 0.97


def test_properties_identity_and_involution():
    rng = np.random.default_rng(7)
    imgs = rng.integers(0, 256, (3, 200, 180, 3), dtype=np.uint8)
    x = torch.from_numpy(imgs).cuda()
    tf = TrainTransform((300, 300))
    ident = tf(x, params=[IDENT] * 3)
    assert torch.equal(ident, PP.resize_normalize(x, (300, 300)))           # = the eval transform
    assert torch.equal(tf(x, params=[dict(IDENT, angle=0.0)] * 3), ident)   # rotation by zero
    h = tf(x, params=[dict(IDENT, hflip=True)] * 3)
    assert torch.equal(h.flip(-1), ident)
    v = tf(x, params=[dict(IDENT, vflip=True)] * 3)
    assert torch.equal(v.flip(-2), ident)


def test_sampling_follows_the_reference_probabilities():
    tf = TrainTransform((32, 32), seed=123)
    ps = tf.sample(20000)
    frac = lambda f: np.mean([f(p) for p in ps])
    assert abs(frac(lambda p: p["hflip"]) - 0.5) < 0.02 and abs(frac(lambda p: p["vflip"]) - 0.3) < 0.02
    assert abs(frac(lambda p: p["angle"] is not None) - 0.5) < 0.02 and abs(frac(lambda p: p["blur"]) - 0.3) < 0.02
    assert abs(frac(lambda p: p["alpha"] is not None) - 0.5) < 0.02
    angles = [p["angle"] for p in ps if p["angle"] is not None]
    assert -15.0 <= min(angles) and max(angles) <= 15.0
    alphas = [p["alpha"] for p in ps if p["alpha"] is not None]
    assert 0.8 <= min(alphas) and max(alphas) <= 1.2
    # same seed, same parameters; a full random batch runs and stays finite
    assert TrainTransform((32, 32), seed=123).sample(5) == ps[:5]
    x = torch.randint(0, 256, (16, 64, 64, 3), dtype=torch.uint8, device="cuda")
    assert torch.isfinite(tf(x)).all()
