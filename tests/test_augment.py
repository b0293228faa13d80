"""SURVEY §8 f4 — device-side augmentation (csrc/augment.hip, dataloaders/gpu_augment.py) against the numpy restatement of the
reference's cv2 / PIL sequence (oracle/augment_ref.py, base/base_dataset.py:63-136).  cv2 is absent from the image, so the
restatement itself is PARITY-UNPINNED against the library; its own invariants are checked on CPU, and the HIP kernels are held
to it on the GPU: labels bit-exact, normalised images within one uint8 level on at most 0.1 % of the pixels."""
import random

import numpy as np
import pytest
import torch

from oracle import augment_ref as R

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _sample(h, w, seed, classes=21):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(yy * 3 + xx * 2 + 40 * np.sin(xx / 7.0)) % 256, (xx * 5 + 30 * np.cos(yy / 5.0)) % 256, g.integers(0, 256, (h, w))], -1)
    lab = ((yy // 16) * 3 + xx // 16) % classes
    return img.astype(np.uint8), lab.astype(np.int32)


def test_restatement_invariants():
    img, lab = _sample(60, 84, 1)
    assert np.array_equal(R.resize_linear(img, 60, 84), img) and np.array_equal(R.resize_nearest(lab, 60, 84), lab)   # identity size
    up = R.resize_nearest(lab, 120, 168)
    assert np.array_equal(up[::2, ::2], lab)                                          # nearest 2x: floor(dst / 2)
    r0, l0 = R.warp_affine(img, lab, R.rotation_inverse(60, 84, 0))
    assert np.array_equal(r0, img) and np.array_equal(l0, lab)                        # angle 0 is the identity
    _, l180 = R.warp_affine(img[:60, :60], lab[:60, :60], R.rotation_inverse(60, 60, 180))
    assert np.array_equal(l180[1:, 1:], lab[:60, :60][::-1, ::-1][:-1, :-1])         # half turn about the centre (w/2, h/2) = (30, 30)
    k = R.gaussian_kernel_half(5, 0.9)
    assert abs(k[0] + 2 * k[1] + 2 * k[2] - 1) < 1e-6 and k[0] > k[1] > k[2] > 0 and k[3] == 0
    flat = np.full((9, 9, 3), 77, np.uint8)
    assert np.array_equal(R.gaussian_blur(flat, 5, 0.9), flat)                        # a constant image is a fixed point
    x, t, d = R.augment(img, lab, MEAN, STD, base_size=80, crop_size=64, scale=True, flip=True, rotate=True, blur=True, rng=random.Random(3))
    assert x.shape == (3, 64, 64) and x.dtype == np.float32 and t.shape == (64, 64) and t.dtype == np.int64
    assert set(d) == {"rs", "angle", "start", "flip", "sigma"} and -10 <= d["angle"] <= 10 and 40 <= max(d["rs"]) <= 160
    # padding region (image smaller than the crop): zeros before normalisation, label 0
    x2, t2, d2 = R.augment(img[:20, :30], lab[:20, :30], MEAN, STD, base_size=None, crop_size=64, scale=False, flip=False, rng=random.Random(1))
    assert d2["start"] == (0, 0) and np.all(t2[20:] == 0) and np.allclose(x2[:, 40, 40], (0 - np.array(MEAN)) / np.array(STD), atol=1e-6)


def test_decisions_follow_the_reference_draw_order():
    """Same seed -> the same sequence of random.* calls as base/base_dataset.py:67-116 (randint long side, randint angle, randint
    start_h, randint start_w, random flip, random sigma)."""
    rng = random.Random(11)
    d = R.draw_decisions(rng, 100, 150, 120, 96, True, True, True, True)
    ref = random.Random(11)
    longside = ref.randint(60, 240)
    h, w = (int(1.0 * longside * 100 / 150 + 0.5), longside)
    angle = ref.randint(-10, 10)
    sy = ref.randint(0, max(h, 96) - 96)
    sx = ref.randint(0, max(w, 96) - 96)
    fl = ref.random() > 0.5
    sg = ref.random()
    assert d == {"rs": (h, w), "angle": angle, "start": (sy, sx), "flip": fl, "sigma": sg}


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(base_size=96, crop_size=80, scale=True, flip=True, rotate=True, blur=True),
                                 dict(base_size=None, crop_size=64, scale=False, flip=True, rotate=False, blur=False),
                                 dict(base_size=70, crop_size=128, scale=True, flip=False, rotate=True, blur=False)])
def test_gpu_augment_matches_the_restatement(cuda, cfg):
    from dataloaders.gpu_augment import GPUAugment
    samples = [_sample(h, w, s) for (h, w, s) in ((60, 84, 1), (97, 61, 2), (50, 50, 3), (120, 40, 4))]
    aug = GPUAugment(MEAN, STD, device=cuda, seed=5, **cfg)
    x, t = aug(samples)
    torch.cuda.synchronize()
    assert tuple(x.shape) == (4, 3, cfg["crop_size"], cfg["crop_size"]) and t.dtype == torch.int64
    rng = random.Random(5)
    lvl = 1.0 / 255 / min(STD)
    for i, (img, lab) in enumerate(samples):
        xr, tr, d = R.augment(img, lab, MEAN, STD, rng=rng, **cfg)
        assert d == aug.decisions[i]
        assert torch.equal(t[i].cpu(), torch.from_numpy(tr)), i
        diff = (x[i].cpu() - torch.from_numpy(xr)).abs()
        assert float(diff.max()) <= 1.001 * lvl, (i, float(diff.max()), lvl)          # at most one uint8 level
        assert float((diff > 1e-5).float().mean()) <= 1e-3, (i, float((diff > 1e-5).float().mean()))
    # the batch is NHWC-backed with a zero padding channel: the model's first convolution consumes it without a layout pass
    from segmi import ops
    assert ops.is_nhwc(x)
