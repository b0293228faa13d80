"""numpy restatement of the reference's training augmentation (test infrastructure — only tests/ may import it).

Follows base/base_dataset.py:63-120 (`BaseDataSet._augmentation`) and :125-136 (`__getitem__`: label -> int64, ToTensor,
Normalize) step by step, with the cv2 / PIL calls restated from their documented algorithms:
  cv2.resize INTER_LINEAR   src = (dst + 0.5) * (src_size / dst_size) - 0.5, neighbours clamped to the image, round to nearest uint8
  cv2.resize INTER_NEAREST  src = min(floor(dst * src_size / dst_size), src_size - 1)
  cv2.getRotationMatrix2D((cx, cy), angle, 1): [[a, b, (1-a) cx - b cy], [-b, a, b cx + (1-a) cy]], a = cos, b = sin (degrees)
  cv2.warpAffine            dst(x, y) = src(M^-1 (x, y)); bilinear on a 1/32-pixel coordinate grid (INTER_BITS = 5) resp. nearest;
                            BORDER_CONSTANT 0 takes part in the interpolation
  cv2.GaussianBlur          separable, kernel exp(-(i - (k-1)/2)^2 / (2 sigma^2)) normalised, BORDER_REFLECT_101
PARITY UNPINNED: cv2 (opencv-python, unpinned in the reference's requirements.txt) is absent from this image and from
/root/reference, so these restatements could not be checked against the library; cv2's uint8 paths use fixed-point
coefficients and may differ from this fp32 arithmetic by one level at isolated pixels.  The random decisions are drawn with
Python's `random` in the reference's order, so a seeded run takes the same decisions as the reference would.
"""
import math
import random

import numpy as np

f32 = np.float32


def _round_u8(v):
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def resize_linear(img, dh, dw):
    sh, sw = img.shape[:2]
    fy, fx = f32(sh) / f32(dh), f32(sw) / f32(dw)
    sy = (np.arange(dh, dtype=f32) + f32(0.5)) * fy - f32(0.5)
    sx = (np.arange(dw, dtype=f32) + f32(0.5)) * fx - f32(0.5)

    def taps(s, n):
        i0 = np.floor(s).astype(np.int64)
        w = (s - i0.astype(f32)).astype(f32)
        lo, hi = i0 < 0, i0 >= n - 1
        i0 = np.where(lo, 0, np.where(hi, n - 1, i0))
        w = np.where(lo | hi, f32(0), w).astype(f32)
        return i0, np.minimum(i0 + 1, n - 1), w

    y0, y1, wy = taps(sy, sh)
    x0, x1, wx = taps(sx, sw)
    im = img.astype(f32)
    wy, wx = wy[:, None, None], wx[None, :, None]
    one = f32(1)
    top = (one - wx) * im[y0][:, x0] + wx * im[y0][:, x1]
    bot = (one - wx) * im[y1][:, x0] + wx * im[y1][:, x1]
    return _round_u8((one - wy) * top + wy * bot)


def resize_nearest(lab, dh, dw):
    sh, sw = lab.shape
    fy, fx = f32(sh) / f32(dh), f32(sw) / f32(dw)
    ny = np.minimum(np.floor(np.arange(dh, dtype=f32) * fy).astype(np.int64), sh - 1)
    nx = np.minimum(np.floor(np.arange(dw, dtype=f32) * fx).astype(np.int64), sw - 1)
    return lab[ny][:, nx]


def rotation_inverse(h, w, angle_deg):
    """Inverse of cv2.getRotationMatrix2D((w/2, h/2), angle, 1.0) as 6 float32 {m00 m01 m02 m10 m11 m12} (computed in float64
    like cv2 and rounded once)."""
    a, b = math.cos(math.radians(angle_deg)), math.sin(math.radians(angle_deg))
    cx, cy = w / 2.0, h / 2.0
    M = np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy], [0, 0, 1]], dtype=np.float64)
    return np.linalg.inv(M)[:2].reshape(-1).astype(f32)


def warp_affine(img, lab, inv6):
    h, w = lab.shape
    m = inv6.astype(f32)
    ys, xs = np.meshgrid(np.arange(h, dtype=f32), np.arange(w, dtype=f32), indexing="ij")
    sx = m[0] * xs + m[1] * ys + m[2]
    sy = m[3] * xs + m[4] * ys + m[5]
    X, Y = np.rint(sx * f32(32)).astype(np.int64), np.rint(sy * f32(32)).astype(np.int64)
    x0, y0 = X >> 5, Y >> 5
    wx, wy = ((X & 31).astype(f32) * f32(1 / 32))[..., None], ((Y & 31).astype(f32) * f32(1 / 32))[..., None]
    im = img.astype(f32)

    def px(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return np.where(ok[..., None], im[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], f32(0))

    one = f32(1)
    v = (one - wy) * ((one - wx) * px(y0, x0) + wx * px(y0, x0 + 1)) + wy * ((one - wx) * px(y0 + 1, x0) + wx * px(y0 + 1, x0 + 1))
    nx, ny = np.rint(sx).astype(np.int64), np.rint(sy).astype(np.int64)
    ok = (ny >= 0) & (ny < h) & (nx >= 0) & (nx < w)
    return _round_u8(v), np.where(ok, lab[np.clip(ny, 0, h - 1), np.clip(nx, 0, w - 1)], 0).astype(lab.dtype)


def gaussian_kernel_half(ksize, sigma):
    """{centre, +-1, +-2, +-3} taps of cv2.getGaussianKernel(ksize, sigma) (sigma > 0), float32."""
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    k = (k / k.sum()).astype(f32)
    half = np.zeros(4, dtype=f32)
    half[: ksize // 2 + 1] = k[ksize // 2:]
    return half


def _reflect101(i, n):
    if n == 1:
        return np.zeros_like(i)
    i = np.abs(i)
    i = np.where(i >= n, 2 * (n - 1) - i, i)
    return np.abs(i)


def gaussian_blur(img, ksize, sigma):
    if ksize <= 1:
        return img.copy()
    half = gaussian_kernel_half(ksize, sigma)
    h, w = img.shape[:2]
    r = ksize // 2
    im = img.astype(f32)
    tmp = np.zeros_like(im)
    for d in range(-r, r + 1):
        tmp = tmp + half[abs(d)] * im[:, _reflect101(np.arange(w) + d, w)]
    out = np.zeros_like(im)
    for d in range(-r, r + 1):
        out = out + half[abs(d)] * tmp[_reflect101(np.arange(h) + d, h)]
    return _round_u8(out)


def draw_decisions(rng, h, w, base_size, crop_size, scale, flip, rotate, blur):
    """The reference's random draws, in its order (base/base_dataset.py:67-116).  rng: a `random.Random`."""
    d = {"rs": None, "angle": None, "start": None, "flip": False, "sigma": None}
    if base_size:
        longside = rng.randint(int(base_size * 0.5), int(base_size * 2.0)) if scale else base_size
        h, w = (longside, int(1.0 * longside * w / h + 0.5)) if h > w else (int(1.0 * longside * h / w + 0.5), longside)
        d["rs"] = (h, w)
    if rotate:
        d["angle"] = rng.randint(-10, 10)
    if crop_size:
        ph, pw = max(h, crop_size), max(w, crop_size)
        d["start"] = (rng.randint(0, ph - crop_size), rng.randint(0, pw - crop_size))
    if flip:
        d["flip"] = rng.random() > 0.5
    if blur:
        d["sigma"] = rng.random()
    return d


def augment(image, label, mean, std, base_size=None, crop_size=321, scale=True, flip=True, rotate=False, blur=False, rng=None):
    """image uint8 [H,W,3], label int32 [H,W] -> (float32 [3,crop,crop] normalised, int64 [crop,crop]) like the reference's
    `__getitem__` with augment=True."""
    rng = rng or random
    h, w = label.shape
    d = draw_decisions(rng, h, w, base_size, crop_size, scale, flip, rotate, blur)
    if d["rs"] is not None:
        image, label = resize_linear(image, *d["rs"]), resize_nearest(label, *d["rs"])
    h, w = label.shape
    if d["angle"] is not None:
        image, label = warp_affine(image, label, rotation_inverse(h, w, d["angle"]))
    if crop_size:
        ph, pw = max(crop_size - h, 0), max(crop_size - w, 0)
        if ph or pw:
            image = np.pad(image, ((0, ph), (0, pw), (0, 0)))
            label = np.pad(label, ((0, ph), (0, pw)))
        sy, sx = d["start"]
        image, label = image[sy:sy + crop_size, sx:sx + crop_size], label[sy:sy + crop_size, sx:sx + crop_size]
    if d["flip"]:
        image, label = image[:, ::-1], label[:, ::-1]
    if d["sigma"] is not None:
        sigma = d["sigma"]
        ksize = int(3.3 * sigma)
        ksize = ksize + 1 if ksize % 2 == 0 else ksize
        image = gaussian_blur(np.ascontiguousarray(image), ksize, sigma)
    x = (image.astype(f32) / f32(255) - np.asarray(mean, dtype=f32)) / np.asarray(std, dtype=f32)
    return np.ascontiguousarray(x.transpose(2, 0, 1)), np.ascontiguousarray(label).astype(np.int64), d
