"""numpy restatement of the reference's training / validation augmentation (test infrastructure — only tests/ may import it).

Follows base/base_dataset.py:40-61 (`_val_augmentation`), :63-120 (`_augmentation`) and :125-136 (`__getitem__`: label -> int64,
ToTensor, Normalize) step by step.  The cv2 calls are restated from OpenCV's PUBLISHED FIXED-POINT ALGORITHMS for 8-bit images
(modules/imgproc/src of OpenCV 4.5+; uint8 paths never go through float), bit for bit in integer arithmetic:

  cv2.resize INTER_LINEAR (resize.cpp: resize(), HResizeLinear, VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>)
      per axis  f = (float)((d + 0.5) * scale - 0.5), s = floor(f), f -= s   with  scale = 1 / ((double)dst / src);
      x axis: s < 0 -> (s, f) = (0, 0); s >= src-1 -> (src-1, 0);   y axis: f kept, the two ROWS are clipped to [0, src-1];
      coefficients  short(rint((1-f) * 2048)), short(rint(f * 2048))    (INTER_RESIZE_COEF_BITS = 11, saturate_cast = round half even)
      rows:   D = S[s] * a0 + S[s+1] * a1            (int)
      out :   uchar((((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2)
      exact 2x2 decimation (both scales == 2): OpenCV switches INTER_LINEAR to INTER_AREA: (a + b + c + d + 2) >> 2
  cv2.resize INTER_NEAREST (resizeNN)            s = min(floor(d * (1 / ((double)dst / src))), src - 1)
  PIL Image.resize(NEAREST) (validation labels)  xo = 0.5 * a, s[d] = min((int)xo, src - 1), xo += a with a = (double)src / dst
      (Geometry.c ImagingScaleAffine: the coordinate is accumulated; PINNED to the installed Pillow by tests/test_augment.py)
  cv2.getRotationMatrix2D(Point2f(w/2, h/2), angle, 1)   double alpha = cos, beta = sin of angle * pi / 180
  cv2.warpAffine (imgwarp.cpp WarpAffineInvoker + remapBilinear<FixedPtCast<int,uchar,15>>): M inverted in double as the code does,
      adelta[x] = rint(M0 * x * 1024), bdelta[x] = rint(M3 * x * 1024), X0 = rint((M1 * y + M2) * 1024) + delta, Y0 likewise
      (AB_BITS = 10; delta = 16 for INTER_LINEAR, 512 for INTER_NEAREST);
      linear : X = (X0 + adelta[x]) >> 5 on the 1/32-pixel grid (INTER_BITS = 5), integer pixel X >> 5, fraction X & 31, weights
               32 * (32 - fy | fy) * (32 - fx | fx) (they sum to 2^15 exactly, so initInterTab2D's correction never fires),
               out = (sum S * w + 2^14) >> 15, taps outside the image = border value 0, fully outside -> 0
      nearest: X = (X0 + adelta[x]) >> 10, outside -> 0
  cv2.GaussianBlur on CV_8U (smooth.dispatch.cpp: getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED, fixed point 8.8;
      smooth.simd.hpp hlineSmooth3N / vlineSmooth3N on ufixedpoint16): ksize = int(3.3 sigma) made odd is 1 (copy) or 3 for the
      reference's sigma in [0, 1):  k = exp(-x^2 / (8 sigma^2)) for x = -2, 0, 2, normalised in double;  m0 = rint(k0 * 256),
      m1 = 256 - 2 m0;  rows t = m0 * (s[x-1] + s[x+1]) + m1 * s[x] (uint16), columns (m0 * (t[y-1] + t[y+1]) + m1 * t[y] + 2^15) >> 16,
      BORDER_REFLECT_101 on both axes.

PARITY: the PIL piece (validation label resize) is PINNED against the installed Pillow 12.2.0 by
tests/test_augment.py::test_pil_nearest_resize_is_pinned_to_the_installed_pillow (the pin corrected this file: Pillow accumulates the
source coordinate).  The cv2 pieces are UNPINNED: cv2 (opencv-python, unpinned in the reference's requirements.txt) and its source are
absent from this image and from /root/reference, so they are restated from the library's published algorithm, not checked against it; a
build of OpenCV that routes these calls through IPP / OpenCL / a platform HAL may round differently, and softfloat's exp() in
getGaussianKernelBitExact could differ from numpy's by one ulp (it matters only if k0 * 256 lands within 1e-13 of a half).
The random decisions are drawn with Python's `random` in the reference's order, so a seeded run takes the same decisions.
"""
import math
import random

import numpy as np

f32 = np.float32


def _rint(v):
    """cvRound / saturate_cast<int|short>: round half to even."""
    return np.rint(v).astype(np.int64)


# ------------------------------------------------------------------------------------------------ cv2.resize
def linear_axis_tables(dst, src, clamp_coeffs):
    """(s, c0, c1) of one axis of cv::resize INTER_LINEAR for 8-bit images.  clamp_coeffs: the x axis (offsets clamped, the
    coefficient zeroed at the borders); the y axis keeps its coefficients and clips the rows when they are read."""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(f32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(f32)).astype(f32)
    if clamp_coeffs:
        lo, hi = s < 0, s >= src - 1
        f = np.where(lo | hi, f32(0), f).astype(f32)
        s = np.where(lo, 0, np.where(hi, src - 1, s))
    c0 = _rint((f32(1) - f) * f32(2048))
    c1 = _rint(f * f32(2048))
    return s, c0, c1


def _is_area_2x(dst, src):
    inv = float(dst) / float(src)
    scale = 1.0 / inv
    isc = int(np.rint(scale))
    return isc == 2 and abs(inv - 1.0 / isc) < np.finfo(np.float64).eps


def resize_linear(img, dh, dw):
    sh, sw = img.shape[:2]
    if _is_area_2x(dh, sh) and _is_area_2x(dw, sw):                      # INTER_LINEAR -> INTER_AREA (resizeAreaFast, 2x2 box)
        s = img.astype(np.int64)
        return ((s[0:2 * dh:2, 0:2 * dw:2] + s[0:2 * dh:2, 1:2 * dw:2] + s[1:2 * dh:2, 0:2 * dw:2] + s[1:2 * dh:2, 1:2 * dw:2] + 2) >> 2).astype(np.uint8)
    xs, a0, a1 = linear_axis_tables(dw, sw, True)
    ys, b0, b1 = linear_axis_tables(dh, sh, False)
    S = img.astype(np.int64)
    x1 = np.minimum(xs + 1, sw - 1)                                      # (a1 == 0 wherever xs + 1 would leave the image)
    rows = S[:, xs] * a0[None, :, None] + S[:, x1] * a1[None, :, None]   # HResizeLinear: [sh, dw, 3] ints
    y0, y1 = np.clip(ys, 0, sh - 1), np.clip(ys + 1, 0, sh - 1)
    D0, D1 = rows[y0], rows[y1]
    b0, b1 = b0[:, None, None], b1[:, None, None]
    return ((((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2).astype(np.uint8)


def nearest_axis_table(dst, src):
    ifx = 1.0 / (float(dst) / float(src))
    return np.minimum(np.floor(np.arange(dst, dtype=np.float64) * ifx).astype(np.int64), src - 1)


def resize_nearest(lab, dh, dw):
    sh, sw = lab.shape
    return lab[nearest_axis_table(dh, sh)][:, nearest_axis_table(dw, sw)]


def pil_nearest_axis_table(dst, src):
    """Pillow's Image.resize(NEAREST) = ImagingScaleAffine (libImaging/Geometry.c): a = (double)src / dst; xo = a * 0.5; per output
    index: xin = (int)xo, then xo += a — the source coordinate is ACCUMULATED in double, not computed as (d + 0.5) * a, so an exact
    boundary such as (3 + 0.5) * 2 / 7 = 1.0 is reached as 0.999...: PINNED against the installed Pillow (12.2.0) by
    tests/test_augment.py::test_pil_nearest_resize_is_pinned_to_the_installed_pillow (the closed form differs from the library
    on 2 -> 7, 3 -> 7 and similar ratios)."""
    a = float(src) / float(dst)
    steps = np.full(dst, a, dtype=np.float64)
    steps[0] = a * 0.5
    return np.minimum(np.add.accumulate(steps).astype(np.int64), src - 1)


def resize_nearest_pil(lab, dh, dw):
    sh, sw = lab.shape
    return lab[pil_nearest_axis_table(dh, sh)][:, pil_nearest_axis_table(dw, sw)]


# ------------------------------------------------------------------------------------------------ cv2.warpAffine
def rotation_tables(h, w, angle_deg):
    """(adelta[w], bdelta[w], X0[h], Y0[h]) of cv::warpAffine for M = getRotationMatrix2D((w/2, h/2), angle, 1) — the fixed-point
    (AB_BITS = 10) source coordinates WITHOUT the interpolation's rounding offset."""
    ang = angle_deg * (math.pi / 180.0)
    alpha, beta = math.cos(ang), math.sin(ang)
    cx, cy = float(f32(w / 2)), float(f32(h / 2))                        # Point2f centre
    M = [alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy]
    D = M[0] * M[4] - M[1] * M[3]                                        # invertAffineTransform as written in cv::warpAffine
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    x, y = np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64)
    return _rint(M[0] * x * 1024), _rint(M[3] * x * 1024), _rint((M[1] * y + M[2]) * 1024), _rint((M[4] * y + M[5]) * 1024)


def warp_affine(img, lab, h, w, angle_deg):
    ad, bd, X0, Y0 = rotation_tables(h, w, angle_deg)
    # bilinear image on the 1/32-pixel grid
    X = (X0[:, None] + 16 + ad[None, :]) >> 5
    Y = (Y0[:, None] + 16 + bd[None, :]) >> 5
    sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31
    S = img.astype(np.int64)

    def px(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return np.where(ok[..., None], S[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0)

    w00, w01 = (32 * (32 - fy) * (32 - fx))[..., None], (32 * (32 - fy) * fx)[..., None]
    w10, w11 = (32 * fy * (32 - fx))[..., None], (32 * fy * fx)[..., None]
    v = (px(sy, sx) * w00 + px(sy, sx + 1) * w01 + px(sy + 1, sx) * w10 + px(sy + 1, sx + 1) * w11 + (1 << 14)) >> 15
    out = np.clip(v, 0, 255).astype(np.uint8)
    # nearest label
    nx = (X0[:, None] + 512 + ad[None, :]) >> 10
    ny = (Y0[:, None] + 512 + bd[None, :]) >> 10
    ok = (ny >= 0) & (ny < h) & (nx >= 0) & (nx < w)
    return out, np.where(ok, lab[np.clip(ny, 0, h - 1), np.clip(nx, 0, w - 1)], 0).astype(lab.dtype)


# ------------------------------------------------------------------------------------------------ cv2.GaussianBlur (CV_8U)
def gaussian_kernel_fixed3(sigma):
    """(m0, m1): the 3-tap kernel {m0, m1, m0} in 8.8 fixed point, getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED."""
    scale2x = -0.125 / (sigma * sigma)
    t = math.exp(4.0 * scale2x)                                          # x = 1 - n = -2: exp(x*x * scale2X)
    total = 2.0 * t + 1.0
    k0 = t * (1.0 / total)
    m0 = int(np.rint(k0 * 256.0))
    return m0, 256 - 2 * m0


def _reflect101(i, n):
    if n == 1:
        return np.zeros_like(i)
    i = np.abs(i)
    i = np.where(i >= n, 2 * (n - 1) - i, i)
    return np.abs(i)


def gaussian_blur(img, ksize, sigma):
    if ksize <= 1:
        return img.copy()
    if ksize != 3:
        raise ValueError("the reference draws sigma in [0, 1): ksize is 1 or 3")
    m0, m1 = gaussian_kernel_fixed3(sigma)
    h, w = img.shape[:2]
    S = img.astype(np.int64)
    xl, xr = _reflect101(np.arange(w) - 1, w), _reflect101(np.arange(w) + 1, w)
    t = m0 * (S[:, xl] + S[:, xr]) + m1 * S                              # ufixedpoint16, <= 255 * 256
    yu, yd = _reflect101(np.arange(h) - 1, h), _reflect101(np.arange(h) + 1, h)
    return np.clip((m0 * (t[yu] + t[yd]) + m1 * t + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ the reference's sequences
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


def _to_tensor_normalize(image, label, mean, std):
    x = (image.astype(f32) / f32(255) - np.asarray(mean, dtype=f32)) / np.asarray(std, dtype=f32)
    return np.ascontiguousarray(x.transpose(2, 0, 1)), np.ascontiguousarray(label).astype(np.int64)


def augment(image, label, mean, std, base_size=None, crop_size=321, scale=True, flip=True, rotate=False, blur=False, rng=None):
    """image uint8 [H,W,3], label int32 [H,W] -> (float32 [3,crop,crop] normalised, int64 [crop,crop], decisions) like the
    reference's `__getitem__` with augment=True."""
    rng = rng or random
    h, w = label.shape
    d = draw_decisions(rng, h, w, base_size, crop_size, scale, flip, rotate, blur)
    if d["rs"] is not None:
        image, label = resize_linear(image, *d["rs"]), resize_nearest(label, *d["rs"])
    h, w = label.shape
    if d["angle"] is not None:
        image, label = warp_affine(image, label, h, w, d["angle"])
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
    x, t = _to_tensor_normalize(image, label, mean, std)
    return x, t, d


def val_augment(image, label, mean, std, crop_size):
    """base/base_dataset.py:40-61 + :125-136: smaller side -> crop_size (cv2 linear image, PIL nearest label), centre crop."""
    if crop_size:
        h, w = label.shape
        h, w = (crop_size, int(crop_size * w / h)) if h < w else (int(crop_size * h / w), crop_size)
        image, label = resize_linear(image, h, w), resize_nearest_pil(label, h, w)
        sh, sw = (h - crop_size) // 2, (w - crop_size) // 2
        image, label = image[sh:sh + crop_size, sw:sw + crop_size], label[sh:sh + crop_size, sw:sw + crop_size]
    return _to_tensor_normalize(image, label, mean, std)
