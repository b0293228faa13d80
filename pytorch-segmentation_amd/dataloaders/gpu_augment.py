"""Training / validation augmentation on the MI355X: the reference's `BaseDataSet._augmentation`,
`_val_augmentation` and `__getitem__` (base/base_dataset.py:40-136: random rescale of the long side, +-10 degree rotation, zero
padding + random crop, horizontal flip, Gaussian blur, ToTensor, Normalize) with the pixel work in libsegmi kernels
(csrc/augment.hip) instead of cv2 / PIL on the host cores.

    aug = GPUAugment(mean, std, base_size=400, crop_size=380, scale=True, flip=True, rotate=True, blur=False)   # reference kwargs
    images, labels = aug([(image_u8_hwc, label_hw), ...])        # -> fp32 [N,3,crop,crop] NHWC-backed, int64 [N,crop,crop], on the GPU

The random decisions are drawn with a `random.Random` in the REFERENCE'S ORDER (long side, angle, crop origin, flip, sigma), so a
run seeded like the reference takes the same decisions.  The pixel arithmetic is OpenCV's fixed-point arithmetic for 8-bit images,
evaluated in integers by the kernels; what OpenCV derives in double / float per output row and column (source offsets, 11-bit
coefficients, the inverted affine map scaled by 2^10, the 8.8 Gaussian taps) is computed HERE with numpy in the same way and
handed to the kernels as small int32 tables (layouts: include/segmi.h).  Raw samples travel as uint8 / int32 (4x less PCIe
traffic than the fp32 tensors the reference's workers produce) and the normalised crops are written straight into the batch the
model consumes.  Blur placement: the reference blurs AFTER the crop and flip; so does this class.
"""
import ctypes as C
import math
import random

import numpy as np
import torch

from segmi import ops
from segmi._lib import check, lib


def _f32(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


# ---- host-side tables (OpenCV's own double / float derivations, imgproc/src/resize.cpp and imgwarp.cpp)
def _linear_axis(dst, src, clamp):
    """cv::resize INTER_LINEAR, one axis: source index and the two 11-bit coefficients per output index.  clamp: the x axis
    (index clamped and coefficient zeroed at the borders); the y axis keeps both and the kernel clips the rows."""
    scale = 1.0 / (float(dst) / float(src))
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp:
        lo, hi = s < 0, s >= src - 1
        f[lo | hi] = 0
        s = np.where(lo, 0, np.where(hi, src - 1, s))
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s.astype(np.int32), ((c0 & 0xFFFF) | (c1 << 16)).astype(np.int32)


def _is_area_2x(dst, src):
    inv = float(dst) / float(src)
    isc = int(np.rint(1.0 / inv))
    return isc == 2 and abs(inv - 1.0 / isc) < np.finfo(np.float64).eps


def _pil_nearest_axis(dst, src):
    """Source index per output index of Pillow's Image.resize(NEAREST) (libImaging/Geometry.c ImagingScaleAffine): the source
    coordinate starts at 0.5 * src/dst and is ACCUMULATED in double, one addition per output index, then truncated."""
    a = float(src) / float(dst)
    steps = np.full(dst, a, dtype=np.float64)
    steps[0] = a * 0.5
    return np.minimum(np.add.accumulate(steps).astype(np.int64), src - 1)


def resize_tables(sh, sw, dh, dw, label_filter="cv2"):
    """(int32 table xs | xa | ys | yb | lx | ly, area2x flag) for segmi_aug_resize.  label_filter: "cv2" = cv2.resize INTER_NEAREST
    (training, base_dataset.py:72), "pil" = PIL Image.resize(NEAREST) (validation, :49)."""
    xs, xa = _linear_axis(dw, sw, True)
    ys, yb = _linear_axis(dh, sh, False)
    if label_filter == "pil":
        lx, ly = _pil_nearest_axis(dw, sw), _pil_nearest_axis(dh, sh)
    else:
        lx = np.minimum(np.floor(np.arange(dw, dtype=np.float64) * (1.0 / (float(dw) / float(sw)))), sw - 1)
        ly = np.minimum(np.floor(np.arange(dh, dtype=np.float64) * (1.0 / (float(dh) / float(sh)))), sh - 1)
    tab = np.concatenate([xs, xa, ys, yb, lx.astype(np.int32), ly.astype(np.int32)])
    return tab, bool(_is_area_2x(dh, sh) and _is_area_2x(dw, sw))


def rotation_tables(h, w, angle_deg):
    """int32 table adelta[w] | bdelta[w] | X0[h] | Y0[h] for segmi_aug_rotate: cv::getRotationMatrix2D(Point2f(w/2, h/2), angle, 1)
    inverted and scaled by 2^10 exactly as cv::warpAffine does it (double arithmetic, cvRound = round half to even)."""
    ang = angle_deg * (math.pi / 180.0)
    alpha, beta = math.cos(ang), math.sin(ang)
    cx, cy = float(np.float32(w / 2)), float(np.float32(h / 2))
    m = [alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy]
    det = m[0] * m[4] - m[1] * m[3]
    det = 1.0 / det if det != 0 else 0.0
    a11, a22 = m[4] * det, m[0] * det
    m[0] = a11
    m[1] *= -det
    m[3] *= -det
    m[4] = a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    x, y = np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64)
    return np.concatenate([np.rint(m[0] * x * 1024), np.rint(m[3] * x * 1024), np.rint((m[1] * y + m[2]) * 1024),
                           np.rint((m[4] * y + m[5]) * 1024)]).astype(np.int32)


def gaussian_taps_3(sigma):
    """(m0, m1): cv2.GaussianBlur's 3-tap kernel {m0, m1, m0} for CV_8U in 8.8 fixed point (getGaussianKernelBitExact: exp(-x^2 / (8
    sigma^2)) at x = -2, 0, 2 normalised in double; getGaussianKernelFixedPoint_ED: m0 = round(k0 * 256), centre = 256 - 2 m0)."""
    t = math.exp(4.0 * (-0.125 / (sigma * sigma)))
    k0 = t * (1.0 / (2.0 * t + 1.0))
    m0 = int(np.rint(k0 * 256.0))
    return m0, 256 - 2 * m0


class GPUAugment:
    def __init__(self, mean, std, base_size=None, crop_size=321, scale=True, flip=True, rotate=False, blur=False, device="cuda", seed=None):
        self.mean, self.std = _f32(mean), _f32(std)
        self.base_size, self.crop_size = base_size, int(crop_size or 0)      # crop_size 0 / None: only plain() is available
        self.scale, self.flip, self.rotate, self.blur = scale, flip, rotate, blur
        self.device = torch.device(device)
        self.rng = random.Random(seed) if seed is not None else random
        self.decisions = []          # of the last batch (tests / logging)

    def _draw(self, h, w):
        """base/base_dataset.py:67-116, same draws in the same order."""
        d = {"rs": None, "angle": None, "start": None, "flip": False, "sigma": None}
        if self.base_size:
            longside = self.rng.randint(int(self.base_size * 0.5), int(self.base_size * 2.0)) if self.scale else self.base_size
            h, w = (longside, int(1.0 * longside * w / h + 0.5)) if h > w else (int(1.0 * longside * h / w + 0.5), longside)
            d["rs"] = (h, w)
        if self.rotate:
            d["angle"] = self.rng.randint(-10, 10)
        ph, pw = max(h, self.crop_size), max(w, self.crop_size)
        d["start"] = (self.rng.randint(0, ph - self.crop_size), self.rng.randint(0, pw - self.crop_size))
        if self.flip:
            d["flip"] = self.rng.random() > 0.5
        if self.blur:
            d["sigma"] = self.rng.random()
        return d

    # ---- staging
    def _upload(self, samples):
        """Raw samples -> device tensors.  Host arrays of the whole batch go through ONE pinned buffer and one asynchronous copy;
        tensors that already live on the device are used in place."""
        dev = self.device
        out, host, total = [], [], 0
        for image, label in samples:
            if torch.is_tensor(image) and image.is_cuda:
                out.append((image.to(torch.uint8).contiguous(), label.to(dev, torch.int32).contiguous()))
                continue
            img = np.ascontiguousarray(image.cpu().numpy() if torch.is_tensor(image) else image)
            lab = np.ascontiguousarray(label.cpu().numpy() if torch.is_tensor(label) else label)
            if img.dtype != np.uint8:
                img = img.astype(np.uint8)           # the reference's `Image.fromarray(np.uint8(image))`, base_dataset.py:133
            lab = lab.astype(np.int32, copy=False)
            if img.ndim != 3 or img.shape[2] != 3 or img.shape[:2] != lab.shape:
                raise ValueError("GPUAugment: image must be [H,W,3] matching the label [H,W]; got %s / %s" % (img.shape, lab.shape))
            nb = (img.nbytes + 15) & ~15
            host.append((len(out), img, lab, total, total + nb))
            total += nb + ((lab.nbytes + 15) & ~15)
            out.append(None)
        if host:
            pinned = torch.empty(total, dtype=torch.uint8, pin_memory=True)
            pv = pinned.numpy()
            for _, img, lab, o0, o1 in host:
                pv[o0:o0 + img.nbytes] = img.reshape(-1)
                pv[o1:o1 + lab.nbytes] = lab.reshape(-1).view(np.uint8)
            devbuf = pinned.to(dev, non_blocking=True)
            self._keep += [pinned, devbuf]
            for i, img, lab, o0, o1 in host:
                h, w = lab.shape
                out[i] = (devbuf[o0:o0 + img.nbytes].view(h, w, 3), devbuf[o1:o1 + lab.nbytes].view(torch.int32).view(h, w))
        return out

    def _tables(self, arr):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)).to(self.device, non_blocking=True)
        self._keep.append(t)
        return t

    def _resize(self, img, lab, h, w, h2, w2, label_filter, st):
        dev = self.device
        tab, area = resize_tables(h, w, h2, w2, label_filter)
        tab = self._tables(tab)
        img2 = torch.empty((h2, w2, 3), dtype=torch.uint8, device=dev)
        lab2 = torch.empty((h2, w2), dtype=torch.int32, device=dev)
        check(lib.segmi_aug_resize(img.data_ptr(), lab.data_ptr(), h, w, img2.data_ptr(), lab2.data_ptr(), h2, w2, tab.data_ptr(), 1 if area else 0, st),
              "aug_resize")
        self._keep += [img, lab]
        return img2, lab2

    def _finish(self, i, img, lab, h, w, sy, sx, flip, out, labels, st):
        ch, cw, ld = int(out.shape[2]), int(out.shape[3]), ops.ld_of(out)
        check(lib.segmi_aug_finish(img.data_ptr(), lab.data_ptr(), h, w, ch, cw, sy, sx, flip, self.mean, self.std,
                                   out.data_ptr() + 4 * i * ch * cw * ld, ld, labels.data_ptr() + 8 * i * ch * cw, st), "aug_finish")
        self._keep += [img, lab]

    def __call__(self, samples):
        """samples: iterable of (image uint8 [H,W,3], label integer [H,W]) numpy arrays or tensors (host or device)."""
        samples = list(samples)
        n, crop, dev = len(samples), self.crop_size, self.device
        if not crop:
            raise ValueError("GPUAugment batches samples: crop_size is required (the reference's training configs all set it)")
        out = ops.empty_nhwc(n, 3, crop, crop, dev)               # pixel stride 4, channel 3 zeroed by the kernel
        labels = torch.empty((n, crop, crop), dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        self.decisions = []
        self._keep = []                                            # staging buffers stay alive until the stream has consumed them
        for i, (img, lab) in enumerate(self._upload(samples)):
            h, w = int(lab.shape[0]), int(lab.shape[1])
            d = self._draw(h, w)
            self.decisions.append(d)
            if d["rs"] is not None and d["rs"] != (h, w):
                img, lab = self._resize(img, lab, h, w, d["rs"][0], d["rs"][1], "cv2", st)
                h, w = d["rs"]
            if d["angle"] is not None:
                tab = self._tables(rotation_tables(h, w, d["angle"]))
                img2, lab2 = torch.empty_like(img), torch.empty_like(lab)
                check(lib.segmi_aug_rotate(img.data_ptr(), lab.data_ptr(), h, w, tab.data_ptr(), img2.data_ptr(), lab2.data_ptr(), st), "aug_rotate")
                self._keep += [img, lab]
                img, lab = img2, lab2
            sy, sx = d["start"]
            sigma = d["sigma"]
            ksize = 1
            if sigma is not None:
                ksize = int(3.3 * sigma)
                ksize = ksize + 1 if ksize % 2 == 0 else ksize
            if ksize > 1:
                # the reference blurs the CROPPED (and flipped) image: crop first into a uint8 staging image, blur it, then normalise
                cimg = torch.zeros((crop, crop, 3), dtype=torch.uint8, device=dev)
                clab = torch.zeros((crop, crop), dtype=torch.int32, device=dev)
                hh, ww = min(crop, h - sy), min(crop, w - sx)
                cimg[:hh, :ww] = img[sy:sy + hh, sx:sx + ww]
                clab[:hh, :ww] = lab[sy:sy + hh, sx:sx + ww]
                if d["flip"]:
                    cimg, clab = cimg.flip(1).contiguous(), clab.flip(1).contiguous()
                m0, m1 = gaussian_taps_3(sigma)
                scratch = torch.empty((crop, crop, 3), dtype=torch.int16, device=dev)
                bimg = torch.empty_like(cimg)
                check(lib.segmi_aug_blur(cimg.data_ptr(), crop, crop, m0, m1, scratch.data_ptr(), bimg.data_ptr(), st), "aug_blur")
                self._keep += [img, lab, cimg, scratch]
                self._finish(i, bimg, clab, crop, crop, 0, 0, 0, out, labels, st)
            else:
                self._finish(i, img, lab, h, w, sy, sx, 1 if d["flip"] else 0, out, labels, st)
        return out, labels

    def validation(self, samples):
        """`_val_augmentation` (base/base_dataset.py:40-61): smaller side -> crop_size (cv2 bilinear image, PIL nearest label),
        centre crop, ToTensor + Normalize.  Deterministic: no draws."""
        samples = list(samples)
        n, crop, dev = len(samples), self.crop_size, self.device
        if not crop:
            return self.plain(samples)                             # `if self.crop_size:` of the reference (:41)
        out = ops.empty_nhwc(n, 3, crop, crop, dev)
        labels = torch.empty((n, crop, crop), dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        self._keep = []
        for i, (img, lab) in enumerate(self._upload(samples)):
            h, w = int(lab.shape[0]), int(lab.shape[1])
            h2, w2 = (crop, int(crop * w / h)) if h < w else (int(crop * h / w), crop)
            if (h2, w2) != (h, w):
                img, lab = self._resize(img, lab, h, w, h2, w2, "pil", st)
            self._finish(i, img, lab, h2, w2, (h2 - crop) // 2, (w2 - crop) // 2, 0, out, labels, st)
        return out, labels

    def plain(self, samples):
        """augment=False, val=False (base/base_dataset.py:125-136 alone): ToTensor + Normalize of equally-sized samples."""
        samples = list(samples)
        dev = self.device
        st = torch.cuda.current_stream().cuda_stream
        self._keep = []
        up = self._upload(samples)
        h, w = int(up[0][1].shape[0]), int(up[0][1].shape[1])
        if any(tuple(lab.shape) != (h, w) for _, lab in up):
            raise ValueError("GPUAugment.plain: samples of different sizes cannot be batched without crop_size / augmentation")
        out = ops.empty_nhwc(len(up), 3, h, w, dev)
        labels = torch.empty((len(up), h, w), dtype=torch.int64, device=dev)
        for i, (img, lab) in enumerate(up):
            self._finish(i, img, lab, h, w, 0, 0, 0, out, labels, st)
        return out, labels
