"""Training-time augmentation on the MI355X: the reference's `BaseDataSet._augmentation` + `__getitem__`
(base/base_dataset.py:63-136: random rescale of the long side, +-10 degree rotation, zero padding + random crop, horizontal
flip, Gaussian blur, ToTensor, Normalize) with the pixel work in libsegmi kernels (csrc/augment.hip) instead of cv2 / PIL on
the host cores.

    aug = GPUAugment(mean, std, base_size=400, crop_size=380, scale=True, flip=True, rotate=True, blur=False)   # reference kwargs
    images, labels = aug([(image_u8_hwc, label_hw), ...])        # -> fp32 [N,3,crop,crop] NHWC-backed, int64 [N,crop,crop], on the GPU

The random decisions are drawn with a `random.Random` in the REFERENCE'S ORDER (long side, angle, crop origin, flip, sigma), so a
run seeded like the reference takes the same decisions; raw samples are uploaded as uint8 / int32 (4x less PCIe traffic than
the fp32 tensors the reference's workers produce) and the normalised crops are written straight into the batch the model
consumes.  Blur placement: the reference blurs AFTER the crop and flip; so does this class.
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


class GPUAugment:
    def __init__(self, mean, std, base_size=None, crop_size=321, scale=True, flip=True, rotate=False, blur=False, device="cuda", seed=None):
        if not crop_size:
            raise ValueError("GPUAugment batches samples: crop_size is required (the reference's training configs all set it)")
        self.mean, self.std = _f32(mean), _f32(std)
        self.base_size, self.crop_size = base_size, int(crop_size)
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

    def __call__(self, samples):
        """samples: iterable of (image uint8 [H,W,3], label integer [H,W]) numpy arrays or tensors (host or device)."""
        samples = list(samples)
        n, crop, dev = len(samples), self.crop_size, self.device
        out = ops.empty_nhwc(n, 3, crop, crop, dev)               # pixel stride 4, channel 3 zeroed by the kernel
        labels = torch.empty((n, crop, crop), dtype=torch.int64, device=dev)
        ld = ops.ld_of(out)
        st = torch.cuda.current_stream().cuda_stream
        self.decisions = []
        keep = []                                                  # staging buffers stay alive until the stream has consumed them
        for i, (image, label) in enumerate(samples):
            img = torch.as_tensor(np.ascontiguousarray(image) if isinstance(image, np.ndarray) else image).to(dev, torch.uint8, non_blocking=True).contiguous()
            lab = torch.as_tensor(np.ascontiguousarray(label) if isinstance(label, np.ndarray) else label).to(dev, torch.int32, non_blocking=True).contiguous()
            h, w = int(lab.shape[0]), int(lab.shape[1])
            if tuple(img.shape) != (h, w, 3):
                raise ValueError("GPUAugment: image must be uint8 [H,W,3] matching the label [H,W]; got %s / %s" % (tuple(img.shape), tuple(lab.shape)))
            d = self._draw(h, w)
            self.decisions.append(d)
            if d["rs"] is not None and d["rs"] != (h, w):
                h2, w2 = d["rs"]
                img2 = torch.empty((h2, w2, 3), dtype=torch.uint8, device=dev)
                lab2 = torch.empty((h2, w2), dtype=torch.int32, device=dev)
                check(lib.segmi_aug_resize(img.data_ptr(), lab.data_ptr(), h, w, img2.data_ptr(), lab2.data_ptr(), h2, w2, st), "aug_resize")
                keep += [img, lab]
                img, lab, h, w = img2, lab2, h2, w2
            if d["angle"] is not None:
                a, b = math.cos(math.radians(d["angle"])), math.sin(math.radians(d["angle"]))
                cx, cy = w / 2.0, h / 2.0
                M = np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy], [0, 0, 1]], dtype=np.float64)
                inv = _f32(np.linalg.inv(M)[:2].reshape(-1))
                img2, lab2 = torch.empty_like(img), torch.empty_like(lab)
                check(lib.segmi_aug_rotate(img.data_ptr(), lab.data_ptr(), h, w, inv, img2.data_ptr(), lab2.data_ptr(), st), "aug_rotate")
                keep += [img, lab]
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
                i64 = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
                k = np.exp(-(i64 * i64) / (2.0 * sigma * sigma))
                k = (k / k.sum()).astype(np.float32)
                half = np.zeros(4, dtype=np.float32)
                half[: ksize // 2 + 1] = k[ksize // 2:]
                scratch = torch.empty((crop, crop, 3), dtype=torch.float32, device=dev)
                bimg = torch.empty_like(cimg)
                check(lib.segmi_aug_blur(cimg.data_ptr(), crop, crop, ksize, _f32(half), scratch.data_ptr(), bimg.data_ptr(), st), "aug_blur")
                keep += [img, lab, cimg, clab, scratch]
                img, lab, h, w, sy, sx, flip = bimg, clab, crop, crop, 0, 0, 0
            else:
                flip = 1 if d["flip"] else 0
            check(lib.segmi_aug_finish(img.data_ptr(), lab.data_ptr(), h, w, crop, crop, sy, sx, flip, self.mean, self.std,
                                       out.data_ptr() + 4 * i * crop * crop * ld, ld, labels.data_ptr() + 8 * i * crop * crop, st), "aug_finish")
            keep += [img, lab]
        self._keep = keep
        return out, labels
