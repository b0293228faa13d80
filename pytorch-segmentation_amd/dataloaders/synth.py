"""Synthetic data loader with the reference's loader surface (base/base_dataloader.py, base/base_dataset.py:125-136).

The reference's real loaders decode images with cv2/PIL on the host; they are outside the hot path and the metric excludes
data loading.  `Synth` yields the same thing the trainer consumes — (image fp32 [N,3,H,W], label int64 [N,H,W]) batches — and
exposes the attributes the trainer reads: `batch_size`, `MEAN`, `STD`, `dataset.num_classes`, `dataset.palette`, `__len__`.
Batches are a pure function of (seed, index), so the reference trainer on CPU and this trainer on the GPU see identical data.
`device=` keeps the batches resident in HBM (what a DataPrefetcher, base/base_dataloader.py:49-85, delivers).

Data parallelism: the reference's nn.DataParallel scatters ONE loader batch over the GPUs; with one process per GPU each rank
draws its own shard instead: iteration i of rank r is global batch i * world + r (`rank` / `world` default to the
torch.distributed group when it is initialised), so W ranks consume W different batches per step and the averaged gradient is
that of the W-times larger global batch.  `len()` is the number of iterations PER RANK.
"""
import numpy as np
import torch
import torch.distributed as dist

from base import BaseDataLoader, BaseDataSet


class _SynthDataset:
    def __init__(self, num_classes):
        self.num_classes = num_classes
        self.palette = [(37 * i) % 256 for i in range(3 * num_classes)]


class Synth:
    MEAN = [0.485, 0.456, 0.406]
    STD = [0.229, 0.224, 0.225]

    def __init__(self, num_classes=2, batch_size=2, height=256, width=256, iters=4, ignore_index=255, seed=1234, block=16,
                 device=None, rank=None, world=None, **_):
        self.dataset = _SynthDataset(num_classes)
        self.batch_size = batch_size
        self.shape = (batch_size, 3, height, width)
        self.iters = iters
        self.ignore_index = ignore_index
        self.seed = seed
        self.block = block
        self.device = device
        ddp = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank() if ddp else 0) if rank is None else int(rank)
        self.world = (dist.get_world_size() if ddp else 1) if world is None else int(world)

    def __len__(self):
        return self.iters

    def batch(self, i):
        n, _, h, w = self.shape
        g = torch.Generator().manual_seed(self.seed + 7919 * i)
        x = torch.randn(self.shape, generator=g)
        b = self.block
        t = torch.randint(0, self.dataset.num_classes, (n, (h + b - 1) // b, (w + b - 1) // b), generator=g)
        t = t.repeat_interleave(b, 1).repeat_interleave(b, 2)[:, :h, :w].contiguous()
        t[:, : max(1, h // 20), :] = self.ignore_index
        x = x + 0.5 * (t.clamp(0, self.dataset.num_classes - 1).unsqueeze(1).float() - 0.5 * (self.dataset.num_classes - 1))   # learnable signal
        if self.device is not None:
            x, t = x.to(self.device), t.to(self.device)
        return x, t

    def __iter__(self):
        for i in range(self.iters):
            yield self.batch(i * self.world + self.rank)      # this rank's shard of global step i


class _SynthImageDataset(BaseDataSet):
    """In-memory stand-in for a file dataset: `num_samples` deterministic uint8 images of RAGGED sizes (a textured background plus
    class-coloured blocks the label map names), produced by `_load_data` exactly where a real dataset decodes a file."""

    def __init__(self, num_classes, num_samples, min_size, max_size, ignore_index, seed, **kwargs):
        self.num_classes = num_classes
        self.palette = [(37 * i) % 256 for i in range(3 * num_classes)]
        self.num_samples, self.min_size, self.max_size, self.ignore_index, self.seed = num_samples, min_size, max_size, ignore_index, seed
        super().__init__(**kwargs)

    def _set_files(self):
        self.files = ["synth_%05d" % i for i in range(self.num_samples)]

    def _load_data(self, index):
        g = np.random.default_rng(self.seed + 104729 * index)
        h, w = (int(v) for v in g.integers(self.min_size, self.max_size + 1, 2))
        b = 16
        lab = g.integers(0, self.num_classes, ((h + b - 1) // b, (w + b - 1) // b)).repeat(b, 0).repeat(b, 1)[:h, :w].astype(np.int32)
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(yy * 3 + xx * 2) % 64, (xx * 5 + yy) % 64, g.integers(0, 64, (h, w))], -1)
        img = (base + (lab[..., None] * (180 // max(1, self.num_classes - 1)))).clip(0, 255).astype(np.uint8)     # learnable signal
        lab[: max(1, h // 20)] = self.ignore_index
        return img, lab, self.files[index]


class SynthImages(BaseDataLoader):
    """`train_loader: {"type": "SynthImages", "args": {...}}` — the reference's loader arguments (config.json:14-31: batch_size,
    base_size, crop_size, augment, shuffle, scale, flip, rotate, blur, num_workers, val) over an in-memory dataset of raw uint8
    images: everything downstream of "a file was decoded" is the real input pipeline (host threads -> pinned staging -> device
    augmentation -> NHWC batch), with no dataset files needed."""
    MEAN = [0.485, 0.456, 0.406]
    STD = [0.229, 0.224, 0.225]

    def __init__(self, num_classes=2, batch_size=2, num_samples=8, min_size=96, max_size=160, crop_size=None, base_size=None, scale=True,
                 num_workers=1, val=False, shuffle=False, flip=False, rotate=False, blur=False, augment=False, val_split=None,
                 return_id=False, ignore_index=255, seed=1234, device=None, rank=None, world=None, **_):
        dataset = _SynthImageDataset(num_classes, num_samples, min_size, max_size, ignore_index, seed, root=None, split="synth",
                                     mean=self.MEAN, std=self.STD, augment=augment, crop_size=crop_size, base_size=base_size, scale=scale,
                                     flip=flip, blur=blur, rotate=rotate, return_id=return_id, val=val)
        super().__init__(dataset, batch_size, shuffle, num_workers, val_split or 0.0, device=device, seed=seed, rank=rank, world=world)
