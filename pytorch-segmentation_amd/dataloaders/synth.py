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
import torch
import torch.distributed as dist


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
