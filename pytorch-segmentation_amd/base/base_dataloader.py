"""BaseDataLoader + DataPrefetcher — the reference's loader surface (base/base_dataloader.py:7-85, used at trainer.py:31-33).

`BaseDataLoader(dataset, batch_size, shuffle, num_workers, val_split)` has the reference's constructor and `get_val_loader()`;
iterating it yields `(input fp32 [N,3,crop,crop] NHWC-backed, target int64 [N,crop,crop])` batches ALREADY ON THE DEVICE: the raw
uint8 samples of a `base.BaseDataSet` are decoded by `num_workers` host threads one batch ahead, staged through one pinned buffer
per batch, and augmented / normalised by the libsegmi kernels (dataloaders/gpu_augment.py) according to the dataset's
`base_size / crop_size / augment / scale / flip / rotate / blur / val` attributes — the values `train_loader.args` of config.json
carries (config.json:14-31 of the reference).  Under torch.distributed every rank draws its own shard of each global step.

DataPrefetcher — host->HBM input staging for loaders that still produce host tensors.

Same surface as the reference class: `DataPrefetcher(loader, device, stop_after=None)`, `len()`, iteration yields
`(input, target)` already on the device, `.dataset` / `.loader` pass-through (plus `batch_size`, `MEAN`, `STD`, which the trainer
reads from its loader).  What is different underneath:

  * the reference calls `.cuda(non_blocking=True)` on whatever the DataLoader produced; from pageable memory that copy is
    staged and synchronous.  Here every batch goes through one of TWO pinned staging slots (allocated once per tensor shape and
    reused), so the H2D transfer is a real asynchronous DMA on a side HIP stream while the compute stream runs the previous
    step (one cfg2 batch is 25 MB + 17 MB of labels: ~0.8 ms of PCIe time hidden behind a 90 ms step);
  * a slot is reused only after the event recorded behind its H2D copy has completed; the device tensors handed to the caller
    are `record_stream`-ed on the compute stream, so the caching allocator cannot recycle them under a kernel still reading them;
  * batches that are already device tensors (e.g. `dataloaders.Synth(device=...)`) pass through untouched.
"""
import random
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.distributed as dist


class BaseDataLoader:
    def __init__(self, dataset, batch_size, shuffle, num_workers, val_split=0.0, device=None, seed=None, drop_last=False,
                 rank=None, world=None, _indices=None, _is_val_split=False):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = bool(shuffle)
        self.num_workers = max(1, int(num_workers or 1))
        self.drop_last = drop_last
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        self.seed = 0 if seed is None else int(seed)
        self.epoch = 0
        ddp = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank() if ddp else 0) if rank is None else int(rank)
        self.world = (dist.get_world_size() if ddp else 1) if world is None else int(world)
        self.nbr_examples = len(dataset)
        self.val_indices = None
        if _indices is not None:
            self.indices = np.asarray(_indices)
        elif val_split:
            # the reference's split (base/base_dataloader.py:25-44): a fixed permutation under np.random.seed(0), the first
            # `val_split` share validates; the training subset is then drawn through a SubsetRandomSampler, i.e. in a NEW random
            # order every epoch whatever `shuffle` said (:43-44 sets shuffle = False only because a sampler is given)
            self.shuffle = True
            split = int(self.nbr_examples * val_split)
            order = np.random.RandomState(0).permutation(self.nbr_examples)
            self.indices, self.val_indices = order[split:], order[:split]
        else:
            self.indices = np.arange(self.nbr_examples)
        self.nbr_examples = len(self.indices)
        # validation-type loaders (a dataset in val mode, or the held-out split) visit every sample exactly once, so their ranks
        # may see batch counts that differ by one: the consumer must run NO collective per step (Trainer._valid_epoch evaluates
        # the loss per rank and all-reduces once per epoch; BN is in eval mode); training loaders need the SAME number of steps
        # on every rank (gradient all-reduce, SyncBN, global-batch loss) and wrap around instead
        self._validation = bool(_is_val_split or getattr(dataset, "val", False))
        self._init_kwargs = dict(batch_size=batch_size, shuffle=False, num_workers=num_workers, device=device, seed=seed, drop_last=drop_last,
                                 rank=self.rank, world=self.world)      # the held-out split shards like its parent
        self._augment = None

    def get_val_loader(self):
        if self.val_indices is None:
            return None
        return BaseDataLoader(self.dataset, _indices=self.val_indices, _is_val_split=True, **self._init_kwargs)

    # ---- batching
    def _batches(self):
        idx = np.array(self.indices)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + 1000003 * self.epoch)
            idx = idx[torch.randperm(len(idx), generator=g).numpy()]
        nb = self._num_batches()
        all_b = [idx[i * self.batch_size:(i + 1) * self.batch_size] for i in range(nb)]
        if self.world == 1:
            return all_b
        if nb == 0:
            return []
        if self._validation:
            # every batch exactly once over the ranks (rank r takes batches r, r + world, ...): no sample is dropped or seen twice,
            # the epoch's metric counters are all-reduced; ranks may differ by one batch
            return [all_b[j] for j in range(self.rank, nb, self.world)]
        # training: ceil(nb / world) global steps on EVERY rank; the last global step wraps around to the epoch's first batches
        # (torch's DistributedSampler pads the same way) instead of silently dropping up to world - 1 batches per epoch
        steps = -(-nb // self.world)
        return [all_b[(i * self.world + self.rank) % nb] for i in range(steps)]  # this rank's shard of global step i

    def _num_batches(self):
        n = len(self.indices)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __len__(self):
        nb = self._num_batches()
        if self.world == 1 or nb == 0:
            return nb
        if self._validation:
            return len(range(self.rank, nb, self.world))
        return -(-nb // self.world)

    def _augmenter(self):
        if self._augment is None:
            from dataloaders.gpu_augment import GPUAugment
            ds = self.dataset
            train = bool(ds.augment) and not ds.val
            if train and not ds.crop_size:
                raise ValueError("%s: crop_size is required to batch augmented samples on the device (the reference's configs set it)" % type(self).__name__)
            self._augment = GPUAugment(ds.mean, ds.std, base_size=ds.base_size if train else None, crop_size=ds.crop_size,
                                       scale=ds.scale if train else False, flip=ds.flip if train else False,
                                       rotate=ds.rotate if train else False, blur=ds.blur if train else False, device=self.device,
                                       seed=self.seed * 7919 + self.rank if self.seed else None)
        return self._augment

    def __iter__(self):
        batches = self._batches()
        self.epoch += 1
        aug = self._augmenter()
        ds = self.dataset
        with ThreadPoolExecutor(max_workers=self.num_workers) as pool:
            def submit(b):
                return [pool.submit(ds.__getitem__, int(i)) for i in b]
            pending = submit(batches[0]) if batches else None
            for k in range(len(batches)):
                raw = [f.result() for f in pending]
                pending = submit(batches[k + 1]) if k + 1 < len(batches) else None      # decode the next batch while this one runs
                pairs = [(r[0], r[1]) for r in raw]
                if ds.val:
                    x, t = aug.validation(pairs)
                elif ds.augment:
                    x, t = aug(pairs)
                else:
                    x, t = aug.plain(pairs)     # neither branch of the reference's __getitem__ (:127-130): ToTensor + Normalize only
                if getattr(ds, "return_id", False):
                    yield x, t, [r[2] for r in raw]
                else:
                    yield x, t


class _PinnedSlot:
    """One pinned host buffer per position of the batch tuple, plus the event guarding its reuse."""

    def __init__(self):
        self.buffers = {}
        self.event = None

    def stage(self, idx, t):
        key = (idx, tuple(t.shape), t.dtype)
        buf = self.buffers.get(key)
        if buf is None:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self.buffers[key] = buf
        buf.copy_(t)
        return buf


class DataPrefetcher(object):
    def __init__(self, loader, device, stop_after=None):
        device = torch.device(device)
        if device.type != "cuda":
            raise ValueError("DataPrefetcher stages batches into HBM; got device %s (the trainer disables prefetch on CPU, "
                             "reference trainer.py:30)" % (device,))
        self.loader = loader
        self.dataset = getattr(loader, "dataset", None)
        self.device = device
        self.stop_after = stop_after
        self.stream = torch.cuda.Stream(device=device)
        self.slots = [_PinnedSlot(), _PinnedSlot()]
        self.turn = 0
        self.next_input = None
        self.next_target = None
        for name in ("batch_size", "MEAN", "STD"):
            if hasattr(loader, name):
                setattr(self, name, getattr(loader, name))

    def __len__(self):
        return len(self.loader)

    def _to_device(self, slot, idx, t):
        if t.is_cuda:
            return t if t.device == self.device else t.to(self.device, non_blocking=True)
        return slot.stage(idx, t).to(self.device, non_blocking=True)

    def preload(self):
        try:
            batch = next(self.loaditer)
        except StopIteration:
            self.next_input = self.next_target = None
            return
        slot = self.slots[self.turn]
        self.turn ^= 1
        if slot.event is not None:
            slot.event.synchronize()          # the H2D copy that last read this slot's pinned buffers has finished
        with torch.cuda.stream(self.stream):
            self.next_input = self._to_device(slot, 0, batch[0])
            self.next_target = self._to_device(slot, 1, batch[1])
            slot.event = torch.cuda.Event()
            slot.event.record(self.stream)

    def __iter__(self):
        count = 0
        self.loaditer = iter(self.loader)
        self.preload()
        while self.next_input is not None:
            current = torch.cuda.current_stream(self.device)
            current.wait_stream(self.stream)
            data, target = self.next_input, self.next_target
            data.record_stream(current)
            target.record_stream(current)
            self.preload()
            count += 1
            yield data, target
            if type(self.stop_after) is int and (count > self.stop_after):     # reference semantics: stop_after + 1 batches
                break
