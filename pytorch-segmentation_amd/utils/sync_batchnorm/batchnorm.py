"""SynchronizedBatchNorm2d: batch statistics over the GLOBAL batch of all ranks.

Reference: utils/sync_batchnorm/batchnorm.py:51-145 (`_SynchronizedBatchNorm`), :211-272
(`SynchronizedBatchNorm2d`), :353-394 (`convert_model`).  Differences by design:
  * statistics are merged from per-rank Welford partials (count, mean, M2) with Chan's formula in the
    `segmi_bn_finalize` kernel, which reproduces `F.batch_norm` on the concatenated batch (what the
    reference's own CPU fallback :65-68 computes); the reference's GPU branch uses E[x^2]-E[x]^2 with a
    variance clamp (:128-145) — available as `clamp_var=True`;
  * every rank updates its running statistics with the same global values (the reference updates the
    master copy only, :137-143);
  * outside an initialised process group (or world size 1) it is exactly segmi.nn.BatchNorm2d.
"""
import contextlib

import torch.nn as nn

from segmi import nn as snn
from segmi.distributed import SyncBNContext


class SynchronizedBatchNorm2d(snn.BatchNorm2d):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, process_group=None, clamp_var=False):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine)
        self.sync = SyncBNContext(process_group, clamp_mode=1 if clamp_var else 0)

    def _check_input_dim(self, input):
        if input.dim() != 4:
            raise ValueError("expected 4D input (got {}D input)".format(input.dim()))


@contextlib.contextmanager
def patch_sync_batchnorm():
    """Temporarily make nn.BatchNorm2d construct synchronized layers (reference :339-350)."""
    backup = nn.BatchNorm2d
    nn.BatchNorm2d = SynchronizedBatchNorm2d
    try:
        yield
    finally:
        nn.BatchNorm2d = backup


def convert_model(module, process_group=None):
    """Recursively replace every nn.BatchNorm2d (incl. segmi.nn.BatchNorm2d) by SynchronizedBatchNorm2d,
    sharing its parameters and running statistics (reference :353-394)."""
    from .replicate import DataParallelWithCallback
    if isinstance(module, nn.DataParallel):
        return DataParallelWithCallback(convert_model(module.module, process_group))
    mod = module
    if isinstance(module, nn.BatchNorm2d) and not isinstance(module, SynchronizedBatchNorm2d):
        mod = SynchronizedBatchNorm2d(module.num_features, module.eps, module.momentum, module.affine, process_group)
        mod.running_mean = module.running_mean
        mod.running_var = module.running_var
        mod.num_batches_tracked = module.num_batches_tracked
        if module.affine:
            mod.weight = module.weight
            mod.bias = module.bias
        mod.train(module.training)
    for name, child in module.named_children():
        mod.add_module(name, convert_model(child, process_group))
    return mod
