"""Per-pixel losses of the plugin surface (reference: utils/losses.py), on libsegmi kernels.

`getattr(losses, config['loss'])(ignore_index=...)` (train.py:30) resolves these by name;
forward(logits [N,C,H,W] fp32, target [N,H,W] int64) -> scalar.
"""
import os

import torch.nn as nn

from segmi import ops

# Data parallelism: the reference computes every loss on the gathered global batch (nn.DataParallel, trainer.py:56-66).  With one
# process per GPU the losses below reproduce that when `process_group` is "auto" (default: the default torch.distributed group
# once it is initialised with more than one rank), a group object, or None (strictly per-rank maths).  See segmi.ops.


def _group(pg):
    return True if pg == "auto" else pg


class CrossEntropyLoss2d(nn.Module):
    """nn.CrossEntropyLoss(weight, ignore_index, reduction) (reference utils/losses.py:24-31) as one
    fused log-softmax + NLL pass; backward recomputes softmax from the saved log-sum-exp.
    weight: per-class rescaling (weighted mean = sum w_t l / sum w_t over valid pixels); reduction 'mean' | 'sum'."""

    def __init__(self, weight=None, ignore_index=255, reduction="mean", process_group="auto", fuse_upsample=True):
        super().__init__()
        self.fuse_upsample = bool(fuse_upsample)
        if reduction not in ("mean", "sum"):
            raise NotImplementedError("reduction=%r: the fused CE kernel reduces on the device ('mean' | 'sum'); the reference's "
                                      "trainer needs a scalar (trainer.py:66-70)" % (reduction,))
        self.register_buffer("weight", None if weight is None else weight.detach().clone().float())
        self.reduction = reduction
        self.ignore_index = ignore_index
        self.process_group = process_group

    def forward(self, output, target):
        src = ops.upsample_source(output) if self.fuse_upsample else None
        if src is not None and tuple(output.shape[2:]) == tuple(target.shape[1:]):
            # `output` is the model's final F.interpolate of low-resolution logits, untouched (models/pspnet.py:85-91,
            # models/deeplabv3_plus.py:361): evaluate the loss on the low-resolution tensor with the interpolation inside the
            # kernel — same value and gradient, without reading the full-resolution logits or creating their gradient
            return ops.upsampled_cross_entropy(src[0], target, src[1], self.ignore_index, self.weight, self.reduction,
                                               _group(self.process_group))
        return ops.cross_entropy(output, target, self.ignore_index, self.weight, self.reduction, _group(self.process_group))


class DiceLoss(nn.Module):
    """Reference utils/losses.py:33-50: softmax -> one-hot -> whole-batch Dice with smooth=1.  Like the reference it
    rewrites ignored pixels of the caller's `target` to target.min() in place."""

    def __init__(self, smooth=1., ignore_index=255, process_group="auto"):
        super().__init__()
        self.ignore_index = ignore_index
        self.smooth = smooth
        self.process_group = process_group

    def forward(self, output, target):
        return ops.dice_loss(output, target, self.ignore_index, self.smooth, _group(self.process_group))


class FocalLoss(nn.Module):
    """Reference utils/losses.py:52-65: ce = CrossEntropy(reduce=False, weight=alpha) (0 where ignored);
    ((1 - exp(-ce))^gamma * ce).mean() over all pixels, or .sum() with size_average=False."""

    def __init__(self, gamma=2, alpha=None, ignore_index=255, size_average=True, process_group="auto"):
        super().__init__()
        self.register_buffer("alpha", None if alpha is None else alpha.detach().clone().float())
        self.gamma = gamma
        self.size_average = size_average
        self.ignore_index = ignore_index
        self.process_group = process_group

    def forward(self, output, target):
        return ops.focal_loss(output, target, self.ignore_index, self.gamma, self.alpha, self.size_average, _group(self.process_group))


class CE_DiceLoss(nn.Module):
    """Reference utils/losses.py:67-77: CrossEntropy(weight, reduction, ignore_index) + DiceLoss() — the Dice term is built with
    its DEFAULT ignore_index (255) whatever this module was given, and CE is evaluated first (on the not-yet-rewritten target).
    Unlike the reference, backward works when ignored pixels exist (there the in-place rewrite invalidates the
    tensor autograd saved for the CE term)."""

    def __init__(self, smooth=1, reduction="mean", ignore_index=255, weight=None, process_group="auto"):
        super().__init__()
        self.smooth = smooth
        self.dice = DiceLoss(process_group=process_group)
        self.cross_entropy = CrossEntropyLoss2d(weight=weight, ignore_index=ignore_index, reduction=reduction, process_group=process_group)

    def forward(self, output, target):
        ce = self.cross_entropy(output, target.clone())     # CE keeps its own copy of the un-rewritten target for backward
        return ce + self.dice(output, target)


class LovaszSoftmax(nn.Module):
    """Reference utils/losses.py:79-89: softmax + Lovasz-Softmax over the whole batch, classes present in the labels.
    (`classes` is stored in an attribute the reference never reads — utils/losses.py:82 — so 'present' is what runs.)
    Data parallel: the batch-level sort is not shard-decomposable, so each rank evaluates its own shard.
    fuse_upsample (default off; SEGMI_LOVASZ_FUSE_UP=1 turns it on): evaluate the loss of the model's final F.interpolate from the
    LOW-resolution logits (ops.upsampled_lovasz_softmax: bit-identical value and gradient, the 1.26 GB logits gradient of cfg5
    never exists) — built and measured in round 6: the Lovasz passes are latency- and ALU-bound, not HBM-bound, at 150 classes,
    and interpolating on the fly made the step 0.1-0.7 ms SLOWER (profiles/r06_lovasz_fused_upsample.txt)."""

    def __init__(self, classes="present", per_image=False, ignore_index=255, fuse_upsample=None):
        super().__init__()
        if per_image:
            raise NotImplementedError("per_image=True is never passed through by the reference's LovaszSoftmax.forward")
        self.smooth = classes
        self.per_image = per_image
        self.ignore_index = ignore_index
        self.fuse_upsample = os.environ.get("SEGMI_LOVASZ_FUSE_UP", "0") == "1" if fuse_upsample is None else bool(fuse_upsample)

    def forward(self, output, target):
        src = ops.upsample_source(output) if self.fuse_upsample else None
        if src is not None and tuple(output.shape[2:]) == tuple(target.shape[1:]):
            return ops.upsampled_lovasz_softmax(src[0], target, src[1], self.ignore_index)
        return ops.lovasz_softmax(output, target, self.ignore_index)
