"""Per-pixel losses of the plugin surface (reference: utils/losses.py), on libsegmi kernels.

`getattr(losses, config['loss'])(ignore_index=...)` (train.py:30) resolves these by name;
forward(logits [N,C,H,W] fp32, target [N,H,W] int64) -> scalar.
"""
import torch.nn as nn

from segmi import ops


class CrossEntropyLoss2d(nn.Module):
    """nn.CrossEntropyLoss(ignore_index, reduction='mean') (reference utils/losses.py:24-31) as one
    fused log-softmax + NLL pass; backward recomputes softmax from the saved log-sum-exp."""

    def __init__(self, weight=None, ignore_index=255, reduction="mean"):
        super().__init__()
        if weight is not None:
            raise NotImplementedError("class weights are not supported by the fused CE kernel yet")
        if reduction != "mean":
            raise NotImplementedError("only reduction='mean' is on the hot path (reference default)")
        self.ignore_index = ignore_index

    def forward(self, output, target):
        return ops.cross_entropy(output, target, self.ignore_index)


class DiceLoss(nn.Module):
    """Reference utils/losses.py:33-50: softmax -> one-hot -> whole-batch Dice with smooth=1.  Like the reference it
    rewrites ignored pixels of the caller's `target` to target.min() in place."""

    def __init__(self, smooth=1., ignore_index=255):
        super().__init__()
        self.ignore_index = ignore_index
        self.smooth = smooth

    def forward(self, output, target):
        return ops.dice_loss(output, target, self.ignore_index, self.smooth)


class FocalLoss(nn.Module):
    """Reference utils/losses.py:52-65: ((1 - exp(-ce))^gamma * ce).mean() over all pixels, ce = 0 where ignored."""

    def __init__(self, gamma=2, alpha=None, ignore_index=255, size_average=True):
        super().__init__()
        if alpha is not None:
            raise NotImplementedError("class weights (alpha) are not supported by the fused focal kernel")
        self.gamma = gamma
        self.size_average = size_average
        self.ignore_index = ignore_index

    def forward(self, output, target):
        loss = ops.focal_loss(output, target, self.ignore_index, self.gamma)
        if self.size_average:
            return loss
        return loss * float(target.numel())      # reference: loss.sum()


class CE_DiceLoss(nn.Module):
    """Reference utils/losses.py:67-77: CrossEntropy(ignore_index) + DiceLoss() — the Dice term is built with its DEFAULT
    ignore_index (255) whatever this module was given, and CE is evaluated first (on the not-yet-rewritten target).
    Unlike the reference, backward works when ignored pixels exist (there the in-place rewrite invalidates the
    tensor autograd saved for the CE term)."""

    def __init__(self, smooth=1, reduction="mean", ignore_index=255, weight=None):
        super().__init__()
        self.smooth = smooth
        self.dice = DiceLoss()
        self.cross_entropy = CrossEntropyLoss2d(weight=weight, ignore_index=ignore_index, reduction=reduction)

    def forward(self, output, target):
        ce = self.cross_entropy(output, target.clone())     # CE keeps its own copy of the un-rewritten target for backward
        return ce + self.dice(output, target)


class LovaszSoftmax(nn.Module):
    """Reference utils/losses.py:79-89: softmax + Lovasz-Softmax over the whole batch, classes present in the labels.
    (`classes` is stored in an attribute the reference never reads — utils/losses.py:82 — so 'present' is what runs.)"""

    def __init__(self, classes="present", per_image=False, ignore_index=255):
        super().__init__()
        if per_image:
            raise NotImplementedError("per_image=True is never passed through by the reference's LovaszSoftmax.forward")
        self.smooth = classes
        self.per_image = per_image
        self.ignore_index = ignore_index

    def forward(self, output, target):
        return ops.lovasz_softmax(output, target, self.ignore_index)
