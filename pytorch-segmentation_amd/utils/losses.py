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
