"""Segmentation metrics of the training loop (reference: utils/metrics.py).

`eval_metrics(output, target, num_class)` keeps the reference's name, arguments and return value
`[correct, labeled, inter[C], union[C]]` (numpy), but the work is one fused libsegmi kernel (argmax + counting) instead of
torch.max + three torch.histc passes.  `SegMetrics` is the accumulate-on-device form the re-hosted trainer uses: it only
synchronises with the host when numbers are read, not twice per iteration like trainer.py:72,84-86.
"""
import numpy as np
import torch

from segmi import ops


class AverageMeter(object):
    """Running weighted average (reference utils/metrics.py:6-40)."""

    def __init__(self):
        self.initialized = False
        self.val = self.avg = self.sum = self.count = None

    def update(self, val, weight=1):
        if not self.initialized:
            self.val, self.avg, self.sum, self.count, self.initialized = val, val, val * weight, weight, True
        else:
            self.val = val
            self.sum = self.sum + val * weight
            self.count = self.count + weight
            self.avg = self.sum / self.count

    @property
    def value(self):
        return self.val

    @property
    def average(self):
        return np.round(self.avg, 5)


class SegMetrics:
    """Device-resident accumulator {correct, labeled, inter[C], pred_area[C], label_area[C]} (int64)."""

    def __init__(self, num_classes, device):
        self.num_classes = num_classes
        self.acc = torch.zeros(2 + 3 * num_classes, dtype=torch.int64, device=device)

    def reset(self):
        self.acc.zero_()

    def update(self, output, target):
        ops.seg_metrics_accumulate(output, target, self.acc)

    def all_reduce(self, group=None):
        """Sum the counters over all ranks of a data-parallel job (exact: int64), so that every rank derives the SAME epoch
        metrics — and therefore takes the same monitor / early-stop / checkpoint decision — from the global batch, as the
        reference does on its gathered outputs (trainer.py:84-86).  A collective: call it on every rank.  Returns a summed COPY
        semantics-wise by reducing in place; call once per epoch, after the last update."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.acc, group=group)
        return self

    def counts(self):
        """(correct, labeled, inter[C], union[C]) as numpy — synchronises."""
        a = self.acc.cpu().numpy()
        C = self.num_classes
        inter, pred, lab = a[2:2 + C], a[2 + C:2 + 2 * C], a[2 + 2 * C:2 + 3 * C]
        return a[0], a[1], inter, pred + lab - inter

    def summary(self):
        """Same dictionary as Trainer._get_seg_metrics (trainer.py:181-193)."""
        correct, labeled, inter, union = self.counts()
        pix_acc = 1.0 * correct / (np.spacing(1) + labeled)
        iou = 1.0 * inter / (np.spacing(1) + union)
        return {"Pixel_Accuracy": np.round(pix_acc, 3), "Mean_IoU": np.round(iou.mean(), 3),
                "Class_IoU": dict(zip(range(self.num_classes), np.round(iou, 3)))}


def eval_metrics(output, target, num_class):
    """Drop-in for utils/metrics.py:59-67: [correct, labeled, inter, union] for one batch."""
    if output.size(1) != num_class:
        raise ValueError("eval_metrics: output has %d channels, num_class is %d" % (output.size(1), num_class))
    m = SegMetrics(num_class, output.device)
    m.update(output, target)
    correct, labeled, inter, union = m.counts()
    return [np.round(np.float32(correct), 5), np.round(np.float32(labeled), 5), np.round(inter.astype(np.float32), 5),
            np.round(union.astype(np.float32), 5)]
