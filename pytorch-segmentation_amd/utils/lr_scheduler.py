"""Per-iteration learning-rate schedules (reference: utils/lr_scheduler.py).  Host-side arithmetic, stepped once per
iteration with `step(epoch=epoch-1)` exactly as trainer.py:52 does."""
import math

from torch.optim.lr_scheduler import _LRScheduler


class Poly(_LRScheduler):
    """lr = base_lr * (1 - T/N)^0.9 with T the global iteration, optional linear warm-up (reference :4-21)."""

    def __init__(self, optimizer, num_epochs, iters_per_epoch=0, warmup_epochs=0, last_epoch=-1):
        self.iters_per_epoch = iters_per_epoch
        self.cur_iter = 0
        self.N = num_epochs * iters_per_epoch
        self.warmup_iters = warmup_epochs * iters_per_epoch
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        T = self.last_epoch * self.iters_per_epoch + self.cur_iter
        factor = pow((1 - 1.0 * T / self.N), 0.9)
        if self.warmup_iters > 0 and T < self.warmup_iters:
            factor = 1.0 * T / self.warmup_iters
        self.cur_iter %= self.iters_per_epoch
        self.cur_iter += 1
        return [base_lr * factor for base_lr in self.base_lrs]


class OneCycle(_LRScheduler):
    """Cosine one-cycle schedule with inverse momentum cycling (reference :24-60)."""

    def __init__(self, optimizer, num_epochs, iters_per_epoch=0, last_epoch=-1, momentums=(0.85, 0.95), div_factor=25, phase1=0.3):
        self.iters_per_epoch = iters_per_epoch
        self.cur_iter = 0
        self.N = num_epochs * iters_per_epoch
        self.phase1_iters = int(self.N * phase1)
        self.phase2_iters = self.N - self.phase1_iters
        self.momentums = momentums
        self.mom_diff = momentums[1] - momentums[0]
        self.low_lrs = [g["lr"] / div_factor for g in optimizer.param_groups]
        self.final_lrs = [g["lr"] / (div_factor * 1e4) for g in optimizer.param_groups]
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        T = self.last_epoch * self.iters_per_epoch + self.cur_iter
        self.cur_iter %= self.iters_per_epoch
        self.cur_iter += 1
        if T <= self.phase1_iters:
            cos = (1 + math.cos(math.pi * T / self.phase1_iters)) / 2
            for g in self.optimizer.param_groups:
                g["momentum"] = self.momentums[0] + self.mom_diff * cos
            return [b - (b - lo) * cos for b, lo in zip(self.base_lrs, self.low_lrs)]
        T -= self.phase1_iters
        cos = (1 + math.cos(math.pi * T / self.phase2_iters)) / 2
        for g in self.optimizer.param_groups:
            g["momentum"] = self.momentums[1] - self.mom_diff * cos
        return [f + (b - f) * cos for b, f in zip(self.base_lrs, self.final_lrs)]
