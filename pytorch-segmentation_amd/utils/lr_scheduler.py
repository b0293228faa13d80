"""Per-iteration learning-rate schedules resolved by name from config.json (`lr_scheduler.type`), stepped by the trainer
once per iteration as `scheduler.step(epoch=epoch-1)` (reference call site trainer.py:52; schedules utils/lr_scheduler.py).

Both schedules are functions of the global iteration  T = epoch * iters_per_epoch + i  (i = position inside the epoch,
tracked by the scheduler itself because the trainer only passes the epoch):

  Poly      lr_g(T) = base_g * (1 - T / N)^0.9, with an optional linear warm-up  lr_g = base_g * T / W  for T < W
  OneCycle  cosine ramp from base/div_factor up to base during the first `phase1` fraction of training, then a cosine decay
            to base/(div_factor*1e4); momentum runs the opposite way between momentums[0] and momentums[1]

Host-side scalar arithmetic only; the values reach the device as kernel arguments of the fused SGD step.
"""
import math

from torch.optim.lr_scheduler import _LRScheduler


class _IterationSchedule(_LRScheduler):
    """Keeps the within-epoch position and hands the global iteration to `rates(T)`."""

    def __init__(self, optimizer, num_epochs, iters_per_epoch, last_epoch=-1):
        self.iters_per_epoch = iters_per_epoch
        self.total_iters = num_epochs * iters_per_epoch
        self.pos = 0
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        t_global = self.last_epoch * self.iters_per_epoch + self.pos
        self.pos = self.pos % self.iters_per_epoch + 1      # wraps to 1 at the start of the next epoch
        return self.rates(t_global)

    def rates(self, t_global):
        raise NotImplementedError


class Poly(_IterationSchedule):
    def __init__(self, optimizer, num_epochs, iters_per_epoch=0, warmup_epochs=0, last_epoch=-1):
        self.warmup_iters = warmup_epochs * iters_per_epoch
        super().__init__(optimizer, num_epochs, iters_per_epoch, last_epoch)

    def rates(self, t_global):
        if 0 < self.warmup_iters and t_global < self.warmup_iters:
            scale = t_global / float(self.warmup_iters)
        else:
            scale = math.pow(1.0 - t_global / float(self.total_iters), 0.9)
        return [base * scale for base in self.base_lrs]


def _half_cosine(x):
    """1 at x = 0, 0 at x = 1."""
    return 0.5 * (1.0 + math.cos(math.pi * x))


class OneCycle(_IterationSchedule):
    def __init__(self, optimizer, num_epochs, iters_per_epoch=0, last_epoch=-1, momentums=(0.85, 0.95), div_factor=25, phase1=0.3):
        total = num_epochs * iters_per_epoch
        self.up_iters = int(total * phase1)
        self.down_iters = total - self.up_iters
        self.momentums = momentums
        self.start_lrs = [g["lr"] / div_factor for g in optimizer.param_groups]
        self.end_lrs = [g["lr"] / (div_factor * 1e4) for g in optimizer.param_groups]
        super().__init__(optimizer, num_epochs, iters_per_epoch, last_epoch)

    def _set_momentum(self, value):
        for group in self.optimizer.param_groups:
            group["momentum"] = value

    def rates(self, t_global):
        m_lo, m_hi = self.momentums
        if t_global <= self.up_iters:                     # warm-up: lr start -> base, momentum hi -> lo
            c = _half_cosine(t_global / float(self.up_iters))
            self._set_momentum(m_lo + (m_hi - m_lo) * c)
            return [base - (base - start) * c for base, start in zip(self.base_lrs, self.start_lrs)]
        c = _half_cosine((t_global - self.up_iters) / float(self.down_iters))   # anneal: lr base -> end, momentum lo -> hi
        self._set_momentum(m_hi - (m_hi - m_lo) * c)
        return [end + (base - end) * c for base, end in zip(self.base_lrs, self.end_lrs)]
