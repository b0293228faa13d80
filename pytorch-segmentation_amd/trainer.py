"""Trainer — the hot loop (reference: trainer.py:12-193), re-hosted on the segmi path.

Per iteration, as trainer.py:49-72: lr_scheduler.step(epoch-1) -> zero_grad -> model(data) -> loss (+0.4*aux for PSP*, keyed on
the arch name like the reference) -> backward -> [gradient all-reduce finishes] -> optimizer.step.
What changed: the reference synchronises with the host twice per iteration (`loss.item()` :72, `eval_metrics` -> numpy :84-86);
here the loss sum and the accuracy / IoU counters stay on the device (segmi_seg_metrics) and are read every `log_step`
iterations and at the end of the epoch.  `iteration_losses` keeps the per-iteration loss tensors of the last epoch for tests.
"""
import contextlib
import time

import numpy as np
import torch

from base import BaseTrainer, DataPrefetcher
from utils.metrics import AverageMeter, SegMetrics


class Trainer(BaseTrainer):
    def __init__(self, model, loss, resume, config, train_loader, val_loader=None, train_logger=None, prefetch=True):
        super().__init__(model, loss, resume, config, train_loader, val_loader, train_logger)
        self.wrt_mode, self.wrt_step = "train_", 0
        self.log_step = config["trainer"].get("log_per_iter", int(np.sqrt(self.train_loader.batch_size)))
        if config["trainer"].get("log_per_iter"):
            self.log_step = int(self.log_step / self.train_loader.batch_size) + 1
        self.log_step = max(1, self.log_step)
        self.num_classes = self.train_loader.dataset.num_classes
        self.metrics = SegMetrics(self.num_classes, self.device)
        self.psp = self.config["arch"]["type"][:3] == "PSP"
        self.iteration_losses = []
        # side-stream H2D staging of the next batch (reference trainer.py:30-33); loaders that already yield device tensors pass through
        if prefetch and self.device.type == "cuda":
            self.train_loader = DataPrefetcher(self.train_loader, device=self.device)
            if self.val_loader is not None:
                self.val_loader = DataPrefetcher(self.val_loader, device=self.device)

    def _forward_loss(self, data, target):
        output = self.model(data)
        if self.psp and isinstance(output, tuple):
            assert output[0].size()[2:] == target.size()[1:]
            assert output[0].size()[1] == self.num_classes
            loss = self.loss(output[0], target) + self.loss(output[1], target) * 0.4
            output = output[0]
        else:
            assert output.size()[2:] == target.size()[1:]
            assert output.size()[1] == self.num_classes
            loss = self.loss(output, target)
        return output, loss

    def _train_epoch(self, epoch):
        self.model.train()
        if self.config["arch"]["args"].get("freeze_bn"):
            self.model.module.freeze_bn()
        self.wrt_mode = "train"
        self._reset_metrics()
        self.iteration_losses = []
        loss_sum = torch.zeros((), device=self.device)
        tic = time.time()
        n_iter = 0
        for batch_idx, (data, target) in enumerate(self.train_loader):
            self.data_time.update(time.time() - tic)
            data, target = data.to(self.device, non_blocking=True), target.to(self.device, non_blocking=True)
            self.lr_scheduler.step(epoch=epoch - 1)

            self.model.zero_grad()
            output, loss = self._forward_loss(data, target)
            loss.backward()
            if self.bucket_step:
                self.model.finish_gradients(self.optimizer)       # all-reduce wait + fused SGD, bucket by bucket
            else:
                self.model.finish_gradients()
                self.optimizer.step()

            loss_d = loss.detach()
            loss_sum += loss_d
            self.iteration_losses.append(loss_d)
            self.metrics.update(output, target)           # device-side accumulation, no sync
            n_iter += 1
            self.batch_time.update(time.time() - tic)
            tic = time.time()

            if batch_idx % self.log_step == 0 and self.rank == 0:
                self.wrt_step = (epoch - 1) * len(self.train_loader) + batch_idx
                s = self.metrics.summary()                # the only host synchronisation of the loop
                avg = float(loss_sum) / n_iter
                self.writer.add_scalar("%s/loss" % self.wrt_mode, float(loss_d), self.wrt_step)
                self.logger.info("TRAIN (%d) [%d/%d] | Loss: %.3f | Acc %.2f mIoU %.2f | B %.2f D %.2f |" % (
                    epoch, batch_idx, len(self.train_loader), avg, s["Pixel_Accuracy"], s["Mean_IoU"],
                    self.batch_time.average, self.data_time.average))

        self.total_loss.update(self._epoch_mean(loss_sum, n_iter), n_iter)
        seg_metrics = self.metrics.all_reduce().summary()     # global-batch counters: identical results on every rank
        for k, v in list(seg_metrics.items())[:-1]:
            self.writer.add_scalar("%s/%s" % (self.wrt_mode, k), v, self.wrt_step)
        for i, g in enumerate(self.optimizer.param_groups):
            self.writer.add_scalar("%s/Learning_rate_%d" % (self.wrt_mode, i), g["lr"], self.wrt_step)
        return {"loss": self.total_loss.average, **seg_metrics}

    def _valid_epoch(self, epoch):
        if self.val_loader is None:
            self.logger.warning("Not data loader was passed for the validation step, No validation is performed !")
            return {}
        self.logger.info("\n###### EVALUATION ######")
        # one set of BatchNorm running statistics for the whole validation set — rank 0's, what the reference's single-process
        # DataParallel validates (and checkpoints) with; without SyncBN the ranks' buffers follow their own shards
        if self.world > 1 and hasattr(self.model, "broadcast_buffers"):
            self.model.broadcast_buffers(src=0)
        self.model.eval()
        self.wrt_mode = "val"
        self._reset_metrics()
        loss_sum = torch.zeros((), device=self.device)
        n_iter = 0
        # validation batches are sharded over the ranks WITHOUT padding (every sample exactly once, base/base_dataloader.py), so
        # ranks may run different numbers of iterations: nothing inside the loop may be a collective.  The losses' global-batch
        # weighting IS one (an all-reduce of the valid-pixel count in forward), so it is switched off here — every rank takes
        # the reference's per-batch mean of ITS batches and `_epoch_mean` all-reduces (sum, count) once, after the loop: the
        # mean over all batches of the epoch, what the single-process reference computes (trainer.py:134-141)
        with torch.no_grad(), self._per_rank_loss():
            for data, target in self.val_loader:
                data, target = data.to(self.device, non_blocking=True), target.to(self.device, non_blocking=True)
                output = self.model(data)
                loss_sum += self.loss(output, target)
                self.metrics.update(output, target)
                n_iter += 1
        self.total_loss.update(self._epoch_mean(loss_sum, n_iter), n_iter)
        self.wrt_step = epoch * len(self.val_loader)
        self.writer.add_scalar("%s/loss" % self.wrt_mode, self.total_loss.average, self.wrt_step)
        seg_metrics = self.metrics.all_reduce().summary()
        for k, v in list(seg_metrics.items())[:-1]:
            self.writer.add_scalar("%s/%s" % (self.wrt_mode, k), v, self.wrt_step)
        return {"val_loss": self.total_loss.average, **seg_metrics}

    @contextlib.contextmanager
    def _per_rank_loss(self):
        """Inside: every loss module evaluates strictly per rank (process_group=None) — no collective in its forward."""
        saved = [(m, m.process_group) for m in self.loss.modules() if hasattr(m, "process_group")]
        for m, _ in saved:
            m.process_group = None
        try:
            yield
        finally:
            for m, pg in saved:
                m.process_group = pg

    def _epoch_mean(self, loss_sum, n_iter):
        """Mean per-iteration loss of the epoch over ALL ranks (every rank must take the same monitor / early-stop decision
        in BaseTrainer.train, otherwise one of them leaves the loop and the others hang in the next collective).  The per-rank
        losses already carry the global-batch weighting (utils/losses.py), so their average is the global-batch loss."""
        t = torch.stack([loss_sum.detach().double().reshape(()), torch.tensor(float(n_iter), dtype=torch.float64, device=loss_sum.device)])
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t)
        return float(t[0]) / max(float(t[1]), 1.0)

    def _reset_metrics(self):
        self.batch_time = AverageMeter()
        self.data_time = AverageMeter()
        self.total_loss = AverageMeter()
        self.metrics.reset()
