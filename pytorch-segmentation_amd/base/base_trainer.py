"""BaseTrainer — the training runtime around the hot loop (reference: base/base_trainer.py:18-182), re-hosted for one
process per GPU.

Kept from the reference: constructor signature, config keys (`n_gpu`, `use_synch_bn`, `optimizer{type,args,differential_lr}`,
`lr_scheduler`, `trainer{epochs,save_dir,save_period,monitor,early_stop,val,val_per_epochs,log_dir}`), differential learning
rates (decoder lr, backbone lr/10, :46-56), checkpoint dictionary layout with `module.`-prefixed keys (:139-151), resume
(:157-173), monitor / early stopping (:110-132).
Changed by design: `nn.DataParallel` (single process, gather/scatter through GPU 0, :33-38) becomes
`segmi.distributed.DistributedModel` — this process drives ONE GPU (LOCAL_RANK) and gradients are averaged with a bucketed
RCCL all-reduce overlapped with backward; with `use_synch_bn` the BN layers exchange statistics over RCCL.  Only rank 0
writes checkpoints and logs.  TensorBoard is optional (the package is not a dependency of the hot path).
"""
import datetime
import json
import logging
import math
import os

import torch
import torch.distributed as dist

import utils.lr_scheduler
from segmi import optim as segmi_optim
from segmi.distributed import DistributedModel
from utils import helpers
from utils.sync_batchnorm import convert_model


def get_instance(module, name, config, *args):
    return getattr(module, config[name]["type"])(*args, **config[name]["args"])


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass


def _make_writer(log_dir):
    try:
        from torch.utils import tensorboard
        return tensorboard.SummaryWriter(log_dir)
    except Exception:   # tensorboard not installed: logging to it is not part of the hot path
        return _NullWriter()


class BaseTrainer:
    def __init__(self, model, loss, resume, config, train_loader, val_loader=None, train_logger=None):
        self.loss = loss
        self.config = config
        self.train_loader = train_loader
        self.val_loader = val_loader
        self.train_logger = train_logger
        self.logger = logging.getLogger(self.__class__.__name__)
        self.do_validation = self.config["trainer"]["val"]
        self.start_epoch = 1
        self.improved = False
        self.not_improved_count = 0

        # DEVICE: one process per GPU
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.device = self._get_device(self.config["n_gpu"])
        model = model.to(self.device)
        if config["use_synch_bn"]:
            model = convert_model(model)
        self.model = DistributedModel(model)          # `.module`, `module.`-prefixed state_dict like nn.DataParallel
        self.loss.to(self.device)

        cfg_trainer = self.config["trainer"]
        self.epochs = cfg_trainer["epochs"]
        self.save_period = cfg_trainer["save_period"]

        # OPTIMIZER (differential learning rates: decoder lr, backbone lr / 10)
        if self.config["optimizer"]["differential_lr"]:
            trainable_params = [
                {"params": [p for p in self.model.module.get_decoder_params() if p.requires_grad]},
                {"params": [p for p in self.model.module.get_backbone_params() if p.requires_grad],
                 "lr": config["optimizer"]["args"]["lr"] / 10}]
            trainable_params = [g for g in trainable_params if g["params"]]
        else:
            trainable_params = [p for p in self.model.parameters() if p.requires_grad]
        # same by-name resolution as the reference (`getattr(torch.optim, type)`); optimizers that exist as fused libsegmi
        # kernels (SGD) are taken from segmi.optim — identical update rule and state layout
        otype = config["optimizer"]["type"]
        oargs = config["optimizer"]["args"]
        fused = hasattr(segmi_optim, otype) and not oargs.get("nesterov") and not oargs.get("dampening")
        self.optimizer = get_instance(segmi_optim if fused else torch.optim, "optimizer", config, trainable_params)
        # data parallel + fused SGD: the update of a gradient bucket is launched right after ITS all-reduce (segmi.distributed)
        self.bucket_step = bool(self.model.reducer.collective and fused and self.model.attach_optimizer(self.optimizer))
        self.lr_scheduler = getattr(utils.lr_scheduler, config["lr_scheduler"]["type"])(self.optimizer, self.epochs, len(train_loader))

        # MONITORING
        self.monitor = cfg_trainer.get("monitor", "off")
        if self.monitor == "off":
            self.mnt_mode, self.mnt_best = "off", 0
        else:
            self.mnt_mode, self.mnt_metric = self.monitor.split()
            assert self.mnt_mode in ["min", "max"]
            self.mnt_best = -math.inf if self.mnt_mode == "max" else math.inf
            self.early_stoping = cfg_trainer.get("early_stop", math.inf)

        # CHECKPOINTS & LOG WRITER (rank 0)
        start_time = datetime.datetime.now().strftime("%m-%d_%H-%M")
        self.checkpoint_dir = os.path.join(cfg_trainer["save_dir"], self.config["name"], start_time)
        self.writer = _NullWriter()
        if self.rank == 0:
            helpers.dir_exists(self.checkpoint_dir)
            with open(os.path.join(self.checkpoint_dir, "config.json"), "w") as handle:
                json.dump(self.config, handle, indent=4, sort_keys=True)
            self.writer = _make_writer(os.path.join(cfg_trainer["log_dir"], self.config["name"], start_time))
        if resume:
            self._resume_checkpoint(resume)

    def _get_device(self, n_gpu):
        if not torch.cuda.is_available():
            raise RuntimeError("the segmi training path needs an MI355X: there is no CPU fallback")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if n_gpu and self.world > n_gpu:
            self.logger.warning("config n_gpu=%d but %d ranks were launched; every rank drives one GPU" % (n_gpu, self.world))
        torch.cuda.set_device(local)
        return torch.device("cuda", local)

    def train(self):
        for epoch in range(self.start_epoch, self.epochs + 1):
            results = self._train_epoch(epoch)
            if self.do_validation and epoch % self.config["trainer"]["val_per_epochs"] == 0:
                results = self._valid_epoch(epoch)
                self.logger.info("\n         ## Info for epoch %d ## " % epoch)
                for k, v in results.items():
                    self.logger.info("         %-15s: %s" % (str(k), v))
            log = {"epoch": epoch, **results}
            if self.train_logger is not None:
                self.train_logger.add_entry(log)

            if self.mnt_mode != "off" and epoch % self.config["trainer"]["val_per_epochs"] == 0:
                try:
                    if self.mnt_mode == "min":
                        self.improved = log[self.mnt_metric] < self.mnt_best
                    else:
                        self.improved = log[self.mnt_metric] > self.mnt_best
                except KeyError:
                    self.logger.warning("The metrics being tracked (%s) has not been calculated. Training stops." % self.mnt_metric)
                    break
                if self.improved:
                    self.mnt_best = log[self.mnt_metric]
                    self.not_improved_count = 0
                else:
                    self.not_improved_count += 1
                if self.not_improved_count > self.early_stoping:
                    self.logger.info("\nPerformance didn't improve for %s epochs" % self.early_stoping)
                    self.logger.warning("Training Stoped")
                    break

            if epoch % self.save_period == 0:
                self._save_checkpoint(epoch, save_best=self.improved)

    def _save_checkpoint(self, epoch, save_best=False):
        if self.rank != 0:
            return
        state = {"arch": type(self.model).__name__, "epoch": epoch,
                 "state_dict": {k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()},
                 "optimizer": self.optimizer.state_dict(), "monitor_best": self.mnt_best, "config": self.config}
        filename = os.path.join(self.checkpoint_dir, "checkpoint-epoch%d.pth" % epoch)
        self.logger.info("\nSaving a checkpoint: %s ..." % filename)
        torch.save(state, filename)
        if save_best:
            torch.save(state, os.path.join(self.checkpoint_dir, "best_model.pth"))
            self.logger.info("Saving current best: best_model.pth")

    def _resume_checkpoint(self, resume_path):
        self.logger.info("Loading checkpoint : %s" % resume_path)
        checkpoint = torch.load(resume_path, map_location="cpu", weights_only=False)
        self.start_epoch = checkpoint["epoch"] + 1
        self.mnt_best = checkpoint["monitor_best"]
        self.not_improved_count = 0
        if checkpoint["config"]["arch"] != self.config["arch"]:
            self.logger.warning("Warning! Current model is not the same as the one in the checkpoint")
        self.model.load_state_dict(checkpoint["state_dict"])
        if checkpoint["config"]["optimizer"]["type"] != self.config["optimizer"]["type"]:
            self.logger.warning("Warning! Current optimizer is not the same as the one in the checkpoint")
        self.optimizer.load_state_dict(checkpoint["optimizer"])
        self.logger.info("Checkpoint <%s> (epoch %d) was loaded" % (resume_path, self.start_epoch))

    def _train_epoch(self, epoch):
        raise NotImplementedError

    def _valid_epoch(self, epoch):
        raise NotImplementedError
