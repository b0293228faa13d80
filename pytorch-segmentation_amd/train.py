"""`python train.py -c config.json [-r checkpoint.pth]` — the reference's entry point (train.py:18-60) on the segmi path.

Single GPU: run it directly.  Multi GPU: one process per GPU,
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py -c config.json
(the reference's `-d/--device` GPU list becomes the launcher's job).  `config.json` is the reference's schema; everything is
resolved by name exactly as train.py:14-16,26,30 does.
"""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dataloaders  # noqa: E402
import models  # noqa: E402
from trainer import Trainer  # noqa: E402
from utils import losses  # noqa: E402


def get_instance(module, name, config, *args):
    return getattr(module, config[name]["type"])(*args, **config[name]["args"])


def main(config, resume):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")          # RCCL over xGMI
    train_loader = get_instance(dataloaders, "train_loader", config)
    val_loader = get_instance(dataloaders, "val_loader", config) if config.get("val_loader") else None
    model = get_instance(models, "arch", config, train_loader.dataset.num_classes)
    loss = getattr(losses, config["loss"])(ignore_index=config["ignore_index"])
    trainer = Trainer(model=model, loss=loss, resume=resume, config=config, train_loader=train_loader, val_loader=val_loader)
    trainer.train()
    return trainer


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="segmi training")
    parser.add_argument("-c", "--config", default="config.json", type=str, help="Path to the config file (default: config.json)")
    parser.add_argument("-r", "--resume", default=None, type=str, help="Path to the .pth model checkpoint to resume training")
    args = parser.parse_args()
    cfg = json.load(open(args.config))
    if args.resume:
        cfg = torch.load(args.resume, map_location="cpu", weights_only=False)["config"]
    main(cfg, args.resume)
