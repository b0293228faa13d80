#!/usr/bin/env python
"""BUILD CONTAINER ONLY (reads /root/reference): the REAL reference trainer timed beside bench.py's `cpu_baseline` port on the same
host (VERDICT r5 missing #5: "no build-container timing of the real train.main at cfg2 is committed beside it to show the two agree").

    python tools/cpu_reference_vs_port.py ref  [iters]     the reference's own Trainer (trainer.py:37-107 through base/base_trainer.py)
                                                           on BASELINE cfg2: PSPNet-R50, 8 x 3x512x512, 21 classes, CE + 0.4*aux,
                                                           SGD(0.01, 0.9, 1e-4) with differential lr, synthetic loader; per-iteration
                                                           wall time from timestamps taken inside the loss call
    python tools/cpu_reference_vs_port.py port             bench.cpu_baseline("cfg2") = what bench.py reports on the GPU box

Two processes (the reference and the drop-in share top-level package names); run one after the other and keep both outputs
(profiles/r06_cpu_reference_trainer_vs_port.txt).
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def run_reference(iters):
    from oracle import reference_harness
    models, losses = reference_harness.load()
    import importlib.util
    from trainer import Trainer              # the REFERENCE's (sys.path: /root/reference first)
    from utils import Logger
    spec = importlib.util.spec_from_file_location("segmi_synth", os.path.join(ROOT, "pytorch-segmentation_amd", "dataloaders", "synth.py"))
    sys.path.insert(1, os.path.join(ROOT, "pytorch-segmentation_amd"))        # synth.py imports base.BaseDataLoader of the drop-in ...
    import types
    base_stub = types.ModuleType("base_stub")
    synth_src = open(spec.origin).read().replace("from base import BaseDataLoader, BaseDataSet", "BaseDataLoader = BaseDataSet = object")
    synth = types.ModuleType("segmi_synth")
    exec(compile(synth_src, spec.origin, "exec"), synth.__dict__)             # ... which the Synth class itself does not need
    tmp = tempfile.mkdtemp()
    config = {"name": "PSPNet-cfg2-cpu", "n_gpu": 0, "use_synch_bn": False,
              "arch": {"type": "PSPNet", "args": {"backbone": "resnet50", "freeze_bn": False, "freeze_backbone": False}},
              "optimizer": {"type": "SGD", "differential_lr": True, "args": {"lr": 0.01, "weight_decay": 1e-4, "momentum": 0.9}},
              "loss": "CrossEntropyLoss2d", "ignore_index": 255, "lr_scheduler": {"type": "Poly", "args": {}},
              "trainer": {"epochs": 1, "save_dir": tmp, "save_period": 100, "monitor": "off", "early_stop": 10, "tensorboard": False,
                          "log_dir": tmp, "log_per_iter": 20, "val": False, "val_per_epochs": 5}}
    loader = synth.Synth(num_classes=21, batch_size=8, height=512, width=512, iters=iters)
    torch.manual_seed(0)
    model = models.PSPNet(21, pretrained=False, **config["arch"]["args"])
    stamps = []

    class Stamped(losses.CrossEntropyLoss2d):
        def forward(self, output, target):
            stamps.append(time.perf_counter())
            return super().forward(output, target)

    t0 = time.perf_counter()
    trainer = Trainer(model=model, loss=Stamped(ignore_index=255), resume=None, config=config, train_loader=loader, val_loader=None,
                      train_logger=Logger())
    trainer.train()
    t1 = time.perf_counter()
    main = stamps[0::2]                       # PSPNet: two loss calls per iteration (main + aux, trainer.py:57-62)
    steps = [b - a for a, b in zip(main, main[1:])]
    print(json.dumps({"what": "REAL reference Trainer (trainer.py:49-72) on torch CPU, cfg2 batch 8 x 3x512x512, %d iterations" % iters,
                      "threads": torch.get_num_threads(), "cores": os.cpu_count(),
                      "seconds_between_consecutive_loss_calls": [round(s, 2) for s in steps],
                      "img_per_s_best": round(8 / min(steps), 4), "img_per_s_median": round(8 / sorted(steps)[len(steps) // 2], 4),
                      "first_iteration_to_loss_s": round(main[0] - t0, 2), "whole_train_call_s": round(t1 - t0, 2)}))


def run_port():
    sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd"))
    import bench
    r = bench.cpu_baseline("cfg2", seconds_cap=240.0)
    r["what"] = "bench.py cpu_baseline (oracle port: oracle/pspnet_ref.py + losses_ref.py + torch.optim.SGD), same workload, this container"
    r["cores_available"] = os.cpu_count()
    print(json.dumps(r))


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    if sys.argv[1] == "ref":
        run_reference(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    else:
        run_port()
