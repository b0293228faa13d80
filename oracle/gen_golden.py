"""Generate tests/golden/*.pt by running the REAL reference (/root/reference) on torch CPU.

    python oracle/gen_golden.py            (build container only; takes ~1 min)

The fixtures pin the oracle restatements (tests/test_oracle_golden.py) and, through them, the HIP
path.  Weights are not stored: they are re-created from the stored manifest by
oracle.weights.synth_state_dict (same seed), inputs by oracle.weights.synth_batch.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import reference_harness  # noqa: E402
from oracle.weights import manifest_of, synth_batch, synth_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _grad_digest(named_params):
    out = {}
    for k, p in named_params:
        if p.grad is None:
            continue
        g = p.grad.detach().reshape(-1)
        step = max(1, g.numel() // 64)
        out[k] = {"norm": g.norm().item(), "absmax": g.abs().max().item(), "head": g[:8].clone(), "sample": g[::step][:64].clone()}
    return out


def gen_losses(losses):
    cases = []
    # known-answer example of SURVEY.md App. C
    logits = (2 * torch.sin(0.37 * torch.arange(24.0))).reshape(1, 3, 2, 4)
    target = torch.tensor([[[0, 1, 2, 255], [2, 2, 0, 1]]])
    cases.append(("appendix_c", logits, target, 255))
    g = torch.Generator().manual_seed(99)
    for name, (N, C, H, W), ign in [("c21", (2, 21, 12, 10), 255), ("c5_absent", (2, 5, 6, 7), 255), ("c150", (1, 150, 8, 8), 255),
                                    ("c2", (2, 2, 9, 9), 255), ("noignore", (2, 7, 5, 5), 255)]:
        lg = torch.randn(N, C, H, W, generator=g) * 2
        tg = torch.randint(0, C if name != "c5_absent" else C - 1, (N, H, W), generator=g)
        if name != "noignore":
            tg[:, 0, :] = ign
        cases.append((name, lg, tg, ign))
    out = {}
    for name, lg, tg, ign in cases:
        rec = {"logits": lg, "target": tg, "ignore_index": ign}
        for lname in ("CrossEntropyLoss2d", "DiceLoss", "FocalLoss", "LovaszSoftmax", "CE_DiceLoss"):
            x = lg.clone().requires_grad_(True)
            crit = getattr(losses, lname)(ignore_index=ign)
            tgt = tg.clone()
            val = crit(x, tgt)
            r = {"loss": val.detach().clone(), "target_after": tgt.clone()}   # DiceLoss rewrites ignored pixels in place
            try:
                val.backward()
                r["grad"] = x.grad.clone()
            except RuntimeError as e:
                # CE_DiceLoss: the in-place target rewrite invalidates the tensor saved for CE's backward (reference bug,
                # SURVEY.md App. E) -> the gradient of the same expression is taken with a protected copy of the target
                assert lname == "CE_DiceLoss", (lname, e)
                x = lg.clone().requires_grad_(True)
                val2 = crit.cross_entropy(x, tg.clone()) + crit.dice(x, tg.clone())
                val2.backward()
                r["grad"] = x.grad.clone()
                r["reference_backward_raises"] = True
            rec[lname] = r
        out[name] = rec
    torch.save(out, os.path.join(GOLD, "losses.pt"))
    print("losses.pt:", list(out))


def gen_pspnet(models, losses):
    """Two regimes: (train) BN batch statistics as in BASELINE cfg2 — forward/loss/running stats are
    well conditioned, parameter gradients are NOT (the reference's own fp32 vs fp64 runs differ by
    percents at this size), so only digests are kept for a coarse check;
    (frozen) the reference's freeze_bn=True configuration — gradients are well conditioned and pinned tightly."""
    torch.manual_seed(0)
    C, N, H, W = 5, 4, 104, 104
    crit = losses.CrossEntropyLoss2d(ignore_index=255)
    x, t = synth_batch(N, 3, H, W, C)
    rec = {"num_classes": C, "input_shape": (N, 3, H, W)}
    for regime in ("train", "frozen"):
        model = models.PSPNet(C, backbone="resnet50", pretrained=False)
        man = manifest_of(model.state_dict())
        model.load_state_dict(synth_state_dict(man, seed=0))
        model.train()
        if regime == "frozen":
            model.freeze_bn()
        for m in model.modules():  # parity runs neutralise dropout
            if isinstance(m, torch.nn.Dropout2d):
                m.eval()
        out, aux = model(x)
        loss = crit(out, t) + 0.4 * crit(aux, t)
        loss.backward()
        sd_after = model.state_dict()
        rec["manifest"] = man
        rec[regime] = {
            # full-resolution main head only for the train regime (argmax audit); the rest at pixel stride 2
            "out": out.detach().clone() if regime == "train" else out.detach()[:, :, ::2, ::2].clone(),
            "aux": aux.detach()[:, :, ::2, ::2].clone(), "loss": loss.detach().clone(),
            "grads": _grad_digest(model.named_parameters()),
            "running": {k: sd_after[k].clone() for k in ("initial.0.1.running_mean", "initial.0.1.running_var",
                                                         "layer4.2.bn3.running_mean", "layer4.2.bn3.running_var",
                                                         "master_branch.0.stages.0.2.running_var",
                                                         "master_branch.0.bottleneck.1.running_mean",
                                                         "initial.1.num_batches_tracked")},
        }
        print("pspnet_r50.pt[%s]: loss %.6f, logit max %.3f" % (regime, loss.item(), out.abs().max().item()))
    model.eval()
    with torch.no_grad():
        rec["eval_out"] = model(x)[:, :, ::2, ::2].clone()
    torch.save(rec, os.path.join(GOLD, "pspnet_r50.pt"))


def gen_unet(models, losses):
    """cfg1 regime (UNet, 2 classes, CE) at two sizes: 64x64 (clean halving) and 50x70 (ceil-mode pooling and the
    bilinear re-alignment of models/unet.py:43-47 both active).  Full outputs + gradient digests, BN batch stats."""
    crit = losses.CrossEntropyLoss2d(ignore_index=255)
    rec = {}
    for name, (N, H, W), C in (("s64", (2, 64, 64), 2), ("s50x70", (2, 50, 70), 3)):
        torch.manual_seed(0)
        model = models.UNet(C)
        man = manifest_of(model.state_dict())
        model.load_state_dict(synth_state_dict(man, seed=1))
        model.train()
        x, t = synth_batch(N, 3, H, W, C, seed=4321)
        out = model(x)
        loss = crit(out, t)
        loss.backward()
        sd_after = model.state_dict()
        rec[name] = {"manifest": man, "num_classes": C, "input_shape": (N, 3, H, W), "out": out.detach().clone(),
                     "loss": loss.detach().clone(), "grads": _grad_digest(model.named_parameters()),
                     "running": {k: sd_after[k].clone() for k in ("start_conv.1.running_mean", "middle_conv.4.running_var",
                                                                  "up4.up_conv.4.running_mean")}}
        model.eval()
        with torch.no_grad():
            rec[name]["eval_out"] = model(x).clone()
        print("unet.pt[%s]: loss %.6f" % (name, loss.item()))
    torch.save(rec, os.path.join(GOLD, "unet.pt"))


def gen_deeplab(models, losses):
    """DeepLabV3+ train steps run by the REAL reference class (its torchvision dependency replaced by oracle/tv_resnet.py):
    aligned Xception at output_stride 16 (cfg5 family; `block2.relu.inplace=False` is the numerically neutral workaround for
    the reference's in-place-ReLU autograd error under torch >= 1.5) and ResNet-50 at output_stride 8 and 16
    (cfg3 family: same code path as ResNet-101, 1/2 the layers)."""
    crit = losses.CrossEntropyLoss2d(ignore_index=255)
    rec = {}
    cases = (("xception_os16", dict(backbone="xception", output_stride=16), (2, 96, 96), 7),
             ("resnet50_os8", dict(backbone="resnet50", output_stride=8), (2, 72, 88), 5),
             ("resnet50_os16", dict(backbone="resnet50", output_stride=16), (2, 97, 97), 19))
    for name, kw, (N, H, W), C in cases:
        torch.manual_seed(0)
        model = models.DeepLab(C, pretrained=False, **kw)
        if kw["backbone"] == "xception":
            model.backbone.block2.relu.inplace = False
        man = manifest_of(model.state_dict())
        model.load_state_dict(synth_state_dict(man, seed=2))
        x, t = synth_batch(N, 3, H, W, C, seed=555)
        r = {"manifest": man, "num_classes": C, "input_shape": (N, 3, H, W), "kwargs": kw}
        for regime in ("train", "frozen"):
            model.zero_grad()
            model.train()
            if regime == "frozen":
                model.freeze_bn()
            for m in model.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.eval()
            out = model(x)
            loss = crit(out, t)
            loss.backward()
            r[regime] = {"out": out.detach().clone(), "loss": loss.detach().clone(), "grads": _grad_digest(model.named_parameters())}
            print("deeplab.pt[%s/%s]: loss %.6f, |logit| max %.3f" % (name, regime, loss.item(), out.abs().max().item()))
        model.eval()
        with torch.no_grad():
            r["eval_out"] = model(x).clone()
        rec[name] = r
    torch.save(rec, os.path.join(GOLD, "deeplab.pt"))


def gen_trainer(models, losses):
    """BASELINE configs[0]: the REAL reference Trainer (trainer.py + base/base_trainer.py) run on CPU through its config.json
    path for one epoch of 4 iterations of UNet / 2 classes / batch 2 / 256x256 on the synthetic loader
    (pytorch-segmentation_amd/dataloaders/synth.py, loaded by file path: batches are a pure function of (seed, index)).
    Records the per-iteration losses, the epoch's metrics and digests of the trained weights."""
    import importlib.util
    import json
    import tempfile
    sys.path.insert(0, reference_harness.REFERENCE)
    from trainer import Trainer
    from utils import Logger          # train.main passes one (train.py:19); without it base_trainer.py:118 hits an unbound `log`
    spec = importlib.util.spec_from_file_location("segmi_synth", os.path.join(ROOT, "pytorch-segmentation_amd", "dataloaders", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    config = json.load(open(os.path.join(ROOT, "pytorch-segmentation_amd", "config.json")))
    tmp = tempfile.mkdtemp()
    config["trainer"].update(epochs=1, val=False, save_dir=tmp, log_dir=tmp, save_period=100)
    loader = synth.Synth(**config["train_loader"]["args"])
    torch.manual_seed(0)
    model = models.UNet(loader.dataset.num_classes, **config["arch"]["args"])
    man = manifest_of(model.state_dict())
    model.load_state_dict(synth_state_dict(man, seed=11))
    seen = []

    class Recording(losses.CrossEntropyLoss2d):
        def forward(self, output, target):
            v = super().forward(output, target)
            seen.append(v.item())
            return v

    trainer = Trainer(model=model, loss=Recording(ignore_index=config["ignore_index"]), resume=None, config=config,
                      train_loader=loader, val_loader=None, train_logger=Logger())
    trainer.train()
    log = trainer._get_seg_metrics()
    sd = trainer.model.state_dict()
    rec = {"config": config, "manifest": man, "losses": seen, "pixel_accuracy": float(log["Pixel_Accuracy"]), "mean_iou": float(log["Mean_IoU"]),
           "total_correct": float(trainer.total_correct), "total_label": float(trainer.total_label),
           "total_inter": torch.as_tensor(trainer.total_inter), "total_union": torch.as_tensor(trainer.total_union),
           "lrs": [g["lr"] for g in trainer.optimizer.param_groups],
           "weights": {k: {"norm": v.float().norm().item(), "head": v.flatten()[:8].clone()} for k, v in sd.items()
                       if k.endswith(("start_conv.0.weight", "middle_conv.3.weight", "up4.up.weight", "final_conv.bias", "down2.down_conv.1.running_var"))}}
    torch.save(rec, os.path.join(GOLD, "trainer_unet.pt"))
    print("trainer_unet.pt: losses", ["%.5f" % v for v in seen], "acc", rec["pixel_accuracy"], "mIoU", rec["mean_iou"], "keys", list(rec["weights"]))


def gen_misc():
    sys.path.insert(0, reference_harness.REFERENCE)
    from utils.metrics import eval_metrics
    from utils.sync_batchnorm import SynchronizedBatchNorm2d
    logits = (2 * torch.sin(0.37 * torch.arange(24.0))).reshape(1, 3, 2, 4)
    target = torch.tensor([[[0, 1, 2, 255], [2, 2, 0, 1]]])
    correct, labeled, inter, union = eval_metrics(logits, target, 3)
    g = torch.Generator().manual_seed(5)
    lg = torch.randn(2, 6, 9, 9, generator=g)
    tg = torch.randint(0, 6, (2, 9, 9), generator=g)
    tg[:, 0] = 255
    m2 = eval_metrics(lg, tg, 6)
    bn = SynchronizedBatchNorm2d(2)
    mean, inv_std = bn._compute_mean_std(torch.tensor([10.0, -4.0]), torch.tensor([30.0, 4.000001]), 8)
    # learning-rate / momentum sequences of the reference's schedulers, stepped exactly as trainer.py:52 does
    from utils import lr_scheduler as ref_sched

    def run_sched(cls, **kw):
        ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(2)]
        opt = torch.optim.SGD([{"params": ps[:1]}, {"params": ps[1:], "lr": 0.001}], lr=0.01, momentum=0.9)
        sch = cls(opt, 3, 5, **kw)
        seq = []
        for epoch in range(1, 4):
            for _ in range(5):
                sch.step(epoch=epoch - 1)
                seq.append([g["lr"] for g in opt.param_groups] + [g["momentum"] for g in opt.param_groups])
        return torch.tensor(seq, dtype=torch.float64)

    sched = {"Poly": run_sched(ref_sched.Poly), "Poly_warmup": run_sched(ref_sched.Poly, warmup_epochs=1),
             "OneCycle": run_sched(ref_sched.OneCycle)}
    rec = {
        "schedulers": sched,
        "metrics_appendix_c": {"logits": logits, "target": target, "correct": float(correct), "labeled": float(labeled),
                               "inter": torch.as_tensor(inter), "union": torch.as_tensor(union)},
        "metrics_rand": {"logits": lg, "target": tg, "correct": float(m2[0]), "labeled": float(m2[1]),
                         "inter": torch.as_tensor(m2[2]), "union": torch.as_tensor(m2[3])},
        "syncbn_mean_std": {"sum": torch.tensor([10.0, -4.0]), "ssum": torch.tensor([30.0, 4.000001]), "n": 8,
                            "mean": mean.clone(), "inv_std": inv_std.clone(),
                            "running_mean": bn.running_mean.clone(), "running_var": bn.running_var.clone()},
    }
    torch.save(rec, os.path.join(GOLD, "misc.pt"))
    print("misc.pt ok")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    models, losses = reference_harness.load()
    which = sys.argv[1:] or ["losses", "pspnet", "unet", "deeplab", "trainer", "misc"]
    if "losses" in which:
        gen_losses(losses)
    if "pspnet" in which:
        gen_pspnet(models, losses)
    if "unet" in which:
        gen_unet(models, losses)
    if "deeplab" in which:
        gen_deeplab(models, losses)
    if "trainer" in which:
        gen_trainer(models, losses)
    if "misc" in which:
        gen_misc()
