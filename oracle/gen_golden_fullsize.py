"""Generate tests/golden/full_*.pt: the REAL reference (/root/reference) run on torch CPU at the BASELINE.json shapes.

    python oracle/gen_golden_fullsize.py [cfg2 cfg3 cfg4 cfg5]      (build container only; ~5 min, peak RSS ~13 GB)
    python oracle/gen_golden_fullsize.py f64 [cfg...]               fp64 oracle logits + the reference's distance from them
    python oracle/gen_golden_fullsize.py f64grads cfgN              fp64 oracle BACKWARD: gradient digests + the reference's distance

One training step per config (dropout neutralised, BN batch statistics, weights from oracle.weights.synth_state_dict,
inputs from oracle.weights.synth_batch) with exactly the loss expression of the reference's hot loop
(trainer.py:56-66: CE + 0.4*CE(aux) for PSP*, the configured loss otherwise).  Full-resolution logits are too large to commit
(cfg2: 176 MB), so each fixture keeps what the north-star acceptance sentence needs:

  mask     uint8  [N,H,W]   argmax of the main head            -> mismatch COUNT of the HIP path's masks, bit for bit
  margin   fp16   [N,H,W]   top-1 minus top-2 logit            -> "0 mismatches among pixels whose margin > 2*max|dlogit|"
  logits   fp32   [N,C,H/s,W/s]  main head at pixel stride s   -> max|dlogit| (and aux head at stride 2s for PSP)
  loss, per-tensor gradient digests (norm / absmax / 64 samples), a few running statistics.
  logits_f64 [N,C,H/s,W/s]  the same forward by the oracle restatement (oracle/pspnet_ref.py, oracle/deeplab_ref.py) in fp64 —
           first checked to reproduce the reference's fp32 logits BIT FOR BIT in fp32 — and ref_err_f64 = max|reference fp32 -
           fp64|: the rounding-noise floor of this config.  Deep batch-statistics stacks at batch 2 (cfg3) amplify fp32 rounding
           to ~1e-3 of the logit scale, so "1e-3 * max|logit| from the fp32 reference" is not attainable by ANY second fp32
           implementation there; the test then requires the HIP path to be as close to fp64 as the reference is.

  metrics  the reference's own `eval_metrics` (utils/metrics.py:59-67: correct, labeled, inter[C], union[C]) on its fp32 logits -> the
           checkable form of "argmax masks bit-identical => mIoU parity"
  wide     for the 6 largest filter gradients: per-output-channel L2 norms, per-input-channel L2 norms and per-tap L2 norms — EXACT
           reductions of the whole tensor (fp64 accumulation), so that a defect confined to one tile of a 72 MB gradient, which the 64
           strided samples can miss, moves a stored number; `wide_f64` = the same from the fp64 oracle backward.

cfg4_sync8 = BASELINE cfg4's defining regime, SyncBN, at a global batch the build container can hold: the reference PSPNet-R50 after
its own `convert_model` (utils/sync_batchnorm/batchnorm.py:353-394) on ONE CPU process with the concatenated global batch of 8 x 769^2
— `_SynchronizedBatchNorm.forward` then IS `F.batch_norm` over the global batch (batchnorm.py:65-68), the semantics the data-parallel
path has to reproduce from 2 ranks x 4 images (tests/test_distributed_gpu.py).  Its fp64 backward recomputes every bottleneck
(oracle/pspnet_ref.py `checkpoint=True`; 70 GB otherwise).

cfg2 = PSPNet-R50 8x3x512x512 21 classes (the bench line); cfg3 = DeepLabV3+ R101 OS16 513x513 19 classes at batch 2 (of 16);
cfg4 = one SyncBN shard's shape, PSPNet-R50 4x3x769x769 19 classes (97x97 maps), local BN; cfg5 = DeepLabV3+ Xception 512x512
150 classes + LovaszSoftmax at batch 2 (of 8), ignore_index -1.
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import reference_harness  # noqa: E402
from oracle.gen_golden import _grad_digest  # noqa: E402
from oracle.weights import manifest_of, synth_batch, synth_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# name: (arch, kwargs, classes, N, H, W, loss, ignore_index, logits stride, weight seed, batch seed, running-stat keys)
FULL = {
    "cfg2": ("PSPNet", dict(backbone="resnet50"), 21, 8, 512, 512, "CrossEntropyLoss2d", 255, 8, 0, 1234,
             ("initial.0.1.running_mean", "layer4.2.bn3.running_var", "master_branch.0.bottleneck.1.running_mean")),
    "cfg3": ("DeepLab", dict(backbone="resnet101", output_stride=16), 19, int(os.environ.get("SEGMI_GOLDEN_CFG3_BATCH", "16")), 513, 513, "CrossEntropyLoss2d", 255, 9, 2, 555,
             ("backbone.layer4.2.bn3.running_var", "decoder.bn1.running_mean")),
    "cfg4": ("PSPNet", dict(backbone="resnet50"), 19, 4, 769, 769, "CrossEntropyLoss2d", 255, 12, 0, 4321,
             ("initial.0.1.running_mean", "layer4.2.bn3.running_var", "master_branch.0.bottleneck.1.running_mean")),
    "cfg5": ("DeepLab", dict(backbone="xception", output_stride=16), 150, 8, 512, 512, "LovaszSoftmax", -1, 16, 2, 777,
             ("ASSP.aspp4.1.running_var", "decoder.bn1.running_mean")),
    "cfg4_sync8": ("PSPNet", dict(backbone="resnet50"), 19, 8, 769, 769, "CrossEntropyLoss2d", 255, 12, 0, 4321,
                   ("initial.0.1.running_mean", "initial.1.running_var", "layer4.2.bn3.running_mean", "layer4.2.bn3.running_var",
                    "master_branch.0.stages.3.2.running_var", "master_branch.0.bottleneck.1.running_mean", "auxiliary_branch.1.running_var")),
}
SYNCBN = {"cfg4_sync8"}        # the reference's convert_model() is applied: SynchronizedBatchNorm2d modules on one CPU process
CHECKPOINT_F64 = {"cfg4_sync8"}
WIDE_TOP = 6


def wide_keys(manifest):
    """The 6 largest 4-D parameters (filters) of a model, by element count, ties in manifest order."""
    filt = [(-int(torch.Size(shape).numel()), i, k) for i, (k, shape) in enumerate(manifest) if len(shape) == 4]
    return [k for _, _, k in sorted(filt)[:WIDE_TOP]]


def wide_stats(g):
    """Exact whole-tensor reductions of a filter gradient [K,C,R,S] in fp64: O(K + C + R*S) numbers."""
    g = g.detach().double()
    # (norms, not sums: a per-tap SUM cancels to ~1e-3 of the tap's norm, so its relative error is rounding noise amplified 100-1000x —
    #  measured ratios 0.3-5.8 between two correct fp32 evaluations — and a dropped tap would barely move it)
    return {"knorm": g.flatten(1).norm(dim=1), "cnorm": g.transpose(0, 1).flatten(1).norm(dim=1), "tapnorm": g.pow(2).sum(dim=(0, 1)).sqrt().reshape(-1)}


def wide_rel_err(a, b):
    """Per statistic: relative L2 distance of a's vector from b's."""
    return {k: ((a[k].double() - b[k].double()).norm() / (b[k].double().norm() + 1e-300)).item() for k in ("knorm", "cnorm", "tapnorm")}


def gen(name, models, losses):
    arch, kw, C, N, H, W, loss_name, ign, stride, wseed, bseed, run_keys = FULL[name]
    torch.manual_seed(0)
    model = getattr(models, arch)(C, pretrained=False, **kw)
    if kw.get("backbone") == "xception":
        model.backbone.block2.relu.inplace = False          # numerically neutral autograd workaround
    man = manifest_of(model.state_dict())
    model.load_state_dict(synth_state_dict(man, seed=wseed))
    if name in SYNCBN:
        from utils.sync_batchnorm import SynchronizedBatchNorm2d, convert_model     # the REFERENCE's (sys.path: /root/reference first)
        model = convert_model(model)
        assert sum(isinstance(m, SynchronizedBatchNorm2d) for m in model.modules()) == 61 and manifest_of(model.state_dict()) == man
    model.train()
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.eval()
    crit = getattr(losses, loss_name)(ignore_index=ign)
    x, t = synth_batch(N, 3, H, W, C, ignore_index=ign, seed=bseed)
    t0 = time.time()
    out = model(x)
    aux = None
    if arch[:3] == "PSP":                       # trainer.py:57-62
        out, aux = out
        loss = crit(out, t) + 0.4 * crit(aux, t)
    else:
        loss = crit(out, t)
    t1 = time.time()
    loss.backward()
    t2 = time.time()
    o = out.detach()
    top2 = o.topk(2, dim=1).values
    sd_after = model.state_dict()
    from utils import metrics as ref_metrics                # the REFERENCE's utils/metrics.py
    correct, labeled, inter, union = ref_metrics.eval_metrics(o, t, C)
    named = dict(model.named_parameters())
    rec = {"name": name, "syncbn": name in SYNCBN,
           "metrics": {"correct": int(correct), "labeled": int(labeled), "inter": torch.from_numpy(inter.astype("int64")),
                       "union": torch.from_numpy(union.astype("int64"))},
           "wide": {k: wide_stats(named[k].grad) for k in wide_keys(man)}, "arch": arch, "kwargs": kw, "num_classes": C, "input_shape": (N, 3, H, W), "loss_name": loss_name,
           "ignore_index": ign, "weight_seed": wseed, "batch_seed": bseed, "stride": stride, "manifest": man,
           "mask": o.argmax(1).to(torch.uint8), "margin": (top2[:, 0] - top2[:, 1]).to(torch.float16),
           "logits": o[:, :, ::stride, ::stride].clone(), "logit_absmax": o.abs().max().item(),
           "loss": loss.detach().clone(), "grads": _grad_digest(model.named_parameters()),
           "running": {k: sd_after[k].clone() for k in run_keys},
           "cpu_seconds": {"forward_loss": t1 - t0, "backward": t2 - t1, "threads": torch.get_num_threads()}}
    if aux is not None:
        rec["aux"] = aux.detach()[:, :, ::2 * stride, ::2 * stride].clone()
    path = os.path.join(GOLD, "full_%s.pt" % name)
    if os.path.exists(path):                    # re-generation: the fp32 record must come out bit for bit; the fp64 passes are kept
        old = torch.load(path, weights_only=False)
        assert torch.equal(old["logits"], rec["logits"]) and torch.equal(old["mask"], rec["mask"]) and torch.equal(old["loss"], rec["loss"]), name
        assert all(torch.equal(old["grads"][k]["sample"], v["sample"]) for k, v in rec["grads"].items()), name
        for k in ("logits_f64", "ref_err_f64", "grads_f64", "ref_grad_err_f64", "loss_f64"):      # (wide_f64: rewritten by the f64grads pass)
            if k in old:
                rec[k] = old[k]
        print("%s: identical to the committed fixture in logits, masks, loss and gradient samples" % name, flush=True)
    torch.save(rec, path)
    print("%s: loss %.6f |logit| max %.3f, margin<1e-4 on %d px, fwd %.1fs bwd %.1fs -> %s (%.1f MB)"
          % (name, loss.item(), rec["logit_absmax"], int((rec["margin"].float() < 1e-4).sum()), t1 - t0, t2 - t1, path,
             os.path.getsize(path) / 1e6), flush=True)


def add_f64(name):
    """Append the fp64 oracle logits (and the reference's own distance from them) to an existing fixture."""
    from oracle import deeplab_ref, pspnet_ref
    path = os.path.join(GOLD, "full_%s.pt" % name)
    rec = torch.load(path, weights_only=False)
    sd = synth_state_dict(rec["manifest"], seed=rec["weight_seed"])
    N, _, H, W = rec["input_shape"]
    s = rec["stride"]
    x, _ = synth_batch(N, 3, H, W, rec["num_classes"], ignore_index=rec["ignore_index"], seed=rec["batch_seed"])

    def fwd(dt):
        st = pspnet_ref.clone_state({k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}, requires_grad=False)
        with torch.no_grad():
            if rec["arch"] == "PSPNet":
                return pspnet_ref.pspnet_forward(st, x.to(dt), training=True, backbone=rec["kwargs"]["backbone"])[0]
            return deeplab_ref.deeplab_forward(st, x.to(dt), rec["kwargs"]["backbone"], rec["kwargs"]["output_stride"], training=True)

    o32 = fwd(torch.float32)[:, :, ::s, ::s]
    assert torch.equal(o32, rec["logits"]), "oracle restatement is not bit-identical to the reference in fp32 (%s)" % name
    o64 = fwd(torch.float64)[:, :, ::s, ::s].clone()
    rec["ref_err_f64"] = (rec["logits"].double() - o64).abs().max().item()
    rec["logits_f64"] = o64.float()          # the fp64 result rounded ONCE to fp32 (1e-7; the floors measured against it are >= 1e-4)
    torch.save(rec, path)
    print("%s: oracle fp32 == reference fp32 bit for bit; max|reference fp32 - fp64| = %.3e (%.2e of max|logit|) -> %.1f MB"
          % (name, rec["ref_err_f64"], rec["ref_err_f64"] / rec["logit_absmax"], os.path.getsize(path) / 1e6), flush=True)


def _lovasz_softmax_any_dtype(logits, target, ignore_index):
    """oracle.losses_ref.lovasz_softmax with every intermediate in the logits' dtype (the reference hard-codes .float() in
    utils/lovasz_losses.py:185,194-198 and therefore raises for fp64 input): the same piecewise-linear function, evaluated in fp64."""
    import torch.nn.functional as F
    C = logits.shape[1]
    p = F.softmax(logits, dim=1).permute(0, 2, 3, 1).reshape(-1, C)
    t = target.reshape(-1)
    keep = t != ignore_index
    p, t = p[keep], t[keep]
    losses = []
    for c in range(C):
        fg = (t == c).to(p.dtype)
        total = fg.sum()
        if total == 0:
            continue
        err = (fg - p[:, c]).abs()
        err_sorted, perm = torch.sort(err, 0, descending=True)
        g = fg[perm]
        jac = 1.0 - (total - g.cumsum(0)) / (total + (1 - g).cumsum(0))
        if g.numel() > 1:
            jac[1:] = jac[1:] - jac[:-1].clone()
        losses.append(torch.dot(err_sorted, jac.detach()))
    return sum(losses) / len(losses)


def add_f64_grads(name):
    """Append the fp64 oracle BACKWARD to an existing fixture: per-tensor digests of d loss / d parameter
    evaluated in fp64 at the same sample positions as `grads` ("grads_f64": norm, 64 strided samples, first 8 values), and
    `ref_grad_err_f64` = the REAL reference's own fp32 digests' relative L2 distance from them, per tensor, with its median and
    maximum — the measured rounding-noise floor of this config's gradients that tests/test_fullsize_golden_gpu.py holds the HIP
    path to.  One process per config (cfg2: ~25 GB, cfg3 at batch 16 and cfg5: more; run them one at a time)."""
    import statistics
    from oracle import deeplab_ref, losses_ref, pspnet_ref
    path = os.path.join(GOLD, "full_%s.pt" % name)
    rec = torch.load(path, weights_only=False)
    sd = synth_state_dict(rec["manifest"], seed=rec["weight_seed"])
    N, _, H, W = rec["input_shape"]
    C, ign = rec["num_classes"], rec["ignore_index"]
    x, t = synth_batch(N, 3, H, W, C, ignore_index=ign, seed=rec["batch_seed"])
    dt = torch.float64
    st = pspnet_ref.clone_state({k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}, requires_grad=True)
    t0 = time.time()
    if rec["arch"] == "PSPNet":
        out, aux = pspnet_ref.pspnet_forward(st, x.to(dt), training=True, backbone=rec["kwargs"]["backbone"], checkpoint=name in CHECKPOINT_F64)
        loss = losses_ref.cross_entropy(out, t, ign) + 0.4 * losses_ref.cross_entropy(aux, t, ign)
        del aux
    else:
        out = deeplab_ref.deeplab_forward(st, x.to(dt), rec["kwargs"]["backbone"], rec["kwargs"]["output_stride"], training=True)
        if rec["loss_name"] == "LovaszSoftmax":
            loss = _lovasz_softmax_any_dtype(out, t, ign)
        else:
            loss = losses_ref.cross_entropy(out, t, ign)
    s = rec["stride"]
    assert (out.detach()[:, :, ::s, ::s].float() - rec["logits_f64"]).abs().max().item() <= 1e-5 * rec["logit_absmax"], "fp64 forward drifted"
    del out
    loss.backward()
    g64, errs = {}, {}
    for k, dg in rec["grads"].items():
        g = st[k].grad.detach().reshape(-1)
        step = max(1, g.numel() // 64)
        g64[k] = {"norm": g.norm().item(), "head": g[:8].clone(), "sample": g[::step][:64].clone()}
        a = torch.cat([dg["sample"], dg["head"]]).double()
        b = torch.cat([g64[k]["sample"], g64[k]["head"]])
        errs[k] = ((a - b).norm() / (b.norm() + 1e-300)).item()
    top = max(v["norm"] for v in g64.values())
    live = [e for k, e in errs.items() if g64[k]["norm"] > 1e-5 * top]     # analytically-zero gradients are pure rounding noise
    if "grads_f64" in rec:                      # a re-run reproduces the committed digests bit for bit
        assert all(torch.equal(rec["grads_f64"][k]["sample"], v["sample"]) for k, v in g64.items()), name
    if "wide" in rec:
        rec["wide_f64"] = {k: wide_stats(st[k].grad) for k in rec["wide"]}
        rec["wide_ref_err_f64"] = {k: wide_rel_err(rec["wide"][k], rec["wide_f64"][k]) for k in rec["wide"]}
        for k, e in rec["wide_ref_err_f64"].items():
            print("  wide %-45s reference fp32 vs fp64: knorm %.2e cnorm %.2e tapnorm %.2e" % (k, e["knorm"], e["cnorm"], e["tapnorm"]), flush=True)
    rec["grads_f64"] = g64
    rec["ref_grad_err_f64"] = {"per_tensor": errs, "median": statistics.median(live), "max": max(live),
                               "worst": max((e, k) for k, e in errs.items() if g64[k]["norm"] > 1e-5 * top)[1]}
    rec["loss_f64"] = loss.item()
    torch.save(rec, path)
    import resource
    print("%s: fp64 backward in %.0f s (peak RSS %.1f GB); reference fp32 gradient digests vs fp64: rel-L2 median %.3e max %.3e (%s); "
          "loss fp64 %.8f vs reference %.8f" % (name, time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6,
                                                rec["ref_grad_err_f64"]["median"], rec["ref_grad_err_f64"]["max"], rec["ref_grad_err_f64"]["worst"],
                                                rec["loss_f64"], rec["loss"].item()), flush=True)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    args = sys.argv[1:]
    if args and args[0] == "f64":           # second pass, separate process (the oracle and the reference share module names)
        for name in (args[1:] or list(FULL)):
            add_f64(name)
        sys.exit(0)
    if args and args[0] == "f64grads":      # third pass, ONE config per process (memory)
        for name in (args[1:] or list(FULL)):
            add_f64_grads(name)
        sys.exit(0)
    models, losses = reference_harness.load()
    for name in (args or list(FULL)):
        gen(name, models, losses)
