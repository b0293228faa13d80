"""GPU: parity of the HIP path with the REAL reference AT THE BASELINE.json SHAPES (north_star: "PSPNet-R50 at 512x512 ...
argmax masks bit-identical to the CPU reference").

tests/golden/full_cfg{2,3,4,5}.pt were written by oracle/gen_golden_fullsize.py from the imported reference running on torch
CPU: uint8 argmax masks, fp16 top-2 margins, main-head logits at a pixel stride, loss, per-tensor gradient digests.
For every config this test runs ONE training step of the drop-in model on the same weights / inputs and PRINTS the audit
numbers of SURVEY.md §7 (pytest -s, or the captured stdout of a failure):

    mismatch count of the full-resolution masks, the largest oracle margin among mismatching pixels, max|dlogit|

and asserts
  * logits:  max|dlogit| <= 1e-3 * max|logit| for EVERY config (strided sample of the main head; aux head for PSPNet), and the
             noise-floor criterion of tests/test_pspnet_gpu.py: the HIP logits are at most 2.0x as far from the fp64 oracle as the
             reference's own fp32 run is (measured 1.01-1.45x).  Both distances are printed for every config.  cfg3's backbone oracle is the restated
             torchvision ResNet-v1.5 (torchvision is absent): the audit line says "backbone oracle unpinned".
  * masks :  0 mismatches among pixels whose oracle top-2 margin exceeds 2*max|dlogit| — bit-identity on EVERY pixel is not
             attainable between two fp32 summation orders (torch-CPU NCHW vs channels_last already differ on 341 of 1 M
             pixels); every remaining mismatch is a numerical tie, and the count is printed
  * loss  :  |d| < 1e-4
  * gradients (BN batch statistics => ill conditioned), pinned to the MEASURED rounding-noise floor of each config:
             the fixtures carry the fp64 oracle's gradient digests (`grads_f64`: 64 strided samples + the first 8 values of every
             parameter gradient, oracle/gen_golden_fullsize.py `add_f64_grads`) and `ref_grad_err_f64` = the REAL reference's own
             fp32 digests' relative L2 distance from them (median / max over tensors: cfg2 1.85e-2 / 2.9e-2, cfg3 4.4e-2 / 7.0e-2,
             cfg4 1.9e-2 / 2.7e-2, cfg5 1.6e-2 / 2.9e-2).  The HIP path's distance from the SAME fp64 digests must be
             median <= 1.5 x and max <= 2 x the reference's own: a second fp32 evaluation cannot be expected closer to fp64 than
             the first, and anything systematically wrong (a dropped Winograd sub-grid, a lost split-K partial, 5 % of a tensor's
             energy corrupted) lands far above it.  Per-tensor norms: median within 1 %, max within 3 % of the reference's.
             The distance from the reference's fp32 digests (~ sqrt(2) x the floor) is printed as well; everything is recorded in
             gpurun_out/audit.json.
Batches: cfg2 8 (= BASELINE), cfg3 the batch stored in the fixture (16 = BASELINE when the build container could hold it), cfg4 one
shard of 4 (= BASELINE per GPU), cfg5 8 (= BASELINE per GPU); the test id carries the batch.
"""
import os
import statistics

import pytest
import torch

from oracle.weights import synth_batch, synth_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


WIDE_STATS = ("knorm", "cnorm", "tapnorm")


def wide_stats(g):
    """Exact whole-tensor reductions of a filter gradient [K,C,R,S] (fp64 accumulation on the device): per-output-channel L2 norms,
    per-input-channel L2 norms, per-tap L2 norms — the same definition as oracle/gen_golden_fullsize.py `wide_stats`."""
    g = g.detach().double()
    return {"knorm": g.flatten(1).norm(dim=1).cpu(), "cnorm": g.transpose(0, 1).flatten(1).norm(dim=1).cpu(),
            "tapnorm": g.pow(2).sum(dim=(0, 1)).sqrt().reshape(-1).cpu()}


def collect_step(rec, module, out, aux, loss):
    """Everything the audit needs from ONE rank's finished training step, as CPU tensors / numbers (small enough to cross a
    process boundary): mask, strided logits, loss, gradient digests at the fixture's sample positions, the wide whole-tensor
    statistics, segmentation counters from the fused device kernel, running statistics."""
    s = rec["stride"]
    o = out.detach()
    got = {"mask": o.argmax(1).to(torch.uint8).cpu(), "osub": o[:, :, ::s, ::s].cpu(), "loss": loss.item(), "shape": tuple(out.shape)}
    if aux is not None:
        got["aux"] = aux.detach()[:, :, ::2 * s, ::2 * s].cpu()
    named = dict(module.named_parameters())
    norms, samples = [], []
    for k in rec["grads"]:
        g = named[k].grad.detach().reshape(-1)
        step = max(1, g.numel() // 64)
        norms.append(g.norm().item())
        samples.append(torch.cat([g[::step][:64], g[:8]]).cpu())
    got["grad_norms"], got["grad_samples"] = norms, samples
    got["wide"] = {k: wide_stats(named[k].grad) for k in rec.get("wide", {})}
    sd_after = module.state_dict()
    got["running"] = {k: sd_after[k].detach().cpu().float() for k in rec["running"]}
    return got


def collect_metrics(rec, out, target):
    """[correct, labeled, inter[C], pred[C], label[C]] int64 counters of seg_metrics_kernel on this rank's logits (exact, summable)."""
    from utils.metrics import SegMetrics
    m = SegMetrics(rec["num_classes"], out.device)
    m.update(out.detach(), target)
    return m.acc.cpu()


def _miou_pixacc(correct, labeled, inter, union):
    """Trainer._get_seg_metrics (reference trainer.py:181-193) without the display rounding."""
    import numpy as np
    pix = 1.0 * float(correct) / (np.spacing(1) + float(labeled))
    iou = 1.0 * inter.double().numpy() / (np.spacing(1) + union.double().numpy())
    return float(iou.mean()), pix


def evaluate_audit(rec, got, name):
    """The audit numbers of one config from the fixture `rec` and the collected step `got` (whole global batch: for a sharded
    run the caller concatenates the ranks' rows and passes the all-reduced gradients of one rank)."""
    N, _, H, W = rec["input_shape"]
    C = rec["num_classes"]
    assert tuple(got["shape"]) == (N, C, H, W), got["shape"]
    mask, osub = got["mask"], got["osub"]
    d = (osub - rec["logits"]).abs().max().item()
    margin = rec["margin"].float()
    mism = mask != rec["mask"]
    n_mis = int(mism.sum())
    max_margin_mis = float(margin[mism].max()) if n_mis else 0.0
    bad = int((mism & (margin > 2 * d)).sum())
    res = {"config": name, "pixels": mask.numel(), "mismatches": n_mis, "max_margin_among_mismatches": max_margin_mis,
           "max_abs_dlogit": d, "logit_absmax": rec["logit_absmax"], "mismatches_outside_margin": bad,
           "hip_err_f64": (osub.double() - rec["logits_f64"].double()).abs().max().item(), "ref_err_f64": rec["ref_err_f64"],
           "near_ties_in_oracle(margin<2d)": int((margin <= 2 * d).sum()),
           "loss": got["loss"], "loss_ref": rec["loss"].item()}
    if "aux" in got:
        res["max_abs_daux"] = (got["aux"] - rec["aux"]).abs().max().item()
        res["aux_absmax"] = rec["aux"].abs().max().item()
    rel, srel, frel = [], [], []
    for (k, dg), gnorm, gs in zip(rec["grads"].items(), got["grad_norms"], got["grad_samples"]):
        rel.append((abs(gnorm - dg["norm"]) / (dg["norm"] + 1e-30), k, dg["norm"]))
        gs = gs.double()
        ref = torch.cat([dg["sample"], dg["head"]]).double()
        srel.append(((gs - ref).norm().item() / (ref.norm().item() + 1e-30), k, dg["norm"]))
        d64 = rec["grads_f64"][k]
        r64 = torch.cat([d64["sample"], d64["head"]]).double()
        frel.append(((gs - r64).norm().item() / (r64.norm().item() + 1e-300), k, d64["norm"]))
    res["grad_norm_rel_err_median"] = statistics.median(r[0] for r in rel)
    floor = 1e-5 * max(r[2] for r in rel)        # analytically-zero gradients (BN bias in front of a batch-stat BN) are rounding noise
    res["grad_norm_rel_err_max"], res["grad_norm_worst"] = max(((r[0], r[1]) for r in rel if r[2] > floor), default=(0.0, ""))
    res["grad_sample_rel_err_median"] = statistics.median(r[0] for r in srel)
    res["grad_sample_rel_err_max"], res["grad_sample_worst"] = max(((r[0], r[1]) for r in srel if r[2] > floor), default=(0.0, ""))
    # distance from the fp64 oracle's digests, over the same "live" tensors as the fixture's own floor (gen_golden_fullsize.add_f64_grads)
    top = max(r[2] for r in frel)
    live = [r for r in frel if r[2] > 1e-5 * top]
    res["grad_f64_rel_err_median"] = statistics.median(r[0] for r in live)
    res["grad_f64_rel_err_max"], res["grad_f64_worst"] = max((r[0], r[1]) for r in live)
    res["ref_grad_f64_rel_err_median"] = rec["ref_grad_err_f64"]["median"]
    res["ref_grad_f64_rel_err_max"] = rec["ref_grad_err_f64"]["max"]
    res["batch"] = N
    res["running_ok"] = all(torch.allclose(got["running"][k], v.float(), rtol=1e-4, atol=1e-5) for k, v in rec["running"].items())
    # ---- whole-tensor statistics of the largest filter gradients: a defect confined to one K-panel / C-panel / tap
    # of a large gradient moves its row of knorm / cnorm / tapnorm.  Worst ratio of the HIP path's distance from the fp64 oracle's
    # statistics to the reference fp32's own distance, and the worst absolute distance
    worst_ratio, worst_abs, worst_key = 0.0, 0.0, ""
    for k, w64 in rec.get("wide_f64", {}).items():
        for st in WIDE_STATS:
            ref_e = rec["wide_ref_err_f64"][k][st]
            e = ((got["wide"][k][st].double() - w64[st].double()).norm() / (w64[st].double().norm() + 1e-300)).item()
            res["wide/%s/%s" % (k, st)] = e
            if e / max(ref_e, WIDE_ABS_FLOOR) > worst_ratio:
                worst_ratio, worst_key = e / max(ref_e, WIDE_ABS_FLOOR), "%s/%s" % (k, st)
            worst_abs = max(worst_abs, e)
    res["wide_tensors"] = len(rec.get("wide_f64", {}))
    res["wide_worst_ratio"], res["wide_worst_abs"], res["wide_worst"] = worst_ratio, worst_abs, worst_key
    # ---- the headline metric's parity clause (BASELINE.json "mIoU parity"): utils/metrics.py:42-67 evaluated by the
    # REAL reference on its own fp32 logits (fixture) against seg_metrics_kernel on the HIP logits
    if "metrics" in rec and "metrics" in got:
        a = got["metrics"]
        correct, labeled = int(a[0]), int(a[1])
        inter, pred, lab = a[2:2 + C], a[2 + C:2 + 2 * C], a[2 + 2 * C:2 + 3 * C]
        union = pred + lab - inter
        rm = rec["metrics"]
        res["metrics_labeled_equal"] = labeled == rm["labeled"]
        res["metrics_dcorrect"] = abs(correct - rm["correct"])
        res["metrics_dinter_max"] = int((inter - rm["inter"]).abs().max())
        res["metrics_dunion_max"] = int((union - rm["union"]).abs().max())
        res["miou"], res["pixacc"] = _miou_pixacc(correct, labeled, inter, union)
        res["miou_ref"], res["pixacc_ref"] = _miou_pixacc(rm["correct"], rm["labeled"], rm["inter"], rm["union"])
    return res


# statistics whose reference-fp32-vs-fp64 distance is below this are compared against this absolute distance instead (a ratio of two
# numbers at the 1e-7 level is noise)
WIDE_ABS_FLOOR = 5e-5
WIDE_BAR = 2.0        # measured, worst statistic per audit over all configs and both algorithms: 1.15-1.94 (profiles/r06_fullsize_audit.json;
                      # the 1.94 is the ONE-number tap norm of a 1x1 filter — 1.06e-4 against the 5e-5 floor; knorm / cnorm stay <= 1.34)


def run_fullsize_audit(name, device):
    """One train step of config `name` on `device` against tests/golden/full_<name>.pt.  Returns the audit dict
    (also used by __graft_entry__.smoke())."""
    import models
    import utils.losses as losses_mod
    rec = torch.load(os.path.join(GOLD, "full_%s.pt" % name), weights_only=False)
    C, kw, ign = rec["num_classes"], rec["kwargs"], rec["ignore_index"]
    N, _, H, W = rec["input_shape"]
    m = getattr(models, rec["arch"])(C, pretrained=False, **kw)
    m.load_state_dict(synth_state_dict(rec["manifest"], seed=rec["weight_seed"]))
    m.to(device).train()
    for mod in m.modules():
        if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout2d)):
            mod.eval()                   # GPU dropout masks differ from aten's by construction; parity runs neutralise dropout
    crit = getattr(losses_mod, rec["loss_name"])(ignore_index=ign)
    x, t = synth_batch(N, 3, H, W, C, ignore_index=ign, seed=rec["batch_seed"])
    xd, td = x.to(device), t.to(device)
    out = m(xd)
    aux = None
    if rec["arch"][:3] == "PSP":
        out, aux = out
        loss = crit(out, td) + 0.4 * crit(aux, td)
    else:
        loss = crit(out, td)
    loss.backward()
    got = collect_step(rec, m, out, aux, loss)
    got["metrics"] = collect_metrics(rec, out, td)
    return evaluate_audit(rec, got, name)


# batch of every fixture (part of the numerics under batch-statistics BN, so it is in the test id); kept here so that collecting
# this module does not load 60 MB of fixtures — the test asserts it against the fixture header
FIXTURE_BATCH = {"cfg2": 8, "cfg3": 16, "cfg4": 4, "cfg5": 8}


def audit_line(r, algo):
    note = " [backbone oracle unpinned (torchvision ResNet-v1.5 restated, oracle/tv_resnet.py)]" if r["config"] == "cfg3" else ""
    head = "[fullsize %s batch %d, fp32 MFMA, %s]" % (r["config"], r["batch"], algo)
    body = ("pixels %d | argmax mismatches %d | max margin among mismatches %.3e | max|dlogit| %.3e "
            "(max|logit| %.3f) | distance from the fp64 oracle: HIP %.3e, reference fp32 %.3e | mismatches outside 2*max|dlogit| %d | "
            "oracle pixels within that margin %d | loss %.6f (ref %.6f) | grad-norm rel err median %.2e max %.2e (%s) | "
            "grad-sample rel-L2 from the reference fp32 median %.2e max %.2e (%s) | from the fp64 oracle: HIP median %.2e max %.2e (%s), "
            "reference fp32 median %.2e max %.2e | %d largest filter gradients, whole-tensor knorm / cnorm / tapnorm vs fp64: worst %.2e "
            "= %.2f x the reference fp32's own (%s) | eval_metrics vs the reference's on its own logits: |d correct| %d, max |d inter| %d, "
            "max |d union| %d, mIoU %.6f (ref %.6f), pixAcc %.6f (ref %.6f)"
            % (r["pixels"], r["mismatches"], r["max_margin_among_mismatches"],
               r["max_abs_dlogit"], r["logit_absmax"], r["hip_err_f64"], r["ref_err_f64"], r["mismatches_outside_margin"],
               r["near_ties_in_oracle(margin<2d)"], r["loss"], r["loss_ref"], r["grad_norm_rel_err_median"], r["grad_norm_rel_err_max"],
               r["grad_norm_worst"], r["grad_sample_rel_err_median"], r["grad_sample_rel_err_max"], r["grad_sample_worst"],
               r["grad_f64_rel_err_median"], r["grad_f64_rel_err_max"], r["grad_f64_worst"], r["ref_grad_f64_rel_err_median"],
               r["ref_grad_f64_rel_err_max"], r["wide_tensors"], r["wide_worst_abs"], r["wide_worst_ratio"], r["wide_worst"],
               r["metrics_dcorrect"], r["metrics_dinter_max"], r["metrics_dunion_max"], r["miou"], r["miou_ref"], r["pixacc"], r["pixacc_ref"]))
    return head + note + " " + body


def record_audit(r, algo):
    """Append the audit to gpurun_out/audit.json (independent of pytest's output capture) and return its one-line form."""
    import json
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "audit.json")
        doc = json.load(open(path)) if os.path.exists(path) else {}
        doc["%s/f32/%s" % (r["config"], algo)] = {k: v for k, v in r.items() if isinstance(v, (int, float, str, bool))}
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    return audit_line(r, algo)


# Bar on (HIP logits' distance from the fp64 oracle) / (the reference fp32 run's own distance), a max statistic over the strided sample:
# measured 1.01-1.45x over all configs and both algorithms (profiles/r0{4,5}_fullsize_audit.json).  (Round 5 had loosened it to 2.5x
# for the second arithmetic, which is retired; VERDICT r5 weak #1.)
FP64_LOGIT_BAR = 2.0


def assert_audit(r):
    """The acceptance criteria of one full-size audit (shared by the single-rank configs, the 2-rank SyncBN run of
    tests/test_distributed_gpu.py and __graft_entry__.smoke())."""
    assert r["max_abs_dlogit"] <= 1e-3 * r["logit_absmax"], r
    # a kernel bug (a dropped tap, a lost partial) lands orders of magnitude above the bar
    assert r["hip_err_f64"] <= FP64_LOGIT_BAR * r["ref_err_f64"], r
    assert r["mismatches_outside_margin"] == 0, r
    if "max_abs_daux" in r:
        assert r["max_abs_daux"] <= 1e-3 * r["aux_absmax"], r
    assert abs(r["loss"] - r["loss_ref"]) < 1e-4, r
    assert r["grad_norm_rel_err_median"] <= 1e-2 and r["grad_norm_rel_err_max"] <= 3e-2, r
    assert r["grad_f64_rel_err_median"] <= 1.5 * r["ref_grad_f64_rel_err_median"], r
    assert r["grad_f64_rel_err_max"] <= 2.0 * r["ref_grad_f64_rel_err_max"], r
    assert r["running_ok"], r
    # whole-tensor statistics of the 6 largest filter gradients: no further from the fp64 oracle's than WIDE_BAR x the reference fp32's own
    assert r["wide_tensors"] == 6 and r["wide_worst_ratio"] <= WIDE_BAR, r
    # the metric clause: every counter within the number of mismatching (= tied) pixels, mIoU / pixel accuracy to 1e-4
    assert r["metrics_labeled_equal"], r
    assert r["metrics_dcorrect"] <= r["mismatches"] and r["metrics_dinter_max"] <= r["mismatches"] and r["metrics_dunion_max"] <= r["mismatches"], r
    assert abs(r["miou"] - r["miou_ref"]) <= 1e-4 and abs(r["pixacc"] - r["pixacc_ref"]) <= 1e-4, r


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4", "cfg5"], ids=lambda n: "%s-batch%d" % (n, FIXTURE_BATCH[n]))
def test_fullsize_step_matches_reference_golden(cuda, name, conv_algorithm):
    r = run_fullsize_audit(name, cuda)
    print("\n" + record_audit(r, conv_algorithm or "default"))
    assert r["batch"] == FIXTURE_BATCH[name]
    assert_audit(r)
