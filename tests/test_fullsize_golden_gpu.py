"""GPU: parity of the HIP path with the REAL reference AT THE BASELINE.json SHAPES (north_star: "PSPNet-R50 at 512x512 ...
argmax masks bit-identical to the CPU reference").

tests/golden/full_cfg{2,3,4,5}.pt were written by oracle/gen_golden_fullsize.py from the imported reference running on torch
CPU: uint8 argmax masks, fp16 top-2 margins, main-head logits at a pixel stride, loss, per-tensor gradient digests.
For every config this test runs ONE training step of the drop-in model on the same weights / inputs and PRINTS the audit
numbers of SURVEY.md §7 (pytest -s, or the captured stdout of a failure):

    mismatch count of the full-resolution masks, the largest oracle margin among mismatching pixels, max|dlogit|

and asserts
  * logits:  max|dlogit| <= 1e-3 * max|logit| for EVERY config (strided sample of the main head; aux head for PSPNet), and the
             noise-floor criterion of tests/test_pspnet_gpu.py: the HIP logits are at most 2.5x as far from the fp64 oracle as the
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
The same file is the acceptance test of any alternative conv arithmetic (SEGMI_CONV_MATH): identical tolerances.
"""
import os
import statistics

import pytest
import torch

from oracle.weights import synth_batch, synth_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_fullsize_audit(name, device):
    """One train step of config `name` on `device` against tests/golden/full_<name>.pt.  Returns the audit dict
    (also used by __graft_entry__.smoke())."""
    import models
    import utils.losses as losses_mod
    rec = torch.load(os.path.join(GOLD, "full_%s.pt" % name), weights_only=False)
    C, kw, ign, s = rec["num_classes"], rec["kwargs"], rec["ignore_index"], rec["stride"]
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
    assert tuple(out.shape) == (N, C, H, W)
    o = out.detach()
    mask = o.argmax(1).to(torch.uint8).cpu()
    osub = o[:, :, ::s, ::s].cpu()
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
           "loss": loss.item(), "loss_ref": rec["loss"].item()}
    if aux is not None:
        res["max_abs_daux"] = (aux.detach()[:, :, ::2 * s, ::2 * s].cpu() - rec["aux"]).abs().max().item()
        res["aux_absmax"] = rec["aux"].abs().max().item()
    named = dict(m.named_parameters())
    rel, srel, frel = [], [], []
    for k, dg in rec["grads"].items():
        g = named[k].grad.detach().reshape(-1)
        rel.append((abs(g.norm().item() - dg["norm"]) / (dg["norm"] + 1e-30), k, dg["norm"]))
        step = max(1, g.numel() // 64)
        got = torch.cat([g[::step][:64], g[:8]]).cpu().double()
        ref = torch.cat([dg["sample"], dg["head"]]).double()
        srel.append(((got - ref).norm().item() / (ref.norm().item() + 1e-30), k, dg["norm"]))
        d64 = rec["grads_f64"][k]
        r64 = torch.cat([d64["sample"], d64["head"]]).double()
        frel.append(((got - r64).norm().item() / (r64.norm().item() + 1e-300), k, d64["norm"]))
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
    sd_after = m.state_dict()
    res["running_ok"] = all(torch.allclose(sd_after[k].cpu().float(), v.float(), rtol=1e-4, atol=1e-5) for k, v in rec["running"].items())
    return res


# batch of every fixture (part of the numerics under batch-statistics BN, so it is in the test id); kept here so that collecting
# this module does not load 60 MB of fixtures — the test asserts it against the fixture header
FIXTURE_BATCH = {"cfg2": 8, "cfg3": 16, "cfg4": 4, "cfg5": 8}


def audit_line(r, algo):
    from segmi import ops
    note = " [backbone oracle unpinned (torchvision ResNet-v1.5 restated, oracle/tv_resnet.py)]" if r["config"] == "cfg3" else ""
    head = "[fullsize %s batch %d, conv math %s, %s]" % (r["config"], r["batch"], ops.get_conv_math(), algo)
    body = ("pixels %d | argmax mismatches %d | max margin among mismatches %.3e | max|dlogit| %.3e "
            "(max|logit| %.3f) | distance from the fp64 oracle: HIP %.3e, reference fp32 %.3e | mismatches outside 2*max|dlogit| %d | "
            "oracle pixels within that margin %d | loss %.6f (ref %.6f) | grad-norm rel err median %.2e max %.2e (%s) | "
            "grad-sample rel-L2 from the reference fp32 median %.2e max %.2e (%s) | from the fp64 oracle: HIP median %.2e max %.2e (%s), "
            "reference fp32 median %.2e max %.2e"
            % (r["pixels"], r["mismatches"], r["max_margin_among_mismatches"],
               r["max_abs_dlogit"], r["logit_absmax"], r["hip_err_f64"], r["ref_err_f64"], r["mismatches_outside_margin"],
               r["near_ties_in_oracle(margin<2d)"], r["loss"], r["loss_ref"], r["grad_norm_rel_err_median"], r["grad_norm_rel_err_max"],
               r["grad_norm_worst"], r["grad_sample_rel_err_median"], r["grad_sample_rel_err_max"], r["grad_sample_worst"],
               r["grad_f64_rel_err_median"], r["grad_f64_rel_err_max"], r["grad_f64_worst"], r["ref_grad_f64_rel_err_median"],
               r["ref_grad_f64_rel_err_max"]))
    return head + note + " " + body


def record_audit(r, algo):
    """Append the audit to gpurun_out/audit.json (independent of pytest's output capture) and return its one-line form."""
    import json
    from segmi import ops
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "audit.json")
        doc = json.load(open(path)) if os.path.exists(path) else {}
        doc["%s/%s/%s" % (r["config"], ops.get_conv_math(), algo)] = {k: v for k, v in r.items() if isinstance(v, (int, float, str, bool))}
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    return audit_line(r, algo)


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4", "cfg5"], ids=lambda n: "%s-batch%d" % (n, FIXTURE_BATCH[n]))
def test_fullsize_step_matches_reference_golden(cuda, name, conv_algorithm):
    r = run_fullsize_audit(name, cuda)
    print("\n" + record_audit(r, conv_algorithm or "default"))
    assert r["batch"] == FIXTURE_BATCH[name]
    assert r["max_abs_dlogit"] <= 1e-3 * r["logit_absmax"], r
    # every config: the HIP logits' largest distance from the fp64 oracle against the reference fp32 run's own (a max statistic over the
    # strided sample).  Measured: 1.01-1.45x on the default arithmetic (all four configs, both algorithms), 1.08-2.02x under
    # SEGMI_CONV_MATH=bf16x3; a kernel bug (a dropped tap, a lost partial) lands orders of magnitude above either
    assert r["hip_err_f64"] <= 2.5 * r["ref_err_f64"], r
    assert r["mismatches_outside_margin"] == 0, r
    if "max_abs_daux" in r:
        assert r["max_abs_daux"] <= 1e-3 * r["aux_absmax"], r
    assert abs(r["loss"] - r["loss_ref"]) < 1e-4, r
    assert r["grad_norm_rel_err_median"] <= 1e-2 and r["grad_norm_rel_err_max"] <= 3e-2, r
    assert r["grad_f64_rel_err_median"] <= 1.5 * r["ref_grad_f64_rel_err_median"], r
    assert r["grad_f64_rel_err_max"] <= 2.0 * r["ref_grad_f64_rel_err_max"], r
    assert r["running_ok"], r
