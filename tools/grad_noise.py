#!/usr/bin/env python
"""Where does the HIP path's gradient rounding noise sit?  One training step of a BASELINE-shape config against the fixture's fp64
gradient digests (tests/golden/full_<cfg>.pt): per parameter tensor, the HIP path's relative L2 distance from fp64 next to the
reference fp32's own (the floor), grouped by model stage and by parameter kind, plus the tensors with the largest ratio.

    python tools/grad_noise.py cfg2 [cfg3 ...]        (GPU box)
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-segmentation_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from oracle.weights import synth_batch, synth_state_dict  # noqa: E402


def run(name, dev):
    import models
    import utils.losses as losses_mod
    rec = torch.load(os.path.join(ROOT, "tests", "golden", "full_%s.pt" % name), weights_only=False)
    C, kw, ign = rec["num_classes"], rec["kwargs"], rec["ignore_index"]
    N, _, H, W = rec["input_shape"]
    m = getattr(models, rec["arch"])(C, pretrained=False, **kw)
    m.load_state_dict(synth_state_dict(rec["manifest"], seed=rec["weight_seed"]))
    m.to(dev).train()
    for mod in m.modules():
        if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout2d)):
            mod.eval()
    crit = getattr(losses_mod, rec["loss_name"])(ignore_index=ign)
    x, t = synth_batch(N, 3, H, W, C, ignore_index=ign, seed=rec["batch_seed"])
    out = m(x.to(dev))
    if rec["arch"][:3] == "PSP":
        loss = crit(out[0], t.to(dev)) + 0.4 * crit(out[1], t.to(dev))
    else:
        loss = crit(out, t.to(dev))
    loss.backward()
    named = dict(m.named_parameters())
    top = max(v["norm"] for v in rec["grads_f64"].values())
    rows = []
    for k, d64 in rec["grads_f64"].items():
        if d64["norm"] <= 1e-5 * top:
            continue
        g = named[k].grad.detach().reshape(-1)
        step = max(1, g.numel() // 64)
        got = torch.cat([g[::step][:64], g[:8]]).cpu().double()
        r64 = torch.cat([d64["sample"], d64["head"]]).double()
        hip = ((got - r64).norm() / r64.norm()).item()
        ref = rec["ref_grad_err_f64"]["per_tensor"][k]
        kind = "bn.weight/bias" if named[k].dim() == 1 else ("conv 1x1" if named[k].shape[-1] == 1 else "conv kxk")
        rows.append((k, kind, hip, ref))
    print("== %s: %d tensors; HIP median %.3e max %.3e | reference fp32 median %.3e max %.3e" % (
        name, len(rows), statistics.median(r[2] for r in rows), max(r[2] for r in rows), statistics.median(r[3] for r in rows), max(r[3] for r in rows)))
    groups = {}
    for k, kind, hip, ref in rows:
        stage = ".".join(k.split(".")[:2]) if k.startswith(("backbone", "master_branch", "auxiliary", "decoder", "ASSP")) else k.split(".")[0]
        for key in (stage, kind):
            groups.setdefault(key, []).append((hip, ref))
    print("   %-28s %5s %12s %12s %7s" % ("group", "n", "HIP median", "ref median", "ratio"))
    for key, v in sorted(groups.items(), key=lambda kv: -len(kv[1])):
        h, r = statistics.median(a for a, _ in v), statistics.median(b for _, b in v)
        print("   %-28s %5d %12.3e %12.3e %7.2f" % (key, len(v), h, r, h / r))
    print("   largest HIP / reference ratios:")
    for k, kind, hip, ref in sorted(rows, key=lambda r: -r[2] / r[3])[:10]:
        print("     %-52s %-14s HIP %.3e ref %.3e ratio %.2f" % (k, kind, hip, ref, hip / ref))


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for name in (sys.argv[1:] or ["cfg2"]):
        run(name, dev)
        torch.cuda.empty_cache()
