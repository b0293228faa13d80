"""Probe (CPU only): would Winograd F(2x2, 3x3) in fp32 keep the parity bars of the stride-1 3x3 convolutions?

The 3x3 stride-1 layers are 27 of the 64 conv milliseconds of a cfg2 step (forward + data gradient).  F(2x2, 3x3) needs 2.25x
fewer multiplications in the SAME arithmetic type (transform constants 0, +-1, +-1/2: no rounding in the filter/input transforms
beyond ordinary fp32 additions), so it is an algorithm change, not a precision change — but the summation is re-associated, and
the end-to-end bars (1e-3 of max|logit|, the UNet gradient check, the cfg3 noise floor) are tight.  This emulates the exact
dataflow in torch fp32 on the CPU (input transform B^T d B per 4x4 tile, filter transform G g G^T, 16 channel contractions,
output transform A^T m A; dilation d handled as d*d dense sub-grids) and measures, against fp64:

  (a) single layers at the reduction lengths of the path
  (b) UNet cfg1 frozen-BN parameter gradients (the tightest existing check: 1e-3 per tensor against torch-CPU fp32)
  (c) PSPNet-R50 logits with batch statistics

for direct fp32 (torch) and the Winograd dataflow.   python tools/probes/winograd_numerics.py [a] [b] [c]
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd"))
import torch
import torch.nn.functional as F

_conv2d = F.conv2d
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def _wino_dense(x, w):
    """3x3, stride 1, padding 1, fp32, as F(2x2,3x3)."""
    N, C, H, W = x.shape
    K = w.shape[0]
    th, tw = (H + 1) // 2, (W + 1) // 2
    xp = F.pad(x, (1, 2 * tw + 1 - W, 1, 2 * th + 1 - H))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                   # [N, C, th, tw, 4, 4]
    V = torch.einsum("ij,nctujk,lk->nctuil", BT, d, BT)
    U = torch.einsum("ij,kcjl,ml->kcim", G, w, G)                            # [K, C, 4, 4]
    # 16 independent contractions over c (the GEMMs of the real kernel)
    M = torch.einsum("kcil,nctuil->nktuil", U, V)
    Y = torch.einsum("ij,nktujl,ml->nktuim", AT, M, AT)                      # [N, K, th, tw, 2, 2]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, 2 * th, 2 * tw)
    return Y[:, :, :H, :W]


def wino_conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    st = stride if isinstance(stride, int) else stride[0]
    pd = padding if isinstance(padding, int) else padding[0]
    dl = dilation if isinstance(dilation, int) else dilation[0]
    if x.dtype != torch.float32 or groups != 1 or w.shape[2:] != (3, 3) or st != 1 or pd != dl:
        return _conv2d(x, w, bias, stride, padding, dilation, groups)
    if dl == 1:
        y = _wino_dense(x, w)
    else:
        # dilation d = d*d dense problems on the sub-grids (h % d, w % d); assembled out of place so that autograd sees plain adds
        y = torch.zeros(x.shape[0], w.shape[0], x.shape[2], x.shape[3], dtype=x.dtype)
        for i in range(dl):
            for j in range(dl):
                part = torch.zeros_like(y)
                part[:, :, i::dl, j::dl] = _wino_dense(x[:, :, i::dl, j::dl], w)
                y = y + part
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return y


def rel_max(a, ref):
    return ((a.double() - ref).abs().max() / ref.abs().max()).item()


def l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def part_a():
    print("(a) single layers: max|err| / max|ref| against an fp64 convolution")
    print("    %-34s %12s %12s %8s" % ("layer", "direct fp32", "winograd", "ratio"))
    g = torch.Generator().manual_seed(0)
    for name, N, C, K, H, dl in (("C64->K64 64x64", 2, 64, 64, 64, 1), ("C256->K256 32x32 d2", 2, 256, 256, 32, 2),
                                 ("C512->K512 32x32 d4", 1, 512, 512, 32, 4), ("C2048->K512 32x32", 1, 2048, 512, 32, 1),
                                 ("C1024->K256 33x33", 1, 1024, 256, 33, 1)):
        x = torch.randn(N, C, H, H, generator=g)
        x = torch.relu(x)                                        # post-ReLU activations: non-negative, like the real operands
        w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5
        ref = _conv2d(x.double(), w.double(), None, 1, dl, dl)
        e_d = rel_max(_conv2d(x, w, None, 1, dl, dl), ref)
        e_w = rel_max(wino_conv2d(x, w, None, 1, dl, dl), ref)
        print("    %-34s %12.3e %12.3e %8.2f" % (name, e_d, e_w, e_w / e_d))


def part_b():
    from oracle import losses_ref, pspnet_ref, unet_ref
    from oracle.weights import synth_batch, synth_state_dict
    rec = torch.load(os.path.join(ROOT, "tests", "golden", "unet.pt"), weights_only=False)["s64"]
    sd = synth_state_dict(rec["manifest"], seed=2)
    x, t = synth_batch(2, 3, 256, 256, 2, seed=99)
    grads = {}
    for name, dt, conv in (("f64", torch.float64, _conv2d), ("direct", torch.float32, _conv2d), ("winograd", torch.float32, wino_conv2d)):
        F.conv2d = conv
        try:
            ref = pspnet_ref.clone_state({k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()})
            losses_ref.cross_entropy(unet_ref.unet_forward(ref, x.to(dt), training=True, bn_training=False), t).backward()
            grads[name] = {k: v.grad.double() for k, v in ref.items() if v.grad is not None}
        finally:
            F.conv2d = _conv2d
    print("(b) UNet cfg1 frozen-BN parameter gradients, per-tensor relative L2")
    for a, b in (("direct", "f64"), ("winograd", "f64"), ("winograd", "direct")):
        v = [l2(grads[a][k], grads[b][k]) for k in grads["f64"]]
        print("    %-9s vs %-7s median %.2e  max %.2e" % (a, b, statistics.median(v), max(v)))


def part_c():
    from oracle import pspnet_ref
    from oracle.weights import synth_batch, synth_state_dict
    man = torch.load(os.path.join(ROOT, "tests", "golden", "full_cfg2.pt"), weights_only=False)["manifest"]
    res = []
    for seed in (1, 2, 3):
        sd = synth_state_dict(man, seed=seed)
        x, _ = synth_batch(2, 3, 128, 128, 21, seed=10 + seed)
        outs = {}
        for name, dt, conv in (("f64", torch.float64, _conv2d), ("direct", torch.float32, _conv2d), ("winograd", torch.float32, wino_conv2d)):
            F.conv2d = conv
            try:
                ref = pspnet_ref.clone_state({k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()})
                with torch.no_grad():
                    o, _ = pspnet_ref.pspnet_forward(ref, x.to(dt), training=True, backbone="resnet50")
                outs[name] = o.double()
            finally:
                F.conv2d = _conv2d
        res.append((rel_max(outs["direct"], outs["f64"]), rel_max(outs["winograd"], outs["f64"])))
    print("(c) PSPNet-R50 2x3x128x128, batch statistics: max|dlogit| / max|logit| against fp64, three weight seeds")
    for d, w in res:
        print("    direct fp32 %.3e   winograd %.3e   ratio %.2f" % (d, w, w / d))


if __name__ == "__main__":
    which = sys.argv[1:] or ["a", "b", "c"]
    torch.manual_seed(0)
    for p in which:
        {"a": part_a, "b": part_b, "c": part_c}[p]()
