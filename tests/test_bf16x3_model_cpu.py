"""CPU: the three-plane bf16 split behind SEGMI_CONV_MATH_BF16X3 (csrc/conv_igemm.hip: split_pair, mma_bf16x3) is exact
as a decomposition and fp32-accurate as a product — checked on a numpy model of the kernel's arithmetic
(oracle/bf16x3_model.py).  The GPU counterpart (tests/test_conv_bf16x3_gpu.py) holds the kernels to the same bound."""
import numpy as np

from oracle import bf16x3_model as M


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n).astype(np.float32)
    scale = np.exp2(rng.integers(-40, 40, n)).astype(np.float32)      # wide dynamic range: gradients are tiny, activations not
    return x * scale


def test_split_is_exact_and_planes_are_bf16():
    x = np.concatenate([_samples(200000, 0), np.float32([0.0, -0.0, 1.0, -1.0, 1.0 + 2 ** -8, 1.0 + 2 ** -9, 3.0e38 * 0.5, 1e-30])])
    h, m, l = M.split3(x)
    for p in (h, m, l):
        assert np.all((p.view(np.uint32) & 0xFFFF) == 0)              # representable in bf16
    s = (h.astype(np.float64) + m.astype(np.float64)) + l.astype(np.float64)
    assert np.array_equal(s, x.astype(np.float64))                    # x == h + m + l, exactly
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8)         # |m| <= half an 8-bit ulp of x
    assert np.all(np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_dropped_plane_products_are_below_one_fp32_rounding():
    a, b = _samples(100000, 1), _samples(100000, 2)
    ah, am, al = M.split3(a)
    bh, bm, bl = M.split3(b)
    kept = sum(p.astype(np.float64) * q.astype(np.float64) for p, q in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)))
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(kept - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -23                                    # worst case 2*2^-8*2^-16 (+ l*l')
    assert rel.mean() <= 2.0 ** -26                                   # typical: well under an fp32 rounding (2^-24)
    assert abs(np.mean((kept - exact) / np.abs(exact))) <= 2.0 ** -29  # and unbiased (round-to-nearest planes)


def test_dot_products_match_fp64_as_well_as_the_fp32_chain_does():
    rng = np.random.default_rng(3)
    for K in (64, 576, 4608):                                         # 1x1 C=64 ... 3x3 C=512 reductions
        a = rng.standard_normal((256, K)).astype(np.float32)
        b = rng.standard_normal((256, K)).astype(np.float32)
        ref = (a.astype(np.float64) * b.astype(np.float64)).sum(-1)
        scale = (np.abs(a).astype(np.float64) * np.abs(b)).sum(-1)
        e3 = np.abs(M.dot_bf16x3(a, b) - ref) / scale
        e1 = np.abs(M.dot_f32_chain(a, b) - ref) / scale
        # same order of magnitude as the fp32 FMA chain (it has 6K/16 accumulator roundings against the chain's K)
        assert e3.max() <= 4 * e1.max() + 2.0 ** -24, (K, e3.max(), e1.max())
        assert np.sqrt((e3 ** 2).mean()) <= 2 * np.sqrt((e1 ** 2).mean()) + 2.0 ** -26, (K,)
    # all-positive operands (post-ReLU activations x positive weights): no cancellation to hide a bias
    a = np.abs(rng.standard_normal((64, 4096))).astype(np.float32)
    b = np.abs(rng.standard_normal((64, 4096))).astype(np.float32)
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(-1)
    e3 = np.abs(M.dot_bf16x3(a, b) - ref) / ref
    e1 = np.abs(M.dot_f32_chain(a, b) - ref) / ref
    assert e3.max() <= 2 * e1.max() and abs(np.mean((M.dot_bf16x3(a, b) - ref) / ref)) <= 2.0 ** -22, (e3.max(), e1.max())


def test_pspnet_logits_under_the_bf16x3_model_are_as_close_to_fp64_as_fp32_is(monkeypatch):
    """Network-scale evidence without a GPU: PSPNet-R50 forward (oracle/pspnet_ref.py) with every convolution computed the
    bf16x3 way — operands split into three bf16 planes (torch's RNE conversion, as v_cvt_pk_bf16_f32), the six kept plane
    products summed exactly (fp64), the result rounded to fp32 — against the same network in fp64 and in plain fp32.
    The dropped plane products must not move the logits more than fp32 rounding already does."""
    import torch
    import torch.nn.functional as F
    import models
    from oracle import pspnet_ref

    torch.manual_seed(0)
    sd32 = {k: v.detach().clone().contiguous() for k, v in models.PSPNet(5, backbone="resnet50", pretrained=False).state_dict().items()}
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 3, 64, 64, generator=g)

    def split3(t):
        h = t.to(torch.bfloat16).to(torch.float32)
        r = t - h
        m = r.to(torch.bfloat16).to(torch.float32)
        l = (r - m).to(torch.bfloat16).to(torch.float32)
        assert torch.equal((h.double() + m.double()) + l.double(), t.double())
        return h, m, l

    def conv_x3(sd, key, inp, stride=1, pad=0, dil=1):
        w, b = sd[key + ".weight"], sd.get(key + ".bias")
        (xh, xm, xl), (wh, wm, wl) = split3(inp), split3(w)
        acc = None
        for pa, pb in ((xl, wh), (xh, wl), (xm, wm), (xm, wh), (xh, wm), (xh, wh)):
            y = F.conv2d(pa.double(), pb.double(), None, stride, pad, dil)
            acc = y if acc is None else acc + y
        y = acc.to(torch.float32)
        return y if b is None else y + b.view(1, -1, 1, 1)

    with torch.no_grad():
        ref = pspnet_ref.pspnet_forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd32.items()}, x.double(),
                                        training=False)
        f32 = pspnet_ref.pspnet_forward(sd32, x, training=False)
        monkeypatch.setattr(pspnet_ref, "_conv", conv_x3)
        x3 = pspnet_ref.pspnet_forward(sd32, x, training=False)
    scale = ref.abs().max().item()
    e32 = (f32.double() - ref).abs().max().item() / scale
    ex3 = (x3.double() - ref).abs().max().item() / scale
    assert e32 < 1e-4 and ex3 < 1e-4, (e32, ex3)                # both at fp32 level through 50+ layers ...
    assert ex3 <= 2.0 * e32 + 1e-7, (ex3, e32)                  # ... and the split costs nothing beyond fp32 rounding
    assert int((x3.argmax(1) != f32.argmax(1)).sum()) <= 2      # masks agree (flips only possible at exact near-ties)
