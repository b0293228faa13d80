"""GPU: the hot-path kernels at BASELINE.json's FULL sizes (cfg2: 8x512x512 PSPNet-R50 shapes; cfg5: 150 classes, 8x512x512
logits), where the torch-CPU oracle takes minutes to hours, checked through size-independent properties:

  * convolution: the adjoint identities  <conv(x, w), g> == <x, dgrad(g, w)> == <w, wgrad(x, g)>  tie the three implicit-GEMM
    kernels to each other (each is the transpose of the same bilinear form), plus linearity in x and a strided-sample check of
    output pixels against a direct fp64 evaluation of the definition;
  * batch norm (batch statistics): output moments are (beta, gamma^2) per channel; sum over pixels of dx is 0;
  * cross entropy: every pixel's gradient sums to 0 over classes, ignored pixels get exactly 0, loss == mean(lse - x_t);
  * Lovasz-Softmax: 0 <= loss <= 1, invariance under a permutation of the pixels, zero-sum softmax gradients, ignored -> 0;
  * a whole cfg2 training step is bit-reproducible run to run (deterministic split-K, no atomics on the fp32 path).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dot(a, b):
    return float((a.detach().double() * b.detach().double()).sum())


# (N, C, H, W, K, R, stride, pad, dil): the PSP bottleneck, a dilated layer4 3x3, the strided layer2 3x3, a 1x1 expansion, stem
FULL_CONVS = [(8, 4096, 64, 64, 512, 3, 1, 1, 1), (8, 512, 64, 64, 512, 3, 1, 4, 4), (8, 128, 128, 128, 128, 3, 2, 1, 1),
              (8, 256, 64, 64, 1024, 1, 1, 0, 1), (8, 3, 512, 512, 64, 3, 2, 1, 1),
              # cfg3: DeepLab's ASPP branch at dilation 18 on the 33x33 map (taps / pixel chunks that miss the image are skipped per tile)
              (16, 2048, 33, 33, 256, 3, 1, 18, 18)]


@pytest.mark.parametrize("case", FULL_CONVS)
def test_conv_adjoint_identities_and_definition_at_full_size(cuda, case):
    from segmi import ops
    N, C, H, W, K, R, stride, pad, dil = case
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(N, C, H, W, device=cuda, generator=g).requires_grad_(True)
    w = (torch.randn(K, C, R, R, device=cuda, generator=g) / (C * R * R) ** 0.5).requires_grad_(True)
    y = ops.conv2d(x, w, None, stride, pad, dil)
    gy = torch.randn(y.shape, device=cuda, generator=g)
    dx, dw = torch.autograd.grad(y, (x, w), gy)
    a, b, c = _dot(y, gy), _dot(x, dx), _dot(w, dw)
    scale = (float(y.double().pow(2).sum()) * float(gy.double().pow(2).sum())) ** 0.5
    assert abs(a - b) <= 2e-6 * scale and abs(a - c) <= 2e-6 * scale, (a, b, c, scale)
    # linearity in the input
    x2 = torch.randn(N, C, H, W, device=cuda, generator=g)
    with torch.no_grad():
        lhs = ops.conv2d(0.5 * x.detach() - 2.0 * x2, w.detach(), None, stride, pad, dil)
        rhs = 0.5 * y.detach() - 2.0 * ops.conv2d(x2, w.detach(), None, stride, pad, dil)
    assert (lhs - rhs).abs().max().item() <= 1e-4 * rhs.abs().max().item()
    # a strided sample of output pixels against the definition in fp64 (incl. corners: zero padding, dilation)
    P, Q = y.shape[2], y.shape[3]
    xs, ws = x.detach().double(), w.detach().double()
    pts = [(0, 0, 0), (N - 1, P - 1, Q - 1), (1, P // 2, 0), (N // 2, 1, Q - 2), (2, P - 1, Q // 3)]
    # the image rows / columns where a filter row or column enters and leaves the image (the edges of the skipped tap ranges)
    pts += [(e % N, e, (7 * e) % Q) for e in (pad - 1, pad, P - 1 - pad, P - pad) if 0 <= e < P and pad > 1]
    pts += [((e + 1) % N, (5 * e) % P, e) for e in (pad - 1, pad, Q - 1 - pad, Q - pad) if 0 <= e < Q and pad > 1]
    for (n, p, q) in pts:
        acc = torch.zeros(K, dtype=torch.float64, device=cuda)
        for r in range(R):
            for s in range(R):
                h, ww = p * stride - pad + r * dil, q * stride - pad + s * dil
                if 0 <= h < H and 0 <= ww < W:
                    acc += ws[:, :, r, s] @ xs[n, :, h, ww]
        got = y.detach()[n, :, p, q].double()
        assert (got - acc).abs().max().item() <= 1e-4 * acc.abs().max().item() + 1e-6, (n, p, q)


def test_batchnorm_moments_and_zero_sum_gradient_at_full_size(cuda):
    from segmi import ops
    g = torch.Generator(device="cuda").manual_seed(2)
    x = (torch.randn(8, 256, 128, 128, device=cuda, generator=g) * 3 + 1.5).requires_grad_(True)   # layer1 output size of cfg2
    gamma = (torch.rand(256, device=cuda, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(256, device=cuda, generator=g).requires_grad_(True)
    rm, rv = torch.zeros(256, device=cuda), torch.ones(256, device=cuda)
    y = ops.batch_norm_act(x, gamma, beta, rm, rv, None, training=True, momentum=0.1, eps=1e-5, relu=False)
    yd = y.detach().double()
    assert (yd.mean((0, 2, 3)) - beta.detach().double()).abs().max().item() < 1e-4
    assert (yd.var((0, 2, 3), unbiased=False) - gamma.detach().double() ** 2).abs().max().item() < 2e-3
    xd = x.detach().double()
    assert torch.allclose(rm.double(), 0.1 * xd.mean((0, 2, 3)), rtol=1e-4, atol=1e-5)
    gy = torch.randn(y.shape, device=cuda, generator=g)
    dx, dg, db = torch.autograd.grad(y, (x, gamma, beta), gy)
    n = x.numel() // 256
    assert (dx.double().sum((0, 2, 3)).abs() / n).max().item() < 1e-6            # projection property of BN backward
    assert torch.allclose(db.double(), gy.double().sum((0, 2, 3)), rtol=1e-4, atol=1e-2)


def test_cross_entropy_properties_at_full_size(cuda):
    from segmi import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.randn(8, 21, 512, 512, device=cuda, generator=g) * 2).requires_grad_(True)          # cfg2 logits
    t = torch.randint(0, 21, (8, 512, 512), device=cuda, generator=g)
    t[:, :25, :] = 255
    loss = ops.cross_entropy(x, t, 255)
    (dl,) = torch.autograd.grad(loss, x)
    assert dl.sum(1).abs().max().item() < 1e-9
    assert float(dl[:, :, :25, :].abs().max()) == 0.0
    xd = x.detach().double()
    valid = t != 255
    ref = (torch.logsumexp(xd, 1) - xd.gather(1, t.clamp(0, 20).unsqueeze(1)).squeeze(1))[valid].mean()
    assert abs(loss.item() - ref.item()) < 1e-5


def test_lovasz_properties_at_full_size(cuda):
    import utils.losses as L
    g = torch.Generator(device="cuda").manual_seed(4)
    N, C, H, W = 8, 150, 512, 512                                                                 # cfg5 logits (1.26 GB)
    x = (torch.randn(N, C, H, W, device=cuda, generator=g)).requires_grad_(True)
    t = torch.randint(0, C, (N, H // 16, W // 16), device=cuda, generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()
    t[:, :25, :] = -1
    crit = L.LovaszSoftmax(ignore_index=-1)
    loss = crit(x, t)
    (dl,) = torch.autograd.grad(loss, x)
    assert 0.0 <= loss.item() <= 1.0
    assert dl.sum(1).abs().max().item() < 1e-9 and float(dl[:, :, :25, :].abs().max()) == 0.0
    # the loss is a function of the multiset of (pixel probabilities, label): permute pixels within the batch
    perm = torch.randperm(N * H * W, device=cuda, generator=g)
    xp = x.detach().permute(0, 2, 3, 1).reshape(-1, C)[perm].reshape(N, H, W, C).permute(0, 3, 1, 2).contiguous()
    tp = t.reshape(-1)[perm].reshape(N, H, W).contiguous()
    assert abs(crit(xp, tp).item() - loss.item()) < 1e-5


@pytest.mark.parametrize("boost", [0.0, 5.0])
def test_lovasz_tail_pruning_equals_the_full_sort_at_full_size(cuda, boost):
    """2 M pixels x 19 classes: class segments of up to 2 M keys — 512 sort tiles and 1024 scan chunks per class, walked by capped
    grids (64 workgroups per class and pass; 128 for the Jaccard pass) — pruned and full sort agree BIT FOR BIT in loss and gradient,
    on random-init-like and on confident logits (more survivors)."""
    import utils.losses as L
    from segmi import lib, ops
    g = torch.Generator(device="cuda").manual_seed(6)
    N, C, H, W = 8, 19, 512, 512
    x = torch.randn(N, C, H, W, device=cuda, generator=g) * 2
    t = torch.randint(0, C - 2, (N, H, W), device=cuda, generator=g)
    t[:, :9, :] = 255
    if boost:
        hit = (torch.rand(N, H, W, device=cuda, generator=g) < 0.8) & (t != 255)
        x.scatter_add_(1, t.clamp(0, C - 1).unsqueeze(1), hit.float().unsqueeze(1) * boost)
    crit = L.LovaszSoftmax(ignore_index=255)
    res = []
    try:
        for prune in (1, 0):
            assert lib.segmi_lovasz_set_prune(prune) == 0
            xd = x.detach().clone().requires_grad_(True)
            loss = crit(xd, t)
            (dl,) = torch.autograd.grad(loss, xd)
            res.append((loss.detach().clone(), dl, ops.lovasz_last_stats()))
    finally:
        lib.segmi_lovasz_set_prune(1)
    (l1, d1, (k1, f1)), (l0, d0, (k0, f0)) = res
    assert torch.equal(l1, l0) and torch.equal(d1, d0), ((l1 - l0).item(), (d1 - d0).abs().max().item())
    assert k0 == f0 == f1 and k1 < f1 and torch.isfinite(d1).all()
    print("survivors %d of %d (%.2f %%)" % (k1, f1, 100.0 * k1 / f1))


def test_cfg2_training_step_is_bit_reproducible(cuda):
    import models
    from utils.losses import CrossEntropyLoss2d

    def run():
        torch.manual_seed(0)
        m = models.PSPNet(21, backbone="resnet50", pretrained=False).to(cuda).train()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(8, 3, 512, 512, generator=g).to(cuda)
        t = torch.randint(0, 21, (8, 512, 512), generator=g).to(cuda)
        crit = CrossEntropyLoss2d(ignore_index=255)
        torch.manual_seed(123)                      # dropout seeds are drawn from torch's CPU generator
        out, aux = m(x)
        loss = crit(out, t) + 0.4 * crit(aux, t)
        loss.backward()
        return loss.detach().clone(), out.detach().clone(), m.layer1[0].conv1.weight.grad.clone(), m.master_branch[0].bottleneck[0].weight.grad.clone()

    a, b = run(), run()
    for u, v in zip(a, b):
        assert torch.equal(u, v)
