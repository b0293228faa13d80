"""GPU: degenerate inputs of the hot-path operators, held to what the reference's operators (torch CPU) do with them —
everything ignored, a single class, one pixel, one image, pads larger than the map, resizing to the same size."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_losses_with_every_pixel_ignored(cuda):
    """nn.CrossEntropyLoss(mean) over zero valid pixels is NaN (0/0) in torch and here; Focal averages zeros over all pixels;
    the reference's lovasz_softmax returns 0 for an empty selection (utils/lovasz_losses.py:176-178); gradients are zero / NaN
    exactly where torch's are."""
    import utils.losses as L
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 5, 6, 7, generator=g)
    t = torch.full((2, 6, 7), 255, dtype=torch.int64)
    xd = x.to(cuda).requires_grad_(True)
    ce = L.CrossEntropyLoss2d(ignore_index=255)(xd, t.to(cuda))
    assert math.isnan(ce.item()) and math.isnan(F.cross_entropy(x, t, ignore_index=255).item())
    fo = L.FocalLoss(ignore_index=255)(xd, t.to(cuda))
    assert fo.item() == 0.0
    fo.backward()
    assert float(xd.grad.abs().max()) == 0.0
    lv = L.LovaszSoftmax(ignore_index=255)(x.to(cuda).requires_grad_(True), t.to(cuda))
    assert lv.item() == 0.0


def test_losses_with_a_single_class_and_a_single_pixel(cuda):
    import utils.losses as L
    from oracle import losses_ref
    g = torch.Generator().manual_seed(2)
    for shape, cls in (((1, 4, 1, 1), 2), ((2, 3, 5, 4), 1)):
        x = torch.randn(shape, generator=g)
        t = torch.full((shape[0], shape[2], shape[3]), cls, dtype=torch.int64)
        for name, ref in (("CrossEntropyLoss2d", losses_ref.cross_entropy), ("DiceLoss", losses_ref.dice), ("FocalLoss", losses_ref.focal),
                          ("LovaszSoftmax", losses_ref.lovasz_softmax)):
            xr = x.clone().requires_grad_(True)
            lr = ref(xr, t.clone(), 255)
            lr.backward()
            xd = x.to(cuda).requires_grad_(True)
            ld = getattr(L, name)(ignore_index=255)(xd, t.clone().to(cuda))
            ld.backward()
            assert abs(ld.item() - lr.item()) <= 1e-5 * abs(lr.item()) + 1e-6, (name, shape, ld.item(), lr.item())
            assert torch.allclose(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-7), (name, shape)


def test_batch_norm_needs_more_than_one_value_per_channel(cuda):
    from segmi import nn as snn
    bn = snn.BatchNorm2d(8).to(cuda).train()
    with pytest.raises(ValueError):
        bn(torch.randn(1, 8, 1, 1, device=cuda))
    ref = torch.nn.BatchNorm2d(8).train()
    with pytest.raises(ValueError):
        ref(torch.randn(1, 8, 1, 1))
    x = torch.randn(2, 8, 1, 1)                      # two values per channel: fine, and equal to torch
    ref.eval()
    bn.eval()
    assert torch.allclose(bn(x.to(cuda)).cpu(), ref(x), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", [(1, 4, 1, 1, 8, 3, 1, 1, 1), (1, 8, 2, 3, 4, 3, 1, 4, 4), (3, 4, 1, 7, 4, 1, 1, 0, 1), (1, 12, 5, 5, 20, 3, 2, 1, 1)])
def test_convolution_on_degenerate_maps(cuda, case):
    """One-pixel maps, padding / dilation larger than the map (every tap but the centre falls outside), batch 1."""
    from segmi import ops
    N, C, H, W, K, R, stride, pad, dil = case
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, R, R, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad, dil)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.to(cuda).requires_grad_(True)
    wd = w.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yd = ops.conv2d(xd, wd, None, stride, pad, dil)
    yd.backward(gy.to(cuda))
    for a, b in ((yd, yr), (xd.grad, xr.grad), (wd.grad, wr.grad)):
        assert a.shape == b.shape
        assert (a.detach().cpu() - b.detach()).abs().max().item() <= 1e-4 * b.abs().max().item() + 1e-7


def test_resize_and_pool_identities(cuda):
    from segmi import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 8, 5, 7, generator=g)
    for ac in (False, True):
        y = ops.interpolate_bilinear(x.to(cuda), (5, 7), ac)               # same size: the identity
        assert torch.allclose(y.cpu(), x, rtol=0, atol=1e-6)
        one = ops.interpolate_bilinear(x[:, :, :1, :1].contiguous().to(cuda), (4, 6), ac)      # 1x1 source: a constant map
        assert torch.allclose(one.cpu(), x[:, :, :1, :1].expand(2, 8, 4, 6), rtol=0, atol=1e-6)
    assert torch.allclose(ops.adaptive_avg_pool2d(x.to(cuda), (5, 7)).cpu(), x, rtol=0, atol=1e-6)
    p = ops.max_pool2d(x.to(cuda), 2, 2, 0, ceil_mode=True)
    assert torch.equal(p.cpu(), F.max_pool2d(x, 2, 2, 0, ceil_mode=True))


def test_metrics_without_labeled_pixels(cuda):
    from utils.metrics import SegMetrics
    m = SegMetrics(4, cuda)
    m.update(torch.randn(1, 4, 3, 3, device=cuda), torch.full((1, 3, 3), 255, dtype=torch.int64, device=cuda))
    correct, labeled, inter, union = m.counts()
    assert correct == 0 and labeled == 0 and inter.sum() == 0 and union.sum() == 0
    s = m.summary()                                   # 0 / eps like the reference (utils/metrics.py, np.spacing(1))
    assert s["Pixel_Accuracy"] == 0 and s["Mean_IoU"] == 0
