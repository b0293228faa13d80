"""CPU: properties of the host-side planners behind the C ABI (no launches): split-K plans of the weight-gradient and
small-output forward kernels (csrc/conv_igemm.hip: plan_wgrad, plan_fwd_split) and the dispatcher's variant choice, over
randomly drawn convolution descriptors (hypothesis), plus the BASELINE configs' extreme shapes."""
import re

import pytest

hypothesis = pytest.importorskip("hypothesis")          # a missing optional package must not break collection of the suite
from hypothesis import given, settings, strategies as st  # noqa: E402


def _desc(N, H, W, C, K, R, stride, dil):
    from segmi._lib import ConvDesc
    from segmi.ops import conv_out_size
    pad = dil * (R // 2)
    P, Q = conv_out_size(H, R, stride, pad, dil), conv_out_size(W, R, stride, pad, dil)
    if P <= 0 or Q <= 0:
        return None
    C4, K4 = (C + 3) & ~3, (K + 3) & ~3
    return ConvDesc(N, H, W, C4, K, R, R, P, Q, stride, pad, dil, C4, K4)


descs = st.builds(_desc, st.integers(1, 16), st.integers(1, 200), st.integers(1, 200), st.sampled_from([3, 4, 21, 48, 64, 150, 256, 304, 728, 2048]),
                  st.sampled_from([2, 19, 21, 32, 48, 64, 150, 256, 512, 728]), st.sampled_from([1, 3, 7]), st.sampled_from([1, 2]),
                  st.sampled_from([1, 2, 4, 6, 12, 18])).filter(lambda d: d is not None)

VARIANT = re.compile(r"^(conv_dma_kernel<(64|128), (32|64|128), (2, 2|4, 1), [01], (true|false), [01]>"
                     r"|conv_wgrad_dma_kernel<(64|128), (64|128), (true|false), [01]> splitk=\d+"
                     r"|conv_gather_kernel<128, (32|64|128), (16|32), (2, 2|4, 1), [01]>"
                     r"|conv_wgrad_kernel<(64|128), (64|128), 32, 2, 2> splitk=\d+)$")


@settings(max_examples=300, deadline=None)
@given(descs)
def test_plans_are_consistent(d):
    from segmi import lib
    from segmi.ops import conv_variant
    M = d.N * d.P * d.Q
    dw_bytes = d.K * d.R * d.S * d.C * 4
    ws = lib.segmi_conv2d_wgrad_workspace(d)
    assert ws % dw_bytes == 0
    nsplit = ws // dw_bytes                       # 0 = unsplit (partials would be the result itself)
    assert nsplit != 1 and nsplit <= 512
    if nsplit:
        assert (M + 31) // 32 >= nsplit           # every split owns at least one 32-pixel chunk
    fws = lib.segmi_conv2d_fwd_workspace(d)
    assert fws % (M * d.ldy * 4) == 0
    ks = fws // (M * d.ldy * 4)
    assert ks == 0 or 2 <= ks <= 64
    if ks:                                        # only problems with very few output tiles are split in the forward pass
        bn = 128 if d.K > 64 else (64 if d.K > 32 else 32)
        assert ((M + 127) // 128) * ((d.K + bn - 1) // bn) <= 32
        assert ((d.C + 31) // 32) * d.R * d.S >= 16
    names = [conv_variant(d, op) for op in (0, 1, 2)]
    for n in names:
        assert VARIANT.match(n), n
    assert ("splitk=%d" % max(nsplit, 1)) in names[2]


@pytest.mark.parametrize("shape", [
    (8, 64, 64, 4096, 512, 3, 1, 1),      # cfg2 PSP bottleneck
    (4, 97, 97, 4096, 512, 3, 1, 1),      # cfg4 (769^2 input -> 97^2 maps)
    (16, 129, 129, 304, 256, 3, 1, 1),    # cfg3 decoder
    (8, 128, 128, 256, 150, 1, 1, 1),     # cfg5 classifier
    (2, 256, 256, 3, 32, 3, 1, 1),        # cfg1 first layer (C padded to 4)
    (16, 513, 513, 3, 64, 7, 2, 1),       # cfg3 stem
])
def test_baseline_shapes_take_the_lds_dma_kernels(shape):
    from segmi.ops import conv_variant
    d = _desc(*shape)
    assert conv_variant(d, 0).startswith("conv_dma_kernel<")
    assert conv_variant(d, 1).startswith("conv_dma_kernel<")
    assert conv_variant(d, 2).startswith("conv_wgrad_dma_kernel<")
