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

VARIANT = re.compile(r"^(conv_dma_kernel<(64|128), (32|64|128), (2, 2|4, 1), [01], (true|false), (true|false)>"
                     r"|conv_wgrad_dma_kernel<(64|128), (64|128), (true|false), (true|false)> splitk=\d+"
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


@pytest.mark.parametrize("shape,op,bm,bn", [
    ((16, 33, 33, 256, 1024, 1, 1, 1), 0, 64, 128),     # cfg3 layer3 256 -> 1024: 1096 tiles of 128 rows, fifth per-CU round holds 72 -> 64-row tiles
    ((16, 33, 33, 1024, 256, 1, 1, 1), 0, 64, 64),      # 274 tiles: under-filled (the round-4 rule); 546 tiles of 64 x 128 fill 71 % of the 768 slots -> 64 x 64 (round 6)
    ((16, 33, 33, 512, 2048, 1, 1, 1), 1, 64, 128),     # dgrad, Cd = 512: 548 tiles, 36 in the third round
    ((16, 33, 33, 1024, 2048, 1, 1, 1), 0, 128, 128),   # 2192 tiles: the last round is more than half full -> 128 rows stay
    ((8, 64, 64, 512, 2048, 1, 1, 1), 0, 128, 128),     # cfg2 layer4: 4096 tiles = 16 full rounds
    ((8, 64, 64, 256, 256, 1, 1, 1), 1, 128, 128),      # 512 tiles = 2 full rounds (64-row tiles measured 11 % slower here)
    ((8, 32, 32, 728, 728, 1, 1, 1), 0, 64, 128),       # Xception middle flow: 384 tiles, under-filled; 768 tiles of 64 x 128 = every slot
])
def test_tile_height_follows_the_per_cu_round_rule(shape, op, bm, bn):
    """Round 5 (csrc/conv_igemm.hip dma_half_m, profiles/r05_half_m_layers.txt): 64-row tiles for under-filled launches (< 512 tiles
    of 128 rows) and when the last per-CU round of the 128-row tiling is less than half full (tiles mod 256 in (0, 128]); the BN
    statistics epilogue emits one partial per row tile of the tiling actually launched."""
    from segmi import lib
    from segmi.ops import conv_variant
    d = _desc(*shape)
    name = conv_variant(d, op)
    assert name.startswith("conv_dma_kernel<%d, %d, " % (bm, bn)), name
    M, Cd = d.N * d.H * d.W, (d.K if op == 0 else d.C)
    assert (bn == 64) == (bm == 64 and -(-M // 64) * -(-Cd // 128) * 100 <= 75 * 768)   # round 6: quarter tiles below 75 % of the slots
    tiles = -(-M // 128) * -(-Cd // 128)
    want64 = tiles < 512 or 0 < tiles % 256 <= 128
    assert (bm == 64) == want64
    if op == 0:
        assert lib.segmi_conv2d_fwd_stats_parts(d) == -(-M // bm)


wino_descs = st.builds(lambda N, H, W, C, K, dil: _desc(N, H, W, C, K, 3, 1, dil), st.integers(1, 16), st.integers(1, 130), st.integers(1, 130),
                       st.sampled_from([4, 64, 256, 304, 512, 2048]), st.sampled_from([19, 21, 64, 256, 512]),
                       st.sampled_from([1, 2, 4, 6, 12])).filter(lambda d: d is not None)


@settings(max_examples=200, deadline=None)
@given(wino_descs)
def test_winograd_plans_are_consistent(d):
    """Winograd F(2x2,3x3) planners (csrc/conv_winograd.hip): tile count, workspace decomposition of the three passes, the kept
    transformed input, and the batched filter-gradient split, for random 3x3 stride-1 "same" problems incl. dilation > map."""
    import ctypes
    from segmi import lib

    def al(b):
        return (b + 255) & ~255

    assert lib.segmi_conv2d_winograd_ok(d, 0) == 1
    th, tw = (-(-d.H // d.dil) + 1) // 2, (-(-d.W // d.dil) + 1) // 2
    T = d.N * d.dil * d.dil * th * tw
    assert lib.segmi_conv2d_winograd_tiles(d) == T and 4 * T >= d.N * d.H * d.W          # the tiles cover every output pixel
    Tp, Kp = (T + 31) & ~31, (d.K + 3) & ~3
    assert lib.segmi_conv2d_winograd_v_bytes(d) == 16 * Tp * d.C * 4
    assert lib.segmi_conv2d_winograd_workspace(d, 0) == al(16 * d.K * d.C * 4) + al(16 * Tp * d.C * 4) + al(16 * T * Kp * 4)
    if lib.segmi_conv2d_winograd_ok(d, 1) == 1:
        TpK = Tp                                       # dgrad: the transformed input is dy (Kp channels), the product has C columns
        assert lib.segmi_conv2d_winograd_workspace(d, 1) == al(16 * d.C * Kp * 4) + al(16 * TpK * Kp * 4) + al(16 * T * d.C * 4)
    assert lib.segmi_conv2d_winograd_wgrad_ok(d) == 1
    buf = ctypes.create_string_buffer(128)
    assert lib.segmi_conv2d_winograd_wgrad_variant(d, buf, 128) == 0
    m = re.match(r"^winograd_f2x2_3x3 wgrad: 16 x conv_wgrad_dma_kernel<(64|128), (64|128), true, true> splitk=(\d+)$", buf.value.decode())
    assert m, buf.value
    ns = int(m.group(3))
    assert 1 <= ns <= 512 and ns <= Tp // 32                                   # every split owns at least one 32-row chunk of tiles
    assert lib.segmi_conv2d_winograd_wgrad_workspace(d) == al(16 * Tp * d.C * 4) + al(16 * Tp * Kp * 4) + al(ns * 16 * d.K * d.C * 4)


@settings(max_examples=100, deadline=None)
@given(st.integers(1, 1 << 22), st.sampled_from([2, 3, 19, 21, 150, 255]))
def test_lovasz_workspace_layout(rows, C):
    """segmi_lovasz_workspace covers the worst case of the tail pruning (every element survives): two 64-bit key buffers +
    the survivor counts per (class, 256-pixel unit) + scan scratch + per-tile digit histograms [C][ceil(rows/4096)][256] of the
    segmented sort, 256-byte aligned pieces; it does not depend on the prune switch, which is validated."""
    from segmi import lib
    assert lib.segmi_lovasz_set_prune(7) != 0
    try:
        assert lib.segmi_lovasz_set_prune(0) == 0
        full = lib.segmi_lovasz_workspace(rows, C)
    finally:
        assert lib.segmi_lovasz_set_prune(1) == 0
    seg = lib.segmi_lovasz_workspace(rows, C)
    assert seg == full
    keys = 2 * ((rows * C * 8 + 255) & ~255)
    hist = C * ((rows + 4095) // 4096) * 256 * 4
    units = C * ((rows + 255) // 256) * 4
    assert seg % 256 == 0
    assert seg >= keys + hist + units
    assert seg - keys - ((hist + 255) & ~255) - ((units + 255) & ~255) < (1 << 20) + C * ((rows + 2047) // 2048) * 12 + 4096
    assert lib.segmi_lovasz_workspace(1 << 24, C) == 0          # fp32-exact rank arithmetic ends at 2^24 pixels, like the reference's cumsums


def test_issued_fraction_of_launches_that_skip_taps():
    """The instrumentation's EXECUTED-FLOP accounting (ops._conv_issued): 1 for pointwise layers and for 3x3 layers whose taps all
    reach the image from every tile; for DeepLab's ASPP branches on the 33x33 map (dilation 6 / 12 / 18) the share of (tile, tap) pairs /
    (chunk, tap) pairs the kernels issue — never below the share of (pixel, tap) pairs inside the image, falling with the dilation."""
    from segmi.ops import _conv_issued
    assert all(_conv_issued(_desc(8, 64, 64, 512, 2048 // 4, 1, 1, 1), op) == 1.0 for op in (0, 1, 2))
    prev = [1.0, 1.0, 1.0]
    for dil in (6, 12, 18):
        d = _desc(16, 33, 33, 2048, 256, 3, 1, dil)
        inside = ((1 + 2 * (33 - dil) / 33.0) / 3.0) ** 2          # (pixel, tap) pairs inside the image / all pairs
        rows_only = (1 + 2 * (33 - dil) / 33.0) / 3.0             # what whole image rows can skip
        for op in (0, 1, 2):
            f = _conv_issued(d, op)
            assert inside < rows_only <= f < prev[op], (dil, op, f, rows_only, prev[op])
            prev[op] = f
