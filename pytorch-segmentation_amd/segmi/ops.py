"""Functional ops + autograd glue over the libsegmi C ABI.

Tensor convention ("NHWC-backed"): an activation is a logical NCHW fp32 CUDA tensor whose memory is
NHWC with a pixel stride ld >= round_up(C, 4):  strides == (H*W*ld, 1, W*ld, ld).  Callers of the
drop-in modules (trainer.py:56-66 of the reference) only look at `.size()`, so the convention is
invisible to them; plain NCHW-contiguous inputs are converted once at the model boundary by our own
transpose kernel.  Channels C..round_up(C,4)-1 of a buffer are always zero.

PyTorch is used here for device memory (caching allocator), streams and autograd bookkeeping only:
every arithmetic op below is a libsegmi kernel, and there is no CPU path — CPU tensors raise.
"""
import ctypes
import os
import re
import weakref

import torch

from ._lib import ConvDesc, FilterTx, SegmiError, check, lib
from .profile import span

__all__ = [
    "conv2d", "conv2d_skip", "conv2d_fan", "depthwise_conv2d", "conv_transpose2x2", "batch_norm_act", "relu", "add", "max_pool2d", "adaptive_avg_pool2d", "pyramid_pool", "pyramid_bottleneck_conv", "interpolate_bilinear",
    "cat", "dropout", "cross_entropy", "upsampled_cross_entropy", "upsample_source", "dice_loss", "focal_loss", "lovasz_softmax", "upsampled_lovasz_softmax", "lovasz_last_stats", "seg_metrics_accumulate", "to_nhwc", "empty_nhwc", "is_nhwc", "pad4", "set_conv_winograd", "get_conv_winograd", "set_wgrad_stream", "get_wgrad_stream", "set_conv_bn_stats", "get_conv_bn_stats", "wgrad_stream_join", "register_grad_slots", "reset_grad_slots", "sync_batch_norm_group", "sync_batch_norm_residual_tail", "batch_norm_depthwise", "batch_norm_depthwise_ok", "sync_groupable", "set_dropout_epoch",
]


def pad4(c):
    return (c + 3) & ~3


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(t, what):
    if not t.is_cuda:
        raise SegmiError("segmi.%s: tensor is on %s — the segmi hot path only runs on the MI355X "
                         "(there is deliberately no CPU / eager fallback)" % (what, t.device))
    if t.dtype != torch.float32:
        raise SegmiError("segmi.%s: expected float32, got %s" % (what, t.dtype))


# --------------------------------------------------------------------------- workspace (per stream)
_WS = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer; kernels on one stream run in order, so one buffer per stream is safe."""
    key = (device.index, _stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# --------------------------------------------------------------------------- layout helpers
def empty_nhwc(N, C, H, W, device, ld=None):
    ld = ld or pad4(C)
    buf = torch.empty((N, H, W, ld), device=device, dtype=torch.float32)
    t = buf.permute(0, 3, 1, 2)
    return t if ld == C else t[:, :C]


def ld_of(t):
    """Pixel stride (elements) of an NHWC-backed tensor, or None if t does not follow the convention."""
    if t.dim() != 4:
        return None
    N, C, H, W = t.shape
    s = t.stride()
    if W > 1:
        ld = s[3]
    elif H > 1:
        ld = s[2]
    elif N > 1:
        ld = s[0]
    else:
        ld = pad4(C)
    if ld < pad4(C) or (ld & 3):
        return None
    if C > 1 and s[1] != 1:
        return None
    if W > 1 and s[3] != ld:
        return None
    if H > 1 and s[2] != W * ld:
        return None
    if N > 1 and s[0] != H * W * ld:
        return None
    if (C & 3) and ld != pad4(C):
        return None  # ragged channel count must own its zero padding
    if t.data_ptr() & 15:
        return None
    return ld


def is_nhwc(t):
    return t.is_cuda and t.dtype == torch.float32 and ld_of(t) is not None


def to_nhwc(t, what="to_nhwc"):
    """Bring any 4-D fp32 CUDA tensor into the NHWC-backed convention (no-op when already there)."""
    _need_cuda(t, what)
    if ld_of(t) is not None:
        return t
    N, C, H, W = t.shape
    out = empty_nhwc(N, C, H, W, t.device)
    ld = pad4(C)
    if t.is_contiguous():
        check(lib.segmi_nchw_to_nhwc(t.data_ptr(), out.data_ptr(), N, C, H, W, ld, _stream()), "nchw_to_nhwc")
    elif t.is_contiguous(memory_format=torch.channels_last):
        check(lib.segmi_copy_rows(t.data_ptr(), C, out.data_ptr(), ld, N * H * W, C, ld, _stream()), "copy_rows")
    else:
        src = t.contiguous()  # arbitrary strided view: allocator-level gather by torch, then our transpose
        check(lib.segmi_nchw_to_nhwc(src.data_ptr(), out.data_ptr(), N, C, H, W, ld, _stream()), "nchw_to_nhwc")
    return out


def to_nchw_contiguous(t):
    """NHWC-backed -> plain contiguous NCHW (only for callers that insist on it)."""
    t = to_nhwc(t)
    N, C, H, W = t.shape
    out = torch.empty((N, C, H, W), device=t.device, dtype=torch.float32)
    check(lib.segmi_nhwc_to_nchw(t.data_ptr(), out.data_ptr(), N, C, H, W, ld_of(t), _stream()), "nhwc_to_nchw")
    return out


def _rows(t):
    N, C, H, W = t.shape
    return N * H * W


# --------------------------------------------------------------------------- convolution
def _filter_is_krsc(w):
    K, C, R, S = w.shape
    if R == 1 and S == 1:
        return w.is_contiguous() or w.is_contiguous(memory_format=torch.channels_last)
    return w.is_contiguous(memory_format=torch.channels_last)


def _filter_krsc(w, Ce):
    """Filter as a flat [K,R,S,Ce] device buffer (Ce = channels padded to 4, zero filled).  Returns a
    tensor that owns or aliases the memory."""
    K, C, R, S = w.shape
    if _filter_is_krsc(w):
        if Ce == C and not (w.data_ptr() & 15):
            return w
        out = torch.empty(K * R * S * Ce, device=w.device, dtype=torch.float32)
        check(lib.segmi_copy_rows(w.data_ptr(), C, out.data_ptr(), Ce, K * R * S, C, Ce, _stream()), "filter pad")
        return out
    if not w.is_contiguous():
        w = w.contiguous()
    out = torch.empty(K * R * S * Ce, device=w.device, dtype=torch.float32)
    check(lib.segmi_nchw_to_nhwc(w.data_ptr(), out.data_ptr(), K, C, R, S, Ce, _stream()), "filter kcrs->krsc")
    return out


# Gradient slots: a data-parallel reducer (segmi.distributed.GradAllReducer) keeps every parameter's gradient as a view into a
# flat all-reduce bucket.  When it registers those views here, the filter-gradient kernels write STRAIGHT into the bucket: the
# tensor handed to autograd is a fresh alias of the slot (its own TensorImpl, so AccumulateGrad adopts it without a clone or an
# in-place add) and the reducer's hook finds the gradient already in place — no per-parameter copy in the N > 1 path.
# A slot is handed out once per iteration (a filter used twice in one graph accumulates through the ordinary path).
_GRAD_SLOTS = {}          # id(parameter) -> [view into the bucket, taken this iteration, weakref to the parameter]
_HOOK_SAFE = {}           # id(parameter) -> weakref: parameters whose post-accumulate hook is a GradAllReducer's (it joins the side stream itself)


def register_grad_slots(views):
    """views: {parameter: gradient view with the parameter's shape and strides, or None to unregister that parameter}.
    Entries hold only a weak reference to their parameter and are verified by identity when handed out: a slot whose parameter
    died (its id may be reused by a new tensor) is dropped instead of being written."""
    for p, v in views.items():
        if v is None:
            e = _GRAD_SLOTS.get(id(p))
            if e is not None and e[2]() is p:
                del _GRAD_SLOTS[id(p)]
        else:
            _GRAD_SLOTS[id(p)] = [v, False, weakref.ref(p)]
    for k in [k for k, e in _GRAD_SLOTS.items() if e[2]() is None]:
        del _GRAD_SLOTS[k]


def mark_reducer_hooks(params, on=True):
    """A GradAllReducer declares that the post-accumulate hook on these parameters is its own (it waits for the filter-gradient
    side stream before it reads a gradient): such a hook does not force the filter gradient in order (see _on_wgrad_stream)."""
    for p in params:
        if on:
            _HOOK_SAFE[id(p)] = weakref.ref(p)
        else:
            _HOOK_SAFE.pop(id(p), None)


def _hook_safe(p):
    r = _HOOK_SAFE.get(id(p))
    return r is not None and r() is p


def reset_grad_slots(params=None):
    """Start of an iteration: every slot (of `params`, or all) may be handed out again."""
    for e in (_GRAD_SLOTS.values() if params is None else (_GRAD_SLOTS[id(p)] for p in params if id(p) in _GRAD_SLOTS)):
        e[1] = False


def _take_grad_slot(w):
    e = _GRAD_SLOTS.get(id(w))
    if e is not None and e[2]() is not w:          # stale entry under a reused id
        del _GRAD_SLOTS[id(w)]
        return None
    if e is None or e[1] or w.grad is not None or e[0].shape != w.shape or e[0].stride() != w.stride() or (e[0].data_ptr() & 15):
        return None
    e[1] = True
    v = e[0]
    return v.as_strided(v.shape, v.stride(), v.storage_offset())


def _filter_grad_buffer(w, Ce):
    """(flat KRSC buffer for the wgrad kernel, gradient tensor or None).  When the filter itself is KRSC in memory and needs no
    channel padding, the buffer IS the gradient tensor with the parameter's own (dense) strides — an owning tensor, not a view,
    so autograd's AccumulateGrad adopts it instead of cloning it (44 device copies per PSPNet-R50 step otherwise) — or, under a
    data-parallel reducer, a fresh alias of the parameter's slot in the all-reduce bucket (see _GRAD_SLOTS)."""
    K, C, R, S = w.shape
    if Ce == C and _filter_is_krsc(w):
        dw = _take_grad_slot(w) if _GRAD_SLOTS else None
        if dw is not None:
            return dw, dw
        mf = torch.channels_last if (R > 1 or S > 1 or not w.is_contiguous()) else torch.contiguous_format
        dw = torch.empty((K, C, R, S), device=w.device, dtype=torch.float32, memory_format=mf)
        return dw, dw
    return torch.empty(K * R * S * Ce, device=w.device, dtype=torch.float32), None


def _filter_grad_like(dw_krsc, w, Ce):
    """[K,R,S,Ce] wgrad buffer -> gradient tensor with weight's shape and memory layout."""
    K, C, R, S = w.shape
    if _filter_is_krsc(w):
        if Ce != C:
            crop = torch.empty(K * R * S * C, device=w.device, dtype=torch.float32)
            check(lib.segmi_copy_rows(dw_krsc.data_ptr(), Ce, crop.data_ptr(), C, K * R * S, C, C, _stream()), "filter crop")
            dw_krsc = crop
        return dw_krsc.view(K, R, S, C).permute(0, 3, 1, 2)
    out = torch.empty((K, C, R, S), device=w.device, dtype=torch.float32)
    check(lib.segmi_nhwc_to_nchw(dw_krsc.data_ptr(), out.data_ptr(), K, C, R, S, Ce, _stream()), "filter krsc->kcrs")
    return out


def conv_out_size(H, k, stride, pad, dil):
    return (H + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv_variant(d, op):
    """Kernel variant name for (desc, op in {0 fwd, 1 dgrad, 2 wgrad}) as a rocprofv3 trace shows it."""
    buf = ctypes.create_string_buffer(128)
    check(lib.segmi_conv2d_variant(d, op, buf, 128), "conv2d_variant")
    return buf.value.decode()


def _geom(d):
    return "N%d %dx%d C%d K%d %dx%d s%d d%d" % (d.N, d.H, d.W, d.C, d.K, d.R, d.S, d.stride, d.dil)


def _conv_flops(d, C):
    """Algorithmic FLOPs of one conv pass (true channel count C, not the 4-padded d.C)."""
    return 2 * d.N * d.P * d.Q * d.K * d.R * d.S * C


def _conv_issued(d, op):
    """Fraction of a direct launch's (tile x reduction chunk x tap) iteration space that the kernel issues (instrumentation only;
    1.0 wherever nothing is skipped).  Restates two rules of csrc/conv_igemm.hip: conv_dma_kernel's tap-mask form (FAST, not pointwise,
    unsplit, stride 1 for the data gradient) walks only the taps that reach the image from SOME row of its tile of BM consecutive
    pixels; conv_wgrad_dma_kernel's generic form (not ROWQ, not pointwise, Q >= 32, padded, C != 4) skips the 32-pixel chunks both
    image rows of which miss the image for the workgroup's tap.  Skipped products have a zero-filled operand: they are part of
    the algorithmic FLOP count (`_conv_flops`) and not of the EXECUTED one."""
    import numpy as np
    RS = d.R * d.S
    v = conv_variant(d, op)
    m = re.match(r"conv_(wgrad_)?dma_kernel<([^>]*)>", v)
    if RS <= 1 or m is None:
        return 1.0
    args = [a.strip() for a in m.group(2).split(",")]
    taps_r, taps_s = np.arange(d.R), np.arange(d.S)
    if op == 2:
        if args[2] != "false" or args[3] != "false" or d.Q < 32 or d.pad <= 0 or d.C == 4:
            return 1.0
        M = d.N * d.P * d.Q
        m0 = np.arange(0, M, 32)
        p0, q0 = (m0 % (d.P * d.Q)) // d.Q, m0 % d.Q
        p1 = np.where(p0 + 1 == d.P, 0, p0 + 1)
        ok = lambda pr: ((pr[:, None] * d.stride - d.pad + taps_r[None, :] * d.dil >= 0) &
                         (pr[:, None] * d.stride - d.pad + taps_r[None, :] * d.dil < d.H))
        issued = ok(p0) | ((q0 + 32 > d.Q)[:, None] & ok(p1))          # [chunks, R]: every tap column of a filter row alike
        return float(issued.mean())
    if args[5] != "true" or args[6] != "false" or " splitk=" in v or (op == 1 and d.stride != 1):
        return 1.0
    BM = int(args[0])
    if op == 0:       # rows = output pixels, taps read x at (p * stride - pad + r * dil, ...)
        Hd, Wd = d.P, d.Q
        ih = np.arange(Hd)[:, None] * d.stride - d.pad + taps_r[None, :] * d.dil
        iw = np.arange(Wd)[:, None] * d.stride - d.pad + taps_s[None, :] * d.dil
        vh, vw = (ih >= 0) & (ih < d.H), (iw >= 0) & (iw < d.W)
    else:             # rows = input pixels, taps read dy at (h + pad - r * dil, ...)
        Hd, Wd = d.H, d.W
        ih = np.arange(Hd)[:, None] + d.pad - taps_r[None, :] * d.dil
        iw = np.arange(Wd)[:, None] + d.pad - taps_s[None, :] * d.dil
        vh, vw = (ih >= 0) & (ih < d.P), (iw >= 0) & (iw < d.Q)
    pix = (vh[:, None, :, None] & vw[None, :, None, :]).reshape(Hd * Wd, RS)       # [pixel of one image, tap]
    rows = np.tile(pix, (d.N, 1))
    pad_rows = (-rows.shape[0]) % BM
    if pad_rows:
        rows = np.concatenate([rows, np.zeros((pad_rows, RS), dtype=bool)])
    return float(rows.reshape(-1, BM, RS).any(axis=1).mean())


def _conv_exec_flops(d, C, op):
    """EXECUTED FLOPs of a direct launch, as a callable for `span` (evaluated only while a KernelTimer is active)."""
    return lambda: int(round(_conv_flops(d, C) * _conv_issued(d, op)))


def _conv_bytes(d, C):
    """Algorithmic bytes of one conv pass: each of the three tensors (input, filter, output) crosses HBM once, fp32."""
    return 4 * (d.N * d.H * d.W * C + d.K * d.R * d.S * C + d.N * d.P * d.Q * d.K)


# Winograd F(2x2, 3x3) for the stride-1 3x3 layers (csrc/conv_winograd.hip): the DEFAULT algorithm of the eligible layers for
# all three passes since round 3 (the whole GPU suite runs under it; SEGMI_CONV_WINOGRAD=0 / set_conv_winograd(False) selects the
# direct implicit-GEMM kernels everywhere); `min_channels` = smallest min(C, K) it is used for (128 since round 4: measured per
# config in one call each, 256 -> 128 gives cfg1 7.74 -> 7.15 ms, cfg3 82.0 -> 80.9, cfg5 69.4 -> 68.7, cfg2 56.13 -> 56.22; at 64
# the transform traffic eats the saved multiplications: cfg1 7.27, cfg3 83.5 vs 83.2, cfg5 70.6 vs 69.7); `min_subgrid` = smallest ceil(H / dilation) it is used for
# (a dilated layer runs as dilation^2 dense sub-grids: ASPP's d = 12..36 on 33x33 maps would be 2x2 tiles of mostly padding).
_WINOGRAD = {"on": os.environ.get("SEGMI_CONV_WINOGRAD", "1") == "1",
             "min_channels": int(os.environ.get("SEGMI_CONV_WINOGRAD_MIN_CHANNELS", "128")),
             "min_subgrid": int(os.environ.get("SEGMI_CONV_WINOGRAD_MIN_SUBGRID", "8")),
             "wgrad": os.environ.get("SEGMI_CONV_WINOGRAD_WGRAD", "1") == "1", "calls": 0,
             # keep the forward pass's transformed input V (4x the layer input) for the filter gradient instead of transforming x
             # again in backward: 3.2 GB at the bench shape, one HBM pass of 5x|x| less per eligible layer
             "keep_v": os.environ.get("SEGMI_CONV_WINOGRAD_KEEP_V", "1") == "1"}


def set_conv_winograd(on, min_channels=None, min_subgrid=None, wgrad=None, keep_v=None):
    """Route eligible 3x3 stride-1 convolutions (forward and data gradient; with wgrad=True also the filter gradient) through
    the Winograd F(2x2,3x3) kernels.  keep_v: keep the forward pass's transformed input for the filter gradient."""
    _WINOGRAD["on"] = bool(on)
    if wgrad is not None:
        _WINOGRAD["wgrad"] = bool(wgrad)
    if keep_v is not None:
        _WINOGRAD["keep_v"] = bool(keep_v)
    if min_channels is not None:
        _WINOGRAD["min_channels"] = int(min_channels)
    if min_subgrid is not None:
        _WINOGRAD["min_subgrid"] = int(min_subgrid)


def get_conv_winograd():
    return dict(_WINOGRAD)


def _winograd(d, op):
    if not (_WINOGRAD["on"] and d.R == 3 and d.S == 3 and d.stride == 1 and min(d.C, d.K) >= _WINOGRAD["min_channels"]):
        return False
    sub = min(-(-d.H // d.dil), -(-d.W // d.dil))
    return sub >= _WINOGRAD["min_subgrid"] and lib.segmi_conv2d_winograd_ok(d, op) == 1


def _winograd_variant(d, op):
    buf = ctypes.create_string_buffer(128)
    check(lib.segmi_conv2d_winograd_variant(d, op, buf, 128), "conv2d_winograd_variant")
    return buf.value.decode()


def _winograd_inner(d, C):
    """(executed FLOPs, operand bytes) of the 16 transform-domain contractions of one Winograd pass (any of the three passes:
    [T x C] x [C x K] per plane with the true channel counts)."""
    T = lib.segmi_conv2d_winograd_tiles(d)
    return 32 * T * C * d.K, 64 * (T * C + C * d.K + T * d.K)


def _winograd_wgrad_variant(d):
    buf = ctypes.create_string_buffer(128)
    check(lib.segmi_conv2d_winograd_wgrad_variant(d, buf, 128), "conv2d_winograd_wgrad_variant")
    return buf.value.decode()


# BatchNorm statistics from the producing convolution's epilogue (segmi_conv2d_fwd_stats, default on; SEGMI_CONV_BN_STATS=0 keeps the
# separate statistics pass): a convolution whose output goes to a training-mode BatchNorm writes the per-row-tile Welford partials
# of its output while the tile is in registers, and the BN layer merges those instead of reading the tensor again.  The pairing
# is discovered at run time: the convolution tags its output with its module, the BN layer that receives a tagged tensor marks
# that module (`_bn_consumer`), and from the next step on the convolution emits the partials (`bn_stats=True`), attached to
# its output as `_segmi_bn_stats` = (partials, nparts, tensor version).
_BN_FUSE = {"on": os.environ.get("SEGMI_CONV_BN_STATS", "1") == "1", "emitted": 0, "consumed": 0, "last": None, "fan": None}


def set_conv_bn_stats(on):
    _BN_FUSE["on"] = bool(on)


def get_conv_bn_stats():
    return {k: _BN_FUSE[k] for k in ("on", "emitted", "consumed")}


def _conv_fwd(d, C, x, w, bias, y, accumulate=0, keep_v=False, bn_stats=False):
    """segmi_conv2d_fwd (or its pre-split-filter / Winograd form when that applies) with its workspace and roofline span.
    bn_stats: also emit the BN-statistics partials of y when the launch has that epilogue (left in _BN_FUSE["last"]).
    w: flat KRSC filter tensor of d.K * d.R * d.S * d.C floats.  keep_v: the caller will need this layer's filter gradient —
    returns the Winograd-transformed input V (a tensor to keep for `_conv_wgrad(v=...)`) when the layer runs on the Winograd
    kernels with a Winograd filter gradient, else None."""
    dev, st = x.device, _stream()
    if _winograd(d, 0):
        _WINOGRAD["calls"] += 1
        nws = lib.segmi_conv2d_winograd_workspace(d, 0)
        ws = workspace(nws, dev)
        v = None
        if keep_v and _WINOGRAD["keep_v"] and _WINOGRAD["wgrad"] and lib.segmi_conv2d_winograd_wgrad_ok(d) == 1:
            v = torch.empty(lib.segmi_conv2d_winograd_v_bytes(d) // 4, device=dev, dtype=torch.float32)
        part, parts = None, 0
        if bn_stats and not accumulate:
            parts = lib.segmi_conv2d_winograd_fwd_stats_parts(d)         # the output transform's BN-statistics epilogue
            if parts > 0:
                part = torch.empty(parts * 3 * d.K, device=dev, dtype=torch.float32)
        with span(lambda: _winograd_variant(d, 0), _conv_flops(d, C), _conv_bytes(d, C), detail=lambda: _geom(d), inner=lambda: _winograd_inner(d, C)):
            check(lib.segmi_conv2d_winograd_fwd(d, x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                y.data_ptr(), accumulate, v.data_ptr() if v is not None else None,
                                                part.data_ptr() if part is not None else None, ws.data_ptr(), nws, st),
                  "conv2d_winograd_fwd")
        if part is not None:
            _BN_FUSE["last"] = (part, parts)
            _BN_FUSE["emitted"] += 1
        return v
    bp = bias.data_ptr() if bias is not None else None
    if bn_stats and not accumulate:
        parts = lib.segmi_conv2d_fwd_stats_parts(d)
        if parts > 0:
            part = torch.empty(parts * 3 * d.K, device=dev, dtype=torch.float32)
            with span(lambda: conv_variant(d, 0), _conv_exec_flops(d, C, 0), _conv_bytes(d, C), detail=lambda: _geom(d), eff=_conv_flops(d, C)):
                check(lib.segmi_conv2d_fwd_stats(d, x.data_ptr(), w.data_ptr(), bp, y.data_ptr(), part.data_ptr(), st), "conv2d_fwd_stats")
            _BN_FUSE["last"] = (part, parts)
            _BN_FUSE["emitted"] += 1
            return None
    nws = lib.segmi_conv2d_fwd_workspace(d) if (bias is None and not accumulate) else 0
    ws = workspace(nws, dev) if nws else None
    wsp = ws.data_ptr() if ws is not None else None
    with span(lambda: conv_variant(d, 0), _conv_exec_flops(d, C, 0), _conv_bytes(d, C), detail=lambda: _geom(d), eff=_conv_flops(d, C)):
        check(lib.segmi_conv2d_fwd(d, x.data_ptr(), w.data_ptr(), bp, y.data_ptr(), accumulate, wsp, nws, st), "conv2d_fwd")


def _conv_wgrad(d, C, x, dy, dwb, v=None):
    """segmi_conv2d_wgrad (or its Winograd form when that applies) into the flat KRSC buffer dwb, with workspace and span.
    v: the transformed input the forward pass kept (`_conv_fwd(keep_v=True)`), or None."""
    dev, st = x.device, _stream()
    if v is not None or (_WINOGRAD["wgrad"] and _winograd(d, 0) and lib.segmi_conv2d_winograd_wgrad_ok(d) == 1):
        _WINOGRAD["calls"] += 1
        nws = lib.segmi_conv2d_winograd_wgrad_workspace(d)
        ws = workspace(nws, dev)
        with span(lambda: _winograd_wgrad_variant(d), _conv_flops(d, C), _conv_bytes(d, C), detail=lambda: _geom(d), inner=lambda: _winograd_inner(d, C)):
            check(lib.segmi_conv2d_winograd_wgrad(d, x.data_ptr(), v.data_ptr() if v is not None else None, dy.data_ptr(), dwb.data_ptr(),
                                                  ws.data_ptr(), nws, st), "conv2d_winograd_wgrad")
        return
    nws = lib.segmi_conv2d_wgrad_workspace(d)
    ws = workspace(nws, dev) if nws else None
    with span(lambda: conv_variant(d, 2), _conv_exec_flops(d, C, 2), _conv_bytes(d, C), detail=lambda: _geom(d), eff=_conv_flops(d, C)):
        check(lib.segmi_conv2d_wgrad(d, x.data_ptr(), dy.data_ptr(), dwb.data_ptr(), ws.data_ptr() if ws is not None else None, nws, st),
              "conv2d_wgrad")


# Filter gradients on a side HIP stream (default on; SEGMI_WGRAD_STREAM=0 / set_wgrad_stream(False) keeps them in order): nothing in the backward pass consumes dW —
# only the optimizer (or the gradient all-reduce) does — so the wgrad launch of a layer may run concurrently with the data-gradient
# chain of the layers below it.  The wgrad kernels leave most of a CU's register file and 32 KB of LDS free (116 VGPRs x 2 waves per
# SIMD), so the HBM-bound BN-backward / transform kernels of the main stream co-reside with them instead of queueing behind them.
# Rules that keep it exact: the side stream first waits for the main stream (dy is complete), x / dy / dW are record_stream()ed so the
# caching allocator does not recycle them early, the main stream re-joins at the END of the backward pass (autograd engine
# callback), and the path is only taken when the parameter has no gradient yet (AccumulateGrad then adopts the tensor without
# launching anything; an accumulating or bucket-view gradient keeps the in-order path).  Two more cases stay in order because
# something on the compute stream reads dW before the end-of-backward join: a filter used MORE THAN ONCE in one graph (the engine
# sums the two gradients in its input buffer — the second use first joins the side stream, then runs in order), and a parameter
# carrying tensor hooks or post-accumulate hooks other than a GradAllReducer's (a clipping hook: it sees dW during backward; the
# reducer's own hook joins the side stream before it touches a gradient).  Hooks registered on the parameter's AccumulateGrad NODE
# (torch DistributedDataParallel, FSDP, torch.autograd.graph hooks) cannot be enumerated from Python: whenever a process group is
# initialised, only parameters a GradAllReducer has claimed (_HOOK_SAFE) take the side stream — under a foreign data-parallel
# wrapper every filter gradient stays in order.  Node hooks installed by hand without torch.distributed need set_wgrad_stream(False).
_WGRAD_SIDE = {"on": os.environ.get("SEGMI_WGRAD_STREAM", "1") == "1", "streams": {}, "armed": None, "launches": 0,
               "task": None, "inflight": set(), "in_order_reuse": 0, "in_order_hooks": 0}


def set_wgrad_stream(on):
    _WGRAD_SIDE["on"] = bool(on)


def get_wgrad_stream():
    return {"on": _WGRAD_SIDE["on"], "launches": _WGRAD_SIDE["launches"], "in_order_reuse": _WGRAD_SIDE["in_order_reuse"],
            "in_order_hooks": _WGRAD_SIDE["in_order_hooks"]}


def _join_wgrad_stream():
    _WGRAD_SIDE["armed"] = None
    _WGRAD_SIDE["task"] = None
    _WGRAD_SIDE["inflight"].clear()
    for side in _WGRAD_SIDE["streams"].values():
        torch.cuda.current_stream(side.device).wait_stream(side)


def wgrad_stream_join():
    """Make the current stream wait for every filter gradient launched on the side stream so far (a gradient bucket is about to
    be all-reduced in the middle of the backward pass).  No-op when the side stream is off."""
    if _WGRAD_SIDE["streams"]:
        for side in _WGRAD_SIDE["streams"].values():
            torch.cuda.current_stream(side.device).wait_stream(side)


def _on_wgrad_stream(weight, tensors, fn):
    """Run fn() — launches that only produce (part of) `weight`'s gradient from `tensors` — on the filter-gradient side stream
    when that is enabled and safe (see _WGRAD_SIDE), in order on the current stream otherwise."""
    if not (_WGRAD_SIDE["on"] and weight.grad is None and weight.is_leaf and torch.is_grad_enabled() is False):
        return fn()
    task = torch._C._current_graph_task_id()
    if _WGRAD_SIDE["task"] != task:
        _WGRAD_SIDE["task"] = task
        _WGRAD_SIDE["inflight"].clear()
    if id(weight) in _WGRAD_SIDE["inflight"]:
        # second use of this filter in the same graph: the engine will add the two gradients on the compute stream
        wgrad_stream_join()
        _WGRAD_SIDE["in_order_reuse"] += 1
        return fn()
    if not _hook_safe(weight) and (weight._backward_hooks or getattr(weight, "_post_accumulate_grad_hooks", None)
                                   or (torch.distributed.is_available() and torch.distributed.is_initialized())):
        _WGRAD_SIDE["in_order_hooks"] += 1
        return fn()
    if weight._backward_hooks:
        _WGRAD_SIDE["in_order_hooks"] += 1
        return fn()
    if task >= 0:
        _WGRAD_SIDE["inflight"].add(id(weight))
    dev = weight.device
    side = _WGRAD_SIDE["streams"].get(dev.index)
    if side is None:
        # SEGMI_WGRAD_STREAM_PRIORITY: HIP stream priority of the side stream (lower number = higher priority; the MI355X range is
        # {0, -1}).  Default -1: measured in one call at cfg2 (profiles/r03_wgrad_stream_priority.txt) 140.1 img/s against 138.4 /
        # 138.9 at the compute stream's priority — a filter-gradient launch that is dispatched as soon as it is issued overlaps the
        # HBM-bound BN-backward kernels that follow on the compute stream instead of queueing behind them.
        side = _WGRAD_SIDE["streams"][dev.index] = torch.cuda.Stream(device=dev, priority=int(os.environ.get("SEGMI_WGRAD_STREAM_PRIORITY", "-1")))
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        fn()
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    _WGRAD_SIDE["launches"] += 1
    # one join per backward pass, keyed by the engine's graph-task id (a pass that died with an exception never ran its
    # callback: the next pass has a new id and arms again)
    if task < 0:                           # not inside an engine-driven backward pass: join right away
        _join_wgrad_stream()
    elif _WGRAD_SIDE["armed"] != task:
        torch.autograd.Variable._execution_engine.queue_callback(_join_wgrad_stream)
        _WGRAD_SIDE["armed"] = task


def _conv_wgrad_param(weight, d, C, x, dy, dwb, v=None):
    """_conv_wgrad for a parameter's filter gradient: on the side stream when enabled and safe, in order otherwise."""
    _on_wgrad_stream(weight, (x, dy, dwb, v), lambda: _conv_wgrad(d, C, x, dy, dwb, v))


def _conv_dgrad(d, C, dy, wt, dx, accumulate=0):
    """segmi_conv2d_dgrad (or its Winograd form).  wt: flat CRSK filter of d.C * d.R * d.S * pad4(d.K) floats."""
    dev, st = dy.device, _stream()
    if _winograd(d, 1):
        _WINOGRAD["calls"] += 1
        nws = lib.segmi_conv2d_winograd_workspace(d, 1)
        ws = workspace(nws, dev)
        with span(lambda: _winograd_variant(d, 1), _conv_flops(d, C), _conv_bytes(d, C), detail=lambda: _geom(d), inner=lambda: _winograd_inner(d, C)):
            check(lib.segmi_conv2d_winograd_dgrad(d, dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), accumulate, ws.data_ptr(), nws, st),
                  "conv2d_winograd_dgrad")
        return
    with span(lambda: conv_variant(d, 1), _conv_exec_flops(d, C, 1), _conv_bytes(d, C), detail=lambda: _geom(d), eff=_conv_flops(d, C)):
        check(lib.segmi_conv2d_dgrad(d, dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), accumulate, st), "conv2d_dgrad")


class _FilterTransposes:
    """[C,R,S,K] copies of the convolution filters for the data-gradient pass, produced for ALL filters of a training step by
    ONE launch (segmi_filter_krsc_to_crsk_multi) instead of one ~5 us launch per convolution (66 per PSPNet-R50 step).

    A forward pass `note`s every filter that lives in parameter memory and whose input needs a gradient; the first data-gradient
    call after that transposes everything noted into one pooled buffer, later calls of the same backward only look their
    slice up.  Filters change between steps (the optimizer writes them), never between a forward pass and its backward, so the
    pool is rebuilt once per step.  Filters that are temporaries (channel-padded stems, filter slices) are not noted and keep
    the single-filter call."""

    def __init__(self):
        self.notes = {}          # key -> (weakref to the parameter, floats of its transposed copy)
        self.clean = False       # pool holds the transposition of everything in `notes`
        self.pool = None
        self.offsets = {}
        self.table = None        # (signature, device table, n, total tiles)
        self.launches = 0

    @staticmethod
    def key(w, K, R, S, Ce, Kp):
        return (w.data_ptr(), w.device.index, K, R, S, Ce, Kp)

    def note(self, weight, w, K, R, S, Ce, Kp):
        if w.data_ptr() != weight.data_ptr():
            return
        if self.clean:           # first filter of a new step: forget the previous step's set
            self.notes, self.clean = {}, False
        self.notes[self.key(w, K, R, S, Ce, Kp)] = (weakref.ref(weight), Ce * R * S * Kp)

    def get(self, weight, w, K, R, S, Ce, Kp):
        """The transposed filter as a flat tensor, or None when this filter was not noted (caller transposes it alone)."""
        k = self.key(w, K, R, S, Ce, Kp)
        if w.data_ptr() != weight.data_ptr() or k not in self.notes:
            return None
        if not self.clean:
            self._run(w.device)
        off = self.offsets.get(k)
        return None if off is None else self.pool[off[0]:off[0] + off[1]]

    def _run(self, dev):
        live = [(k, n) for k, (ref, n) in self.notes.items() if ref() is not None and k[1] == dev.index]
        self.notes = {k: self.notes[k] for k, _ in live}
        total = sum(n for _, n in live)
        if self.pool is None or self.pool.device != dev or self.pool.numel() < total:
            self.pool, self.table = torch.empty(max(total, 1), device=dev, dtype=torch.float32), None
        sig = (tuple(k for k, _ in live), self.pool.data_ptr())
        if self.table is None or self.table[0] != sig:
            arr = (FilterTx * max(len(live), 1))()
            self.offsets, off, tiles = {}, 0, 0
            for i, (k, n) in enumerate(live):
                ptr, _, K, R, S, Ce, Kp = k
                arr[i] = FilterTx(ptr, self.pool.data_ptr() + 4 * off, K, R, S, Ce, Kp, tiles)
                self.offsets[k] = (off, n)
                off += n
                tiles += lib.segmi_filter_tx_tiles(K, R, S, Ce, Kp)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.table = (sig, host.to(dev), len(live), tiles)
        _, tab, n, tiles = self.table
        if n:
            check(lib.segmi_filter_krsc_to_crsk_multi(tab.data_ptr(), n, tiles, _stream()), "krsc_to_crsk_multi")
            self.launches += 1
        self.clean = True


_filter_transposes = _FilterTransposes()


def _filter_crsk(weight, w, K, R, S, Ce, Kp):
    """[Ce,R,S,Kp] copy of the KRSC filter `w` of parameter `weight` (pooled per step when possible, see _FilterTransposes)."""
    wt = _filter_transposes.get(weight, w, K, R, S, Ce, Kp)
    if wt is None:
        wt = torch.empty(Ce * R * S * Kp, device=w.device, dtype=torch.float32)
        check(lib.segmi_filter_krsc_to_crsk(w.data_ptr(), wt.data_ptr(), K, R, S, Ce, Kp, _stream()), "krsc_to_crsk")
    return wt


# Grad mode at the time a convolution wrapper was called (inside Function.forward it is always off, and needs_input_grad reflects
# requires_grad only): under torch.no_grad() validation nothing will ever ask for the filter gradient, so the Winograd layers must
# not allocate the 4x-size V buffer they keep for it.
_FWD = {"grad": True}


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil, bn_stats=False):
        x = to_nhwc(x, "conv2d")
        _need_cuda(weight, "conv2d")
        N, C, H, W = x.shape
        K, Cw, R, S = weight.shape
        if Cw != C:
            raise SegmiError("conv2d: input has %d channels, filter expects %d (groups != 1 goes through dwconv)" % (C, Cw))
        Ce = pad4(C)
        w = _filter_krsc(weight, Ce)
        P, Q = conv_out_size(H, R, stride, pad, dil), conv_out_size(W, S, stride, pad, dil)
        y = empty_nhwc(N, K, P, Q, x.device)
        d = ConvDesc(N, H, W, Ce, K, R, S, P, Q, stride, pad, dil, ld_of(x), ld_of(y))
        v = _conv_fwd(d, C, x, w, bias, y, keep_v=ctx.needs_input_grad[1] and _FWD["grad"], bn_stats=bn_stats)
        if ctx.needs_input_grad[0]:
            _filter_transposes.note(weight, w, K, R, S, Ce, pad4(K))
        ctx.save_for_backward(x, weight, v)
        ctx.geom = (N, C, H, W, K, R, S, P, Q, stride, pad, dil)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, v = ctx.saved_tensors
        N, C, H, W, K, R, S, P, Q, stride, pad, dil = ctx.geom
        dy = to_nhwc(dy, "conv2d.backward")
        Ce = pad4(C)
        st = _stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            Kp = pad4(K)
            w = _filter_krsc(weight, Ce)
            wt = _filter_crsk(weight, w, K, R, S, Ce, Kp)
            dx = empty_nhwc(N, C, H, W, x.device)
            d = ConvDesc(N, H, W, Ce, K, R, S, P, Q, stride, pad, dil, ld_of(dx), ld_of(dy))
            _conv_dgrad(d, C, dy, wt, dx)
        if ctx.needs_input_grad[1]:
            d = ConvDesc(N, H, W, Ce, K, R, S, P, Q, stride, pad, dil, ld_of(x), ld_of(dy))
            dwb, dw_owned = _filter_grad_buffer(weight, Ce)
            if dw_owned is not None:
                _conv_wgrad_param(weight, d, C, x, dy, dwb, v)
            else:
                _conv_wgrad(d, C, x, dy, dwb, v)
            dw = dw_owned if dw_owned is not None else _filter_grad_like(dwb, weight, Ce)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            rows = N * P * Q
            nws = lib.segmi_colsum_workspace(rows, K)
            ws = workspace(nws, x.device)
            db = torch.empty(K, device=x.device, dtype=torch.float32)
            check(lib.segmi_colsum(dy.data_ptr(), ld_of(dy), rows, K, db.data_ptr(), ws.data_ptr(), nws, st), "colsum")
        return dx, dw, db, None, None, None, None


def _tag_bn_stats(y, producer):
    """Attach what a following BatchNorm can use: the producing module (pairing discovery) and, when the convolution emitted
    them, the statistics partials of exactly this tensor version."""
    last, _BN_FUSE["last"] = _BN_FUSE["last"], None
    if producer is not None:
        y._segmi_producer = weakref.ref(producer)
    if last is not None:
        y._segmi_bn_stats = (last[0], last[1], y._version)
    return y


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, bn_stats=False, producer=None):
    """aten::conv2d replacement (groups == 1, symmetric stride/padding/dilation).  bn_stats: the output feeds a training-mode
    BatchNorm — emit its statistics partials from the convolution's epilogue when the launch supports it."""
    _FWD["grad"] = torch.is_grad_enabled()
    _BN_FUSE["last"] = None
    y = _Conv2dFn.apply(x, weight, bias, int(stride), int(padding), int(dilation), bool(bn_stats) and _BN_FUSE["on"])
    return _tag_bn_stats(y, producer)


class _Conv2dSkipFn(torch.autograd.Function):
    """y, skip = conv(x, w), x — the residual fork of a ResNet block expressed as ONE autograd node, so that its backward can
    ACCUMULATE the data gradient of the convolution onto the gradient that arrives through the skip (dgrad kernel with
    accumulate=1) instead of leaving a separate full-size `add` to autograd (16 such adds per PSPNet-R50 step).
    Contract: the gradient arriving at `skip` must be exclusively owned by this node (it is overwritten in place); that holds
    for batch_norm_act(residual=skip, relu=True), whose backward allocates the residual gradient."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, dil, bn_stats=False):
        x = to_nhwc(x, "conv2d")
        y = _Conv2dFn.forward(ctx, x, weight, None, stride, pad, dil, bn_stats)
        return y, x[:]

    @staticmethod
    def backward(ctx, dy, dskip):
        if dskip is None or not ctx.needs_input_grad[0]:
            dx, dw = _Conv2dFn.backward(ctx, dy)[:2]
            if dskip is not None and dx is not None:
                dx = add(dx, dskip)
            return (dx if dx is not None else dskip), dw, None, None, None, None
        x, weight, v = ctx.saved_tensors
        N, C, H, W, K, R, S, P, Q, stride, pad, dil = ctx.geom
        dy = to_nhwc(dy, "conv2d.backward")
        dskip = to_nhwc(dskip, "conv2d.backward")
        Ce, Kp, st = pad4(C), pad4(K), _stream()
        w = _filter_krsc(weight, Ce)
        wt = _filter_crsk(weight, w, K, R, S, Ce, Kp)
        d = ConvDesc(N, H, W, Ce, K, R, S, P, Q, stride, pad, dil, ld_of(dskip), ld_of(dy))
        _conv_dgrad(d, C, dy, wt, dskip, accumulate=1)
        dw = None
        if ctx.needs_input_grad[1]:
            d = ConvDesc(N, H, W, Ce, K, R, S, P, Q, stride, pad, dil, ld_of(x), ld_of(dy))
            dwb, dw_owned = _filter_grad_buffer(weight, Ce)
            if dw_owned is not None:
                _conv_wgrad_param(weight, d, C, x, dy, dwb, v)
            else:
                _conv_wgrad(d, C, x, dy, dwb, v)
            dw = dw_owned if dw_owned is not None else _filter_grad_like(dwb, weight, Ce)
        return dskip, dw, None, None, None, None


def conv2d_skip(x, weight, stride=1, padding=0, dilation=1, bn_stats=False, producer=None):
    """(conv2d(x, weight), x) with a fused backward: see _Conv2dSkipFn for the ownership contract of the skip gradient."""
    _FWD["grad"] = torch.is_grad_enabled()
    _BN_FUSE["last"] = None
    y, skip = _Conv2dSkipFn.apply(x, weight, int(stride), int(padding), int(dilation), bool(bn_stats) and _BN_FUSE["on"])
    return _tag_bn_stats(y, producer), skip


class _Conv2dFanFn(torch.autograd.Function):
    """ys = [conv2d(x, w_i, stride_i, pad_i, dil_i) for i] — several bias-free convolutions of ONE input (conv1 and the projection
    shortcut of a residual block, models/resnet.py:105-121) as one autograd node: the data gradients of all branches are summed by
    the dgrad kernels themselves (accumulate=1 onto the first branch's dx) instead of by an autograd-engine `aten::add` over the
    full-size input (4 per PSPNet-R50 step, 0.25 ms).  args: geoms = ((stride, pad, dil, bn_stats), ...), then the filters."""

    @staticmethod
    def forward(ctx, x, geoms, *weights):
        x = to_nhwc(x, "conv2d")
        N, C, H, W = x.shape
        Ce = pad4(C)
        ys, vs, gs, stats = [], [], [], []
        for i, (weight, (stride, pad, dil, bn_stats)) in enumerate(zip(weights, geoms)):
            _need_cuda(weight, "conv2d")
            K, Cw, R, S = weight.shape
            if Cw != C:
                raise SegmiError("conv2d: input has %d channels, filter expects %d" % (C, Cw))
            w = _filter_krsc(weight, Ce)
            P, Q = conv_out_size(H, R, stride, pad, dil), conv_out_size(W, S, stride, pad, dil)
            y = empty_nhwc(N, K, P, Q, x.device)
            d = ConvDesc(N, H, W, Ce, K, R, S, P, Q, stride, pad, dil, ld_of(x), ld_of(y))
            _BN_FUSE["last"] = None
            v = _conv_fwd(d, C, x, w, None, y, keep_v=ctx.needs_input_grad[2 + i] and _FWD["grad"], bn_stats=bn_stats)
            stats.append(_BN_FUSE["last"])
            _BN_FUSE["last"] = None
            if ctx.needs_input_grad[0]:
                _filter_transposes.note(weight, w, K, R, S, Ce, pad4(K))
            ys.append(y)
            vs.append(v)
            gs.append((K, R, S, P, Q, stride, pad, dil))
        ctx.save_for_backward(x, *weights, *[v for v in vs if v is not None])
        ctx.has_v = [v is not None for v in vs]
        ctx.geom = (N, C, H, W, tuple(gs))
        _BN_FUSE["fan"] = stats
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        saved = list(ctx.saved_tensors)
        N, C, H, W, gs = ctx.geom
        n = len(gs)
        x, weights, rest = saved[0], saved[1:1 + n], saved[1 + n:]
        vs = [rest.pop(0) if h else None for h in ctx.has_v]
        Ce = pad4(C)
        dx, dws = None, []
        for i, (dy, weight, v, (K, R, S, P, Q, stride, pad, dil)) in enumerate(zip(dys, weights, vs, gs)):
            if dy is None:
                dws.append(None)
                continue
            dy = to_nhwc(dy, "conv2d.backward")
            if ctx.needs_input_grad[0]:
                w = _filter_krsc(weight, Ce)
                wt = _filter_crsk(weight, w, K, R, S, Ce, pad4(K))
                first = dx is None
                if first:
                    dx = empty_nhwc(N, C, H, W, x.device)
                d = ConvDesc(N, H, W, Ce, K, R, S, P, Q, stride, pad, dil, ld_of(dx), ld_of(dy))
                _conv_dgrad(d, C, dy, wt, dx, accumulate=0 if first else 1)
            dw = None
            if ctx.needs_input_grad[2 + i]:
                d = ConvDesc(N, H, W, Ce, K, R, S, P, Q, stride, pad, dil, ld_of(x), ld_of(dy))
                dwb, dw_owned = _filter_grad_buffer(weight, Ce)
                if dw_owned is not None:
                    _conv_wgrad_param(weight, d, C, x, dy, dwb, v)
                else:
                    _conv_wgrad(d, C, x, dy, dwb, v)
                dw = dw_owned if dw_owned is not None else _filter_grad_like(dwb, weight, Ce)
            dws.append(dw)
        return (dx, None) + tuple(dws)


def conv2d_fan(x, branches):
    """[conv2d(x, weight, None, stride, padding, dilation) for every branch] as ONE autograd node (see _Conv2dFanFn).
    branches: [(weight, stride, padding, dilation, bn_stats, producer module or None), ...]; the FIRST branch must have stride 1
    (its data gradient initialises dx, the others accumulate onto it)."""
    if not branches or int(branches[0][1]) != 1:
        raise SegmiError("conv2d_fan: the first branch must be a stride-1 convolution")
    _FWD["grad"] = torch.is_grad_enabled()
    geoms = tuple((int(s), int(p), int(d), bool(b) and _BN_FUSE["on"]) for _, s, p, d, b, _ in branches)
    ys = _Conv2dFanFn.apply(x, geoms, *[b[0] for b in branches])
    stats, _BN_FUSE["fan"] = _BN_FUSE.get("fan") or [None] * len(ys), None
    for y, st, b in zip(ys, stats, branches):
        _BN_FUSE["last"] = st
        _tag_bn_stats(y, b[5])
    return list(ys)


# --------------------------------------------------------------------------- depthwise convolution
_DW_WGRAD_SIDE = os.environ.get("SEGMI_DW_WGRAD_STREAM", "1") == "1"      # A/B hook: 0 keeps the depthwise filter gradients in order


def dw_filter_rsc_layout(w):
    """The depthwise filter [C, 1, R, S] re-laid over [R, S, C] memory (same logical tensor): what segmi.nn.Conv2d stores."""
    C, one, R, S = w.shape
    return w.permute(2, 3, 0, 1).contiguous().permute(2, 3, 0, 1)


def _dw_rsc_view(w):
    """Flat [R*S*C] view of a depthwise filter whose memory already is tap-major (dw_filter_rsc_layout), else None."""
    C, one, R, S = w.shape
    s = w.stride()
    if one == 1 and s[0] == 1 and (S == 1 or s[3] == C) and (R == 1 or s[2] == S * C) and (w.data_ptr() & 15) == 0:
        return torch.as_strided(w, (R * S * C,), (1,), w.storage_offset())
    return None


class _DepthwiseConv2dFn(torch.autograd.Function):
    """nn.Conv2d(C, C, k, groups=C, bias=False): the kernels read the filter as [R,S,C]; a parameter stored in that memory order
    (segmi.nn.Conv2d does) is used in place and receives its gradient in the same order, any other filter is re-laid per call."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, dil, bn_stats=False):
        x = to_nhwc(x, "depthwise_conv2d")
        _need_cuda(weight, "depthwise_conv2d")
        N, C, H, W = x.shape
        Cw, one, R, S = weight.shape
        if Cw != C or one != 1:
            raise SegmiError("depthwise_conv2d: filter %s does not match %d input channels" % (tuple(weight.shape), C))
        if C & 3:
            raise SegmiError("depthwise_conv2d: channel count must be a multiple of 4 (got %d)" % C)
        st = _stream()
        wrsc = _dw_rsc_view(weight.detach())
        ctx.rsc_param = wrsc is not None
        if wrsc is None:
            wrsc = torch.empty(R * S * C, device=x.device, dtype=torch.float32)
            check(lib.segmi_nchw_to_nhwc(weight.contiguous().data_ptr(), wrsc.data_ptr(), 1, C, R * S, 1, C, st), "dw filter crs->rsc")
        P, Q = conv_out_size(H, R, stride, pad, dil), conv_out_size(W, S, stride, pad, dil)
        y = empty_nhwc(N, C, P, Q, x.device)
        d = ConvDesc(N, H, W, C, C, R, S, P, Q, stride, pad, dil, ld_of(x), ld_of(y))
        parts = lib.segmi_dwconv2d_fwd_stats_parts(d) if bn_stats else 0
        if parts > 0:        # the BatchNorm behind this layer takes its statistics from the kernel's epilogue (see _BN_FUSE)
            part = torch.empty(parts * 3 * C, device=x.device, dtype=torch.float32)
            check(lib.segmi_dwconv2d_fwd_stats(d, x.data_ptr(), wrsc.data_ptr(), y.data_ptr(), part.data_ptr(), st), "dwconv2d_fwd_stats")
            _BN_FUSE["last"] = (part, parts)
            _BN_FUSE["emitted"] += 1
        else:
            check(lib.segmi_dwconv2d_fwd(d, x.data_ptr(), wrsc.data_ptr(), y.data_ptr(), st), "dwconv2d_fwd")
        ctx.save_for_backward(x, wrsc, weight)
        ctx.geom = (N, C, H, W, R, S, P, Q, stride, pad, dil)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wrsc, weight = ctx.saved_tensors
        N, C, H, W, R, S, P, Q, stride, pad, dil = ctx.geom
        dy = to_nhwc(dy, "depthwise_conv2d.backward")
        st = _stream()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = empty_nhwc(N, C, H, W, x.device)
            d = ConvDesc(N, H, W, C, C, R, S, P, Q, stride, pad, dil, ld_of(dx), ld_of(dy))
            check(lib.segmi_dwconv2d_dgrad(d, dy.data_ptr(), wrsc.data_ptr(), dx.data_ptr(), st), "dwconv2d_dgrad")
        if ctx.needs_input_grad[1]:
            d = ConvDesc(N, H, W, C, C, R, S, P, Q, stride, pad, dil, ld_of(x), ld_of(dy))
            # under a data-parallel reducer the kernel writes straight into the parameter's slot of the all-reduce bucket (same
            # tap-major memory order): the reducer's hook then finds the gradient in place — no copy, no join of the side stream
            slot = _take_grad_slot(weight) if (_GRAD_SLOTS and ctx.rsc_param) else None
            dwr = slot if slot is not None else torch.empty(R * S * C, device=x.device, dtype=torch.float32)

            def run_wgrad():
                nws = lib.segmi_dwconv2d_wgrad_workspace(d)
                ws = workspace(nws, x.device)             # (keyed by the stream the call runs on)
                check(lib.segmi_dwconv2d_wgrad(d, x.data_ptr(), dy.data_ptr(), dwr.data_ptr(), ws.data_ptr(), nws, _stream()), "dwconv2d_wgrad")

            if ctx.rsc_param:
                # the parameter's own memory order: the gradient is a view of what the kernel wrote, and — like the dense filter
                # gradients — nothing in the backward pass reads it: the two launches (partials + their sum) go to the
                # filter-gradient side stream when that is safe (round 5: 63 x ~43 us per DeepLab-Xception step off the compute stream)
                if _DW_WGRAD_SIDE:
                    _on_wgrad_stream(weight, (x, dy, dwr), run_wgrad)
                else:
                    run_wgrad()
                dw = slot if slot is not None else dwr.view(R, S, C).permute(2, 0, 1).unsqueeze(1)
            else:
                run_wgrad()
                dw = torch.empty((C, 1, R, S), device=x.device, dtype=torch.float32)
                check(lib.segmi_nhwc_to_nchw(dwr.data_ptr(), dw.data_ptr(), 1, C, R * S, 1, C, st), "dw filter rsc->crs")
        return dx, dw, None, None, None, None


def depthwise_conv2d(x, weight, stride=1, padding=0, dilation=1, bn_stats=False, producer=None):
    """nn.Conv2d(C, C, k, groups=C).  bn_stats / producer: as for conv2d (BN statistics from the kernel's epilogue, pairing tag)."""
    _BN_FUSE["last"] = None
    y = _DepthwiseConv2dFn.apply(x, weight, int(stride), int(padding), int(dilation), bool(bn_stats) and _BN_FUSE["on"])
    return _tag_bn_stats(y, producer)


# --------------------------------------------------------------------------- transposed convolution 2x2 / stride 2
class _ConvTranspose2x2Fn(torch.autograd.Function):
    """nn.ConvTranspose2d(Cin, K, kernel_size=2, stride=2): a 1x1 convolution with 4K outputs on the MFMA
    path (column k*4 + r*2 + s) followed by depth_to_space (+bias); backward = space_to_depth + 1x1 dgrad/wgrad."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = to_nhwc(x, "conv_transpose2x2")
        _need_cuda(weight, "conv_transpose2x2")
        N, C, H, W = x.shape
        Cw, K, R, S = weight.shape
        if Cw != C or R != 2 or S != 2:
            raise SegmiError("conv_transpose2x2: expected weight [%d, K, 2, 2], got %s" % (C, tuple(weight.shape)))
        if (C & 3) or (K & 3):
            raise SegmiError("conv_transpose2x2: channel counts must be multiples of 4 (got %d -> %d)" % (C, K))
        st = _stream()
        K4 = 4 * K
        w2 = torch.empty(K4 * C, device=x.device, dtype=torch.float32)          # [(k,r,s), c]
        check(lib.segmi_nchw_to_nhwc(weight.contiguous().data_ptr(), w2.data_ptr(), 1, C, K4, 1, C, st), "convT filter")
        t = empty_nhwc(N, K4, H, W, x.device)
        d = ConvDesc(N, H, W, C, K4, 1, 1, H, W, 1, 0, 1, ld_of(x), ld_of(t))
        with span(lambda: conv_variant(d, 0), _conv_exec_flops(d, C, 0), _conv_bytes(d, C), detail=lambda: _geom(d), eff=_conv_flops(d, C)):
            nws = lib.segmi_conv2d_fwd_workspace(d)
            ws = workspace(nws, x.device) if nws else None
            check(lib.segmi_conv2d_fwd(d, x.data_ptr(), w2.data_ptr(), None, t.data_ptr(), 0, ws.data_ptr() if ws is not None else None, nws, st),
                  "convT conv2d_fwd")
        y = empty_nhwc(N, K, 2 * H, 2 * W, x.device)
        check(lib.segmi_depth_to_space2(t.data_ptr(), ld_of(t), y.data_ptr(), ld_of(y), bias.data_ptr() if bias is not None else None,
                                        N, H, W, K, st), "depth_to_space2")
        ctx.save_for_backward(x, w2)
        ctx.geom = (N, C, H, W, K)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        N, C, H, W, K = ctx.geom
        K4 = 4 * K
        dy = to_nhwc(dy, "conv_transpose2x2.backward")
        st, dev = _stream(), x.device
        g = empty_nhwc(N, K4, H, W, dev)
        check(lib.segmi_space_to_depth2(dy.data_ptr(), ld_of(dy), g.data_ptr(), ld_of(g), N, H, W, K, st), "space_to_depth2")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = torch.empty(C * K4, device=dev, dtype=torch.float32)
            check(lib.segmi_filter_krsc_to_crsk(w2.data_ptr(), wt.data_ptr(), K4, 1, 1, C, K4, st), "krsc_to_crsk")
            dx = empty_nhwc(N, C, H, W, dev)
            d = ConvDesc(N, H, W, C, K4, 1, 1, H, W, 1, 0, 1, ld_of(dx), ld_of(g))
            with span(lambda: conv_variant(d, 1), _conv_exec_flops(d, C, 1), _conv_bytes(d, C), detail=lambda: _geom(d), eff=_conv_flops(d, C)):
                check(lib.segmi_conv2d_dgrad(d, g.data_ptr(), wt.data_ptr(), dx.data_ptr(), 0, st), "convT conv2d_dgrad")
        if ctx.needs_input_grad[1]:
            d = ConvDesc(N, H, W, C, K4, 1, 1, H, W, 1, 0, 1, ld_of(x), ld_of(g))
            nws = lib.segmi_conv2d_wgrad_workspace(d)
            ws = workspace(nws, dev) if nws else None
            dw2 = torch.empty(K4 * C, device=dev, dtype=torch.float32)
            with span(lambda: conv_variant(d, 2), _conv_exec_flops(d, C, 2), _conv_bytes(d, C), detail=lambda: _geom(d), eff=_conv_flops(d, C)):
                check(lib.segmi_conv2d_wgrad(d, x.data_ptr(), g.data_ptr(), dw2.data_ptr(),
                                             ws.data_ptr() if ws is not None else None, nws, st), "convT conv2d_wgrad")
            dw = torch.empty((C, K, 2, 2), device=dev, dtype=torch.float32)
            check(lib.segmi_nhwc_to_nchw(dw2.data_ptr(), dw.data_ptr(), 1, C, K4, 1, C, st), "convT filter grad")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            rows = N * 4 * H * W
            nws = lib.segmi_colsum_workspace(rows, K)
            ws = workspace(nws, dev)
            db = torch.empty(K, device=dev, dtype=torch.float32)
            check(lib.segmi_colsum(dy.data_ptr(), ld_of(dy), rows, K, db.data_ptr(), ws.data_ptr(), nws, st), "colsum")
        return dx, dw, db


def conv_transpose2x2(x, weight, bias=None):
    """F.conv_transpose2d(x, weight, bias, stride=2) for 2x2 kernels (the U-Net up-convolution)."""
    return _ConvTranspose2x2Fn.apply(x, weight, bias)


# --------------------------------------------------------------------------- batch norm (+ReLU +residual)
def _producer_stats(x, C):
    """(partials, nparts) the producing convolution wrote for exactly this tensor (same version, matching width), else None."""
    st = getattr(x, "_segmi_bn_stats", None)
    if st is None or not _BN_FUSE["on"] or st[2] != x._version or st[0].numel() != st[1] * 3 * C:
        return None
    return st[0], st[1]


def _note_bn_consumer(x, batch_stats):
    """Pairing discovery: tell the module that produced `x` whether a batch-statistics BatchNorm consumes its output."""
    ref = getattr(x, "_segmi_producer", None)
    mod = ref() if ref is not None else None
    if mod is None:
        return
    if batch_stats:
        mod._bn_consumer = True
    elif torch.is_grad_enabled():
        # a BN on running statistics inside a grad-enabled forward = a FROZEN layer: its producer need not emit partials.  A
        # validation pass (model.eval() under torch.no_grad(), trainer.py:109-171) leaves the marks alone — clearing them made the
        # first training step after every validation take the separate statistics pass, i.e. the training trajectory depended
        # (at fp32 rounding level) on whether validation ran
        mod._bn_consumer = False


def _bn_coefficients(x, gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps, sync, pstats):
    """The statistics half of a BatchNorm layer on the NHWC tensor x: coef = mean | invstd | scale | shift | global count (4*C + 4
    floats) from the producer's partials, a pass over x, the gathered partials of all ranks (SyncBN) or the running statistics;
    running statistics updated like nn.BatchNorm2d.  Returns (coef, count) — count None when it lives on the device (SyncBN)."""
    N, C, H, W = x.shape
    rows = N * H * W
    dev, st = x.device, _stream()
    coef = torch.empty(4 * C + 4, device=dev, dtype=torch.float32)  # mean | invstd | scale | shift | global count (SyncBN)
    mean, invstd, scale, shift = (coef[i * C:(i + 1) * C] for i in range(4))
    gp = gamma.data_ptr() if gamma is not None else None
    bp = beta.data_ptr() if beta is not None else None
    count = float(rows)
    rm = running_mean.data_ptr() if running_mean is not None else None
    rv = running_var.data_ptr() if running_var is not None else None
    nbt = num_batches_tracked.data_ptr() if num_batches_tracked is not None else None
    if training and pstats is not None and (sync is not None or rows > 1):
        part, nparts = pstats
        nws = lib.segmi_bn_parts_workspace(nparts, C)
        ws = workspace(nws, dev)
        _BN_FUSE["consumed"] += 1
        if sync is None:
            check(lib.segmi_bn_finalize_from_parts(part.data_ptr(), nparts, C, gp, bp, eps, momentum, 0, rm, rv, nbt,
                                                   mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                   ws.data_ptr(), nws, st), "bn_finalize_from_parts")
        else:
            one = torch.empty(3 * C, device=dev, dtype=torch.float32)
            check(lib.segmi_bn_stats_from_parts(part.data_ptr(), nparts, C, one.data_ptr(), ws.data_ptr(), nws, st), "bn_stats_from_parts")
            allp, nall = sync.gather_stats(one)
            count = None
            check(lib.segmi_bn_finalize(allp.data_ptr(), nall, C, gp, bp, eps, momentum, sync.clamp_mode, rm, rv, nbt,
                                        mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                        coef.data_ptr() + 16 * C, st), "bn_finalize")
    elif training and sync is None:
        if rows <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
        nws = lib.segmi_bn_stats_workspace(rows, C)
        ws = workspace(nws, dev)
        check(lib.segmi_bn_stats_finalize(x.data_ptr(), ld_of(x), rows, C, gp, bp, eps, momentum, 0, rm, rv, nbt,
                                          mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                          ws.data_ptr(), nws, st), "bn_stats_finalize")
    elif training:
        nws = lib.segmi_bn_stats_workspace(rows, C)
        ws = workspace(nws, dev)
        part = torch.empty(3 * C, device=dev, dtype=torch.float32)
        check(lib.segmi_bn_stats(x.data_ptr(), ld_of(x), rows, C, part.data_ptr(), ws.data_ptr(), nws, st), "bn_stats")
        # one all-gather; the global element count is summed from the gathered partials ON THE DEVICE (coef[4C]) — no host-side
        # count exchange, so ragged / changing shard sizes cannot desynchronise the ranks' collectives
        part, nparts = sync.gather_stats(part)
        count = None
        check(lib.segmi_bn_finalize(part.data_ptr(), nparts, C, gp, bp, eps, momentum, sync.clamp_mode, rm, rv, nbt,
                                    mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                    coef.data_ptr() + 16 * C, st), "bn_finalize")
    else:
        check(lib.segmi_bn_eval_coeffs(running_mean.data_ptr(), running_var.data_ptr(), gp, bp, eps, C,
                                       mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), st),
              "bn_eval_coeffs")
    return coef, count


class _BatchNormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, num_batches_tracked, training, momentum,
                eps, relu, sync, pstats=None):
        x = to_nhwc(x, "batch_norm")
        N, C, H, W = x.shape
        if C & 3:
            raise SegmiError("batch_norm: channel count must be a multiple of 4 (got %d)" % C)
        rows = N * H * W
        dev, st = x.device, _stream()
        if residual is not None:
            residual = to_nhwc(residual, "batch_norm.residual")
        coef, count = _bn_coefficients(x, gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps, sync, pstats)
        scale, shift = coef[2 * C:3 * C], coef[3 * C:4 * C]
        y = empty_nhwc(N, C, H, W, dev)
        check(lib.segmi_bn_apply(x.data_ptr(), ld_of(x), residual.data_ptr() if residual is not None else None,
                                 ld_of(residual) if residual is not None else 0, y.data_ptr(), ld_of(y), rows, C,
                                 scale.data_ptr(), shift.data_ptr(), 1 if relu else 0, st), "bn_apply")
        # the ReLU mask is recomputed from x in backward unless a residual was added (then it needs the saved output)
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, coef)
        ctx.cfg = (training, relu, residual is not None, count, sync)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, coef = ctx.saved_tensors
        training, relu, has_res, count, sync = ctx.cfg
        N, C, H, W = x.shape
        rows = N * H * W
        dev, st = x.device, _stream()
        dy = to_nhwc(dy, "batch_norm.backward")
        mean, invstd, scale, shift = coef[0:C], coef[C:2 * C], coef[2 * C:3 * C], coef[3 * C:4 * C]
        sums = torch.empty(2 * C, device=dev, dtype=torch.float32)
        nws = lib.segmi_bn_bwd_reduce_workspace(rows, C)
        ws = workspace(nws, dev)
        yp, ldy = (y.data_ptr(), ld_of(y)) if y is not None else (None, 0)
        check(lib.segmi_bn_bwd_reduce(dy.data_ptr(), ld_of(dy), x.data_ptr(), ld_of(x), yp, ldy, rows, C,
                                      mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                      1 if relu else 0, sums.data_ptr(), ws.data_ptr(), nws, st), "bn_bwd_reduce")
        dgamma = sums[C:2 * C] if ctx.needs_input_grad[1] else None
        dbeta = sums[0:C] if ctx.needs_input_grad[2] else None
        gsums = sums
        if training and sync is not None:
            gsums = sync.reduce_sums(sums)  # global sums for dx; local ones stay the parameter grads
        dx = dres = None
        want_dx = ctx.needs_input_grad[0]
        want_res = has_res and ctx.needs_input_grad[3]
        if want_dx or (want_res and relu):
            dx = empty_nhwc(N, C, H, W, dev)
            if want_res and relu:
                dres = empty_nhwc(N, C, H, W, dev)
            check(lib.segmi_bn_bwd_apply(dy.data_ptr(), ld_of(dy), x.data_ptr(), ld_of(x), yp, ldy, rows, C,
                                         mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), gsums.data_ptr(),
                                         count if count is not None else 0.0, coef.data_ptr() + 16 * C if count is None else None,
                                         1 if relu else 0, 1 if training else 0, dx.data_ptr(), ld_of(dx),
                                         dres.data_ptr() if dres is not None else None,
                                         ld_of(dres) if dres is not None else 0, st), "bn_bwd_apply")
        if want_res and not relu:
            dres = dy
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None


class _BatchNormDepthwiseFn(torch.autograd.Function):
    """depthwise3x3(relu?(batch_norm(z))) as ONE node (round 6): the normalised tensor is never materialised — the depthwise kernels
    apply max(fmaf(z, scale, shift), 0) on every loaded tap (segmi_dwconv2d_fwd_pre / _wgrad_pre), bit-identical to
    segmi_bn_apply followed by the plain depthwise kernel.  Xception: the BatchNorm(+ReLU) between a pointwise convolution and the
    next SeparableConv2d (models/deeplabv3_plus.py:99-119, 225-232 of the reference).  Backward: depthwise data gradient -> the
    BatchNorm backward passes with the ReLU mask recomputed from z (as _BatchNormActFn does without a residual); the depthwise
    filter gradient reads z through the same fused load, on the filter-gradient side stream."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps, relu, pstats,
                weight, pad, dil, bn_stats):
        z = to_nhwc(z, "batch_norm_depthwise")
        _need_cuda(weight, "batch_norm_depthwise")
        N, C, H, W = z.shape
        dev, st = z.device, _stream()
        coef, count = _bn_coefficients(z, gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps, None, pstats)
        wrsc = _dw_rsc_view(weight.detach())
        y = empty_nhwc(N, C, H, W, dev)
        d = ConvDesc(N, H, W, C, C, 3, 3, H, W, 1, pad, dil, ld_of(z), ld_of(y))
        parts = lib.segmi_dwconv2d_fwd_stats_parts(d) if bn_stats else 0
        part = torch.empty(parts * 3 * C, device=dev, dtype=torch.float32) if parts > 0 else None
        check(lib.segmi_dwconv2d_fwd_pre(d, z.data_ptr(), coef.data_ptr() + 8 * C, coef.data_ptr() + 12 * C, 1 if relu else 0, wrsc.data_ptr(),
                                         y.data_ptr(), part.data_ptr() if part is not None else None, st), "dwconv2d_fwd_pre")
        if part is not None:
            _BN_FUSE["last"] = (part, parts)
            _BN_FUSE["emitted"] += 1
        ctx.save_for_backward(z, coef, wrsc, weight)
        ctx.cfg = (training, relu, count, pad, dil)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, coef, wrsc, weight = ctx.saved_tensors
        training, relu, count, pad, dil = ctx.cfg
        N, C, H, W = z.shape
        rows = N * H * W
        dev, st = z.device, _stream()
        dy = to_nhwc(dy, "batch_norm_depthwise.backward")
        dw = dz = dgamma = dbeta = None
        if ctx.needs_input_grad[11]:
            d = ConvDesc(N, H, W, C, C, 3, 3, H, W, 1, pad, dil, ld_of(z), ld_of(dy))
            slot = _take_grad_slot(weight) if _GRAD_SLOTS else None
            dwr = slot if slot is not None else torch.empty(9 * C, device=dev, dtype=torch.float32)

            def run_wgrad():
                nws = lib.segmi_dwconv2d_wgrad_workspace(d)
                ws = workspace(nws, dev)             # (keyed by the stream the call runs on)
                check(lib.segmi_dwconv2d_wgrad_pre(d, z.data_ptr(), coef.data_ptr() + 8 * C, coef.data_ptr() + 12 * C, 1 if relu else 0,
                                                   dy.data_ptr(), dwr.data_ptr(), ws.data_ptr(), nws, _stream()), "dwconv2d_wgrad_pre")

            if _DW_WGRAD_SIDE:
                _on_wgrad_stream(weight, (z, dy, dwr, coef), run_wgrad)
            else:
                run_wgrad()
            dw = slot if slot is not None else dwr.view(3, 3, C).permute(2, 0, 1).unsqueeze(1)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            da = empty_nhwc(N, C, H, W, dev)          # gradient with respect to the (never materialised) normalised tensor
            d = ConvDesc(N, H, W, C, C, 3, 3, H, W, 1, pad, dil, ld_of(da), ld_of(dy))
            check(lib.segmi_dwconv2d_dgrad(d, dy.data_ptr(), wrsc.data_ptr(), da.data_ptr(), st), "dwconv2d_dgrad")
            mean, invstd, scale, shift = (coef.data_ptr() + 4 * i * C for i in range(4))
            sums = torch.empty(2 * C, device=dev, dtype=torch.float32)
            nws = lib.segmi_bn_bwd_reduce_workspace(rows, C)
            ws = workspace(nws, dev)
            check(lib.segmi_bn_bwd_reduce(da.data_ptr(), ld_of(da), z.data_ptr(), ld_of(z), None, 0, rows, C, mean, invstd, scale, shift,
                                          1 if relu else 0, sums.data_ptr(), ws.data_ptr(), nws, st), "bn_bwd_reduce")
            dgamma = sums[C:2 * C] if ctx.needs_input_grad[1] else None
            dbeta = sums[0:C] if ctx.needs_input_grad[2] else None
            if ctx.needs_input_grad[0]:
                dz = empty_nhwc(N, C, H, W, dev)
                check(lib.segmi_bn_bwd_apply(da.data_ptr(), ld_of(da), z.data_ptr(), ld_of(z), None, 0, rows, C, mean, invstd, scale, shift,
                                             sums.data_ptr(), count, None, 1 if relu else 0, 1 if training else 0, dz.data_ptr(), ld_of(dz),
                                             None, 0, st), "bn_bwd_apply")
        return dz, dgamma, dbeta, None, None, None, None, None, None, None, None, dw, None, None, None


_DW_BN_FUSION = os.environ.get("SEGMI_DW_FUSED_BN", "1") == "1"      # A/B switch


def batch_norm_depthwise_ok(bn, conv):
    """Can `conv(relu?(bn(z)))` run as the fused node?  bn: a local (not synchronized) BatchNorm2d; conv: a depthwise 3x3 stride-1
    pad == dil in {1, 2} segmi.nn.Conv2d whose filter lives tap-major (what the module stores)."""
    if not _DW_BN_FUSION or getattr(bn, "sync", None) is not None or bn.momentum is None or not getattr(conv, "depthwise", False):
        return False
    if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.padding != conv.dilation or conv.dilation[0] not in (1, 2):
        return False
    if conv.in_channels != bn.num_features or (conv.in_channels & 3) or not (bn.training or bn.track_running_stats):
        return False
    return _dw_rsc_view(conv.weight.detach()) is not None


def batch_norm_depthwise(z, bn, conv, relu=True, bn_stats=False):
    """conv(relu?(bn(z))) — see _BatchNormDepthwiseFn; the caller checked batch_norm_depthwise_ok(bn, conv)."""
    training = bn.training or not bn.track_running_stats
    _note_bn_consumer(z, training)
    pstats = _producer_stats(z, z.shape[1]) if (training and z.dim() == 4) else None
    _BN_FUSE["last"] = None
    y = _BatchNormDepthwiseFn.apply(z, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                                    bn.running_var if bn.track_running_stats else None,
                                    bn.num_batches_tracked if (bn.track_running_stats and bn.training) else None,
                                    bool(training), float(bn.momentum), float(bn.eps), bool(relu), pstats,
                                    conv.weight, int(conv.padding[0]), int(conv.dilation[0]), bool(bn_stats) and _BN_FUSE["on"])
    return _tag_bn_stats(y, conv)


def batch_norm_act(x, gamma, beta, running_mean, running_var, num_batches_tracked=None, residual=None, training=True,
                   momentum=0.1, eps=1e-5, relu=False, sync=None):
    """BN (batch or running statistics) -> (+ residual) -> (ReLU), one fused apply pass.  Batch statistics come from the
    producing convolution's epilogue when it emitted them for this tensor (see _BN_FUSE), else from a pass over x."""
    _note_bn_consumer(x, training)
    pstats = _producer_stats(x, x.shape[1]) if (training and x.dim() == 4) else None
    return _BatchNormActFn.apply(x, gamma, beta, residual, running_mean, running_var, num_batches_tracked,
                                 bool(training), float(momentum), float(eps), bool(relu), sync, pstats)


# --------------------------------------------------------------------------- SyncBN layers that share their collectives
# Exact SyncBN needs a layer's GLOBAL statistics before the layer can be applied, so its two collectives (all-gather of the
# Welford partials forward, all-reduce of {sum dy, sum dy*xhat} backward) cannot be deferred — but layers whose INPUTS are ready
# together can share them: the four pyramid-stage BNs of _PSPModule (models/pspnet.py:25-31) and, in every residual block with a
# projection, bn3 + the downsample BN (models/resnet.py:137-145).  The second pair is possible in backward as well because the
# gradient that reaches the projection branch is dy * [y > 0] — it does not depend on bn3's statistics — so both reductions run
# before the ONE all-reduce.  Same kernels, same operands, same summation order as the one-layer path: results are bit-identical
# to it (tests/test_distributed_gpu.py); 122 -> 108 collectives per PSPNet-R50 step (reference: one master/slave exchange per
# layer and direction, utils/sync_batchnorm/batchnorm.py:105-126).
def _bn_member_stats(x, pstats=None):
    N, C, H, W = x.shape
    rows, dev = N * H * W, x.device
    if pstats is not None:
        nws = lib.segmi_bn_parts_workspace(pstats[1], C)
        ws = workspace(nws, dev)
        part = torch.empty(3 * C, device=dev, dtype=torch.float32)
        check(lib.segmi_bn_stats_from_parts(pstats[0].data_ptr(), pstats[1], C, part.data_ptr(), ws.data_ptr(), nws, _stream()), "bn_stats_from_parts")
        _BN_FUSE["consumed"] += 1
        return part
    nws = lib.segmi_bn_stats_workspace(rows, C)
    ws = workspace(nws, dev)
    part = torch.empty(3 * C, device=dev, dtype=torch.float32)
    check(lib.segmi_bn_stats(x.data_ptr(), ld_of(x), rows, C, part.data_ptr(), ws.data_ptr(), nws, _stream()), "bn_stats")
    return part


def _bn_member_finalize(parts, nparts, C, gamma, beta, rm, rv, nbt, eps, momentum, clamp_mode, dev):
    coef = torch.empty(4 * C + 4, device=dev, dtype=torch.float32)    # mean | invstd | scale | shift | global count
    check(lib.segmi_bn_finalize(parts.data_ptr(), nparts, C, gamma.data_ptr() if gamma is not None else None,
                                beta.data_ptr() if beta is not None else None, eps, momentum, clamp_mode,
                                rm.data_ptr() if rm is not None else None, rv.data_ptr() if rv is not None else None,
                                nbt.data_ptr() if nbt is not None else None, coef.data_ptr(), coef.data_ptr() + 4 * C,
                                coef.data_ptr() + 8 * C, coef.data_ptr() + 12 * C, coef.data_ptr() + 16 * C, _stream()), "bn_finalize")
    return coef


def _bn_member_apply(x, residual, coef, relu):
    N, C, H, W = x.shape
    y = empty_nhwc(N, C, H, W, x.device)
    check(lib.segmi_bn_apply(x.data_ptr(), ld_of(x), residual.data_ptr() if residual is not None else None,
                             ld_of(residual) if residual is not None else 0, y.data_ptr(), ld_of(y), N * H * W, C,
                             coef.data_ptr() + 8 * C, coef.data_ptr() + 12 * C, 1 if relu else 0, _stream()), "bn_apply")
    return y


def _bn_member_bwd_reduce(dy, x, y, coef, relu):
    N, C, H, W = x.shape
    rows, dev = N * H * W, x.device
    sums = torch.empty(2 * C, device=dev, dtype=torch.float32)
    nws = lib.segmi_bn_bwd_reduce_workspace(rows, C)
    ws = workspace(nws, dev)
    yp, ldy = (y.data_ptr(), ld_of(y)) if y is not None else (None, 0)
    check(lib.segmi_bn_bwd_reduce(dy.data_ptr(), ld_of(dy), x.data_ptr(), ld_of(x), yp, ldy, rows, C, coef.data_ptr(),
                                  coef.data_ptr() + 4 * C, coef.data_ptr() + 8 * C, coef.data_ptr() + 12 * C, 1 if relu else 0,
                                  sums.data_ptr(), ws.data_ptr(), nws, _stream()), "bn_bwd_reduce")
    return sums


def _bn_member_bwd_apply(dy, x, y, coef, gsums, relu, want_dres):
    N, C, H, W = x.shape
    dev = x.device
    dx = empty_nhwc(N, C, H, W, dev)
    dres = empty_nhwc(N, C, H, W, dev) if want_dres else None
    yp, ldy = (y.data_ptr(), ld_of(y)) if y is not None else (None, 0)
    check(lib.segmi_bn_bwd_apply(dy.data_ptr(), ld_of(dy), x.data_ptr(), ld_of(x), yp, ldy, N * H * W, C, coef.data_ptr(),
                                 coef.data_ptr() + 4 * C, coef.data_ptr() + 8 * C, coef.data_ptr() + 12 * C, gsums.data_ptr(),
                                 0.0, coef.data_ptr() + 16 * C, 1 if relu else 0, 1, dx.data_ptr(), ld_of(dx),
                                 dres.data_ptr() if dres is not None else None, ld_of(dres) if dres is not None else 0, _stream()),
          "bn_bwd_apply")
    return dx, dres


class _SyncBNGroupFn(torch.autograd.Function):
    """G independent SyncBN(+ReLU) layers, training mode, ONE all-gather forward and ONE all-reduce backward for all of them.
    args: hyper = [(eps, momentum, relu, sync)] * G, then per member x, gamma, beta, running_mean, running_var, num_batches_tracked."""

    @staticmethod
    def forward(ctx, hyper, *args):
        G = len(hyper)
        mem = [args[6 * i:6 * i + 6] for i in range(G)]
        xs = [to_nhwc(m[0], "sync_bn_group") for m in mem]
        for x in xs:
            if x.shape[1] & 3:
                raise SegmiError("batch_norm: channel count must be a multiple of 4 (got %d)" % x.shape[1])
        parts = [_bn_member_stats(x, h[4] if len(h) > 4 else None) for x, h in zip(xs, hyper)]
        gathered = hyper[0][3].gather_stats_many(parts)
        coefs, ys = [], []
        for x, (_, g, b, rm, rv, nbt), (eps, mom, relu, sync, *_), (pall, npart) in zip(xs, mem, hyper, gathered):
            coef = _bn_member_finalize(pall, npart, x.shape[1], g, b, rm, rv, nbt, eps, mom, sync.clamp_mode, x.device)
            coefs.append(coef)
            ys.append(_bn_member_apply(x, None, coef, relu))
        ctx.save_for_backward(*xs, *coefs)
        ctx.hyper = [h[:4] for h in hyper]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        hyper = ctx.hyper
        G = len(hyper)
        xs, coefs = ctx.saved_tensors[:G], ctx.saved_tensors[G:]
        dys = [to_nhwc(dy, "sync_bn_group.backward") for dy in dys]
        sums = [_bn_member_bwd_reduce(dy, x, None, coef, h[2]) for dy, x, coef, h in zip(dys, xs, coefs, hyper)]
        gsums = hyper[0][3].reduce_sums_many(sums)
        out = [None]
        for i, (dy, x, coef, h, s, gs) in enumerate(zip(dys, xs, coefs, hyper, sums, gsums)):
            C = x.shape[1]
            dx = _bn_member_bwd_apply(dy, x, None, coef, gs, h[2], False)[0] if ctx.needs_input_grad[1 + 6 * i] else None
            out += [dx, s[C:2 * C] if ctx.needs_input_grad[2 + 6 * i] else None, s[0:C] if ctx.needs_input_grad[3 + 6 * i] else None,
                    None, None, None]
        return tuple(out)


class _SyncBNResidualTailFn(torch.autograd.Function):
    """y = relu(bn_main(x_main) + bn_proj(x_proj)): the tail of a residual block with a projection shortcut
    (models/resnet.py:137-145: bn3(conv3(.)) + downsample(x)), both SyncBN layers sharing one all-gather and one all-reduce.
    args: hyper = [(eps, momentum, sync)] * 2 (main, proj), then x, gamma, beta, rm, rv, nbt for main and for proj."""

    @staticmethod
    def forward(ctx, hyper, *args):
        mem = [args[0:6], args[6:12]]
        xm, xp = to_nhwc(mem[0][0], "sync_bn_tail"), to_nhwc(mem[1][0], "sync_bn_tail")
        if xm.shape != xp.shape or (xm.shape[1] & 3):
            raise SegmiError("sync_bn_tail: main and projection branch must have the same shape, channels a multiple of 4")
        parts = [_bn_member_stats(xm, hyper[0][3] if len(hyper[0]) > 3 else None), _bn_member_stats(xp, hyper[1][3] if len(hyper[1]) > 3 else None)]
        gathered = hyper[0][2].gather_stats_many(parts)
        coefs = []
        for x, (_, g, b, rm, rv, nbt), (eps, mom, sync, *_), (pall, npart) in zip((xm, xp), mem, hyper, gathered):
            coefs.append(_bn_member_finalize(pall, npart, x.shape[1], g, b, rm, rv, nbt, eps, mom, sync.clamp_mode, x.device))
        ident = _bn_member_apply(xp, None, coefs[1], False)
        y = _bn_member_apply(xm, ident, coefs[0], True)
        ctx.save_for_backward(xm, xp, y, coefs[0], coefs[1])
        ctx.hyper = [h[:3] for h in hyper]
        return y

    @staticmethod
    def backward(ctx, dy):
        xm, xp, y, cm, cp = ctx.saved_tensors
        hyper = ctx.hyper
        dy = to_nhwc(dy, "sync_bn_tail.backward")
        C = xm.shape[1]
        # the gradient reaching the projection branch is dy * [y > 0] whatever bn_main's statistics are: both reductions (the
        # projection's with the ReLU mask of the block output) run before the one all-reduce
        sm = _bn_member_bwd_reduce(dy, xm, y, cm, True)
        sp = _bn_member_bwd_reduce(dy, xp, y, cp, True)
        gm, gp = hyper[0][2].reduce_sums_many([sm, sp])
        dxm, dres = _bn_member_bwd_apply(dy, xm, y, cm, gm, True, True)
        dxp = _bn_member_bwd_apply(dres, xp, None, cp, gp, False, False)[0]
        ni = ctx.needs_input_grad
        return (None, dxm if ni[1] else None, sm[C:2 * C] if ni[2] else None, sm[0:C] if ni[3] else None, None, None, None,
                dxp if ni[7] else None, sp[C:2 * C] if ni[8] else None, sp[0:C] if ni[9] else None, None, None, None)


def sync_groupable(bns):
    """All layers are SyncBN layers in training mode of one multi-rank process group (else the one-layer path applies)."""
    first = getattr(bns[0], "sync", None)
    if first is None or (first.world <= 1 and not first.force_group):
        return False
    return all(getattr(b, "sync", None) is not None and b.sync.group is first.group and b.training and b.track_running_stats
               and b.momentum is not None for b in bns)


def sync_batch_norm_group(xs, bns, relu=True):
    """[bn(x) (+ReLU) for x, bn in zip(xs, bns)] with the SyncBN collectives of all layers batched into one all-gather (forward)
    and one all-reduce (backward); falls back to layer-by-layer calls when the layers are not synchronized."""
    if not sync_groupable(bns):
        return [bn(x, relu=relu) for x, bn in zip(xs, bns)]
    for x in xs:
        _note_bn_consumer(x, True)
    hyper = [(float(b.eps), float(b.momentum), bool(relu), b.sync, _producer_stats(x, x.shape[1])) for x, b in zip(xs, bns)]
    args = []
    for x, b in zip(xs, bns):
        args += [x, b.weight, b.bias, b.running_mean, b.running_var, b.num_batches_tracked]
    return list(_SyncBNGroupFn.apply(hyper, *args))


def sync_batch_norm_residual_tail(x_main, bn_main, x_proj, bn_proj):
    """relu(bn_main(x_main) + bn_proj(x_proj)) — one all-gather / one all-reduce for the two SyncBN layers; None when the layers
    are not synchronized (the caller then takes the one-layer path)."""
    if not sync_groupable([bn_main, bn_proj]):
        return None
    for x in (x_main, x_proj):
        _note_bn_consumer(x, True)
    hyper = [(float(b.eps), float(b.momentum), b.sync, _producer_stats(x, x.shape[1])) for x, b in ((x_main, bn_main), (x_proj, bn_proj))]
    args = []
    for x, b in ((x_main, bn_main), (x_proj, bn_proj)):
        args += [x, b.weight, b.bias, b.running_mean, b.running_var, b.num_batches_tracked]
    return _SyncBNResidualTailFn.apply(hyper, *args)


# --------------------------------------------------------------------------- relu / add
class _ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = to_nhwc(x, "relu")
        N, C, H, W = x.shape
        y = empty_nhwc(N, C, H, W, x.device)
        check(lib.segmi_relu_fwd(x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), N * H * W, C, _stream()), "relu_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = to_nhwc(dy, "relu.backward")
        N, C, H, W = y.shape
        dx = empty_nhwc(N, C, H, W, y.device)
        check(lib.segmi_relu_bwd(dy.data_ptr(), ld_of(dy), y.data_ptr(), ld_of(y), dx.data_ptr(), ld_of(dx), N * H * W, C,
                                 _stream()), "relu_bwd")
        return dx


def relu(x):
    return _ReluFn.apply(x)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = to_nhwc(a, "add"), to_nhwc(b, "add")
        if a.shape != b.shape:
            raise SegmiError("add: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        N, C, H, W = a.shape
        o = empty_nhwc(N, C, H, W, a.device)
        check(lib.segmi_add(a.data_ptr(), ld_of(a), b.data_ptr(), ld_of(b), o.data_ptr(), ld_of(o), N * H * W, C, _stream()), "add")
        return o

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _AddFn.apply(a, b)


# --------------------------------------------------------------------------- pooling
def pool_out_size(H, k, stride, pad, ceil_mode):
    if ceil_mode:
        o = -((H + 2 * pad - k) // -stride) + 1
        if (o - 1) * stride >= H + pad:  # last window must start inside the input (or left padding)
            o -= 1
        return o
    return (H + 2 * pad - k) // stride + 1


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad, ceil_mode):
        x = to_nhwc(x, "max_pool2d")
        N, C, H, W = x.shape
        P, Q = pool_out_size(H, k, stride, pad, ceil_mode), pool_out_size(W, k, stride, pad, ceil_mode)
        y = empty_nhwc(N, C, P, Q, x.device)
        idx = torch.empty((N * P * Q, pad4(C)), device=x.device, dtype=torch.uint8)
        check(lib.segmi_maxpool_fwd(x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), idx.data_ptr(), N, H, W, C, P, Q, k,
                                    stride, pad, _stream()), "maxpool_fwd")
        ctx.save_for_backward(idx)
        ctx.geom = (N, C, H, W, P, Q, k, stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, C, H, W, P, Q, k, stride, pad = ctx.geom
        dy = to_nhwc(dy, "max_pool2d.backward")
        dx = empty_nhwc(N, C, H, W, dy.device)
        check(lib.segmi_maxpool_bwd(dy.data_ptr(), ld_of(dy), idx.data_ptr(), dx.data_ptr(), ld_of(dx), N, H, W, C, P, Q, k,
                                    stride, pad, _stream()), "maxpool_bwd")
        return dx, None, None, None, None


def max_pool2d(x, kernel_size, stride=None, padding=0, ceil_mode=False):
    return _MaxPoolFn.apply(x, int(kernel_size), int(stride or kernel_size), int(padding), bool(ceil_mode))


class _AdaptiveAvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, OH, OW):
        x = to_nhwc(x, "adaptive_avg_pool2d")
        N, C, H, W = x.shape
        y = empty_nhwc(N, C, OH, OW, x.device)
        check(lib.segmi_adaptive_avgpool_fwd(x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), N, H, W, C, OH, OW, _stream()),
              "adaptive_avgpool_fwd")
        ctx.geom = (N, C, H, W, OH, OW)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W, OH, OW = ctx.geom
        dy = to_nhwc(dy, "adaptive_avg_pool2d.backward")
        dx = empty_nhwc(N, C, H, W, dy.device)
        check(lib.segmi_adaptive_avgpool_bwd(dy.data_ptr(), ld_of(dy), dx.data_ptr(), ld_of(dx), N, H, W, C, OH, OW, 0,
                                             _stream()), "adaptive_avgpool_bwd")
        return dx, None, None


def adaptive_avg_pool2d(x, output_size):
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    return _AdaptiveAvgPoolFn.apply(x, int(output_size[0]), int(output_size[1]))


class _PyramidPoolFn(torch.autograd.Function):
    """(AdaptiveAvgPool2d(b)(x) for b in bins) in one read of x; backward writes dx once (see segmi_pyramid_pool_*)."""

    @staticmethod
    def forward(ctx, x, *bins):
        x = to_nhwc(x, "pyramid_pool")
        N, C, H, W = x.shape
        nl = len(bins)
        barr = (ctypes.c_int * nl)(*bins)
        nws = lib.segmi_pyramid_pool_workspace(N, H, W, C, nl, barr)
        if nws == 0:
            raise SegmiError("pyramid_pool: unsupported bins %s for a %dx%d map (<= 4 levels, bins <= 8)" % (bins, H, W))
        ws = workspace(nws, x.device)
        ys = [empty_nhwc(N, C, b, b, x.device) for b in bins]
        yp = (ctypes.c_void_p * nl)(*[y.data_ptr() for y in ys])
        ld = (ctypes.c_int * nl)(*[ld_of(y) for y in ys])
        check(lib.segmi_pyramid_pool_fwd(x.data_ptr(), ld_of(x), N, H, W, C, nl, barr, yp, ld, ws.data_ptr(), nws, _stream()), "pyramid_pool_fwd")
        ctx.geom = (N, C, H, W, tuple(bins))
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        N, C, H, W, bins = ctx.geom
        nl = len(bins)
        dev = next(d for d in dys if d is not None).device
        dys = [to_nhwc(d, "pyramid_pool.backward") if d is not None else torch.zeros((N, b, b, pad4(C)), device=dev).permute(0, 3, 1, 2)[:, :C]
               for d, b in zip(dys, bins)]
        dx = empty_nhwc(N, C, H, W, dev)
        dp = (ctypes.c_void_p * nl)(*[d.data_ptr() for d in dys])
        ld = (ctypes.c_int * nl)(*[ld_of(d) for d in dys])
        barr = (ctypes.c_int * nl)(*bins)
        nws = lib.segmi_pyramid_pool_workspace(N, H, W, C, nl, barr)
        ws = workspace(nws + 16, dev)
        check(lib.segmi_pyramid_pool_bwd(dp, ld, dx.data_ptr(), ld_of(dx), N, H, W, C, nl, barr, (ws.data_ptr() + 15) & ~15, nws, _stream()),
              "pyramid_pool_bwd")
        return (dx,) + (None,) * nl


def pyramid_pool(x, bins):
    """[F.adaptive_avg_pool2d(x, b) for b in bins] — the PSP pyramid (models/pspnet.py:25-37) — fused."""
    return _PyramidPoolFn.apply(x, *[int(b) for b in bins])


# --------------------------------------------------------------------------- factored PSP bottleneck
def _conv_call(kind, d, C, *args):
    """One libsegmi convolution launch wrapped in a roofline span (kind 0 fwd, 1 dgrad, 2 wgrad)."""
    fn = (lib.segmi_conv2d_fwd, lib.segmi_conv2d_dgrad, lib.segmi_conv2d_wgrad)[kind]
    with span(lambda: conv_variant(d, kind), _conv_exec_flops(d, C, kind), _conv_bytes(d, C), detail=lambda: _geom(d), eff=_conv_flops(d, C)):
        check(fn(d, *args), ("conv2d_fwd", "conv2d_dgrad", "conv2d_wgrad")[kind])


class _PyramidBottleneckFn(torch.autograd.Function):
    """y = conv3x3(cat([x, up(p_1), ..., up(p_L)], 1), weight, padding=1) with up = bilinear(align_corners=True) to x's size —
    the PSP module's concat + bottleneck convolution (models/pspnet.py:32-38) — evaluated in factored form: the convolution
    proper runs over x's channels only; each pyramid branch contributes through T_l = p_l (x) W[:, slice_l] (a 1x1 convolution
    with 9K output channels on the b x b map) and a separable bilinear assembly (csrc/pyramid_bottleneck.hip).  Backward is the
    exact transpose: dgrad / wgrad over x's channels, dy reduced to G_l = dT_l, then dgrad / wgrad of the 1x1 convolutions."""

    @staticmethod
    def forward(ctx, x, weight, *ps):
        x = to_nhwc(x, "pyramid_bottleneck")
        _need_cuda(weight, "pyramid_bottleneck")
        N, Cx, H, W = x.shape
        K, Ct, R, S = weight.shape
        ps = [to_nhwc(p, "pyramid_bottleneck") for p in ps]
        cs = [p.shape[1] for p in ps]
        bins = [(int(p.shape[2]), int(p.shape[3])) for p in ps]          # (rows, columns) of every low-resolution map
        if (R, S) != (3, 3) or Ct != Cx + sum(cs) or not weight.is_contiguous(memory_format=torch.channels_last) or (weight.data_ptr() & 15):
            raise SegmiError("pyramid_bottleneck: needs a channels_last 3x3 filter over %d + %s channels, got %s" % (Cx, cs, tuple(weight.shape)))
        if (Cx & 3) or (K & 3) or any(c & 3 for c in cs) or any(ld_of(p) != p.shape[1] for p in ps) or len(ps) > 4:
            raise SegmiError("pyramid_bottleneck: channel counts must be multiples of 4, the low-resolution maps dense, at most 4 of them")
        dev, st = x.device, _stream()
        nl = len(ps)
        bh, bw = (ctypes.c_int * nl)(*[b[0] for b in bins]), (ctypes.c_int * nl)(*[b[1] for b in bins])
        # contiguous operands: the x-channel slice as a KRSC filter, each pyramid slice as a 1x1 filter with rows (rs, k)
        fx = torch.empty(K * 9 * Cx, device=dev, dtype=torch.float32)
        check(lib.segmi_filter_slice(weight.data_ptr(), K, 9, Ct, 0, Cx, 0, fx.data_ptr(), st), "filter_slice")
        y = empty_nhwc(N, K, H, W, dev)
        Ts, c0 = [], Cx
        for p, c, (b, b2) in zip(ps, cs, bins):
            fs = torch.empty(9 * K * c, device=dev, dtype=torch.float32)
            check(lib.segmi_filter_slice(weight.data_ptr(), K, 9, Ct, c0, c, 1, fs.data_ptr(), st), "filter_slice")
            t = torch.empty((N, b, b2, 9 * K), device=dev, dtype=torch.float32)
            d = ConvDesc(N, b, b2, c, 9 * K, 1, 1, b, b2, 1, 0, 1, c, 9 * K)
            _conv_fwd(d, c, p, fs, None, t)
            Ts.append(t)
            c0 += c
        nws = lib.segmi_pyramid_up_workspace(N, H, W, K, nl, bh, bw)
        ws = workspace(nws + 16, dev)
        tp = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in Ts])
        check(lib.segmi_pyramid_up_fwd(tp, N, H, W, K, nl, bh, bw, y.data_ptr(), ld_of(y), (ws.data_ptr() + 15) & ~15, nws, st), "pyramid_up_fwd")
        d = ConvDesc(N, H, W, Cx, K, 3, 3, H, W, 1, 1, 1, ld_of(x), ld_of(y))
        v = _conv_fwd(d, Cx, x, fx, None, y, accumulate=1, keep_v=ctx.needs_input_grad[1] and _FWD["grad"])             # accumulate onto the pyramid part
        ctx.has_v = v is not None
        ctx.save_for_backward(x, weight, *ps, *([v] if v is not None else []))
        ctx.geom = (N, Cx, H, W, K, Ct, tuple(cs), tuple(bins))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, *ps = ctx.saved_tensors
        v = ps.pop() if ctx.has_v else None
        N, Cx, H, W, K, Ct, cs, bins = ctx.geom
        dy = to_nhwc(dy, "pyramid_bottleneck.backward")
        dev, st = x.device, _stream()
        nl = len(ps)
        bh, bw = (ctypes.c_int * nl)(*[b[0] for b in bins]), (ctypes.c_int * nl)(*[b[1] for b in bins])
        # KRSC memory, filled slice by slice; under a data-parallel reducer this is the parameter's slot in the all-reduce bucket
        dwb = _take_grad_slot(weight) if (_GRAD_SLOTS and ctx.needs_input_grad[1]) else None
        if dwb is None:
            dwb = torch.empty((K, Ct, 3, 3), device=dev, dtype=torch.float32, memory_format=torch.channels_last)
        fx = torch.empty(K * 9 * Cx, device=dev, dtype=torch.float32)
        check(lib.segmi_filter_slice(weight.data_ptr(), K, 9, Ct, 0, Cx, 0, fx.data_ptr(), st), "filter_slice")
        # ---- feature channels: plain dgrad / wgrad of the 3x3 convolution over Cx channels
        dx = None
        if ctx.needs_input_grad[0]:
            wt = torch.empty(Cx * 9 * K, device=dev, dtype=torch.float32)
            check(lib.segmi_filter_krsc_to_crsk(fx.data_ptr(), wt.data_ptr(), K, 3, 3, Cx, K, st), "krsc_to_crsk")
            dx = empty_nhwc(N, Cx, H, W, dev)
            d = ConvDesc(N, H, W, Cx, K, 3, 3, H, W, 1, 1, 1, ld_of(dx), ld_of(dy))
            _conv_dgrad(d, Cx, dy, wt, dx)
        d = ConvDesc(N, H, W, Cx, K, 3, 3, H, W, 1, 1, 1, ld_of(x), ld_of(dy))
        dfx = torch.empty(K * 9 * Cx, device=dev, dtype=torch.float32)

        def feature_wgrad():               # the largest filter gradient of the model: eligible for the side stream like any conv's
            _conv_wgrad(d, Cx, x, dy, dfx, v)
            check(lib.segmi_filter_unslice(dfx.data_ptr(), K, 9, Ct, 0, Cx, 0, dwb.data_ptr(), _stream()), "filter_unslice")
        _on_wgrad_stream(weight, (x, dy, dfx, dwb, v), feature_wgrad)
        # ---- pyramid branches: G_l = dT_l by the transposed interpolation, then the 1x1 convolution's dgrad / wgrad
        Gs = [torch.empty((N, b, b2, 9 * K), device=dev, dtype=torch.float32) for b, b2 in bins]
        nws = lib.segmi_pyramid_up_workspace(N, H, W, K, nl, bh, bw)
        ws = workspace(nws + 16, dev)
        gp = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in Gs])
        check(lib.segmi_pyramid_up_bwd(dy.data_ptr(), ld_of(dy), N, H, W, K, nl, bh, bw, gp, (ws.data_ptr() + 15) & ~15, nws, st), "pyramid_up_bwd")
        dps, c0 = [], Cx
        for li, (p, c, (b, b2), g) in enumerate(zip(ps, cs, bins, Gs)):
            fs = torch.empty(9 * K * c, device=dev, dtype=torch.float32)
            check(lib.segmi_filter_slice(weight.data_ptr(), K, 9, Ct, c0, c, 1, fs.data_ptr(), st), "filter_slice")
            dp = None
            if ctx.needs_input_grad[2 + li]:
                wt = torch.empty(c * 9 * K, device=dev, dtype=torch.float32)
                check(lib.segmi_filter_krsc_to_crsk(fs.data_ptr(), wt.data_ptr(), 9 * K, 1, 1, c, 9 * K, st), "krsc_to_crsk")
                dp = empty_nhwc(N, c, b, b2, dev)
                # dp = G x F as a FORWARD 1x1 convolution over G's 9K channels (filter = F transposed): the forward kernel splits
                # the 4608-long reduction over the chip, the dgrad entry would walk it in 4 workgroups
                d = ConvDesc(N, b, b2, 9 * K, c, 1, 1, b, b2, 1, 0, 1, 9 * K, ld_of(dp))
                _conv_fwd(d, 9 * K, g, wt, None, dp)
            dps.append(dp)
            d = ConvDesc(N, b, b2, c, 9 * K, 1, 1, b, b2, 1, 0, 1, ld_of(p), 9 * K)
            nws = lib.segmi_conv2d_wgrad_workspace(d)
            ws = workspace(nws, dev) if nws else None
            dfs = torch.empty(9 * K * c, device=dev, dtype=torch.float32)
            _conv_call(2, d, c, p.data_ptr(), g.data_ptr(), dfs.data_ptr(), ws.data_ptr() if ws is not None else None, nws, st)
            check(lib.segmi_filter_unslice(dfs.data_ptr(), K, 9, Ct, c0, c, 1, dwb.data_ptr(), st), "filter_unslice")
            c0 += c
        dw = dwb if ctx.needs_input_grad[1] else None
        return (dx, dw) + tuple(dps)


def pyramid_bottleneck_conv(x, pyramid, weight):
    """F.conv2d(torch.cat([x] + [F.interpolate(p, x.shape[2:], mode='bilinear', align_corners=True) for p in pyramid], 1), weight,
    padding=1) without building the upsampled maps or the concatenation (see _PyramidBottleneckFn)."""
    _FWD["grad"] = torch.is_grad_enabled()
    return _PyramidBottleneckFn.apply(x, weight, *pyramid)


# --------------------------------------------------------------------------- bilinear resize
class _BilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, OH, OW, align_corners):
        x = to_nhwc(x, "interpolate")
        N, C, H, W = x.shape
        y = empty_nhwc(N, C, OH, OW, x.device)
        check(lib.segmi_bilinear_fwd(x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), N, H, W, C, OH, OW,
                                     1 if align_corners else 0, _stream()), "bilinear_fwd")
        ctx.geom = (N, C, H, W, OH, OW, align_corners)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W, OH, OW, ac = ctx.geom
        dy = to_nhwc(dy, "interpolate.backward")
        dx = empty_nhwc(N, C, H, W, dy.device)
        nws = lib.segmi_bilinear_bwd_workspace(N, H, W, C, OH, OW)
        ws = workspace(nws, dy.device)
        check(lib.segmi_bilinear_bwd(dy.data_ptr(), ld_of(dy), dx.data_ptr(), ld_of(dx), N, H, W, C, OH, OW,
                                     1 if ac else 0, ws.data_ptr(), nws, _stream()), "bilinear_bwd")
        return dx, None, None, None


def interpolate_bilinear(x, size, align_corners=False):
    """F.interpolate(x, size=size, mode='bilinear', align_corners=align_corners).

    The result remembers what it was interpolated from (`_segmi_src`): a loss that receives it untouched can evaluate itself
    on the LOW-resolution tensor with the interpolation folded into its kernel (`upsampled_cross_entropy`), so the
    full-resolution logits are written once for the caller (the drop-in contract: trainer.py:63-65 checks their size, metrics
    read them) but never read by the loss, and their gradient never exists."""
    y = _BilinearFn.apply(x, int(size[0]), int(size[1]), bool(align_corners))
    if x.dim() == 4 and x.shape[1] <= 256 and (int(size[0]) > x.shape[2] or int(size[1]) > x.shape[3]):
        y._segmi_src = (x, bool(align_corners), y._version, x._version)
    return y


def upsample_source(t):
    """(low-resolution tensor, align_corners) if `t` is an unmodified result of interpolate_bilinear of an unmodified source,
    else None (an in-place edit of either tensor after the interpolation bumps its version counter: the loss then reads `t`)."""
    src = getattr(t, "_segmi_src", None)
    if src is None or t._version != src[2] or src[0]._version != src[3]:
        return None
    return src[0], src[1]


# --------------------------------------------------------------------------- concat / dropout
class _CatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *xs):
        xs = [to_nhwc(x, "cat") for x in xs]
        N, _, H, W = xs[0].shape
        chans = [x.shape[1] for x in xs]
        for x in xs:
            if x.shape[0] != N or x.shape[2] != H or x.shape[3] != W:
                raise SegmiError("cat: spatial/batch mismatch")
        Ct = sum(chans)
        out = empty_nhwc(N, Ct, H, W, xs[0].device)
        ldo, st, off = ld_of(out), _stream(), 0
        for x, c in zip(xs, chans):
            last = off + c == Ct
            fill = (pad4(Ct) - off) if last else c   # zero the row padding after the last slice
            check(lib.segmi_copy_rows(x.data_ptr(), ld_of(x), out.data_ptr() + 4 * off, ldo, N * H * W, c, fill, st), "cat")
            off += c
        ctx.chans = chans
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = to_nhwc(dy, "cat.backward")
        outs, off = [], 0
        for c in ctx.chans:
            outs.append(dy[:, off:off + c])   # zero-copy channel slices (ld = total)
            off += c
        return tuple(outs)


def cat(tensors):
    """torch.cat(tensors, dim=1)"""
    return _CatFn.apply(*tensors)


# Device-side step counter folded into every dropout seed (segmi_dropout's seed_epoch_dev).  None in eager mode (each call
# draws a fresh host seed); segmi.graph.GraphedStep installs one so that a captured step draws fresh masks per replay.
_DROPOUT_EPOCH = None


def set_dropout_epoch(t):
    """t: None or a 1-element int64 CUDA tensor that the caller advances once per training step (after backward)."""
    global _DROPOUT_EPOCH
    if t is not None and not (t.is_cuda and t.dtype == torch.int64 and t.numel() == 1):
        raise SegmiError("segmi.set_dropout_epoch: expected a 1-element int64 CUDA tensor")
    _DROPOUT_EPOCH = t


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, channelwise, seed):
        x = to_nhwc(x, "dropout")
        N, C, H, W = x.shape
        y = empty_nhwc(N, C, H, W, x.device)
        ep = _DROPOUT_EPOCH
        check(lib.segmi_dropout(x.data_ptr(), ld_of(x), y.data_ptr(), ld_of(y), N, H * W, C, p, 1 if channelwise else 0, seed,
                                ep.data_ptr() if ep is not None else None, _stream()), "dropout")
        ctx.cfg = (p, channelwise, seed, ep)     # backward regenerates the mask: same seed, same (not yet advanced) epoch
        return y

    @staticmethod
    def backward(ctx, dy):
        p, channelwise, seed, ep = ctx.cfg
        dy = to_nhwc(dy, "dropout.backward")
        N, C, H, W = dy.shape
        dx = empty_nhwc(N, C, H, W, dy.device)
        check(lib.segmi_dropout(dy.data_ptr(), ld_of(dy), dx.data_ptr(), ld_of(dx), N, H * W, C, p, 1 if channelwise else 0,
                                seed, ep.data_ptr() if ep is not None else None, _stream()), "dropout.backward")
        return dx, None, None, None


def dropout(x, p, training=True, channelwise=False):
    """nn.Dropout (element mask) / nn.Dropout2d (per (n, c) mask).  The seed comes from torch's CPU
    generator (reproducible under torch.manual_seed, no device sync); masks differ from aten's."""
    if not training or p == 0.0:
        return x
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return _DropoutFn.apply(x, float(p), bool(channelwise), seed)


# --------------------------------------------------------------------------- per-pixel losses
# Data-parallel semantics (`group`): the reference evaluates its loss on the GATHERED global batch (trainer.py:56-66 under
# nn.DataParallel), i.e. CE is the mean over the valid pixels of ALL shards and Dice's sums run over the whole batch.  One
# process per GPU averages gradients instead, so each rank's loss is rescaled such that the AVERAGE over ranks of the
# per-rank losses — and of their gradients — equals the global-batch value: CE/Focal: W * local_sum / global_denominator
# (one all-reduce of the denominator), Dice: global sums (all-reduce of 3 doubles) with the upstream gradient times W.
# `group` is a torch.distributed process group (or True = the default group); None / world size 1 = single-device maths.
def _dist_world(group):
    import torch.distributed as dist
    if group is None or not (dist.is_available() and dist.is_initialized()):
        return None, 1
    g = None if group is True else group
    w = dist.get_world_size(g)
    return g, w


def _class_weight(weight, C, dev, what):
    if weight is None:
        return None
    w = torch.as_tensor(weight, dtype=torch.float32).to(dev).contiguous()
    if w.numel() != C:
        raise SegmiError("%s: weight has %d entries, logits have %d classes" % (what, w.numel(), C))
    return w


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index, weight, reduction, group, up):
        # up = None: logits are at the target's resolution.  up = align_corners flag: logits are LOW resolution and the loss is
        # that of their bilinear upsampling to the target's size (interpolation fused into the kernels)
        logits = to_nhwc(logits, "cross_entropy")
        N, C, H, W = logits.shape
        if target.dtype != torch.int64 or not target.is_cuda:
            raise SegmiError("cross_entropy: target must be an int64 CUDA tensor")
        if target.dim() != 3 or target.shape[0] != N or (up is None and tuple(target.shape) != (N, H, W)):
            raise SegmiError("cross_entropy: target shape %s does not match logits %s" % (tuple(target.shape), tuple(logits.shape)))
        target = target.contiguous()
        OH, OW = int(target.shape[1]), int(target.shape[2])
        rows, dev, st = N * OH * OW, logits.device, _stream()
        cw = _class_weight(weight, C, dev, "cross_entropy")
        lse = torch.empty(rows, device=dev, dtype=torch.float32)
        out = torch.empty(3, device=dev, dtype=torch.float32)  # {mean, denominator, numerator}
        if up is None:
            nws = lib.segmi_ce_workspace(rows)
            ws = workspace(nws, dev)
            check(lib.segmi_ce_fwd(logits.data_ptr(), ld_of(logits), target.data_ptr(), rows, C, ignore_index,
                                   cw.data_ptr() if cw is not None else None, lse.data_ptr(), out.data_ptr(), ws.data_ptr(), nws, st), "ce_fwd")
        else:
            nws = lib.segmi_upsample_ce_workspace(N, H, W, C, OH, OW)
            ws = workspace(nws, dev)
            check(lib.segmi_upsample_ce_fwd(logits.data_ptr(), ld_of(logits), N, H, W, C, OH, OW, 1 if up else 0, target.data_ptr(),
                                            ignore_index, cw.data_ptr() if cw is not None else None, lse.data_ptr(), out.data_ptr(),
                                            ws.data_ptr(), nws, st), "upsample_ce_fwd")
        pg, world = _dist_world(group)
        if reduction == "sum":
            norm = torch.ones(3, device=dev, dtype=torch.float32)     # bwd divides by norm[1] = 1
            value = out[2]
            if world > 1:
                norm = norm / world      # gradients are averaged over ranks: the sum over the global batch needs W * local
                value = out[2] * world
        elif world > 1:
            from .distributed import global_batch_mean
            value, den = global_batch_mean(out[2], out[1], pg)       # one all-reduce: valid pixels / weight sum of all shards
            norm = torch.stack([out[0], den, out[2]])
        else:
            norm, value = out, out[0]
        ctx.save_for_backward(logits, target, lse, norm, cw)
        ctx.ignore_index = ignore_index
        ctx.up = up
        return value

    @staticmethod
    def backward(ctx, g):
        logits, target, lse, norm, cw = ctx.saved_tensors
        N, C, H, W = logits.shape
        g = g.contiguous().float()
        dl = empty_nhwc(N, C, H, W, logits.device)
        if ctx.up is None:
            check(lib.segmi_ce_bwd(logits.data_ptr(), ld_of(logits), target.data_ptr(), lse.data_ptr(), N * H * W, C,
                                   ctx.ignore_index, cw.data_ptr() if cw is not None else None, norm.data_ptr(), g.data_ptr(),
                                   dl.data_ptr(), ld_of(dl), _stream()), "ce_bwd")
        else:
            OH, OW = int(target.shape[1]), int(target.shape[2])
            nws = lib.segmi_upsample_ce_workspace(N, H, W, C, OH, OW)
            ws = workspace(nws + 16, logits.device)
            wp = (ws.data_ptr() + 15) & ~15
            check(lib.segmi_upsample_ce_bwd(logits.data_ptr(), ld_of(logits), N, H, W, C, OH, OW, 1 if ctx.up else 0, target.data_ptr(),
                                            lse.data_ptr(), ctx.ignore_index, cw.data_ptr() if cw is not None else None, norm.data_ptr(),
                                            g.data_ptr(), dl.data_ptr(), ld_of(dl), wp, nws, _stream()), "upsample_ce_bwd")
        return dl, None, None, None, None, None, None


def cross_entropy(logits, target, ignore_index=255, weight=None, reduction="mean", group=None):
    """nn.CrossEntropyLoss(weight=..., ignore_index=..., reduction='mean'|'sum') on [N,C,H,W] logits / [N,H,W] int64 target."""
    if reduction not in ("mean", "sum"):
        raise SegmiError("cross_entropy: reduction must be 'mean' or 'sum' (got %r)" % (reduction,))
    return _CrossEntropyFn.apply(logits, target, int(ignore_index), weight, reduction, group, None)


def upsampled_cross_entropy(logits_lo, target, align_corners=False, ignore_index=255, weight=None, reduction="mean", group=None):
    """cross_entropy(F.interpolate(logits_lo, size=target.shape[1:], mode='bilinear', align_corners=...), target, ...) with the
    interpolation evaluated inside the loss kernels: no [N,C,H,W] logits, no [N,C,H,W] gradient (include/segmi.h
    segmi_upsample_ce_fwd / _bwd)."""
    if reduction not in ("mean", "sum"):
        raise SegmiError("cross_entropy: reduction must be 'mean' or 'sum' (got %r)" % (reduction,))
    return _CrossEntropyFn.apply(logits_lo, target, int(ignore_index), weight, reduction, group, bool(align_corners))


def _loss_inputs(logits, target, what):
    logits = to_nhwc(logits, what)
    N, C, H, W = logits.shape
    if target.dtype != torch.int64 or not target.is_cuda:
        raise SegmiError("%s: target must be an int64 CUDA tensor" % what)
    if tuple(target.shape) != (N, H, W):
        raise SegmiError("%s: target shape %s does not match logits %s" % (what, tuple(target.shape), tuple(logits.shape)))
    return logits, N * H * W, C


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index, smooth, group):
        logits, rows, C = _loss_inputs(logits, target, "dice_loss")
        if not target.is_contiguous():
            raise SegmiError("dice_loss: target must be contiguous (it is rewritten in place like the reference does)")
        dev, st = logits.device, _stream()
        lse = torch.empty(rows, device=dev, dtype=torch.float32)
        out = torch.empty(4, device=dev, dtype=torch.float32)
        stats = torch.empty(4, device=dev, dtype=torch.int64)
        nws = lib.segmi_dice_workspace(rows)
        ws = workspace(nws, dev)
        pg, world = _dist_world(group)
        if world == 1:
            check(lib.segmi_dice_fwd(logits.data_ptr(), ld_of(logits), target.data_ptr(), rows, C, ignore_index, smooth,
                                     stats.data_ptr(), lse.data_ptr(), out.data_ptr(), ws.data_ptr(), nws, st), "dice_fwd")
        else:
            import torch.distributed as dist
            # the reference's target.min()/max()/(target == ignore).sum() are taken over the gathered global target (utils/losses.py:40-42)
            check(lib.segmi_target_stats(target.data_ptr(), rows, ignore_index, stats.data_ptr(), ws.data_ptr(), nws, st), "target_stats")
            ext = torch.stack([stats[0], -stats[1]])
            dist.all_reduce(ext, op=dist.ReduceOp.MIN, group=pg)
            cnt = stats[2:3].clone()
            dist.all_reduce(cnt, group=pg)
            tmin, tmax = ext[0], -ext[1]
            in_range = (tmin <= ignore_index) & (tmax > ignore_index)
            stats = torch.stack([tmin, tmax, cnt[0], ((~in_range) & (cnt[0] > 0)).to(torch.int64)]).contiguous()
            sums = torch.empty(3, device=dev, dtype=torch.float64)
            check(lib.segmi_dice_sums(logits.data_ptr(), ld_of(logits), target.data_ptr(), rows, C, ignore_index, stats.data_ptr(),
                                      lse.data_ptr(), sums.data_ptr(), ws.data_ptr(), nws, st), "dice_sums")
            dist.all_reduce(sums, group=pg)
            check(lib.segmi_dice_finalize(sums.data_ptr(), smooth, out.data_ptr(), st), "dice_finalize")
        ctx.save_for_backward(logits, lse, out)
        ctx.target = target     # int64, no autograd involvement; kept by reference (its rewritten content is what backward needs)
        ctx.world = world
        return out[0]

    @staticmethod
    def backward(ctx, g):
        logits, lse, out = ctx.saved_tensors
        N, C, H, W = logits.shape
        g = g.contiguous().float()
        if ctx.world > 1:
            g = g * ctx.world       # this rank's share of the GLOBAL loss' gradient; the gradient all-reduce averages over ranks
        dl = empty_nhwc(N, C, H, W, logits.device)
        check(lib.segmi_dice_bwd(logits.data_ptr(), ld_of(logits), ctx.target.data_ptr(), lse.data_ptr(), N * H * W, C,
                                 out.data_ptr(), g.data_ptr(), dl.data_ptr(), ld_of(dl), _stream()), "dice_bwd")
        return dl, None, None, None, None


def dice_loss(logits, target, ignore_index=255, smooth=1.0, group=None):
    """DiceLoss.forward of the reference (utils/losses.py:39-50), including its in-place rewrite of ignored target pixels."""
    return _DiceFn.apply(logits, target, int(ignore_index), float(smooth), group)


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index, gamma, alpha, size_average, group):
        logits, rows, C = _loss_inputs(logits, target, "focal_loss")
        target = target.contiguous()
        dev, st = logits.device, _stream()
        cw = _class_weight(alpha, C, dev, "focal_loss")
        lse = torch.empty(rows, device=dev, dtype=torch.float32)
        out = torch.empty(3, device=dev, dtype=torch.float32)      # {mean over all pixels, rows, sum}
        nws = lib.segmi_ce_workspace(rows)
        ws = workspace(nws, dev)
        check(lib.segmi_focal_fwd(logits.data_ptr(), ld_of(logits), target.data_ptr(), rows, C, ignore_index, gamma,
                                  cw.data_ptr() if cw is not None else None, lse.data_ptr(), out.data_ptr(), ws.data_ptr(), nws, st), "focal_fwd")
        pg, world = _dist_world(group)
        if not size_average:                                        # reference: loss.sum()
            norm = torch.ones(3, device=dev, dtype=torch.float32) / world
            value = out[2] * world
        elif world > 1:
            from .distributed import global_batch_mean
            value, den = global_batch_mean(out[2], out[1], pg)     # pixels of all shards (ragged shards weigh by their size)
            norm = torch.stack([out[0], den, out[2]])
        else:
            norm, value = out, out[0]
        ctx.save_for_backward(logits, target, lse, norm, cw)
        ctx.cfg = (ignore_index, gamma)
        return value

    @staticmethod
    def backward(ctx, g):
        logits, target, lse, norm, cw = ctx.saved_tensors
        ignore_index, gamma = ctx.cfg
        N, C, H, W = logits.shape
        g = g.contiguous().float()
        dl = empty_nhwc(N, C, H, W, logits.device)
        check(lib.segmi_focal_bwd(logits.data_ptr(), ld_of(logits), target.data_ptr(), lse.data_ptr(), N * H * W, C, ignore_index,
                                  gamma, cw.data_ptr() if cw is not None else None, norm.data_ptr(), g.data_ptr(), dl.data_ptr(),
                                  ld_of(dl), _stream()), "focal_bwd")
        return dl, None, None, None, None, None, None


def focal_loss(logits, target, ignore_index=255, gamma=2.0, alpha=None, size_average=True, group=None):
    """FocalLoss.forward of the reference (utils/losses.py:52-65): ce = alpha_t * (-log p_t) (0 where ignored),
    ((1 - exp(-ce))^gamma * ce).mean() over ALL pixels (or .sum() with size_average=False)."""
    return _FocalFn.apply(logits, target, int(ignore_index), float(gamma), alpha, bool(size_average), group)


_LOVASZ_LAST = {"out": None}


class _LovaszFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index, up):
        # up = None: logits are at the target's resolution.  up = align_corners flag: logits are LOW resolution and the loss is
        # that of their bilinear upsampling to the target's size (interpolation inside the kernels, like _CrossEntropyFn)
        if up is None:
            logits, rows, C = _loss_inputs(logits, target, "lovasz_softmax")
            N, _, H, W = logits.shape
            OH, OW = H, W
        else:
            logits = to_nhwc(logits, "lovasz_softmax")
            N, C, H, W = logits.shape
            if target.dtype != torch.int64 or not target.is_cuda or target.dim() != 3 or target.shape[0] != N:
                raise SegmiError("lovasz_softmax: target must be an int64 CUDA tensor [N,OH,OW] (got %s for logits %s)"
                                 % (tuple(target.shape), tuple(logits.shape)))
            OH, OW = int(target.shape[1]), int(target.shape[2])
            rows = N * OH * OW
        target = target.contiguous()
        dev, st = logits.device, _stream()
        nws = lib.segmi_lovasz_workspace(rows, C) if up is None else lib.segmi_upsample_lovasz_workspace(N, H, W, C, OH, OW)
        if nws == 0:
            raise SegmiError("lovasz_softmax: unsupported size (needs 0 < pixels < 2^24 and at most 1820 classes, got %d x %d)" % (rows, C))
        ws = workspace(nws + 256, dev)
        wp = (ws.data_ptr() + 255) & ~255
        lse = torch.empty(rows, device=dev, dtype=torch.float32)
        # G is written and read at the SURVIVOR entries only (see include/segmi.h): never cleared
        G = empty_nhwc(N, C, OH, OW, dev)
        if os.environ.get("SEGMI_LOVASZ_POISON") == "1":          # tests: any read of an unwritten entry turns the gradient NaN
            G.fill_(float("nan"))
        out = torch.empty(4 + C, device=dev, dtype=torch.float32)
        if up is None:
            check(lib.segmi_lovasz_fwd(logits.data_ptr(), ld_of(logits), target.data_ptr(), rows, C, ignore_index, lse.data_ptr(),
                                       G.data_ptr(), ld_of(G), out.data_ptr(), wp, nws, st), "lovasz_fwd")
        else:
            check(lib.segmi_upsample_lovasz_fwd(logits.data_ptr(), ld_of(logits), N, H, W, C, OH, OW, 1 if up else 0, target.data_ptr(),
                                                ignore_index, lse.data_ptr(), G.data_ptr(), ld_of(G), out.data_ptr(), wp, nws, st),
                  "upsample_lovasz_fwd")
        ctx.save_for_backward(logits, target, lse, G, out)
        ctx.ignore_index = ignore_index
        ctx.up = up
        _LOVASZ_LAST["out"] = out
        return out[0]

    @staticmethod
    def backward(ctx, g):
        logits, target, lse, G, out = ctx.saved_tensors
        N, C, H, W = logits.shape
        g = g.contiguous().float()
        dl = empty_nhwc(N, C, H, W, logits.device)
        if ctx.up is None:
            check(lib.segmi_lovasz_bwd(logits.data_ptr(), ld_of(logits), target.data_ptr(), ctx.ignore_index, lse.data_ptr(), G.data_ptr(),
                                       ld_of(G), N * H * W, C, out.data_ptr(), g.data_ptr(), dl.data_ptr(), ld_of(dl), _stream()), "lovasz_bwd")
        else:
            OH, OW = int(target.shape[1]), int(target.shape[2])
            nws = lib.segmi_upsample_lovasz_workspace(N, H, W, C, OH, OW)
            ws = workspace(nws + 256, logits.device)
            wp = (ws.data_ptr() + 255) & ~255
            check(lib.segmi_upsample_lovasz_bwd(logits.data_ptr(), ld_of(logits), N, H, W, C, OH, OW, 1 if ctx.up else 0, target.data_ptr(),
                                                ctx.ignore_index, lse.data_ptr(), G.data_ptr(), ld_of(G), out.data_ptr(), g.data_ptr(),
                                                dl.data_ptr(), ld_of(dl), wp, nws, _stream()), "upsample_lovasz_bwd")
        return dl, None, None, None


def lovasz_softmax(logits, target, ignore_index=255):
    """LovaszSoftmax.forward of the reference (softmax + lovasz_softmax(classes='present', per_image=False, ignore=...))."""
    return _LovaszFn.apply(logits, target, int(ignore_index), None)


def upsampled_lovasz_softmax(logits_lo, target, align_corners=False, ignore_index=255):
    """lovasz_softmax(F.interpolate(logits_lo, size=target.shape[1:], mode='bilinear', align_corners=...), target, ...) with the
    interpolation evaluated inside the loss kernels: no [N,C,H,W] logits, no [N,C,H,W] gradient (include/segmi.h
    segmi_upsample_lovasz_fwd / _bwd).  Bit-identical to the unfused form (tests/test_ops_gpu.py)."""
    return _LovaszFn.apply(logits_lo, target, int(ignore_index), bool(align_corners))


def lovasz_last_stats():
    """(survivors, keys of the full sort) of the most recent lovasz_softmax call: how many (class, pixel) elements the tail
    pruning kept out of n_present * n_valid.  Synchronises; for reporting only."""
    out = _LOVASZ_LAST["out"]
    if out is None:
        return None
    kept, full = out[2:4].tolist()
    return int(kept), int(full)


def seg_metrics_accumulate(logits, target, acc):
    """Fused argmax + accuracy / IoU counting (utils/metrics.py:59-67 of the reference) accumulated into the int64 device
    tensor `acc` of 2 + 3*C entries {correct, labeled, inter[C], pred_area[C], label_area[C]}; no host synchronisation."""
    logits, rows, C = _loss_inputs(logits.detach(), target, "seg_metrics")
    if acc.dtype != torch.int64 or not acc.is_cuda or acc.numel() != 2 + 3 * C or not acc.is_contiguous():
        raise SegmiError("seg_metrics: acc must be a contiguous int64 CUDA tensor of %d entries" % (2 + 3 * C))
    check(lib.segmi_seg_metrics(logits.data_ptr(), ld_of(logits), target.contiguous().data_ptr(), rows, C, acc.data_ptr(), _stream()),
          "seg_metrics")
    return acc
