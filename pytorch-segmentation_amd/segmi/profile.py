"""Per-launch timing of libsegmi kernels with HIP events (bench.py's roofline leg).

`KernelTimer` brackets each instrumented C-ABI call with a pair of events recorded on the stream the
kernel is launched on (torch's current stream — the stream every segmi launch uses), and attributes
the elapsed time to the kernel variant name the dispatcher reports (`segmi_conv2d_variant`), i.e.
the name a `rocprofv3 --kernel-trace` of the same command shows.  Disabled (the default) it costs one
global lookup per call.
"""
import contextlib
from collections import defaultdict

import torch

_ACTIVE = None
_NULL = contextlib.nullcontext()


def _b4(*elems):
    return 4 * sum(int(e) for e in elems)


def _dw_bytes(d):
    d = d._obj if hasattr(d, "_obj") else d          # ctypes byref() wrapper or the structure itself
    return 4 * (d.N * d.H * d.W * d.C + d.N * d.P * d.Q * d.C)


# HBM-bound entry points of the C ABI -> algorithmic bytes of one call, from its positional arguments (include/segmi.h order):
# every tensor the call must read or write crosses HBM once.  Used by KernelTimer(membound=True), which times these calls
# with HIP events exactly like the convolutions (tools/membound_ops.py prints the table).
MEMBOUND_BYTES = {
    "segmi_bn_stats_finalize": lambda a: _b4(a[2] * a[3]),
    "segmi_bn_stats": lambda a: _b4(a[2] * a[3]),
    "segmi_bn_apply": lambda a: _b4((2 + (a[2] is not None)) * a[6] * a[7]),
    "segmi_bn_bwd_reduce": lambda a: _b4((2 + (a[4] is not None)) * a[6] * a[7]),
    "segmi_bn_bwd_apply": lambda a: _b4((3 + (a[4] is not None) + (a[19] is not None)) * a[6] * a[7]),
    "segmi_maxpool_fwd": lambda a: _b4(a[5] * a[8] * (a[6] * a[7] + a[9] * a[10])) + a[5] * a[8] * a[9] * a[10],
    "segmi_maxpool_bwd": lambda a: _b4(a[5] * a[8] * (a[6] * a[7] + a[9] * a[10])) + a[5] * a[8] * a[9] * a[10],
    "segmi_pyramid_pool_fwd": lambda a: _b4(a[2] * a[3] * a[4] * a[5]),
    "segmi_pyramid_pool_bwd": lambda a: _b4(a[4] * a[5] * a[6] * a[7]),
    "segmi_pyramid_up_fwd": lambda a: _b4(a[1] * a[2] * a[3] * a[4]),
    "segmi_pyramid_up_bwd": lambda a: _b4(a[2] * a[3] * a[4] * a[5]),
    "segmi_upsample_ce_fwd": lambda a: 12 * a[2] * a[6] * a[7] + _b4(a[2] * a[3] * a[4] * a[5]),
    "segmi_upsample_ce_bwd": lambda a: 12 * a[2] * a[6] * a[7] + _b4(2 * a[2] * a[6] * a[4] * ((a[5] + 3) & ~3), 2 * a[2] * a[3] * a[4] * a[5]),
    "segmi_ce_fwd": lambda a: _b4(a[3] * a[4]) + 12 * a[3],
    "segmi_ce_bwd": lambda a: _b4(2 * a[4] * a[5]) + 12 * a[4],
    "segmi_bilinear_fwd": lambda a: _b4(a[4] * a[7] * (a[5] * a[6] + a[8] * a[9])),
    "segmi_bilinear_bwd": lambda a: _b4(a[4] * a[7] * (a[5] * a[6] + a[8] * a[9])),
    "segmi_dropout": lambda a: _b4(2 * a[4] * a[5] * a[6]),
    "segmi_copy_rows": lambda a: _b4(2 * a[4] * a[5]),
    "segmi_nchw_to_nhwc": lambda a: _b4(2 * a[2] * a[3] * a[4] * a[5]),
    # depthwise 3x3 (args: desc*, x, w, y, stream): input + output once; the filter gradient reads x and dy
    "segmi_dwconv2d_fwd": lambda a: _dw_bytes(a[0]),
    "segmi_dwconv2d_fwd_stats": lambda a: _dw_bytes(a[0]),            # + the BN statistics partials (a few KB)
    "segmi_dwconv2d_dgrad": lambda a: _dw_bytes(a[0]),
    "segmi_dwconv2d_wgrad": lambda a: _dw_bytes(a[0]),
    # the same with BatchNorm(+ReLU) applied to the loaded operand (round 6): the pre-normalisation tensor in, the output (or dy) once
    "segmi_dwconv2d_fwd_pre": lambda a: _dw_bytes(a[0]),
    "segmi_dwconv2d_wgrad_pre": lambda a: _dw_bytes(a[0]),
    # Lovasz forward, tail-pruned (round 5): the threshold of a class is a reduction over ALL its pixels, so the logits are read at
    # least twice (4 B each: threshold pass, selection pass) — the survivors' sort traffic (~0.7 % of the elements x 88 B) is below
    # 1 B per element and not counted.  The implementation reads them three times (threshold, count, emit).
    "segmi_lovasz_fwd": lambda a: 8 * a[3] * a[4],
    # backward: logits read, dlogits written (G only at the survivor entries)
    "segmi_lovasz_bwd": lambda a: _b4(2 * a[7] * a[8]),
    # on upsampled logits: the low-resolution logits + target (8 B) and lse (4 B) per output pixel forward; backward the same reads,
    # the per-pixel dot (4 B written + read), the pass-W buffer [N, OH, W, C] written + read, the low-resolution gradient written
    "segmi_upsample_lovasz_fwd": lambda a: _b4(a[2] * a[3] * a[4] * a[5]) + 12 * a[2] * a[6] * a[7],
    "segmi_upsample_lovasz_bwd": lambda a: 20 * a[2] * a[6] * a[7] + _b4(2 * a[2] * a[6] * a[4] * ((a[5] + 3) & ~3), 2 * a[2] * a[3] * a[4] * a[5]),
    "segmi_relu_fwd": lambda a: _b4(2 * a[4] * a[5]),
    "segmi_add": lambda a: _b4(3 * a[6] * a[7]),
}


class KernelTimer:
    def __init__(self, membound=False):
        self.spans = []   # {"name", "flops", "bytes", "a", "b", "detail", "inner"}
        self.membound = membound
        self._patched = {}

    def __enter__(self):
        global _ACTIVE
        self._prev, _ACTIVE = _ACTIVE, self
        if self.membound:
            from ._lib import lib      # the ctypes functions are attributes of the CDLL object: swap in timing wrappers
            for name, nbytes in MEMBOUND_BYTES.items():
                fn = getattr(lib, name)
                self._patched[name] = fn

                def wrapper(*args, _fn=fn, _name=name, _nb=nbytes):
                    with self._span(_name, 0, _nb(args)):
                        return _fn(*args)
                setattr(lib, name, wrapper)
        return self

    def __exit__(self, *exc):
        global _ACTIVE
        _ACTIVE = self._prev
        if self._patched:
            from ._lib import lib
            for name, fn in self._patched.items():
                setattr(lib, name, fn)
            self._patched = {}

    @contextlib.contextmanager
    def _span(self, name, flops, nbytes, detail=None, inner=None, eff=None):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # flops = EXECUTED, eff = the algorithmic FLOPs of the convolution the launch stands for (None: the same — they differ for a
        # direct launch that skips taps / chunks with zero-filled operands, ops._conv_issued, and for the composite Winograd calls)
        # (a callable `flops` is evaluated by _resolve() when the spans are read, not here: host work between the launches of the
        #  instrumented step leaves the GPU idle between kernels and the kernels after such a gap measured 4 % slower)
        rec = {"name": name, "flops": flops, "eff": eff, "bytes": nbytes, "a": a, "b": b, "detail": detail, "inner": None}
        if inner is not None:
            # composite call (a Winograd pass = transforms + ONE batched contraction launch): the library records a second
            # event pair right around the contraction (segmi_conv2d_winograd_trace), so the MFMA-bound kernel is timed apart
            # from the HBM-bound transforms.  The events must exist before their handles can be handed over: record once here.
            from ._lib import lib
            ia, ib = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ia.record()
            ib.record()
            lib.segmi_conv2d_winograd_trace(ia.cuda_event, ib.cuda_event)
            rec["inner"] = {"flops": inner[0], "bytes": inner[1], "a": ia, "b": ib}
        a.record()
        yield
        b.record()
        self.spans.append(rec)

    def _resolve(self):
        for s in self.spans:
            if callable(s["flops"]):
                s["flops"] = s["flops"]()
            if s["eff"] is None:
                s["eff"] = s["flops"]

    def by_detail(self):
        """[(name, detail, launches, total_ms, flops, bytes)] grouped by (name, detail), slowest first (synchronises).  Composite
        (Winograd) calls are ONE row each: whole-call time against the ALGORITHMIC (direct-convolution) FLOPs."""
        torch.cuda.synchronize()
        self._resolve()
        acc = {}
        for s in self.spans:
            r = acc.setdefault((s["name"], s["detail"]), [0, 0.0, 0, 0])
            r[0] += 1
            r[1] += s["a"].elapsed_time(s["b"])
            r[2] += s["eff"]
            r[3] += s["bytes"]
        return sorted(((k[0], k[1], v[0], v[1], v[2], v[3]) for k, v in acc.items()), key=lambda t: -t[3])

    def summary(self):
        """{kernel name: {"launches", "total_ms", "avg_us", "flops", "eff_flops", "bytes"}} (synchronises).

        `flops` are EXECUTED FLOPs, `eff_flops` the algorithmic FLOPs of the direct convolution the launch stands for (equal
        for the direct kernels).  A composite Winograd call contributes two rows: its contraction under the name of the kernel
        that runs it (the batched implicit-GEMM / filter-gradient kernel, executed = 32*T*C*K transform-domain FLOPs, timed by the
        inner event pair) and "winograd transforms" (whole call minus contraction, 0 FLOPs)."""
        torch.cuda.synchronize()
        self._resolve()
        out = defaultdict(lambda: {"launches": 0, "total_ms": 0.0, "flops": 0, "eff_flops": 0, "bytes": 0})
        for s in self.spans:
            whole = s["a"].elapsed_time(s["b"])
            if s["inner"] is None:
                r = out[s["name"]]
                r["launches"] += 1
                r["total_ms"] += whole
                r["flops"] += s["flops"]
                r["eff_flops"] += s["eff"]
                r["bytes"] += s["bytes"]
                continue
            inn = s["inner"]
            core = min(whole, inn["a"].elapsed_time(inn["b"]))
            r = out[s["name"].split(": 16 x ", 1)[-1]]
            r["launches"] += 1
            r["total_ms"] += core
            r["flops"] += inn["flops"]
            r["eff_flops"] += s["eff"]
            r["bytes"] += inn["bytes"]
            t = out["winograd transforms (" + s["name"].split(":", 1)[0].split()[-1] + ")"]
            t["launches"] += 1
            t["total_ms"] += whole - core
            t["bytes"] += s["bytes"]
        for r in out.values():
            r["avg_us"] = 1e3 * r["total_ms"] / r["launches"]
        return dict(out)


def span(name, flops=0, nbytes=0, detail=None, inner=None, eff=None):
    """Context manager around one C-ABI launch; `name`/`detail`/`inner` may be callables evaluated only when timing, a callable
    `flops` only when the timer's spans are read.
    inner = (executed FLOPs, operand bytes) of the contraction inside a composite Winograd call (see KernelTimer._span);
    eff = algorithmic FLOPs where they differ from the executed `flops`."""
    if _ACTIVE is None:
        return _NULL
    return _ACTIVE._span(name() if callable(name) else name, flops, nbytes,
                         detail() if callable(detail) else detail, inner() if callable(inner) else inner, eff)
