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


class KernelTimer:
    def __init__(self):
        self.spans = []   # (name, flops, bytes, start_event, end_event)
        self.details = []

    def __enter__(self):
        global _ACTIVE
        self._prev, _ACTIVE = _ACTIVE, self
        return self

    def __exit__(self, *exc):
        global _ACTIVE
        _ACTIVE = self._prev

    @contextlib.contextmanager
    def _span(self, name, flops, nbytes, detail=None):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self.spans.append((name, flops, nbytes, a, b))
        self.details.append(detail)

    def by_detail(self):
        """[(name, detail, launches, total_ms, flops, bytes)] grouped by (name, detail), slowest first (synchronises)."""
        torch.cuda.synchronize()
        acc = {}
        for (name, flops, nbytes, a, b), det in zip(self.spans, self.details):
            r = acc.setdefault((name, det), [0, 0.0, 0, 0])
            r[0] += 1
            r[1] += a.elapsed_time(b)
            r[2] += flops
            r[3] += nbytes
        return sorted(((k[0], k[1], v[0], v[1], v[2], v[3]) for k, v in acc.items()), key=lambda t: -t[3])

    def summary(self):
        """{name: {"launches", "total_ms", "avg_us", "flops", "bytes"}} (synchronises)."""
        torch.cuda.synchronize()
        out = defaultdict(lambda: {"launches": 0, "total_ms": 0.0, "flops": 0, "bytes": 0})
        for name, flops, nbytes, a, b in self.spans:
            r = out[name]
            r["launches"] += 1
            r["total_ms"] += a.elapsed_time(b)
            r["flops"] += flops
            r["bytes"] += nbytes
        for r in out.values():
            r["avg_us"] = 1e3 * r["total_ms"] / r["launches"]
        return dict(out)


def span(name, flops=0, nbytes=0, detail=None):
    """Context manager around one C-ABI launch; `name`/`detail` may be callables evaluated only when timing."""
    if _ACTIVE is None:
        return _NULL
    return _ACTIVE._span(name() if callable(name) else name, flops, nbytes, detail() if callable(detail) else detail)
