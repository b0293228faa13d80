"""Fused optimizers on libsegmi.

`SGD` is a drop-in for `torch.optim.SGD(params, lr, momentum, weight_decay)` as the reference instantiates it by name from
config.json (`base/base_trainer.py:57` `get_instance(torch.optim, 'optimizer', config, trainable_params)`), including parameter
groups with their own lr (differential learning rates).  One kernel launch updates every parameter (`segmi_sgd_step`) instead
of torch's three foreach passes.  State layout (`state[p]['momentum_buffer']`) matches torch.optim.SGD, so optimizer
state_dicts interchange with checkpoints written by the reference (buffers that arrive contiguous-NCHW for a channels_last
filter are re-laid in the parameter's memory order on the first step).
"""
import ctypes as C

import torch

from ._lib import check, lib


class _Chunk(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("momentum", C.c_void_p), ("count", C.c_long), ("group", C.c_int),
                ("vec4", C.c_int)]


def _same_layout(a, b):
    """Same element order in memory (strides may differ only on size-1 dimensions)."""
    return a.shape == b.shape and all(sa == sb or n == 1 for sa, sb, n in zip(a.stride(), b.stride(), a.shape))


class SGD(torch.optim.Optimizer):
    """capturable=True keeps lr / weight decay / momentum in DEVICE memory so that a step captured into a hipGraph
    (segmi.graph.GraphedStep) follows the lr schedule — kernel arguments would be frozen at capture.  `push_hyper()` sends the
    current `param_groups` values through a ring of pinned staging slots with an ordinary stream-ordered copy (each slot is
    reused only after the event of its previous copy has completed, so a host running several replays ahead cannot overwrite
    values still in flight).  Eager `step()` pushes by itself; around a captured step call `push_hyper()` before each replay."""

    _RING = 8

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0, weight_decay=0.0, nesterov=False, capturable=False):
        if dampening != 0 or nesterov:
            raise NotImplementedError("segmi.optim.SGD implements dampening=0, nesterov=False (the reference's configuration)")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=False))
        if len(self.param_groups) > 8:
            raise ValueError("segmi.optim.SGD supports up to 8 parameter groups")
        self._table = None
        self._sig = None
        self._segments = None          # optional: parameter lists whose chunks are contiguous in the table (gradient buckets)
        self._seg_ranges = []
        self.capturable = bool(capturable)
        self._ring = self._ring_events = self._hyper_dev = None
        self._ring_pos = 0

    def push_hyper(self):
        """Stream-ordered refresh of the device-resident hyper-parameters from `param_groups` (capturable=True only)."""
        if not self.capturable:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("segmi.optim.SGD.push_hyper must stay outside the captured region (call it before graph replay)")
        if self._ring is None:
            n = lib.segmi_sgd_hyper_floats()
            dev = self.param_groups[0]["params"][0].device
            self._ring = torch.zeros(self._RING, n, dtype=torch.float32).pin_memory()
            self._ring_events = [None] * self._RING
            self._hyper_dev = torch.zeros(n, dtype=torch.float32, device=dev)
        i = self._ring_pos
        self._ring_pos = (i + 1) % self._RING
        if self._ring_events[i] is not None:
            self._ring_events[i].synchronize()          # the copy that last read this slot has executed
        h = self._ring[i]
        for gi, g in enumerate(self.param_groups):
            h[gi], h[8 + gi], h[16 + gi] = float(g["lr"]), float(g["weight_decay"]), float(g["momentum"])
        self._hyper_dev.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._ring_events[i] = ev

    def set_segments(self, segments):
        """Lay the chunk table out segment by segment (`segments`: lists of parameters, e.g. the gradient buckets of
        segmi.distributed.GradAllReducer in the order their all-reduces complete) so that `step_segment(i)` can update one
        bucket's parameters as soon as ITS all-reduce has finished, while later buckets are still on the wire
        (reference: one optimizer.step() after the whole gather, base/base_trainer.py:46-57 + trainer.py:70-71)."""
        self._segments = [list(seg) for seg in segments]
        self._sig = None

    def _ordered(self):
        """[(group index, param)] in table order, and the segment boundaries (in parameters)."""
        gi_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
        if not self._segments:
            return [(gi, p) for gi, g in enumerate(self.param_groups) for p in g["params"]], None
        seen, order, bounds = set(), [], []
        for seg in self._segments:
            start = len(order)
            for p in seg:
                if id(p) in gi_of and id(p) not in seen:
                    seen.add(id(p))
                    order.append((gi_of[id(p)], p))
            bounds.append((start, len(order)))
        start = len(order)
        order += [(gi, p) for gi, g in enumerate(self.param_groups) for p in g["params"] if id(p) not in seen]
        bounds.append((start, len(order)))          # parameters outside every segment: a last segment of their own
        return order, bounds

    def _build(self):
        """Chunk table on the device; rebuilt only if a gradient / parameter / buffer pointer changed."""
        ce = lib.segmi_sgd_chunk_elems()
        entries, sig = [], []
        order, bounds = self._ordered()
        first_chunk = []                 # chunk index at which parameter i of `order` starts
        for gi, p in order:
            first_chunk.append(len(entries))
            if p.grad is None:
                continue
            st = self.state[p]
            if "momentum_buffer" not in st or st["momentum_buffer"] is None:
                st["momentum_buffer"] = torch.zeros(p.numel(), dtype=p.dtype, device=p.device).as_strided(p.shape, p.stride())
            buf, g = st["momentum_buffer"], p.grad
            if buf.shape == p.shape and (not _same_layout(buf, p) or buf.device != p.device or buf.dtype != p.dtype):
                # a state_dict written by the reference (or by torch.optim.SGD) holds contiguous NCHW momentum buffers, while
                # k > 1 filters live channels_last here (KRSC in memory): re-lay the buffer once in the parameter's own layout
                fixed = torch.empty_like(p)           # preserve_format: the parameter's (dense, permuted) strides
                fixed.copy_(buf)
                buf = st["momentum_buffer"] = fixed
            if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and _same_layout(g, p) and _same_layout(buf, p)):
                raise RuntimeError("segmi.optim.SGD: parameter, gradient and momentum buffer must be float32 CUDA tensors with "
                                   "identical (dense) strides")
            n = p.numel()
            sig.append((p.data_ptr(), g.data_ptr(), buf.data_ptr(), n, gi))
            for off in range(0, n, ce):
                cnt = min(ce, n - off)
                ptrs = [t.data_ptr() + 4 * off for t in (p, g, buf)]
                entries.append((ptrs[0], ptrs[1], ptrs[2], cnt, gi, int(cnt % 4 == 0 and all(q % 16 == 0 for q in ptrs))))
        sig = tuple(sig)
        if sig != self._sig:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("segmi.optim.SGD: a parameter / gradient / momentum pointer changed while a hipGraph is being "
                                   "captured; the pointer table is uploaded from host memory and cannot be rebuilt inside a capture. "
                                   "Keep gradients persistent (segmi.distributed.DistributedModel holds them as bucket views, also in a "
                                   "single process) and run at least one eager step before capturing (segmi.graph.GraphedStep does).")
            arr = (_Chunk * len(entries))(*[_Chunk(*e) for e in entries])
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self._table = host.to(self.param_groups[0]["params"][0].device)
            self._n = len(entries)
            self._sig = sig
            first_chunk.append(len(entries))
            self._seg_ranges = [(first_chunk[a], first_chunk[b] - first_chunk[a]) for a, b in bounds] if bounds else [(0, len(entries))]

    @torch.no_grad()
    def step_segment(self, i):
        """Update the parameters of segment i only (see set_segments); a full iteration calls it once per segment, plus once
        for index len(segments) (parameters outside every segment)."""
        if i == 0 or self._table is None:
            self._build()                     # once per iteration: pointer signature check (rebuilds only if something moved)
        start, count = self._seg_ranges[i]
        if count:
            self._launch(start, count)

    @property
    def num_segments(self):
        return len(self._segments) + 1 if self._segments else 1

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._build()
        if not self._n:
            return loss
        self._launch(0, self._n)
        return loss

    def _launch(self, start, count):
        loss = None
        table = self._table.data_ptr() + start * C.sizeof(_Chunk)
        if self.capturable:
            if not torch.cuda.is_current_stream_capturing():
                self.push_hyper()                  # eager step: always current.  Captured step: the caller pushes before each replay
            elif self._hyper_dev is None:
                raise RuntimeError("segmi.optim.SGD(capturable=True): run one eager step (or push_hyper()) before capturing")
            check(lib.segmi_sgd_step_dev(table, count, self._hyper_dev.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "sgd_step_dev")
            return loss
        ng = len(self.param_groups)
        f = (C.c_float * ng)
        lr = f(*[float(g["lr"]) for g in self.param_groups])
        wd = f(*[float(g["weight_decay"]) for g in self.param_groups])
        mom = f(*[float(g["momentum"]) for g in self.param_groups])
        check(lib.segmi_sgd_step(table, count, C.addressof(lr), C.addressof(wd), C.addressof(mom), ng,
                                 torch.cuda.current_stream().cuda_stream), "sgd_step")
        return loss
