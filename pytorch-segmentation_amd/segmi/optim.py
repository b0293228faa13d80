"""Fused optimizers on libsegmi.

`SGD` is a drop-in for `torch.optim.SGD(params, lr, momentum, weight_decay)` as the reference instantiates it by name from
config.json (`base/base_trainer.py:57` `get_instance(torch.optim, 'optimizer', config, trainable_params)`), including parameter
groups with their own lr (differential learning rates).  One kernel launch updates every parameter (`segmi_sgd_step`) instead
of torch's three foreach passes.  State layout (`state[p]['momentum_buffer']`) matches torch.optim.SGD, so optimizer
state_dicts interchange with checkpoints written by the reference (buffers that arrive contiguous-NCHW for a channels_last
filter are re-laid in the parameter's memory order on the first step).
"""
import ctypes as C
import os
import time

import numpy as np
import torch

from ._lib import check, lib


class _Chunk(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("momentum", C.c_void_p), ("count", C.c_long), ("group", C.c_int),
                ("vec4", C.c_int)]


_CHUNK_DTYPE = np.dtype([("param", "<u8"), ("grad", "<u8"), ("momentum", "<u8"), ("count", "<i8"), ("group", "<i4"), ("vec4", "<i4")])
assert _CHUNK_DTYPE.itemsize == C.sizeof(_Chunk)


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
        self._tab_ring = self._tab_events = self._step_end = None
        self._tab_pos = 0
        self._auto, self._drains = None, []       # `auto` upload mode: decided from the drain times of the first steps

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
        """Chunk table on the device; rebuilt only if a gradient / parameter / buffer pointer changed.

        The per-step work is the pointer signature alone.  A rebuild — every step when gradients are freshly allocated tensors
        (`zero_grad(set_to_none=True)` without a reducer's persistent buckets: the allocator hands the side-stream filter gradients
        different blocks from step to step) — lays the table out with numpy and hands it to `_upload`."""
        ce = lib.segmi_sgd_chunk_elems()
        sig, first_param = [], []            # first_param[i]: how many live parameters precede parameter i of `order`
        order, bounds = self._ordered()
        for gi, p in order:
            first_param.append(len(sig))
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
            sig.append((p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel(), gi))
        first_param.append(len(sig))
        sig = tuple(sig)
        if sig == self._sig:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("segmi.optim.SGD: a parameter / gradient / momentum pointer changed while a hipGraph is being "
                               "captured; the pointer table is uploaded from host memory and cannot be rebuilt inside a capture. "
                               "Keep gradients persistent (segmi.distributed.DistributedModel holds them as bucket views, also in a "
                               "single process) and run at least one eager step before capturing (segmi.graph.GraphedStep does).")
        a = np.array(sig, dtype=np.int64).reshape(-1, 5)
        nch = (a[:, 3] + ce - 1) // ce                                   # chunks per live parameter
        first_chunk = np.concatenate([[0], np.cumsum(nch)])              # chunk index at which live parameter j starts
        rep = np.repeat(np.arange(len(a)), nch)
        off = (np.arange(int(first_chunk[-1])) - first_chunk[rep]) * ce  # element offset of each chunk inside its parameter
        tab = np.zeros(int(first_chunk[-1]), dtype=_CHUNK_DTYPE)
        tab["param"], tab["grad"], tab["momentum"] = a[rep, 0] + 4 * off, a[rep, 1] + 4 * off, a[rep, 2] + 4 * off
        tab["count"] = np.minimum(ce, a[rep, 3] - off)
        tab["group"] = a[rep, 4]
        tab["vec4"] = (tab["count"] % 4 == 0) & (tab["param"] % 16 == 0) & (tab["grad"] % 16 == 0) & (tab["momentum"] % 16 == 0)
        self._table = self._upload(tab, self.param_groups[0]["params"][0].device)
        self._n = len(tab)
        self._sig = sig
        fc = [int(first_chunk[j]) for j in first_param]
        self._seg_ranges = [(fc[lo], fc[hi] - fc[lo]) for lo, hi in bounds] if bounds else [(0, len(tab))]

    def _upload(self, tab, device):
        """The table as a device tensor.

        `tensor.to(device)` from pageable memory (rounds 1-5) blocks the host until everything queued before it has run — the end
        of backward.  On a GPU-paced step the host then starts the next step's launches on an idle GPU: 1.1 ms of every cfg2 step,
        2 ms of cfg3, 2-4 ms of cfg5 (profiles/r06_gpu_gaps.txt).  `throttled`: a stream-ordered copy out of a ring of pinned staging
        slots (a slot is reused only after the event of its previous copy has completed) and ONE wait per step on the event behind
        the PREVIOUS step's update, so the host stays within one step of the GPU (with no bound at all the caching allocator
        holds three generations of activations: cfg2 19 -> 61 GB reserved; one step ahead: 31 GB).  On a LAUNCH-paced step (cfg1:
        6.3 ms of kernels in a 7 ms step) draining the queue once per step measured FASTER than never draining it (7.0 against
        11.5 ms per step, profiles/r06_sgd_table_upload_ab.txt), so `auto` (default) times the drain of its first three rebuilds and keeps
        the blocking upload when the GPU is within 2 ms of the host.  SEGMI_SGD_TABLE_UPLOAD=blocking|throttled|auto."""
        raw = torch.from_numpy(tab.view(np.uint8).reshape(-1))
        mode = os.environ.get("SEGMI_SGD_TABLE_UPLOAD", "auto")
        if device.type != "cuda":
            return raw.to(device)
        if mode != "blocking" and self._tab_ring is None:
            # every staging slot up front, in the first step: pinning host memory synchronises with the device and can take
            # milliseconds — it must not happen in the middle of a run when `auto` changes its mind
            cap = max(2 * raw.numel(), 1 << 16)
            self._tab_ring = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(self._RING)]
            self._tab_events, self._tab_pos = [None] * self._RING, 0
        if mode == "auto":
            if self._auto is None:
                t0 = time.perf_counter()
                torch.cuda.current_stream(device).synchronize()
                self._drains.append(time.perf_counter() - t0)
                if len(self._drains) >= 3:                       # (the first sample is the cold step)
                    self._auto = "throttled" if min(self._drains[1:]) > 2e-3 else "blocking"
                return raw.to(device)
            mode = self._auto
        if mode != "throttled":
            return raw.to(device)
        if self._step_end is not None:
            self._step_end.synchronize()         # the host runs at most one step ahead of the GPU (see _mark_step_end)
        i = self._tab_pos
        self._tab_pos = (i + 1) % self._RING
        if self._tab_events[i] is not None:
            self._tab_events[i].synchronize()
        if self._tab_ring[i].numel() < raw.numel():
            self._tab_ring[i] = torch.empty(2 * raw.numel(), dtype=torch.uint8).pin_memory()
        stage = self._tab_ring[i][:raw.numel()]
        stage.copy_(raw)
        dev = torch.empty(raw.numel(), dtype=torch.uint8, device=device)
        dev.copy_(stage, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._tab_events[i] = ev
        return dev

    @torch.no_grad()
    def step_segment(self, i):
        """Update the parameters of segment i only (see set_segments); a full iteration calls it once per segment, plus once
        for index len(segments) (parameters outside every segment)."""
        if i == 0 or self._table is None:
            self._build()                     # once per iteration: pointer signature check (rebuilds only if something moved)
        start, count = self._seg_ranges[i]
        if count:
            self._launch(start, count)
        if i == len(self._seg_ranges) - 1:
            self._mark_step_end()

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
        self._mark_step_end()
        return loss

    def _mark_step_end(self):
        """Event behind the iteration's last update: a throttled `_upload` of the NEXT iteration waits for it."""
        if self._tab_ring is not None and not torch.cuda.is_current_stream_capturing():
            if self._step_end is None:
                self._step_end = torch.cuda.Event()
            self._step_end.record()

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
