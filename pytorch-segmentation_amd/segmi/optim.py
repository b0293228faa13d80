"""Fused optimizers on libsegmi.

`SGD` is a drop-in for `torch.optim.SGD(params, lr, momentum, weight_decay)` as the reference instantiates it by name from
config.json (`base/base_trainer.py:57` `get_instance(torch.optim, 'optimizer', config, trainable_params)`), including parameter
groups with their own lr (differential learning rates).  One kernel launch updates every parameter (`segmi_sgd_step`) instead
of torch's three foreach passes.  State layout (`state[p]['momentum_buffer']`) matches torch.optim.SGD, so optimizer
state_dicts interchange with checkpoints written by the reference.
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

    def _build(self):
        """Chunk table on the device; rebuilt only if a gradient / parameter / buffer pointer changed."""
        ce = lib.segmi_sgd_chunk_elems()
        entries, sig = [], []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "momentum_buffer" not in st or st["momentum_buffer"] is None:
                    st["momentum_buffer"] = torch.zeros(p.numel(), dtype=p.dtype, device=p.device).as_strided(p.shape, p.stride())
                buf, g = st["momentum_buffer"], p.grad
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

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._build()
        if not self._n:
            return loss
        if self.capturable:
            if not torch.cuda.is_current_stream_capturing():
                self.push_hyper()                  # eager step: always current.  Captured step: the caller pushes before each replay
            elif self._hyper_dev is None:
                raise RuntimeError("segmi.optim.SGD(capturable=True): run one eager step (or push_hyper()) before capturing")
            check(lib.segmi_sgd_step_dev(self._table.data_ptr(), self._n, self._hyper_dev.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "sgd_step_dev")
            return loss
        ng = len(self.param_groups)
        f = (C.c_float * ng)
        lr = f(*[float(g["lr"]) for g in self.param_groups])
        wd = f(*[float(g["weight_decay"]) for g in self.param_groups])
        mom = f(*[float(g["momentum"]) for g in self.param_groups])
        check(lib.segmi_sgd_step(self._table.data_ptr(), self._n, C.addressof(lr), C.addressof(wd), C.addressof(mom), ng,
                                 torch.cuda.current_stream().cuda_stream), "sgd_step")
        return loss
