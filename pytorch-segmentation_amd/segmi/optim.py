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
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0, weight_decay=0.0, nesterov=False):
        if dampening != 0 or nesterov:
            raise NotImplementedError("segmi.optim.SGD implements dampening=0, nesterov=False (the reference's configuration)")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=False))
        if len(self.param_groups) > 8:
            raise ValueError("segmi.optim.SGD supports up to 8 parameter groups")
        self._table = None
        self._sig = None

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
        ng = len(self.param_groups)
        f = (C.c_float * ng)
        lr = f(*[float(g["lr"]) for g in self.param_groups])
        wd = f(*[float(g["weight_decay"]) for g in self.param_groups])
        mom = f(*[float(g["momentum"]) for g in self.param_groups])
        check(lib.segmi_sgd_step(self._table.data_ptr(), self._n, C.addressof(lr), C.addressof(wd), C.addressof(mom), ng,
                                 torch.cuda.current_stream().cuda_stream), "sgd_step")
        return loss
