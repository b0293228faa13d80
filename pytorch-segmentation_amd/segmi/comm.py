"""AbiCommunicator — the RCCL exchange steps through libsegmi's own C ABI (include/segmi.h segmi_comm_*), the transport a
non-torch host would use.  The Python drop-in's default transport stays torch.distributed (backend "nccl" = RCCL); this class is
the same collectives on a communicator libsegmi owns, selected with SEGMI_COMM=abi (GradAllReducer) or used directly.

Bootstrap: rank 0 draws the RCCL unique id, the bytes travel to the other ranks through whatever process group already exists
(one small broadcast over gloo / the default group), then every rank joins with segmi_comm_init on its current HIP device.
"""
import ctypes as C

import torch
import torch.distributed as dist

from ._lib import SegmiError, check, lib


class AbiCommunicator:
    def __init__(self, world=None, rank=None, group=None, device=None):
        if not lib.segmi_comm_available():
            raise SegmiError("segmi_comm: no librccl could be bound at run time")
        ddp = dist.is_available() and dist.is_initialized()
        self.world = (dist.get_world_size(group) if ddp else 1) if world is None else int(world)
        self.rank = (dist.get_rank(group) if ddp else 0) if rank is None else int(rank)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        n = lib.segmi_comm_unique_id_bytes()
        buf = (C.c_ubyte * n)()
        if self.rank == 0:
            check(lib.segmi_comm_get_unique_id(buf, n), "comm_get_unique_id")
        if self.world > 1:
            if not ddp:
                raise SegmiError("segmi_comm: a process group is needed to ship the unique id to %d ranks" % self.world)
            t = torch.tensor(list(buf), dtype=torch.uint8)
            if dist.get_backend(group) == "nccl":
                t = t.to(self.device)
            dist.broadcast(t, src=0, group=group)
            buf = (C.c_ubyte * n)(*t.cpu().tolist())
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.segmi_comm_init(C.byref(handle), self.world, self.rank, buf, n), "comm_init")
        self._h = handle

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def all_reduce_async(self, t, average=False, out=None):
        """Enqueue all-reduce(t) -> out (in place by default) behind everything on the current stream; returns immediately."""
        out = t if out is None else out
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or not out.is_contiguous():
            raise SegmiError("segmi_comm.all_reduce: contiguous float32 CUDA tensors only")
        check(lib.segmi_comm_allreduce_async(self._h, t.data_ptr(), out.data_ptr(), t.numel(), 1 if average else 0, self._stream()), "comm_allreduce_async")
        return out

    def all_gather_async(self, t):
        out = torch.empty(self.world * t.numel(), dtype=torch.float32, device=t.device)
        check(lib.segmi_comm_allgather_async(self._h, t.contiguous().data_ptr(), out.data_ptr(), t.numel(), self._stream()), "comm_allgather_async")
        return out

    def wait(self):
        """The current stream waits for this communicator's last collective (no host synchronisation)."""
        check(lib.segmi_comm_wait(self._h, self._stream()), "comm_wait")

    def close(self):
        if self._h is not None:
            lib.segmi_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
