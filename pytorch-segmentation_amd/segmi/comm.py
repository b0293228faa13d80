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
        self.last_ticket = 0
        self._pending = {}          # ticket -> tensors the side stream still uses

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    MAX_PENDING = 48      # below the library's ring of 64 per-call events: an unwaited ticket must still be orderable when it is retired

    def _hold(self, ticket, tensors):
        """Keep the buffers of collective `ticket` referenced until a wait covers it.  A caller that never waits (a backward
        without finish(), an exception between enqueue and wait) must not grow this table without bound: beyond
        MAX_PENDING entries the current stream is ordered behind the OLDEST half — collectives run in order on the
        communicator's stream, so by then they are long complete and the wait is free — and their buffers are released."""
        if len(self._pending) >= self.MAX_PENDING:
            self.wait(sorted(self._pending)[len(self._pending) // 2 - 1])
        self._pending[ticket] = tensors

    def _check(self, *tensors):
        for t in tensors:
            if t.dtype != torch.float32 or not t.is_cuda or t.device != self.device or not t.is_contiguous():
                raise SegmiError("segmi_comm: contiguous float32 tensors on %s only (got %s %s on %s)" % (self.device, t.dtype, tuple(t.shape), t.device))

    def all_reduce_async(self, t, average=False, out=None):
        """Enqueue all-reduce(t) -> out (in place by default) behind everything on the current stream; returns `out` immediately.
        The collective's ticket is `self.last_ticket` (pass it to wait() to wait for exactly this call)."""
        out = t if out is None else out
        self._check(t, out)
        ticket = C.c_long(0)
        with torch.cuda.device(self.device):
            check(lib.segmi_comm_allreduce_async(self._h, t.data_ptr(), out.data_ptr(), t.numel(), 1 if average else 0, C.byref(ticket), self._stream()),
                  "comm_allreduce_async")
        self.last_ticket = ticket.value
        # the side stream reads / writes these buffers: they stay referenced until a wait() has ordered a stream behind the call
        self._hold(self.last_ticket, (t, out))
        return out

    def all_gather_async(self, t):
        """Enqueue all-gather(t) -> a new [world * numel] tensor behind everything on the current stream; returns it immediately
        (its contents are valid for streams that have wait()ed on `self.last_ticket`)."""
        src = t.contiguous()
        self._check(src)
        out = torch.empty(self.world * src.numel(), dtype=torch.float32, device=self.device)
        ticket = C.c_long(0)
        with torch.cuda.device(self.device):
            check(lib.segmi_comm_allgather_async(self._h, src.data_ptr(), out.data_ptr(), src.numel(), C.byref(ticket), self._stream()), "comm_allgather_async")
        self.last_ticket = ticket.value
        self._hold(self.last_ticket, (src, out))         # incl. the contiguous copy: freed only after a wait() covers this call
        return out

    def wait(self, ticket=None):
        """The current stream waits for collective `ticket` (default: this communicator's last one) — no host synchronisation.
        Collectives run in order on the communicator's stream, so every earlier call is complete for this stream as well; their
        buffers are released here (the caching allocator may then reuse them on this stream, which is ordered behind them)."""
        ticket = self.last_ticket if ticket is None else int(ticket)
        if ticket <= 0:
            return
        with torch.cuda.device(self.device):
            check(lib.segmi_comm_wait_ticket(self._h, ticket, self._stream()), "comm_wait_ticket")
        for k in [k for k in self._pending if k <= ticket]:
            del self._pending[k]

    def close(self):
        if self._h is not None:
            lib.segmi_comm_destroy(self._h)      # synchronises the side stream
            self._h = None
            self._pending.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
