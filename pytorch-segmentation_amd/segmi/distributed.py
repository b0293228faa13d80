"""Data-parallel training: one process per GPU, RCCL over xGMI through torch.distributed.

The reference parallelises with single-process `nn.DataParallel` (base/base_trainer.py:33-38): scatter the
minibatch, replicate the module, gather outputs, reduce gradients onto GPU 0 — a per-iteration
broadcast + gather pattern that is bound by one GPU's links.  The MI355X design shards the minibatch
along N across ranks instead; the only exchange steps per iteration are

  * `GradAllReducer`  — bucketed gradient all-reduce (average), launched from autograd hooks as soon
    as a bucket's gradients exist, on a side HIP stream so it overlaps the rest of backward;
  * `SyncBNContext`   — SyncBN statistics: all-gather of the per-rank Welford partials [n, mean, M2]
    in forward and all-reduce of [sum dy, sum dy*xhat] in backward
    (utils/sync_batchnorm/batchnorm.py:70-93,105-126 of the reference).

xGMI is point-to-point (7 links per GPU), so collectives are per-link bound: buckets are large where
their transfer hides behind the rest of backward and small where it cannot.  Gradients arrive in
reverse parameter order; every bucket takes HALF of the bytes that are still to come, clamped to
[4 MiB, 64 MiB] (`bucket_schedule`): PSPNet-R50's 206 MB go out as 64 / 64 / 39 / 20 / 10 / 5 / 5 MB, so
the collective that starts when backward ends — the stem and layer1 gradients, which nothing can
overlap — moves 5 MB instead of the 14 MB remainder of a uniform 64 MiB split, and the early, large
ones still amortise the launch latency (DESIGN.md §7 has the expected-scaling model).

Everything here is device-agnostic torch.distributed code (tests run it on CPU with gloo, world 2);
on the GPU box backend "nccl" IS RCCL.
"""
import os

import torch
import torch.distributed as dist


def _is_dist():
    return dist.is_available() and dist.is_initialized()


def global_batch_mean(local_sum, local_den, group=None):
    """Per-rank terms of a GLOBAL-batch mean under gradient averaging.

    The reference computes its loss on the gathered global batch (nn.DataParallel, trainer.py:56-66): for CrossEntropy that is
    sum over ALL shards of the per-pixel losses / number of valid pixels of ALL shards (utils/losses.py:29-31).  One process per
    GPU back-propagates a per-rank loss and AVERAGES gradients; with `W * local_sum / global_den` as the per-rank loss both the
    average of the losses and the average of the gradients equal the global-batch value, also when shards hold different numbers
    of valid pixels (equal shards reduce it to the local mean).  One all-reduce of a single float.

    local_sum, local_den: 0-d tensors (device-agnostic).  Returns (loss term of this rank, denominator to divide the local
    per-pixel gradients by = global_den / W)."""
    world = dist.get_world_size(group) if _is_dist() else 1
    if world == 1:
        return local_sum / local_den, local_den
    den = local_den.detach().clone().reshape(1)
    dist.all_reduce(den, group=group)
    den = den[0] / world
    return local_sum / den, den


def bucket_schedule(total_bytes, max_bytes=64 << 20, min_bytes=4 << 20):
    """Capacities (bytes) of the gradient buckets in FILL order (= the order backward produces gradients): each bucket holds half
    of what remains, clamped to [min_bytes, max_bytes]; the last one takes the rest."""
    caps, left = [], int(total_bytes)
    while left > 0:
        c = min(max(left // 2, min_bytes), max_bytes)
        if left - c < min_bytes // 2:          # do not leave a sliver behind
            c = left
        caps.append(c)
        left -= c
    return caps


class _AbiWork:
    """Handle of one collective on an AbiCommunicator: wait() orders the current stream behind that call only (its own event),
    so finish() steps bucket i while buckets i+1... are still in flight — like torch.distributed's Work objects."""
    __slots__ = ("comm", "ticket")

    def __init__(self, comm, ticket):
        self.comm, self.ticket = comm, ticket

    def wait(self):
        self.comm.wait(self.ticket)


class GradAllReducer:
    """Bucketed, overlapped gradient averaging for the parameters of one model replica.

    Gradients live as views into flat per-bucket buffers (no flatten/unflatten copies in steady
    state); buckets are filled in reverse parameter order (the order backward produces gradients).
    Usage per iteration:  zero_grad() -> forward -> loss.backward() -> finish() -> optimizer.step().
    """

    def __init__(self, params, process_group=None, bucket_bytes=None, average=True, always_reduce=False):
        """bucket_bytes: None = the backward-order schedule (`bucket_schedule`); an int = uniform buckets of that size."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if _is_dist() else 1
        self.average = average
        # always_reduce: issue the collectives even in a single-rank group (exercises the RCCL call path on a 1-GPU box)
        self.collective = self.world > 1 or (always_reduce and _is_dist())
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradAllReducer: no trainable parameters")
        dev = self.params[0].device
        self.device = dev
        self.side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        # reverse registration order ~ order in which backward produces gradients
        order = list(reversed(self.params))
        self.buckets = []          # dict(buf, params, pending, launched, work)
        self._where = {}           # id(param) -> (bucket index, view)
        # bucket_schedule's rule applied to the bytes ACTUALLY still to come when a bucket is opened: a tensor larger than its bucket's
        # capacity (PSPNet's 72 MB bottleneck filter) closes the bucket before it early and travels alone — indexing a precomputed
        # schedule by bucket number then ran out of large capacities and cut the last 40 MB into ten 4 MB buckets (17 buckets for
        # PSPNet-R50: 17 collectives + 17 optimizer launches per step; now 9)
        left = sum(p.numel() * p.element_size() for p in order)
        cap = bucket_bytes if bucket_bytes is not None else bucket_schedule(left)[0]
        cur, cur_bytes = [], 0
        for p in order:
            nb = p.numel() * p.element_size()
            if cur and cur_bytes + nb > cap and (bucket_bytes is not None or cur_bytes >= (1 << 20)):
                # (a bucket still below 1 MiB — a few BN vectors — takes the oversized tensor in instead of travelling alone)
                self._make_bucket(cur)
                left -= cur_bytes
                cur, cur_bytes = [], 0
                cap = bucket_bytes if bucket_bytes is not None else bucket_schedule(left)[0]
            cur.append(p)
            cur_bytes += nb
        if cur:
            self._make_bucket(cur)
        # Convolution filters on the GPU: the filter-gradient kernels write straight into the bucket slot (segmi.ops._GRAD_SLOTS), so
        # their .grad is left None at zero_grad() and autograd ADOPTS the slot alias the kernel filled — no in-place add, no copy.
        self._slot_params = [p for p in self.params if p.device.type == "cuda" and p.dim() == 4]
        self._ops = None
        if self._slot_params:
            from . import ops as _ops
            self._ops = _ops
            _ops.register_grad_slots({p: self._where[id(p)][1] for p in self._slot_params})
            _ops.mark_reducer_hooks(self._slot_params)
            for p in self._slot_params:
                p.grad = None
        # SEGMI_COMM=abi: the buckets travel through libsegmi's own RCCL entry points (include/segmi.h segmi_comm_*, segmi/comm.py)
        # instead of torch.distributed's communicator — the transport of a non-torch host, selectable here for A/B
        self._abi = None
        if self.collective and os.environ.get("SEGMI_COMM") == "abi" and dev.type == "cuda":
            from .comm import AbiCommunicator
            self._abi = AbiCommunicator(group=process_group, device=dev)
        self.counters = {"adopted": 0, "copied": 0}      # gradients found in place in their slot / copied into it (diagnostics, tests)
        self._fired = set()        # id(param) of the parameters whose gradient arrived in the current iteration
        self.check_unused = os.environ.get("SEGMI_DDP_CHECK_UNUSED", "0") == "1"
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _make_bucket(self, params):
        total = sum(((p.numel() + 3) & ~3) for p in params)   # 16-byte aligned slots
        buf = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
        off, idx = 0, len(self.buckets)
        for p in params:
            n = p.numel()
            # same dense (possibly permuted, e.g. channels_last filter) strides as the parameter
            view = buf[off:off + n].as_strided(p.shape, p.stride()) if p.dim() > 0 else buf[off:off + 1].view(())
            self._where[id(p)] = (idx, view)
            p.grad = view
            off += (n + 3) & ~3
        self.buckets.append({"buf": buf, "params": params, "pending": len(params), "launched": False, "work": None})

    # ------------------------------------------------------------------ per-iteration protocol
    def zero_grad(self):
        """One memset per bucket; parameter .grad stays a view of the bucket."""
        for b in self.buckets:
            b["buf"].zero_()
            b["pending"], b["launched"], b["work"] = len(b["params"]), False, None
        for p in self.params:
            p.grad = self._where[id(p)][1]
        for p in self._slot_params:
            p.grad = None                 # adopted from the slot alias the wgrad kernel writes (see __init__)
        if self._ops is not None:
            self._ops.reset_grad_slots(self._slot_params)
        self._fired.clear()

    def _on_grad(self, p):
        self._fired.add(id(p))
        idx, view = self._where[id(p)]
        g = p.grad
        if g is not view:
            # .grad was None: autograd handed us the tensor the backward function returned — an alias of the bucket slot the wgrad
            # kernel wrote (nothing to do), or a fresh tensor (small parameters, optimizer.zero_grad(set_to_none=True)): one copy
            if g.data_ptr() != view.data_ptr():
                if self._ops is not None and g.is_cuda and g.dim() == 4:
                    # a filter gradient that is not the slot alias may still be in flight on the filter-gradient side stream
                    # (optimizer.zero_grad(set_to_none=True), a slot already handed out): the copy below runs on the compute stream
                    self._ops.wgrad_stream_join()
                view.copy_(g)
                self.counters["copied"] += 1
            else:
                self.counters["adopted"] += 1
            p.grad = view
        b = self.buckets[idx]
        b["pending"] -= 1
        if b["pending"] == 0 and not b["launched"]:
            self._launch(b)

    def _launch(self, b):
        b["launched"] = True
        if not self.collective:
            return
        op = dist.ReduceOp.SUM
        if self._abi is not None:
            if self._ops is not None:
                self._ops.wgrad_stream_join()
            self._abi.all_reduce_async(b["buf"], average=self.average)     # on the communicator's side stream, behind the compute stream
            b["work"], b["scale"] = _AbiWork(self._abi, self._abi.last_ticket), False
            return
        if self.side is not None:
            if self._ops is not None:
                self._ops.wgrad_stream_join()     # filter gradients launched on the wgrad side stream belong to this bucket too
            ev = torch.cuda.Event()
            ev.record()                       # gradients of this bucket are complete on the compute stream
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                if self.average and dist.get_backend(self.group) == "nccl":
                    op = dist.ReduceOp.AVG
                b["work"] = dist.all_reduce(b["buf"], op=op, group=self.group, async_op=True)
            b["scale"] = self.average and op != dist.ReduceOp.AVG
        else:
            b["work"] = dist.all_reduce(b["buf"], op=op, group=self.group, async_op=True)
            b["scale"] = self.average

    def finish(self, optimizer=None):
        """Block the compute stream (not the host, on GPU) until every bucket is reduced.  Buckets whose
        parameters got no gradient this iteration (unused branches) are reduced here with zeros.

        optimizer: a segmi.optim.SGD whose table was laid out per bucket (`optimizer.set_segments(reducer.segments())`): the
        fused update of a bucket's parameters is launched right after the wait on ITS all-reduce, so it runs while the later
        buckets (the gradients backward produced last: stem, layer1) are still being reduced on the side stream — the
        optimizer step of the reference (trainer.py:71, after the whole DataParallel gather) moved into the exchange."""
        for b in self.buckets:
            if not b["launched"]:
                for p in b["params"]:       # parameters without a gradient this iteration contribute zeros
                    view = self._where[id(p)][1]
                    if p.grad is not view:
                        if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                            if self._ops is not None:
                                self._ops.wgrad_stream_join()
                            view.copy_(p.grad)
                        p.grad = view
                self._launch(b)
        # Parameters that received no gradient in this iteration (PSPNet(use_aux=False).auxiliary_branch, heads outside the
        # executed path): the reference leaves their .grad None, so torch.optim.SGD skips them — no weight decay, no momentum
        # update.  Their bucket slots take part in the reduction as zeros (all ranks agree); the view is hidden from the
        # optimizer until zero_grad() re-attaches it for the next iteration.
        if self.check_unused and self.collective:
            self._assert_fired_consistent()
        for p in self.params:
            if id(p) not in self._fired:
                p.grad = None
        self._fired.clear()               # (also cleared by zero_grad(); a caller using optimizer.zero_grad() must not see stale marks)
        for i, b in enumerate(self.buckets):
            w = b["work"]
            if w is not None:
                w.wait()                      # nccl / abi: the current stream waits for THIS bucket's collective only
                if b.get("scale"):
                    b["buf"].div_(self.world)
                b["work"] = None
            b["pending"], b["launched"] = len(b["params"]), False
            if optimizer is not None:
                optimizer.step_segment(i)
        if optimizer is not None:
            for i in range(len(self.buckets), optimizer.num_segments):
                optimizer.step_segment(i)     # parameters the reducer does not own (none for a whole-model reducer)

    def _assert_fired_consistent(self):
        """Debug check (SEGMI_DDP_CHECK_UNUSED=1): the set of parameters that received a gradient must be the same on every rank —
        the decision to hide a parameter from the optimizer is rank-local, and ranks that disagree (a data-dependent branch, an
        aux head skipped on one rank) would apply weight decay / momentum differently and let the replicas drift apart."""
        flags = torch.tensor([1.0 if id(p) in self._fired else 0.0 for p in self.params], device=self.device)
        lo, hi = flags.clone(), flags.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        bad = torch.nonzero(lo != hi).flatten().tolist()
        if bad:
            raise RuntimeError("GradAllReducer: ranks disagree on which parameters received gradients this iteration "
                               "(parameter indices %s): unused-parameter handling would let the replicas diverge" % bad[:8])

    def segments(self):
        """Parameter lists per bucket, in completion order — the layout `segmi.optim.SGD.set_segments` wants."""
        return [list(b["params"]) for b in self.buckets]

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._ops is not None:
            self._ops.register_grad_slots({p: None for p in self._slot_params})
            self._ops.mark_reducer_hooks(self._slot_params, on=False)
            self._slot_params = []

    def __del__(self):
        try:
            self.remove()
        except Exception:
            pass


class SyncBNContext:
    """Collective side of SynchronizedBatchNorm2d (attached to each converted BN module as `.sync`).

    Forward: every rank computes its Welford partial [count, mean, M2] (3*C floats); one all-gather
    hands each rank all `world` partials, which the `segmi_bn_finalize` kernel merges with Chan's
    formula — equal to F.batch_norm over the concatenated global batch (the reference's CPU
    behaviour, batchnorm.py:65-68), not the less accurate E[x^2]-E[x]^2 of its GPU branch
    (batchnorm.py:128-145; opt in with clamp_mode=1 for that variance clamp).
    Backward: all-reduce(sum) of [sum dy, sum dy*xhat] (2*C floats).
    """

    def __init__(self, process_group=None, clamp_mode=0):
        self.group = process_group
        self.clamp_mode = clamp_mode
        self.collectives = 0          # issued by this layer so far (bench.py --sync-bn reports the per-step total)
        self.force_group = False      # tests: take the batched-collective code path in a single-rank group too

    @property
    def world(self):
        return dist.get_world_size(self.group) if _is_dist() else 1

    def gather_stats(self, part):
        """part: [3*C] local partial {count, mean, M2} -> ([world*3*C] all partials, world).  The global element count is NOT
        exchanged separately: every partial carries its own count, the merge kernel (segmi_bn_finalize) sums them on the device
        and hands the total to the backward pass through device memory — correct for ragged shards and for shard sizes that
        change from step to step, and no collective ever depends on rank-local state."""
        w = self.world
        if w == 1:
            return part, 1
        out = torch.empty(w * part.numel(), dtype=part.dtype, device=part.device)
        dist.all_gather_into_tensor(out, part.contiguous(), group=self.group)
        self.collectives += 1
        return out, w

    def gather_stats_many(self, parts):
        """gather_stats for several layers whose partials exist at the same time (parallel branches), as ONE all-gather of the
        concatenated partials: [(all partials of layer i as [world*3*C_i], world)].  Values are exactly those of per-layer calls."""
        w = self.world
        if w == 1:
            return [(p, 1) for p in parts]
        flat = torch.cat([p.reshape(-1) for p in parts])
        out = torch.empty(w * flat.numel(), dtype=flat.dtype, device=flat.device)
        dist.all_gather_into_tensor(out, flat, group=self.group)
        self.collectives += 1
        out = out.view(w, -1)
        res, off = [], 0
        for p in parts:
            n = p.numel()
            res.append((out[:, off:off + n].contiguous().view(-1), w))
            off += n
        return res

    def reduce_sums_many(self, sums_list):
        """reduce_sums for several layers in ONE all-reduce of the concatenated sums (new tensors; inputs untouched)."""
        if self.world == 1:
            return list(sums_list)
        flat = torch.cat([s.reshape(-1) for s in sums_list])
        dist.all_reduce(flat, group=self.group)
        self.collectives += 1
        res, off = [], 0
        for s in sums_list:
            n = s.numel()
            res.append(flat[off:off + n])
            off += n
        return res

    def reduce_sums(self, sums):
        """[2*C] local {sum dy, sum dy*xhat} -> global sums (new tensor; the local ones remain the
        per-rank parameter gradients, which the gradient all-reduce averages like any other)."""
        if self.world == 1:
            return sums
        g = sums.clone()
        dist.all_reduce(g, group=self.group)
        self.collectives += 1
        return g


class DistributedModel(torch.nn.Module):
    """Per-rank wrapper with the attribute the reference's trainer looks for (`.module`,
    base/base_trainer.py:47-51, trainer.py:41-43) plus the gradient reducer.  Parameters are
    broadcast from rank 0 at construction so all replicas start identical."""

    def __init__(self, module, process_group=None, bucket_bytes=None, always_reduce=False):
        super().__init__()
        self.module = module
        if _is_dist() and dist.get_world_size(process_group) > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                d = t.data
                if not d.is_contiguous():
                    # permuted-dense parameters (channels_last filters): broadcast the underlying memory run —
                    # collectives on non-contiguous tensors may act on a temporary copy
                    d = torch.as_strided(d, (d.numel(),), (1,), d.storage_offset())
                dist.broadcast(d, src=0, group=process_group)
        self.reducer = GradAllReducer(module.parameters(), process_group, bucket_bytes, always_reduce=always_reduce)

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def broadcast_buffers(self, src=0):
        """Rank `src`'s floating-point buffers (the BatchNorm running statistics) to every rank, as ONE broadcast.  Without SyncBN
        every rank updates its running statistics from ITS shard's batch statistics, so the replicas' buffers drift apart while
        their parameters stay identical; the reference's nn.DataParallel keeps ONE set — the module's own, i.e. replica 0's
        (base/base_trainer.py:33-38: replicas are re-made from the module at every forward) — and validates and checkpoints
        with it.  Called before a validation pass (Trainer._valid_epoch) so that every rank evaluates its share of the validation
        set with the statistics rank 0 would have used on all of it.  A collective: every rank calls it.  No-op for one rank."""
        if not (_is_dist() and dist.get_world_size(self.reducer.group) > 1):
            return
        bufs = [b for b in self.module.buffers() if b.is_floating_point() and b.numel() > 0]
        if not bufs:
            return
        flat = torch.cat([b.detach().reshape(-1) for b in bufs])
        dist.broadcast(flat, src=src, group=self.reducer.group)
        off = 0
        for b in bufs:
            n = b.numel()
            b.detach().copy_(flat[off:off + n].view_as(b))
            off += n

    def zero_grad(self, set_to_none=False):
        self.reducer.zero_grad()

    def finish_gradients(self, optimizer=None):
        """Wait for the gradient all-reduces; with `optimizer` (a segmi.optim.SGD prepared by `attach_optimizer`) also apply
        the update bucket by bucket — the caller then must NOT call optimizer.step() for this iteration."""
        self.reducer.finish(optimizer)

    def attach_optimizer(self, optimizer):
        """Lay a segmi.optim.SGD out per gradient bucket so that finish_gradients(optimizer) can step each bucket as soon as its
        all-reduce is done.  Returns True when the optimizer supports it (other optimizers keep the plain step())."""
        if hasattr(optimizer, "set_segments"):
            optimizer.set_segments(self.reducer.segments())
            return True
        return False
