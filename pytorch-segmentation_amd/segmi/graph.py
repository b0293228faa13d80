"""hipGraph capture of a whole training step.

The reference's inner loop (trainer.py:55-71: zero_grad -> model(data) -> loss -> backward -> optimizer.step) issues
300-900 kernel launches per iteration from Python.  On the large configurations the GPU is the bottleneck and the launches
hide behind it, but small problems are launch-paced (UNet 2x3x256x256: 323 launches in 9.6 ms), and with eight ranks
sharing the host's cores launch jitter becomes step-time jitter.  `GraphedStep` captures one step into a hipGraph
(through torch's capture plumbing: private memory pool + stream capture; every segmi kernel is a plain launch on the
capturing stream, so it is recorded like any other) and replays it with ONE host call per iteration.

What makes a step capturable here
  * no host synchronisation inside the step: losses / metrics stay on the device (`segmi.ops`), SyncBN's global element count
    is summed on the device from the gathered partials (`segmi_bn_finalize(count_out)`);
  * side-stream filter gradients (`ops.set_wgrad_stream`) fork from and re-join the capturing stream inside the step (the join is
    an autograd-engine callback at the end of backward), which stream capture records as ordinary cross-stream dependencies;
  * dropout: the by-value seed would be frozen at capture, so the kernels fold a DEVICE-side epoch counter into the seed
    (`segmi_dropout(..., seed_epoch_dev)`), advanced by one captured `add_` at the end of the step;
  * stable gradient addresses: the fused SGD walks a device table of (param, grad, momentum) pointers that is built on
    the host.  Keep `.grad` persistent — `segmi.distributed.DistributedModel` / `GradAllReducer` hold gradients as views
    of flat buckets also in a single process — otherwise `segmi.optim.SGD.step` refuses to rebuild its table mid-capture;
  * hyper-parameters passed by value are frozen at capture: `segmi.optim.SGD(capturable=True)` keeps lr / momentum /
    weight decay in device memory and refreshes them with a stream-ordered copy issued BEFORE each replay
    (`GraphedStep(step_fn, pre_replay=optimizer.push_hyper)`), so schedulers keep working.
"""
import torch

from . import ops
from ._lib import SegmiError


class GraphedStep:
    """graphed = GraphedStep(step_fn); loss = graphed()   # one hipGraphLaunch per call

    `step_fn()` performs one complete training step on static (pre-allocated, device-resident) inputs and returns a
    tensor or a tuple of tensors (e.g. the loss); the returned tensors are overwritten in place by every replay.
    Refill the static inputs with `copy_` between calls to feed new batches.  Construction runs `warmup` eager steps (real
    training steps) and then captures; the capture itself executes nothing, so `outputs` is defined only after the first
    replay."""

    def __init__(self, step_fn, warmup=3, device=None, pre_replay=None):
        if not torch.cuda.is_available():
            raise SegmiError("segmi.graph.GraphedStep needs the MI355X (hipGraph capture; there is no CPU path)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.epoch = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._step_fn = step_fn
        self._pre_replay = pre_replay     # eager stream work ordered before each replay (e.g. optimizer.push_hyper, input copy_)
        self.replays = 0
        ops.set_dropout_epoch(self.epoch)
        # warm-up on a side stream (lazy allocations, workspace growth, SGD pointer table, hipFuncSetAttribute calls and the
        # SyncBN count exchange all happen here, outside the capture)
        # warmup=0: the caller has already run the step eagerly at least once (a trainer that must not take an extra step)
        if int(warmup) > 0:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(int(warmup)):
                    self._body()
            torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = self._body()

    def _body(self):
        out = self._step_fn()
        self.epoch.add_(1)          # next step (or replay) draws fresh dropout masks
        return out

    def __call__(self):
        if self._pre_replay is not None:
            self._pre_replay()
        self.graph.replay()
        self.replays += 1
        return self.outputs

    def close(self):
        """Detach the dropout epoch (eager steps draw host seeds again) and drop the graph."""
        if ops._DROPOUT_EPOCH is self.epoch:
            ops.set_dropout_epoch(None)
        self.graph = None
