"""GPU: hipGraph capture of a training step (segmi/graph.py) and the device-side dropout epoch that makes it valid.

First run on hardware: round 2.  Replays equal eager steps bit for bit; measured benefit on one GPU: none on cfg2 (88.8 vs 89.0
img/s) and 1 % on the launch-heavy cfg1 (209 vs 206.5 img/s) — the GPU, not the host, paces these steps — so GraphedStep stays an
opt-in tool (bench.py --graph) and the trainer integration of round 1 was removed."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dropout_epoch_changes_masks_and_backward_regenerates_them(cuda):
    from segmi import ops
    x = torch.ones(2, 64, 16, 16, device=cuda, requires_grad=True)
    ep = torch.zeros(1, dtype=torch.int64, device=cuda)
    try:
        ops.set_dropout_epoch(ep)
        torch.manual_seed(3)
        y0 = ops.dropout(x, 0.5)
        y0.sum().backward()
        g0 = x.grad.clone()
        assert torch.equal((y0 != 0), (g0 != 0))                 # backward saw the same mask as forward
        x.grad = None
        ep.add_(1)
        torch.manual_seed(3)                                      # same host seed, next epoch
        y1 = ops.dropout(x, 0.5)
        frac_diff = ((y0 != 0) != (y1 != 0)).float().mean().item()
        assert 0.4 < frac_diff < 0.6                              # independent masks differ on ~half the elements
        assert abs((y1 != 0).float().mean().item() - 0.5) < 0.02
    finally:
        ops.set_dropout_epoch(None)
    torch.manual_seed(3)
    y2 = ops.dropout(x, 0.5)                                      # no epoch installed: the by-value seed alone == epoch 0
    assert torch.equal(y2 != 0, y0 != 0)
    with pytest.raises(Exception):
        ops.set_dropout_epoch(torch.zeros(1))                     # CPU tensor refused


def _make(cuda, dropout_on):
    import models
    from segmi.distributed import DistributedModel
    from segmi.optim import SGD
    from utils.losses import CrossEntropyLoss2d
    torch.manual_seed(0)
    m = models.PSPNet(5, backbone="resnet50", pretrained=False).to(cuda).train()
    if not dropout_on:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout2d):
                mod.eval()
    dm = DistributedModel(m)                                      # persistent gradient buckets, no process group
    opt = SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    crit = CrossEntropyLoss2d(ignore_index=255)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 3, 64, 64, generator=g).to(cuda)
    t = torch.randint(0, 5, (2, 64, 64), generator=g).to(cuda)

    def step():
        dm.zero_grad()
        out, aux = m(x)
        loss = crit(out, t) + 0.4 * crit(aux, t)
        loss.backward()
        dm.finish_gradients()
        opt.step()
        return loss
    return m, step


def test_graph_replays_equal_eager_steps(cuda):
    """Dropout off, every kernel deterministic: warm-up (3 eager) + capture (1) + k replays == 4 + k eager steps, bit for bit."""
    from segmi.graph import GraphedStep
    m1, step1 = _make(cuda, dropout_on=False)
    losses_eager = [step1().item() for _ in range(7)]
    m2, step2 = _make(cuda, dropout_on=False)
    gs = GraphedStep(step2, warmup=3)                              # 3 eager + 1 captured (capture does not execute)
    losses_graph = [gs().item() for _ in range(4)]
    try:
        # capture itself does not run the kernels: replays 1..4 are steps 4..7
        assert losses_graph == losses_eager[3:7], (losses_graph, losses_eager)
        for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert torch.equal(p1, p2), n1
    finally:
        gs.close()


def test_graph_with_dropout_draws_fresh_masks(cuda):
    from segmi.graph import GraphedStep
    m, step = _make(cuda, dropout_on=True)
    gs = GraphedStep(step, warmup=2)
    try:
        e0 = int(gs.epoch.item())
        ls = [gs().item() for _ in range(3)]
        assert int(gs.epoch.item()) == e0 + 3                     # the captured add_ advances the epoch on every replay
        assert all(torch.isfinite(torch.tensor(ls)))
    finally:
        gs.close()


def test_capturable_sgd_follows_the_lr_schedule_inside_a_graph(cuda):
    """lr changes every iteration (the reference's Poly schedule does, trainer.py:52): a graph captured once must apply the
    lr of each replay — device-resident hyper-parameters pushed before the replay — and match eager steps bit for bit."""
    from segmi.graph import GraphedStep
    from segmi.optim import SGD
    torch.manual_seed(1)
    w0 = torch.randn(1000, device=cuda)
    grads = [torch.randn(1000, device=cuda) for _ in range(6)]
    lrs = [0.1 * (1 - i / 6.0) ** 0.9 for i in range(6)]

    def run(capturable, graphed):
        w = torch.nn.Parameter(w0.clone())
        w.grad = torch.zeros_like(w)
        opt = SGD([w], lr=lrs[0], momentum=0.9, weight_decay=1e-4, capturable=capturable)
        src = torch.zeros_like(w)

        def step():
            w.grad.copy_(src)
            opt.step()
            return w

        gs = None
        for i in range(6):
            opt.param_groups[0]["lr"] = lrs[i]
            src.copy_(grads[i])
            (gs() if gs is not None else step())
            if graphed and i == 2:                      # three eager steps, then capture (warmup=0: capture executes nothing)
                gs = GraphedStep(step, warmup=0, pre_replay=opt.push_hyper)
        torch.cuda.synchronize()
        if gs is not None:
            gs.close()
        return w.detach().clone()

    ref = run(False, False)
    assert torch.equal(run(True, False), ref)           # device-resident hyper-parameters, eager
    got = run(True, True)                                # ... and replayed from a graph
    assert torch.equal(got, ref), (got - ref).abs().max().item()
