#!/usr/bin/env python
"""bench.py — training images/sec of the MI355X-native segmentation hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2] [--no-cpu] [--no-roofline] [--no-alt]

`python bench.py --gpus N` drives all N GPUs by itself (it re-executes under torch.distributed.run, one rank per GPU over
RCCL); the driver's own form `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` is what
that re-execution issues and works directly as well.

A "step" is one pass of the hot path over one synthetic batch resident in HBM: zero_grad -> PSPNet-R50 forward (train mode: BN
batch statistics, dropout active, aux head) -> CrossEntropy(main) + 0.4*CE(aux) (global-batch mean over all ranks' valid
pixels) -> backward -> gradient all-reduce (N>1, RCCL, bucketed, overlapped with backward) -> fused SGD(momentum, weight
decay), per bucket right after its all-reduce — the reference's inner loop trainer.py:55-71.  Workload at every N:
BASELINE.json configs[1] (PSPNet-ResNet50, 8 x 3x512x512, 21 classes) PER GPU, i.e. weak scaling; fp32 end to end (the
reference's dtype; convolutions on v_mfma_f32_32x32x2_f32).

Rank 0 prints ONE JSON line.  Beyond the driver contract it carries
  roofline     : the dominant kernel = the conv implicit-GEMM variant (MFMA-bound) with the largest total time, measured with
                 HIP events per launch in an extra instrumented step: `achieved` = its algorithmic FLOPs per launch / its mean
                 launch duration `avg_us` (compare with profiles/*kernel_stats*), `traffic` = its HBM bytes per launch from the
                 committed rocprofv3 PMC passes (null when no profile knows the kernel by its current name); `all_conv` holds
                 the same quantities over ALL conv launches of the step.
  cpu_baseline : the oracle (torch-CPU restatement of the reference, oracle/pspnet_ref.py) timed on this host's cores on a
                 bounded sample: the FULL batch of the same workload, one untimed warm-up step, a one-step sweep over
                 {16, 32, 64, all} threads, then >= 3 timed steps at the best count within a 240 s cap (N = 1 only).
  alt_direct   : the same K steps in fp32 with Winograd OFF (every layer on the direct implicit-GEMM kernels).  The headline
                 runs the eligible 3x3 stride-1 layers on the Winograd F(2x2,3x3) kernels — the default since the whole GPU
                 suite runs under both algorithms at the same tolerances (tests/conftest.py).  N = 1 only.
`roofline.achieved/frac` count EXECUTED FLOPs (a Winograd contraction executes 16/36 of the direct convolution's), `effective`
the direct-convolution FLOPs the launches stand for; `executed_step_frac` = executed FLOPs of one step / step time / peak.  The
reference formulation's FLOPs per step time are reported as a RATE (`reference_formulation.tflops_equivalent`), never
as a fraction of a peak: the step executes fewer FLOPs than the reference formulation has (Winograd, factored PSP bottleneck).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pytorch-segmentation_amd"))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: RCCL's cross-process buffer registration fails without this (set before HIP starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz

# name -> (arch, kwargs, num_classes, per-GPU batch, H, W, train FLOPs/image (conv only), loss, ignore_index)
# cfg2 is the bench line (BASELINE.json configs[1]); the others are the remaining BASELINE configs, runnable with --config
# for profiling (they are parity-test cases, not bench lines).
CONFIGS = {
    "cfg1": ("UNet", dict(), 2, 2, 256, 256, 295_937_507_328, "CrossEntropyLoss2d", 255),
    "cfg2": ("PSPNet", dict(backbone="resnet50", pretrained=False), 21, 8, 512, 512, 1_221_159_026_688, "CrossEntropyLoss2d", 255),
    "cfg3": ("DeepLab", dict(backbone="resnet101", pretrained=False, output_stride=16), 19, 16, 513, 513, 556_871_061_120, "CrossEntropyLoss2d", 255),
    "cfg4": ("PSPNet", dict(backbone="resnet50", pretrained=False), 19, 4, 769, 769, 2_802_443_088_000, "CrossEntropyLoss2d", 255),
    "cfg5": ("DeepLab", dict(backbone="xception", pretrained=False, output_stride=16), 150, 8, 512, 512, 499_565_445_120, "LovaszSoftmax", -1),
}


def build_model(name, device):
    import models
    arch, kw, classes = CONFIGS[name][:3]
    torch.manual_seed(0)
    return getattr(models, arch)(classes, **kw).to(device).train()


def synth_batch(name, device, rank):
    _, _, classes, n, h, w, _, _, ign = CONFIGS[name]
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(n, 3, h, w, generator=g)
    t = torch.randint(0, classes, (n, h, w), generator=g)
    t[:, : h // 20, :] = ign
    return x.to(device), t.to(device)


def cpu_baseline(name, seconds_cap=240.0):
    """Oracle leg: same model family / loss / optimizer on torch-CPU (the reference's own path: Python on ATen CPU kernels),
    at the bench line's FULL batch.  Protocol: one untimed warm-up step on all threads (lazy oneDNN primitives,
    allocator growth), a one-step sweep over {16, 32, 64, all} torch threads, then timed steps at the best thread count until
    three post-warm-up samples exist at it (the sweep step counts as one) or `seconds_cap` is spent.  `value` is the best
    (minimum-time) step, the median is reported beside it.  kind = "port": /root/reference does not exist on the GPU box and
    nothing here may read it at run time, so the leg is the bit-exact restatement (oracle/pspnet_ref.py, unet_ref.py,
    deeplab_ref.py + losses_ref.py: every BASELINE config has one), not the trainer."""
    from oracle import deeplab_ref, losses_ref, pspnet_ref, unet_ref
    import models
    arch, kw, classes, n, h, w, _, loss_name, ign = CONFIGS[name]
    nb = n
    torch.manual_seed(0)
    sd = {k: v.detach().clone().contiguous() for k, v in getattr(models, arch)(classes, **kw).state_dict().items()}
    ref = pspnet_ref.clone_state(sd)
    params = [v for v in ref.values() if v.requires_grad]
    opt = torch.optim.SGD(params, lr=0.01, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(nb, 3, h, w, generator=g)
    t = torch.randint(0, classes, (nb, h, w), generator=g)
    t[:, : h // 20, :] = ign
    loss_fn = {"CrossEntropyLoss2d": losses_ref.cross_entropy, "LovaszSoftmax": losses_ref.lovasz_softmax}[loss_name]

    def forward_loss():
        if arch == "PSPNet":
            out, aux = pspnet_ref.pspnet_forward(ref, x, training=True, backbone=kw["backbone"])
            return loss_fn(out, t, ign) + 0.4 * loss_fn(aux, t, ign)
        if arch == "UNet":
            return loss_fn(unet_ref.unet_forward(ref, x, training=True), t, ign)
        return loss_fn(deeplab_ref.deeplab_forward(ref, x, kw["backbone"], kw["output_stride"], training=True), t, ign)

    all_threads = torch.get_num_threads()
    t_begin = time.perf_counter()

    def one_step():
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = forward_loss()
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    def spent():
        return time.perf_counter() - t_begin

    warm = one_step()
    samples = {}                                   # threads -> [seconds per step]
    # ascending: the small counts are the fast ones on this workload (cfg2: 16 / 32 / 64 / 128 threads = 0.71 / 0.57 / 0.38 / 0.22 img/s),
    # so a tight --cpu-cap cuts off the slow end of the sweep, not the best candidate
    cands = sorted({c for c in (16, 32, 64, all_threads) if c <= all_threads})
    try:
        for c in cands:
            if samples and spent() + min(min(v) for v in samples.values()) * 1.5 > seconds_cap:
                break
            torch.set_num_threads(c)
            samples[c] = [one_step()]
        if not samples:                            # the warm-up alone exhausted the cap
            samples[all_threads] = [warm]
        best_c = min(samples, key=lambda c: min(samples[c]))
        torch.set_num_threads(best_c)
        while len(samples[best_c]) < 3 and spent() + min(samples[best_c]) * 1.2 < seconds_cap:
            samples[best_c].append(one_step())
    finally:
        torch.set_num_threads(all_threads)
    ts = sorted(samples[best_c])
    best, med = ts[0], ts[len(ts) // 2]
    return {"value": round(nb / best, 4), "median": round(nb / med, 4), "unit": "img/s", "cores": best_c, "kind": "port",
            "timed_steps": len(ts), "step_seconds": [round(v, 2) for v in samples[best_c]],
            "thread_sweep": {str(c): round(nb / min(v), 4) for c, v in sorted(samples.items())},
            "sample": "%d timed train step(s) of the FULL batch %d x 3x%dx%d (same model/loss/SGD as the GPU leg) on %d of %d torch "
                      "threads (best of a one-step sweep over %s) after 1 untimed warm-up step (%.1f s); value = best step, median beside it; "
                      "%.0f s of CPU time in total (cap %.0f s)"
                      % (len(ts), nb, h, w, best_c, all_threads, "/".join(str(c) for c in sorted(samples)), warm, spent(), seconds_cap)}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(argv, n, port):
    """The per-GPU launcher `python bench.py --gpus N ...` re-executes itself under (the driver's own form for N > 1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-cap", type=float, default=240.0, help="seconds of host time the cpu_baseline leg may spend (after its warm-up step)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the instrumented roofline step")
    ap.add_argument("--no-alt", action="store_true", help="skip the `alt_direct` leg (same steps with Winograd off)")
    ap.add_argument("--sync-bn", action="store_true", help="SynchronizedBatchNorm across ranks (cfg4 regime)")
    ap.add_argument("--force-ddp", action="store_true", help="run the N>1 code path (process group, bucketed all-reduce) in a 1-rank group")
    ap.add_argument("--graph", action="store_true", help="capture the training step into a hipGraph (segmi.graph.GraphedStep) and time replays: "
                                                         "one host call per step instead of ~800 launches")
    ap.add_argument("--compute-priority", type=int, default=None,
                    help="A/B: run the steps on a HIP stream of this priority (-1 = high) instead of torch's default stream; together with "
                         "SEGMI_WGRAD_STREAM_PRIORITY it decides who wins the CUs when a data-gradient and a filter-gradient kernel compete")
    ap.add_argument("--lovasz-boost", type=float, default=0.0,
                    help="cfg5 A/B only: add this to the target logit on 80 %% of the pixels before the loss (trained-like, confident logits "
                         "instead of random-init ones: more elements survive the Lovasz tail pruning); costs one extra elementwise add per step")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: one command drives all GPUs, like the reference (base/base_trainer.py:33-38, train.py:46-53)
        # — by re-executing itself as one rank per GPU under torch.distributed.run
        cmd = launch_command(sys.argv[1:], args.gpus, free_port())
        print("[bench] " + " ".join(cmd), file=sys.stderr, flush=True)
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    oversubscribed = world > ndev      # more ranks than GPUs (launcher smoke on a 1-GPU box): ranks share devices, RCCL cannot
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    ddp = world > 1 or args.force_ddp
    backend = None
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("SEGMI_BENCH_BACKEND") or ("gloo" if oversubscribed else "nccl")     # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from segmi.distributed import DistributedModel
    from segmi.profile import KernelTimer
    import utils.losses as losses_mod

    from segmi import ops as segmi_ops

    arch, kw, classes, nb, h, w, flops_img, loss_name, ign = CONFIGS[args.config]
    model = build_model(args.config, device)
    if args.sync_bn and ddp:
        from utils.sync_batchnorm import convert_model
        model = convert_model(model)
    # --graph needs gradients at stable addresses: the reducer keeps them as views of flat buckets, also without a process group
    dm = DistributedModel(model, always_reduce=args.force_ddp) if (ddp or args.graph) else None
    from segmi.optim import SGD          # torch.optim.SGD semantics, one fused launch
    if os.environ.get("SEGMI_BENCH_TORCH_SGD") == "1":
        SGD = torch.optim.SGD            # A/B hook
    opt = SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    bucket_step = bool(dm is not None and dm.reducer.collective and dm.attach_optimizer(opt))
    crit = getattr(losses_mod, loss_name)(ignore_index=ign)
    x, t = synth_batch(args.config, device, rank)
    psp = arch[:3] == "PSP"          # the reference keys the (out, aux) convention on the arch name (trainer.py:57-62)
    boost = None
    if args.lovasz_boost and not psp:
        gb = torch.Generator().manual_seed(99 + rank)
        hit = (torch.rand(nb, h, w, generator=gb) < 0.8).to(device) & (t != ign)
        boost = segmi_ops.to_nhwc(torch.zeros(nb, classes, h, w, device=device).scatter_(
            1, t.clamp(0, classes - 1).unsqueeze(1), hit.float().unsqueeze(1) * args.lovasz_boost))

    def step():
        if dm is not None:
            dm.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        if psp:
            out, aux = model(x)
            loss = crit(out, t) + 0.4 * crit(aux, t)
        else:
            out = model(x)
            loss = crit(out if boost is None else out + boost, t)
        loss.backward()
        if bucket_step:
            dm.finish_gradients(opt)         # per bucket: wait for its all-reduce, then its fused SGD launch
        else:
            if dm is not None:
                dm.finish_gradients()
            opt.step()
        return loss

    def fence():
        if ddp:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(run):
        """W untimed warm-up steps, then exactly K steps between two barrier + synchronize fences; max over ranks."""
        for _ in range(args.warmup):
            loss = run()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = run()
        fence()
        dt = time.perf_counter() - t0
        if ddp:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, float(loss.item())

    def roofline_of(value):
        """Instrumented extra step (HIP events around every conv launch on the launch stream) -> the `roofline` object."""
        peak = PEAK_FP32_MFMA_TFLOPS
        side = segmi_ops.get_wgrad_stream()["on"]
        segmi_ops.set_wgrad_stream(False)      # per-kernel durations are taken with every launch in order on ONE stream
        try:
            with KernelTimer(membound=True) as kt:      # every rank runs the instrumented step (it contains collectives)
                step()
        finally:
            segmi_ops.set_wgrad_stream(side)
        every = kt.summary()
        summ = {k: r for k, r in every.items() if not k.startswith("segmi_")}          # the convolution launches (MFMA-bound)
        mem = {k: r for k, r in every.items() if k.startswith("segmi_")}               # the HBM-bound C-ABI calls (BN, depthwise, losses, resize, pooling)
        mem_ms, mem_b = sum(r["total_ms"] for r in mem.values()), sum(r["bytes"] for r in mem.values())
        hbm = {"bound": "hbm", "peak": 8.0, "unit": "TB/s", "ms_per_step": round(mem_ms, 3), "algorithmic_gb_per_step": round(mem_b / 1e9, 3),
               "achieved": round(mem_b / (mem_ms * 1e-3) / 1e12, 3) if mem_ms else None,
               "frac": round(mem_b / (mem_ms * 1e-3) / 8.0e12, 4) if mem_ms else None,
               "scope": "every HBM-bound C-ABI call of the instrumented step (HIP events per call, event pairs add ~2 us to calls shorter than "
                        "~10 us); bytes = algorithmic (each operand tensor once; Lovasz forward: 8 B per (class, pixel) = the logits read "
                        "twice, threshold pass + selection pass, the survivors' sort traffic not counted; backward: logits read + dlogits written)",
               "top": [{"call": k, "launches": r["launches"], "ms_per_step": round(r["total_ms"], 3), "gb": round(r["bytes"] / 1e9, 3),
                        "tbs": round(r["bytes"] / (r["total_ms"] * 1e-3) / 1e12, 2), "frac": round(r["bytes"] / (r["total_ms"] * 1e-3) / 8.0e12, 3)}
                       for k, r in sorted(mem.items(), key=lambda kv: -kv[1]["total_ms"])[:8]]}
        tot_ms = sum(r["total_ms"] for r in summ.values())
        tot_fl = sum(r["flops"] for r in summ.values())              # EXECUTED (Winograd: transform-domain) FLOPs
        tot_eff = sum(r["eff_flops"] for r in summ.values())         # algorithmic FLOPs of the direct convolutions they stand for
        # dominant kernel = the matrix kernel with the largest total time (the HBM-bound "winograd transforms" rows carry no FLOPs)
        top_name, top = max(((k, r) for k, r in summ.items() if r["flops"] > 0), key=lambda kv: kv[1]["total_ms"])
        all_ach = tot_fl / (tot_ms * 1e-3) / 1e12
        ach = top["flops"] / (top["total_ms"] * 1e-3) / 1e12
        # HBM traffic cannot be measured live (PMC passes need rocprofv3): the committed summary of the offline FETCH_SIZE /
        # WRITE_SIZE passes over this same command is reported, per launch of the dominant kernel like `achieved` (cfg2 only).
        # The file is looked up per arithmetic and must know the dominant kernel BY ITS CURRENT NAME: a profile taken before a
        # kernel was renamed / re-tiled yields null and a note, never a stale number.
        traffic = step_traffic = None
        tnote = "no PMC profile for this config/arithmetic under profiles/"
        tag = "f32" + ("" if wino_default else "_direct")
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_cfg2_conv_traffic_%s.json" % tag))
        if args.config == "cfg2" and cands:
            tnote = "no profile under profiles/ describes kernel %r under the current conv algorithm (stale profiles: re-run tools/gpu_round.sh pmc)" % top_name
            for cand in reversed(cands):                              # newest name first; only a profile that knows the kernel counts
                tj = json.load(open(os.path.join(ROOT, "profiles", cand)))
                rec = tj.get("per_kernel", {}).get(top_name.split(" splitk=")[0])
                if rec is not None and tj.get("conv_winograd", False) == bool(wino_default):
                    traffic, step_traffic = rec["traffic_bytes_per_launch"], tj["traffic_bytes_per_step"]
                    tnote = "HBM+MALL bytes per launch from the rocprofv3 PMC passes (profiles/%s)" % cand
                    break
        step_s = nb * world / value                                   # seconds per step of the timed loop
        return {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic,
                "effective": round(top["eff_flops"] / (top["total_ms"] * 1e-3) / 1e12, 2),
                "peak_note": "fp32 MFMA (v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz; MI355X_MICROARCH.md)",
                "kernel": top_name, "launches": top["launches"], "avg_us": round(top["avg_us"], 1),
                "flops_per_launch": top["flops"] // top["launches"], "algorithmic_bytes_per_launch": top["bytes"] // top["launches"],
                "scope": "dominant kernel = the matrix kernel with the largest total time in one step; HIP events per launch on the launch "
                         "stream (for a Winograd pass: the event pair the library records around its ONE batched contraction launch); "
                         "achieved/frac = EXECUTED FLOPs (the 16 transform-domain contractions of a Winograd launch, 2*16*T*C*K; a direct launch that "
                         "skips taps / pixel chunks whose operand is all padding counts only the share it issues, segmi.ops._conv_issued), "
                         "effective = the direct-convolution FLOPs those launches stand for; traffic: " + tnote,
                "all_conv": {"achieved": round(all_ach, 2), "frac": round(all_ach / peak, 4),
                             "effective": round(tot_eff / (tot_ms * 1e-3) / 1e12, 2),
                             "launches": sum(r["launches"] for r in summ.values()), "ms_per_step": round(tot_ms, 2),
                             "flops_per_step": tot_fl, "effective_flops_per_step": tot_eff,
                             "algorithmic_bytes_per_step": sum(r["bytes"] for r in summ.values()),
                             "traffic_bytes_per_step": step_traffic},
                # executed_step_frac: FLOPs the matrix kernels actually executed in one step / step time / peak (a utilisation)
                "hbm_bound_calls": hbm,
                "executed_step_frac": round(tot_fl / step_s / 1e12 / peak, 4),
                # the REFERENCE formulation's conv FLOPs (3 x forward MACs x 2 of models/*.py as written) per step time:
                # an equivalent-throughput RATE, not a roofline fraction — this step executes fewer FLOPs
                "reference_formulation": {"flops_per_step": flops_img * nb, "tflops_equivalent": round(value / world * flops_img / 1e12, 1),
                                          "executed_over_reference_flops": round(tot_fl / (flops_img * nb), 4),
                                          "note": "not a fraction of any peak: Winograd F(2x2,3x3) and the factored PSP bottleneck execute "
                                                  "fewer FLOPs than the reference formulation has; utilisation = executed_step_frac / frac"},
                "variants": {k: {"launches": r["launches"], "avg_us": round(r["avg_us"], 1),
                                 "tflops": round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 1),
                                 "effective_tflops": round(r["eff_flops"] / (r["total_ms"] * 1e-3) / 1e12, 1)} for k, r in sorted(summ.items())}}

    wino_default = segmi_ops.get_conv_winograd()["on"]
    if args.compute_priority is not None:
        cstream = torch.cuda.Stream(device=device, priority=args.compute_priority)
        cstream.wait_stream(torch.cuda.current_stream(device))
        torch.cuda.set_stream(cstream)
    run = step
    if args.graph:
        from segmi.graph import GraphedStep
        run = GraphedStep(step, warmup=3)       # eager warm-up steps + capture; replays below are the timed steps
    dt, final_loss = timed(run)
    if os.environ.get("SEGMI_BENCH_MEMSTATS") == "1":          # A/B diagnostics: did the caching allocator reach a steady state?
        ms_ = torch.cuda.memory_stats(device)
        print("[bench] memstats: reserved %.2f GB, segments %d, device mallocs %d, retries %d" % (
            ms_.get("reserved_bytes.all.current", 0) / 1e9, ms_.get("segment.all.current", 0), ms_.get("num_device_alloc", 0),
            ms_.get("num_alloc_retries", 0)), file=sys.stderr, flush=True)
    ms = 1e3 * dt / args.steps
    value = world * nb * args.steps / dt
    roof = None if args.no_roofline else roofline_of(value)

    # SyncBN: small latency-bound collectives per step (one all-gather forward + one all-reduce backward per BN layer; exact
    # SyncBN needs both before the layer can proceed, so only parallel branches could share one)
    syncbn_per_step = None
    if args.sync_bn and ddp:
        ctxs = [m.sync for m in model.modules() if getattr(m, "sync", None) is not None]
        before = sum(c.collectives for c in ctxs)
        step()
        syncbn_per_step = sum(c.collectives for c in ctxs) - before

    run_alt = not args.no_alt and not args.graph and world == 1   # N > 1: the headline only
    # `alt_direct`: the same K steps of the same job (it simply keeps training), fp32 arithmetic, with Winograd OFF — every layer on
    # the direct implicit-GEMM kernels (the round-2 headline path; csrc/conv_winograd.hip is an ALGORITHM change in the same
    # arithmetic).  When the
    # process runs with SEGMI_CONV_WINOGRAD=0 the roles swap (`alt_winograd`).
    alt_algo = None
    if run_alt:
        wst = segmi_ops.get_conv_winograd()
        segmi_ops.set_conv_winograd(not wino_default, wgrad=not wino_default)
        try:
            wdt, wloss = timed(step)
            wval = world * nb * args.steps / wdt
            alt_algo = {"conv_algorithm": ("direct implicit GEMM for every layer" if wino_default else
                                           "winograd_f2x2_3x3 for the 3x3 stride-1 layers with >= %d channels (all three passes), direct "
                                           "implicit GEMM elsewhere" % wst["min_channels"]),
                        "conv_math": "f32", "value": round(wval, 2), "unit": "img/s", "ms_per_step": round(1e3 * wdt / args.steps, 2),
                        "steps": args.steps, "warmup": args.warmup, "final_loss": round(wloss, 5),
                        "parity": "the model-level GPU tests, the generic convolution test and the four BASELINE-shape audits run once per "
                                  "algorithm at the same tolerances (tests/conftest.py: conv_algorithm)"}
        finally:
            segmi_ops.set_conv_winograd(wst["on"], wgrad=wst["wgrad"])
    # ranks that actually took part in an RCCL collective on this communicator (an all-reduce of ones over the device tensors),
    # not dist.get_world_size(): a rank that fell back to another transport or never joined the communicator shows up here
    rccl_ranks = 0
    if ddp:
        if backend == "nccl":
            ones = torch.ones(1, device=device)
            dist.all_reduce(ones)
            rccl_ranks = int(ones.item())
            pgb = dist.distributed_c10d._get_default_group()._get_backend(device)
            if hasattr(pgb, "_is_initialized") and not pgb._is_initialized():
                rccl_ranks = 0
        dist.barrier()

    # Lovasz tail pruning: survivors of the last step's loss call out of the n_present * n_valid elements of the full sort
    lovasz_stats = None
    if loss_name == "LovaszSoftmax" and segmi_ops.lovasz_last_stats() is not None:
        kept, full = segmi_ops.lovasz_last_stats()
        lovasz_stats = {"survivors": kept, "full_sort_keys": full, "kept_frac": round(kept / max(full, 1), 5),
                        "logits": "random-init model output" if boost is None else "model output + %.1f on the target class of 80 %% of the pixels" % args.lovasz_boost,
                        "prune": os.environ.get("SEGMI_LOVASZ_PRUNE", "1") != "0"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(args.config, seconds_cap=args.cpu_cap)

    if rank == 0:
        line = {
            "metric": "training images/sec @512x512 (PSPNet-R50)" if args.config == "cfg2" else "training images/sec (%s)" % args.config,
            "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: %s%s %dx3x%dx%d per GPU, %d classes, %s%s, SGD(momentum 0.9, wd 1e-4), "
                                   "BN batch stats%s, dropout on" % (args.config, arch, "-" + kw["backbone"] if "backbone" in kw else "", nb, h, w,
                                                                      classes, loss_name, " + 0.4*aux" if psp else "",
                                                                      " (SyncBN)" if args.sync_bn and ddp else ""),
                       "global_batch": nb * world, "parallelism": "dp%d" % world,
                       "collective_backend": ({"nccl": "rccl"}.get(backend, backend) if ddp else None),
                       "rccl_ranks": rccl_ranks, "final_loss": round(final_loss, 5),
                       "conv_math": "f32 (v_mfma_f32_32x32x2_f32)", "conv_winograd": bool(wino_default),
                       "conv_algorithm": ("winograd_f2x2_3x3 (fwd, dgrad, wgrad) for the 3x3 stride-1 layers with >= %d channels, direct implicit "
                                          "GEMM elsewhere" % segmi_ops.get_conv_winograd()["min_channels"]) if wino_default else "direct implicit GEMM",
                       "hip_graph": bool(args.graph), "wgrad_side_stream": bool(segmi_ops.get_wgrad_stream()["on"]),
                       # how the fused SGD sent its chunk table up during the timed steps (segmi/optim.py _upload; None: built once)
                       "sgd_table_upload": os.environ.get("SEGMI_SGD_TABLE_UPLOAD") or getattr(opt, "_auto", None),
                       "bn_stats_from_conv_epilogue": bool(segmi_ops.get_conv_bn_stats()["on"]),
                       "grad_buckets_mb": ([round(b["buf"].numel() * 4 / 2 ** 20, 1) for b in dm.reducer.buckets] if dm is not None else None),
                       "syncbn_collectives_per_step": syncbn_per_step,
                       "lovasz": lovasz_stats},
            "roofline": roof, "cpu_baseline": cpu, ("alt_direct" if wino_default else "alt_winograd"): alt_algo,
        }
    # The JSON line is the LAST thing this process writes to stdout: RCCL prints a version banner through C stdio at communicator
    # creation, which sits in the libc buffer (stdout is a pipe under the driver) until it is flushed — left alone it comes out at
    # exit, AFTER the line.  Flushed here, then the line, then the (silent) teardown — a teardown that stalled on some rank must not
    # cost the result.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stderr.flush()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
