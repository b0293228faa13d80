"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/segmi.h
declares; host-side planning/query functions behave; the drop-in modules keep the reference's
plugin surface (class names, constructor kwargs, state_dict keys) and refuse to run on CPU."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "segmi.h")
GOLD = os.path.join(ROOT, "tests", "golden")


def _header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(segmi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    from segmi import LIB_PATH
    from segmi._lib import SIGNATURES
    names = _header_functions()
    assert len(names) >= 30
    out = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (segmi_[a-z0-9_]+)", out))
    assert set(names) <= exported, sorted(set(names) - exported)
    assert set(names) == set(SIGNATURES), (sorted(set(names) ^ set(SIGNATURES)))


def test_library_is_gfx950_only_and_has_no_rocm_runpath():
    from segmi import LIB_PATH
    dyn = subprocess.run(["readelf", "-d", LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "libamdhip64.so.7" in dyn
    assert "RUNPATH" not in dyn and "RPATH" not in dyn   # must bind to the runtime torch already loaded
    blob = open(LIB_PATH, "rb").read()
    # every device code object bundled in the library targets gfx950 (rocPRIM's host-side config tables mention other
    # architecture NAMES as plain strings, which is not device code)
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets
    assert b"sm_80" not in blob and b"nvptx" not in blob


def test_status_strings_and_queries():
    from segmi import lib
    from segmi._lib import ConvDesc
    assert lib.segmi_abi_version() == 10
    assert lib.segmi_strerror(0) == b"ok"
    assert b"workspace" in lib.segmi_strerror(-3)
    # bad descriptor -> argument error before any launch (no GPU needed)
    d = ConvDesc(1, 8, 8, 4, 8, 3, 3, 7, 8, 1, 1, 1, 4, 8)  # P inconsistent with H/pad/dil
    assert lib.segmi_conv2d_fwd(d, 16, 16, None, 16, 0, None, 0, None) == -1
    d = ConvDesc(1, 8, 8, 6, 8, 3, 3, 8, 8, 1, 1, 1, 6, 8)  # C % 4 != 0
    assert lib.segmi_conv2d_fwd(d, 16, 16, None, 16, 0, None, 0, None) == -2
    # split-K planning: a 64->64 3x3 on a 256x256 map at batch 8 must split its 524288-pixel reduction
    d = ConvDesc(8, 256, 256, 64, 64, 3, 3, 256, 256, 1, 1, 1, 64, 64)
    ws = lib.segmi_conv2d_wgrad_workspace(d)
    assert ws % (64 * 9 * 64 * 4) == 0 and ws // (64 * 9 * 64 * 4) >= 32
    # the 4096->512 3x3 PSP bottleneck has 1152 tiles = 2.25 rounds of the 512 resident workgroups:
    # (110 TF/s measured); the planner splits until the grid is >= 8 rounds: x4
    d = ConvDesc(8, 64, 64, 4096, 512, 3, 3, 64, 64, 1, 1, 1, 4096, 512)
    assert lib.segmi_conv2d_wgrad_workspace(d) == 4 * 512 * 9 * 4096 * 4
    # a 1x1 2048->512 on an 8-pixel map cannot be split at all
    d = ConvDesc(8, 1, 1, 2048, 512, 1, 1, 1, 1, 1, 0, 1, 2048, 512)
    assert lib.segmi_conv2d_wgrad_workspace(d) == 0
    assert lib.segmi_bn_stats_workspace(8 * 64 * 64, 2048) >= 3 * 2048 * 4
    assert lib.segmi_ce_workspace(1 << 21) > 0


def test_variant_names_and_the_retired_second_arithmetic():
    """The variant name reported for profiling is the kernel instantiation exactly as a rocprofv3 kernel trace prints it.  ABI v9:
    the second convolution arithmetic of v3-v8 (bf16x3) is gone — no entry point, no environment switch, no
    trailing MATH template argument."""
    from segmi import lib, ops
    from segmi._lib import ConvDesc
    d = ConvDesc(8, 64, 64, 512, 512, 3, 3, 64, 64, 1, 2, 2, 512, 512)
    assert lib.segmi_abi_version() >= 9
    assert ops.conv_variant(d, 0) == "conv_dma_kernel<128, 128, 2, 2, 0, true, false>"
    assert ops.conv_variant(d, 1) == "conv_dma_kernel<128, 128, 2, 2, 1, true, false>"
    assert ops.conv_variant(d, 2).startswith("conv_wgrad_dma_kernel<128, 128, true, false>")
    for gone in ("segmi_conv_set_math", "segmi_conv_get_math", "segmi_filter_presplit", "segmi_conv2d_fwd_presplit"):
        assert not hasattr(lib, gone), gone
    assert not hasattr(ops, "set_conv_math")
    assert lib.segmi_conv2d_wgrad_workspace(d) % (512 * 9 * 512 * 4) == 0


def test_winograd_planning_and_dispatch_rules():
    """Host side of the Winograd F(2x2,3x3) path (include/segmi.h segmi_conv2d_winograd_*; no GPU): which problems it accepts,
    its workspace = transformed filter + 16 input planes + 16 product planes over T = N * dil^2 * ceil(ceil(H/dil)/2) *
    ceil(ceil(W/dil)/2) tiles, the kernel name it reports, and the opt-in dispatch rule of segmi.ops (off by default; >= 256
    channels and sub-grids of >= 8 pixels when on)."""
    import ctypes
    from segmi import lib, ops
    from segmi._lib import ConvDesc

    def desc(N, H, C, K, R, stride, pad, dil):
        P = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
        return ConvDesc(N, H, H, C, K, R, R, P, P, stride, pad, dil, C, (K + 3) & ~3)

    d = desc(8, 64, 512, 512, 3, 1, 4, 4)
    assert lib.segmi_conv2d_winograd_ok(d, 0) == 1 and lib.segmi_conv2d_winograd_ok(d, 1) == 1
    T = 8 * 16 * 8 * 8
    al = lambda b: (b + 255) & ~255
    assert lib.segmi_conv2d_winograd_workspace(d, 0) == al(16 * 512 * 512 * 4) + al(16 * T * 512 * 4) + al(16 * T * 512 * 4)
    buf = ctypes.create_string_buffer(128)
    assert lib.segmi_conv2d_winograd_variant(d, 0, buf, 128) == 0 and buf.value.decode().startswith("winograd_f2x2_3x3 fwd: 16 x conv_dma_kernel<")
    # odd maps: 97x97 with dilation 2 -> sub-grids of 49 / 48 rows, 25 tile rows each
    d97 = desc(4, 97, 256, 256, 3, 1, 2, 2)
    # the transformed input holds whole 32-row chunks of tiles (zero rows past T) so that the filter gradient can contract a kept V
    T97a, T97p = 4 * 4 * 25 * 25, (4 * 4 * 25 * 25 + 31) & ~31
    assert lib.segmi_conv2d_winograd_workspace(d97, 0) == al(16 * 256 * 256 * 4) + al(16 * T97p * 256 * 4) + al(16 * T97a * 256 * 4)
    assert lib.segmi_conv2d_winograd_v_bytes(d97) == 16 * T97p * 256 * 4 and lib.segmi_conv2d_winograd_v_bytes(d) == 16 * T * 512 * 4
    # K = 21 classes: the product planes are padded to 24 columns; the data gradient contracts over the padded K
    d21 = desc(2, 33, 64, 21, 3, 1, 1, 1)
    T21 = 2 * 17 * 17
    T21p = (T21 + 31) & ~31
    assert lib.segmi_conv2d_winograd_workspace(d21, 0) == al(16 * 21 * 64 * 4) + al(16 * T21p * 64 * 4) + al(16 * T21 * 24 * 4)
    assert lib.segmi_conv2d_winograd_workspace(d21, 1) == al(16 * 64 * 24 * 4) + al(16 * T21p * 24 * 4) + al(16 * T21 * 64 * 4)
    # not Winograd problems: 1x1, stride 2, padding != dilation, channels not padded to 4, unknown pass
    for bad in (desc(8, 64, 512, 512, 1, 1, 0, 1), desc(8, 64, 512, 512, 3, 2, 1, 1), desc(8, 64, 512, 512, 3, 1, 0, 1),
                desc(8, 64, 510, 512, 3, 1, 1, 1)):
        assert lib.segmi_conv2d_winograd_ok(bad, 0) == 0 and lib.segmi_conv2d_winograd_workspace(bad, 0) == 0
    assert lib.segmi_conv2d_winograd_ok(d, 2) == 0
    # filter gradient in the Winograd domain: both transformed operands (tile rows padded to whole 32-row chunks) and the
    # partial sums [nsplit][16][K][C] of the ONE batched contraction launch (nsplit = the "splitk=" of the variant name)
    import ctypes

    def wg_splits(dd):
        buf = ctypes.create_string_buffer(128)
        assert lib.segmi_conv2d_winograd_wgrad_variant(dd, buf, 128) == 0
        name = buf.value.decode()
        assert name.startswith("winograd_f2x2_3x3 wgrad: 16 x conv_wgrad_dma_kernel<"), name
        return int(name.rsplit("splitk=", 1)[1])
    assert lib.segmi_conv2d_winograd_wgrad_ok(d) == 1 and lib.segmi_conv2d_winograd_wgrad_ok(desc(8, 64, 512, 512, 3, 2, 1, 1)) == 0
    ns = wg_splits(d)
    assert 1 <= ns <= 64
    assert lib.segmi_conv2d_winograd_wgrad_workspace(d) == 2 * al(16 * T * 512 * 4) + al(ns * 16 * 512 * 512 * 4)
    T97 = (4 * 4 * 25 * 25 + 31) & ~31
    ns97 = wg_splits(d97)
    assert lib.segmi_conv2d_winograd_wgrad_workspace(d97) == 2 * al(16 * T97 * 256 * 4) + al(ns97 * 16 * 256 * 256 * 4)
    # dispatch rule
    prev = ops.get_conv_winograd()
    try:
        ops.set_conv_winograd(False)
        assert not ops._winograd(d, 0)
        ops.set_conv_winograd(True, min_channels=256, min_subgrid=8)
        assert ops._winograd(d, 0) and ops._winograd(d, 1) and ops._winograd(d97, 0)
        assert not ops._winograd(desc(8, 128, 128, 128, 3, 1, 1, 1), 0)            # too few channels
        assert not ops._winograd(desc(16, 33, 2048, 256, 3, 1, 12, 12), 0)          # ASPP: sub-grids of 3 pixels
        assert not ops._winograd(desc(8, 64, 512, 512, 3, 2, 1, 1), 0)
    finally:
        ops.set_conv_winograd(prev["on"], prev["min_channels"], prev["min_subgrid"], prev["wgrad"])


def test_conv_kernels_are_compiled_without_scratch(tmp_path):
    """Every LDS-DMA convolution instantiation (both arithmetics) is present in the gfx950 code object, none spills
    (a spilling matrix loop would silently run at a fraction of the modelled rate) and all leave room for two workgroups
    per CU (<= 256 VGPRs incl. accumulators)."""
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(ROOT, "pytorch-segmentation_amd", "build", "conv_igemm.o")
    if not (os.path.exists(os.path.join(llvm, "clang-offload-bundler")) and os.path.exists(obj)):
        pytest.skip("llvm tools / object file not present")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "conv.co")
    subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, str(tmp_path / "unused.o")], check=True)
    subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    kern = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        kern[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count")}
        kern[name]["agpr_count"] = int(blk.split()[0])
    dma = {n: v for n, v in kern.items() if "dma_kernel" in n}
    assert len(dma) == 46, sorted(dma)          # 5 tile shapes x {fprop, dgrad} x {pointwise, fast, generic} + 4 tiles x {ROWQ, generic} x {pointwise, taps} wgrad
    for n, v in dma.items():
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0, (n, v)
        # .vgpr_count is the unified total (arch VGPRs up to the accumulator offset + AGPRs); 2 x 256 = one SIMD's file
        assert v["agpr_count"] <= v["vgpr_count"] <= 256, (n, v)


def test_graph_module_refuses_cpu_and_validates_the_dropout_epoch():
    """segmi.graph.GraphedStep is hipGraph capture: no CPU path; the dropout epoch must be a device int64 scalar."""
    from segmi import SegmiError, graph, ops
    with pytest.raises(SegmiError):
        graph.GraphedStep(lambda: None)
    with pytest.raises(SegmiError):
        ops.set_dropout_epoch(torch.zeros(1, dtype=torch.int64))      # CPU tensor
    ops.set_dropout_epoch(None)
    assert ops._DROPOUT_EPOCH is None


def test_pool_output_sizes_match_torch():
    from segmi.ops import conv_out_size, pool_out_size
    import torch.nn.functional as F
    for H in (15, 16, 17, 33, 256):
        for k, s, p, ceil in ((3, 2, 1, False), (2, 2, 0, True), (2, 2, 0, False), (3, 2, 0, True)):
            ref = F.max_pool2d(torch.zeros(1, 1, H, H), k, s, p, ceil_mode=ceil).shape[-1]
            assert pool_out_size(H, k, s, p, ceil) == ref, (H, k, s, p, ceil)
    assert conv_out_size(512, 3, 2, 1, 1) == 256 and conv_out_size(64, 3, 1, 4, 4) == 64


def test_pspnet_plugin_surface_matches_reference_manifest():
    import models
    gold = torch.load(os.path.join(GOLD, "pspnet_r50.pt"), weights_only=False)
    m = models.PSPNet(gold["num_classes"], backbone="resnet50", pretrained=False)
    mine = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert mine == [(k, tuple(s)) for k, s in gold["manifest"]]
    assert sum(p.numel() for p in m.get_backbone_params()) + sum(p.numel() for p in m.get_decoder_params()) == \
        sum(p.numel() for p in m.parameters())
    m.freeze_bn()
    assert all(not b.training for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d))
    assert "Nbr of trainable parameters" in str(m)
    m2 = models.PSPNet(3, backbone="resnet50", pretrained=False, freeze_backbone=True, use_aux=False)
    assert all(not p.requires_grad for p in m2.layer1.parameters()) and all(p.requires_grad for p in m2.master_branch.parameters())
    with pytest.raises(FileNotFoundError):
        models.PSPNet(3, backbone="resnet50", pretrained=True)


def test_product_path_refuses_cpu_tensors():
    import models
    from segmi import SegmiError
    from utils.losses import CrossEntropyLoss2d
    m = models.PSPNet(3, backbone="resnet50", pretrained=False)
    with pytest.raises(SegmiError):
        m(torch.randn(2, 3, 64, 64))
    with pytest.raises(SegmiError):
        CrossEntropyLoss2d()(torch.randn(1, 3, 4, 4), torch.zeros(1, 4, 4, dtype=torch.int64))


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "pytorch-segmentation_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)


def test_unet_and_deeplab_plugin_surface_matches_reference_manifests():
    """state_dict keys/shapes of the drop-ins equal those of the REAL reference models (manifests stored by gen_golden.py)."""
    import models
    un = torch.load(os.path.join(GOLD, "unet.pt"), weights_only=False)["s64"]
    m = models.UNet(un["num_classes"])
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(s)) for k, s in un["manifest"]]
    assert list(m.get_backbone_params()) == [] and len(list(m.get_decoder_params())) == len(list(m.parameters()))
    dl = torch.load(os.path.join(GOLD, "deeplab.pt"), weights_only=False)
    for case, rec in dl.items():
        m = models.DeepLab(rec["num_classes"], pretrained=False, **rec["kwargs"])
        mine = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        assert mine == [(k, tuple(s)) for k, s in rec["manifest"]], case
        nb = sum(p.numel() for p in m.get_backbone_params()) + sum(p.numel() for p in m.get_decoder_params())
        assert nb == sum(p.numel() for p in m.parameters())
    m = models.DeepLab(3, backbone="resnet50", pretrained=False, freeze_backbone=True, freeze_bn=True)   # NameError in the reference
    assert all(not p.requires_grad for p in m.backbone.parameters()) and all(p.requires_grad for p in m.decoder.parameters())
    # re-striding by module name (models/deeplabv3_plus.py:33-53)
    m16 = models.DeepLab(3, backbone="resnet50", pretrained=False, output_stride=16)
    assert m16.backbone.layer3[0].conv2.stride == (2, 2) and m16.backbone.layer4[0].conv2.stride == (1, 1)
    assert m16.backbone.layer4[2].conv2.dilation == (2, 2) and m16.backbone.layer4[0].downsample[0].stride == (1, 1)
    with pytest.raises(FileNotFoundError):
        models.DeepLab(3, backbone="xception", pretrained=True)


def test_lr_schedulers_match_reference_sequences():
    """Poly (with and without warm-up) and OneCycle, two parameter groups, stepped as trainer.py:52 does, against sequences
    recorded from the reference's utils/lr_scheduler.py (tests/golden/misc.pt)."""
    import warnings
    from utils import lr_scheduler
    gold = torch.load(os.path.join(GOLD, "misc.pt"), weights_only=False)["schedulers"]

    def run(cls, **kw):
        ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(2)]
        opt = torch.optim.SGD([{"params": ps[:1]}, {"params": ps[1:], "lr": 0.001}], lr=0.01, momentum=0.9)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sch = cls(opt, 3, 5, **kw)
            seq = []
            for epoch in range(1, 4):
                for _ in range(5):
                    sch.step(epoch=epoch - 1)
                    seq.append([g["lr"] for g in opt.param_groups] + [g["momentum"] for g in opt.param_groups])
        return torch.tensor(seq, dtype=torch.float64)

    assert torch.allclose(run(lr_scheduler.Poly), gold["Poly"], rtol=1e-12, atol=0)
    assert torch.allclose(run(lr_scheduler.Poly, warmup_epochs=1), gold["Poly_warmup"], rtol=1e-12, atol=0)
    assert torch.allclose(run(lr_scheduler.OneCycle), gold["OneCycle"], rtol=1e-12, atol=1e-18)


def test_data_prefetcher_refuses_cpu_device():
    """The prefetcher only stages into HBM; the trainer (like the reference, trainer.py:30) turns prefetch off on CPU."""
    from base import DataPrefetcher
    with pytest.raises(ValueError):
        DataPrefetcher([], device="cpu")


def test_every_cfg2_conv_layer_dispatches_to_an_lds_dma_kernel():
    """Host-side dispatch (no launch): each conv geometry of PSPNet-R50 at 8x512x512 runs on the LDS-DMA
    implicit-GEMM kernels in all three passes — none falls back to the register-staged `conv_gather_kernel`/`conv_wgrad_kernel` —
    and the tile shapes follow the output width: 128-wide n-tiles above 64 channels, the 32-wide tile for the 21-class heads."""
    from segmi.ops import conv_out_size, conv_variant
    from segmi._lib import ConvDesc
    # (C, K, R, stride, pad, dil, H)  — the distinct layers of the deep-base ResNet-50 (dilated layer3/4), PSP head, aux head
    layers = [(4, 64, 3, 2, 1, 1, 512), (64, 64, 3, 1, 1, 1, 256), (64, 128, 3, 1, 1, 1, 256),
              (128, 64, 1, 1, 0, 1, 128), (64, 64, 3, 1, 1, 1, 128), (64, 256, 1, 1, 0, 1, 128), (256, 64, 1, 1, 0, 1, 128),
              (256, 128, 1, 1, 0, 1, 128), (128, 128, 3, 2, 1, 1, 128), (128, 512, 1, 1, 0, 1, 64), (256, 512, 1, 2, 0, 1, 128),
              (512, 256, 1, 1, 0, 1, 64), (256, 256, 3, 1, 2, 2, 64), (256, 1024, 1, 1, 0, 1, 64), (1024, 256, 1, 1, 0, 1, 64),
              (1024, 512, 1, 1, 0, 1, 64), (512, 512, 3, 1, 4, 4, 64), (512, 2048, 1, 1, 0, 1, 64), (2048, 512, 1, 1, 0, 1, 64),
              (4096, 512, 3, 1, 1, 1, 64), (512, 21, 1, 1, 0, 1, 64), (1024, 512, 3, 1, 1, 1, 64), (2048, 512, 1, 1, 0, 1, 6)]
    for C, K, R, stride, pad, dil, H in layers:
        P = conv_out_size(H, R, stride, pad, dil)
        d = ConvDesc(8, H, H, C, K, R, R, P, P, stride, pad, dil, C, (K + 3) & ~3)
        names = [conv_variant(d, op) for op in (0, 1, 2)]
        assert names[0].startswith("conv_dma_kernel<") and names[1].startswith("conv_dma_kernel<"), (C, K, R, names)
        assert names[2].startswith("conv_wgrad_dma_kernel<"), (C, K, R, names)
        if K > 64 and 8 * P * P >= 65536:
            assert names[0].startswith("conv_dma_kernel<128, 128"), names[0]
        if K <= 32:
            assert names[0].startswith("conv_dma_kernel<128, 32"), names[0]


def test_synth_loader_shards_batches_by_rank():
    """Data parallelism: rank r of W draws global batch i*W + r at iteration i, so the W ranks of a job train on W different
    batches per step (every rank used to see the same data) and together cover exactly the single-process sequence."""
    from dataloaders import Synth
    kw = dict(num_classes=3, batch_size=2, height=32, width=32, iters=3, seed=5)
    single = [b for b in Synth(iters=6, **{k: v for k, v in kw.items() if k != "iters"})]
    r0 = [b for b in Synth(rank=0, world=2, **kw)]
    r1 = [b for b in Synth(rank=1, world=2, **kw)]
    assert len(r0) == len(r1) == 3
    for i in range(3):
        assert not torch.equal(r0[i][0], r1[i][0])
        assert torch.equal(r0[i][0], single[2 * i][0]) and torch.equal(r1[i][1], single[2 * i + 1][1])
    assert Synth(**kw).world == 1 and Synth(**kw).rank == 0          # no process group: the whole sequence


def test_bn_stats_epilogue_planning():
    """segmi_conv2d_fwd_stats_parts (no launch): one Welford partial per row tile of the forward kernel — 128-row tiles, 64-row
    tiles for problems that would not fill the chip — and 0 where the launch has no statistics epilogue (K % 4, split reduction of
    tiny outputs); more than 512 partials need the two-level merge workspace."""
    from segmi import lib
    from segmi._lib import ConvDesc
    d = ConvDesc(8, 64, 64, 512, 2048, 1, 1, 64, 64, 1, 0, 1, 512, 2048)
    assert lib.segmi_conv2d_fwd_stats_parts(d) == 8 * 64 * 64 // 128
    d = ConvDesc(8, 32, 32, 728, 728, 1, 1, 32, 32, 1, 0, 1, 728, 728)          # Xception middle flow: 64-row tiles
    assert lib.segmi_conv2d_fwd_stats_parts(d) == 8 * 32 * 32 // 64
    d = ConvDesc(8, 6, 6, 2048, 512, 1, 1, 6, 6, 1, 0, 1, 2048, 512)            # pyramid stage: split reduction
    assert lib.segmi_conv2d_fwd_stats_parts(d) == 0
    d = ConvDesc(8, 64, 64, 512, 21, 1, 1, 64, 64, 1, 0, 1, 512, 24)            # classifier: K % 4
    assert lib.segmi_conv2d_fwd_stats_parts(d) == 0
    assert lib.segmi_bn_parts_workspace(256, 2048) == 16
    assert lib.segmi_bn_parts_workspace(1024, 256) == 16 * 3 * 256 * 4 + 16


def test_depthwise_filters_are_stored_tap_major_behind_the_reference_shape():
    """Round 5: segmi.nn.Conv2d keeps a depthwise filter as [C,1,R,S] (the reference's state_dict key, shape and values:
    models/deeplabv3_plus.py:80) over memory in [R,S,C] order — the order the depthwise kernels read — so no per-step re-layout is
    needed.  Host-side contract, no GPU: load_state_dict / state_dict / deepcopy preserve values and layout, the flat view the
    kernels receive is the tap-major sequence, and a gradient produced in that order satisfies autograd's layout contract."""
    import copy
    from segmi import nn as snn, ops
    from segmi.optim import _same_layout
    torch.manual_seed(0)
    ref = torch.nn.Conv2d(24, 24, 3, padding=2, dilation=2, groups=24, bias=False)
    m = snn.Conv2d(24, 24, 3, padding=2, dilation=2, groups=24, bias=False)
    assert m.depthwise and tuple(m.weight.shape) == (24, 1, 3, 3) and m.weight.stride()[0] == 1 and m.weight.stride()[2:] == (3 * 24, 24)
    m.load_state_dict(ref.state_dict())
    assert torch.equal(m.weight, ref.weight) and torch.equal(m.state_dict()["weight"], ref.weight)
    assert m.weight.stride()[2:] == (72, 24)                                  # the copy went into the tap-major memory
    flat = ops._dw_rsc_view(m.weight.detach())
    assert flat is not None and flat.data_ptr() == m.weight.data_ptr()
    assert torch.equal(flat, ref.weight.detach().permute(2, 3, 0, 1).reshape(-1))   # [r][s][c]
    assert copy.deepcopy(m).weight.stride() == m.weight.stride()
    # a contiguous [C,1,R,S] filter (a foreign module) is NOT mistaken for the tap-major layout: it takes the re-layout path
    assert ops._dw_rsc_view(ref.weight.detach()) is None
    # the gradient the kernel writes ([R*S*C] flat) viewed back as [C,1,R,S]: same element order as the parameter
    g = torch.arange(216.0).view(3, 3, 24).permute(2, 0, 1).unsqueeze(1)
    assert _same_layout(g, m.weight) and tuple(g.shape) == tuple(m.weight.shape)
    # 1x1 "depthwise" (R = S = 1) degenerates gracefully
    one = snn.Conv2d(8, 8, 1, groups=8, bias=False)
    assert ops._dw_rsc_view(one.weight.detach()) is not None
