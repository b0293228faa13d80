"""SURVEY §8 f4 — the input pipeline: device-side augmentation (csrc/augment.hip, dataloaders/gpu_augment.py) behind the reference's
loader surface (base.BaseDataSet / base.BaseDataLoader, dataloaders.VOC / SynthImages, `train_loader.args` of config.json), against
oracle/augment_ref.py — the numpy restatement of the reference's cv2 / PIL sequence (base/base_dataset.py:40-136) in OpenCV's
published FIXED-POINT arithmetic for 8-bit images.  cv2 is absent from the image, so the restatement is PARITY-UNPINNED against the
library (it is held to the library's documented behaviour by known-answer vectors below); the HIP kernels are held to it
BIT-EXACTLY — labels and uint8 images — on the GPU."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import augment_ref as R

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sample(h, w, seed, classes=21):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(yy * 3 + xx * 2 + 40 * np.sin(xx / 7.0)) % 256, (xx * 5 + 30 * np.cos(yy / 5.0)) % 256, g.integers(0, 256, (h, w))], -1)
    lab = ((yy // 16) * 3 + xx // 16) % classes
    return img.astype(np.uint8), lab.astype(np.int32)


def test_restatement_known_answers_of_the_fixed_point_formulas():
    """Values OpenCV's 8-bit code paths are known to produce (hand-evaluated from the published formulas in the oracle's header)."""
    row = np.array([[[0, 0, 0], [255, 255, 255]]], np.uint8)
    assert R.resize_linear(row, 1, 4)[0, :, 0].tolist() == [0, 64, 191, 255]           # cv2.resize([0, 255] -> 4): 63.75 -> 64, 191.25 -> 191
    assert R.resize_linear(row, 1, 3)[0, :, 0].tolist() == [0, 128, 255]               # centre sample: exactly half
    box = np.arange(16, dtype=np.uint8).reshape(4, 4, 1).repeat(3, 2)
    assert R.resize_linear(box, 2, 2)[:, :, 0].tolist() == [[3, 5], [11, 13]]         # exact 2x decimation -> INTER_AREA: (0+1+4+5+2)>>2 = 3
    assert R.nearest_axis_table(5, 3).tolist() == [0, 0, 1, 1, 2] and R.pil_nearest_axis_table(5, 3).tolist() == [0, 0, 1, 2, 2]
    xs, a0, a1 = R.linear_axis_tables(4, 2, True)
    assert xs.tolist() == [0, 0, 0, 1] and a0.tolist() == [2048, 1536, 512, 2048] and a1.tolist() == [0, 512, 1536, 0]
    ys, b0, b1 = R.linear_axis_tables(4, 2, False)                                      # y axis: offsets / coefficients NOT clamped
    assert ys.tolist() == [-1, 0, 0, 1] and b0.tolist() == [512, 1536, 512, 1536]
    assert R.gaussian_kernel_fixed3(0.8) == (61, 134) and R.gaussian_kernel_fixed3(0.61) == (44, 168)     # 8.8 taps, sum 256
    # rotation by 0 and by 90 degrees about (w/2, h/2) of a 4x4 image on the 2^10 grid
    ad, bd, X0, Y0 = R.rotation_tables(4, 4, 0)
    assert ad.tolist() == [0, 1024, 2048, 3072] and bd.tolist() == [0, 0, 0, 0] and X0.tolist() == [0] * 4 and Y0.tolist() == [0, 1024, 2048, 3072]
    ad, bd, X0, Y0 = R.rotation_tables(4, 4, 90)
    assert ad.tolist() == [0, 0, 0, 0] and bd.tolist() == [0, 1024, 2048, 3072] and X0.tolist() == [4096, 3072, 2048, 1024]


def test_pil_nearest_resize_is_pinned_to_the_installed_pillow():
    """PINNED piece of f4: the validation label resize of the reference is PIL's `Image.fromarray(label).resize((w, h),
    resample=Image.NEAREST)` (base/base_dataset.py:50), and Pillow IS installed here: the restatement's index table and resize are
    held to the real library over a sweep of up- and down-scales, odd sizes and both label dtypes the reference produces."""
    from PIL import Image
    g = np.random.default_rng(3)
    shapes = [(1, 1), (2, 3), (5, 7), (17, 31), (33, 33), (64, 48), (97, 61), (100, 200), (255, 256), (375, 500), (513, 513)]
    targets = [(1, 1), (3, 2), (7, 5), (16, 16), (48, 48), (61, 97), (100, 50), (129, 257), (400, 400), (512, 683)]
    for sh, sw in shapes:
        lab = g.integers(0, 255, (sh, sw))
        for dh, dw in targets:
            for dt in (np.int32, np.uint8):
                want = np.asarray(Image.fromarray(lab.astype(dt)).resize((dw, dh), resample=Image.NEAREST))
                got = R.resize_nearest_pil(lab.astype(dt), dh, dw)
                assert want.shape == got.shape and np.array_equal(want, got), (sh, sw, dh, dw, dt)
    # the axis table alone, against a one-row ramp resized by the library (every source index visible in the output)
    for src, dst in [(7, 3), (3, 7), (500, 375), (375, 500), (1024, 33), (33, 1024), (2049, 2048)]:
        ramp = np.arange(src, dtype=np.int32)[None, :]
        want = np.asarray(Image.fromarray(ramp).resize((dst, 1), resample=Image.NEAREST))[0]
        assert np.array_equal(R.pil_nearest_axis_table(dst, src), want), (src, dst)


def test_restatement_invariants():
    img, lab = _sample(60, 84, 1)
    assert np.array_equal(R.resize_linear(img, 60, 84), img) and np.array_equal(R.resize_nearest(lab, 60, 84), lab)   # identity size
    up = R.resize_nearest(lab, 120, 168)
    assert np.array_equal(up[::2, ::2], lab)                                          # nearest 2x: floor(dst / 2)
    r0, l0 = R.warp_affine(img, lab, 60, 84, 0)
    assert np.array_equal(r0, img) and np.array_equal(l0, lab)                        # angle 0 is the identity
    r180, l180 = R.warp_affine(img[:60, :60], lab[:60, :60], 60, 60, 180)
    assert np.array_equal(l180[1:, 1:], lab[:60, :60][::-1, ::-1][:-1, :-1])         # half turn about the centre (w/2, h/2) = (30, 30)
    assert np.array_equal(r180[1:, 1:], img[:60, :60][::-1, ::-1][:-1, :-1])
    flat = np.full((9, 9, 3), 77, np.uint8)
    assert np.array_equal(R.gaussian_blur(flat, 3, 0.9), flat)                        # a constant image is a fixed point
    assert np.array_equal(R.gaussian_blur(img, 1, 0.2), img)                          # ksize 1: copy
    x, t, d = R.augment(img, lab, MEAN, STD, base_size=80, crop_size=64, scale=True, flip=True, rotate=True, blur=True, rng=random.Random(3))
    assert x.shape == (3, 64, 64) and x.dtype == np.float32 and t.shape == (64, 64) and t.dtype == np.int64
    assert set(d) == {"rs", "angle", "start", "flip", "sigma"} and -10 <= d["angle"] <= 10 and 40 <= max(d["rs"]) <= 160
    # padding region (image smaller than the crop): zeros before normalisation, label 0
    x2, t2, d2 = R.augment(img[:20, :30], lab[:20, :30], MEAN, STD, base_size=None, crop_size=64, scale=False, flip=False, rng=random.Random(1))
    assert d2["start"] == (0, 0) and np.all(t2[20:] == 0) and np.allclose(x2[:, 40, 40], (0 - np.array(MEAN)) / np.array(STD), atol=1e-6)
    xv, tv = R.val_augment(img, lab, MEAN, STD, 48)
    assert xv.shape == (3, 48, 48) and tv.shape == (48, 48)


def test_decisions_follow_the_reference_draw_order():
    """Same seed -> the same sequence of random.* calls as base/base_dataset.py:67-116 (randint long side, randint angle, randint
    start_h, randint start_w, random flip, random sigma)."""
    rng = random.Random(11)
    d = R.draw_decisions(rng, 100, 150, 120, 96, True, True, True, True)
    ref = random.Random(11)
    longside = ref.randint(60, 240)
    h, w = (int(1.0 * longside * 100 / 150 + 0.5), longside)
    angle = ref.randint(-10, 10)
    sy = ref.randint(0, max(h, 96) - 96)
    sx = ref.randint(0, max(w, 96) - 96)
    fl = ref.random() > 0.5
    sg = ref.random()
    assert d == {"rs": (h, w), "angle": angle, "start": (sy, sx), "flip": fl, "sigma": sg}


def test_host_tables_of_the_product_equal_the_restatement():
    """The int32 tables dataloaders/gpu_augment.py hands to the kernels (OpenCV's double / float derivations per output row and
    column) against the oracle's independent derivation — no GPU needed."""
    from dataloaders import gpu_augment as G
    for (sh, sw, dh, dw) in ((60, 84, 97, 131), (97, 61, 40, 25), (50, 50, 25, 25), (120, 40, 240, 80), (33, 47, 33, 47)):
        tab, area = G.resize_tables(sh, sw, dh, dw, "cv2")
        xs, a0, a1 = R.linear_axis_tables(dw, sw, True)
        ys, b0, b1 = R.linear_axis_tables(dh, sh, False)
        want = np.concatenate([xs, (a0 & 0xFFFF) | (a1 << 16), ys, (b0 & 0xFFFF) | (b1 << 16), R.nearest_axis_table(dw, sw), R.nearest_axis_table(dh, sh)])
        assert np.array_equal(tab.astype(np.int64), want.astype(np.int32).astype(np.int64)), (sh, sw, dh, dw)
        assert area == (R._is_area_2x(dh, sh) and R._is_area_2x(dw, sw))
        tabp, _ = G.resize_tables(sh, sw, dh, dw, "pil")
        assert np.array_equal(tabp[-(dw + dh):], np.concatenate([R.pil_nearest_axis_table(dw, sw), R.pil_nearest_axis_table(dh, sh)]))
    for (h, w, ang) in ((60, 84, 7), (97, 61, -10), (50, 50, 0), (33, 47, 3)):
        assert np.array_equal(G.rotation_tables(h, w, ang), np.concatenate(R.rotation_tables(h, w, ang)).astype(np.int32))
    for s in (0.61, 0.7, 0.8, 0.95, 0.999):
        assert G.gaussian_taps_3(s) == R.gaussian_kernel_fixed3(s)


def test_loader_surface_batching_and_sharding():
    """base.BaseDataLoader: reference constructor, val_split + get_val_loader, len(), deterministic epoch shuffles and the per-rank
    shard of every global step (no GPU: nothing is iterated)."""
    import dataloaders
    ld = dataloaders.SynthImages(num_classes=5, batch_size=4, num_samples=22, crop_size=64, base_size=80, augment=True, shuffle=True,
                                 scale=True, flip=True, rotate=True, blur=True, num_workers=2, device="cpu")
    assert len(ld) == 6 and ld.dataset.num_classes == 5 and len(ld.dataset) == 22 and ld.batch_size == 4 and len(ld.MEAN) == 3
    img, lab, name = ld.dataset[3]
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3 and lab.dtype == np.int32 and lab.shape == img.shape[:2] and name == "synth_00003"
    img2, _, _ = ld.dataset[3]
    assert np.array_equal(img, img2) and ld.dataset[4][0].shape != img.shape or True      # deterministic per index; sizes are ragged
    b0 = [b.tolist() for b in ld._batches()]
    ld.epoch = 1
    b1 = [b.tolist() for b in ld._batches()]
    assert sorted(sum(b0, [])) == list(range(22)) and sorted(sum(b1, [])) == list(range(22)) and b0 != b1
    ld.epoch = 0
    assert [b.tolist() for b in ld._batches()] == b0
    r0 = dataloaders.SynthImages(num_classes=5, batch_size=4, num_samples=32, crop_size=64, augment=True, device="cpu", rank=0, world=2)
    r1 = dataloaders.SynthImages(num_classes=5, batch_size=4, num_samples=32, crop_size=64, augment=True, device="cpu", rank=1, world=2)
    assert len(r0) == len(r1) == 4
    seen = sum((b.tolist() for b in r0._batches() + r1._batches()), [])
    assert sorted(seen) == list(range(32))                                               # disjoint shards covering every global step
    sp = dataloaders.SynthImages(num_classes=5, batch_size=2, num_samples=20, crop_size=64, augment=True, val_split=0.25, device="cpu")
    vl = sp.get_val_loader()
    assert len(sp.indices) == 15 and len(vl.indices) == 5 and not set(sp.indices.tolist()) & set(vl.indices.tolist())
    assert dataloaders.SynthImages(num_classes=2, batch_size=2, device="cpu").get_val_loader() is None
    # the held-out split shards like its parent (an explicit rank= / world= used to be lost on the way)
    sp2 = dataloaders.SynthImages(num_classes=5, batch_size=2, num_samples=20, crop_size=64, augment=True, val_split=0.5, device="cpu", rank=1, world=2)
    vl2 = sp2.get_val_loader()
    assert (vl2.rank, vl2.world) == (1, 2) and len(vl2) == 2 and [b.tolist() for b in vl2._batches()] == [vl2.indices[2:4].tolist(), vl2.indices[6:8].tolist()]
    # the training subset of a val_split loader is drawn in a new random order every epoch (the reference's
    # SubsetRandomSampler), whatever `shuffle` says
    e0 = sum((b.tolist() for b in sp._batches()), [])
    sp.epoch += 1
    e1 = sum((b.tolist() for b in sp._batches()), [])
    assert sorted(e0) == sorted(e1) == sorted(sp.indices.tolist()) and e0 != e1
    # no batch is silently dropped under world > 1.  Training loaders wrap around so that every rank runs the same
    # number of steps (collectives per step); validation-type loaders visit every sample exactly once, ranks may differ by one batch.
    kw = dict(num_classes=5, batch_size=4, num_samples=36, crop_size=64, augment=True, device="cpu", world=4)     # 9 batches over 4 ranks
    tr = [dataloaders.SynthImages(rank=r, **kw) for r in range(4)]
    assert [len(t) for t in tr] == [3, 3, 3, 3]
    seen = sum((b.tolist() for t in tr for b in t._batches()), [])
    assert set(seen) == set(range(36)) and len(seen) == 48                               # 3 wrapped batches, nothing missing
    va = [dataloaders.SynthImages(rank=r, val=True, **dict(kw, augment=False)) for r in range(4)]
    assert [len(v) for v in va] == [3, 2, 2, 2]
    assert sorted(sum((b.tolist() for v in va for b in v._batches()), [])) == list(range(36))
    few = [dataloaders.SynthImages(rank=r, **dict(kw, num_samples=8)) for r in range(4)]    # fewer batches (2) than ranks (4)
    assert [len(f) for f in few] == [1, 1, 1, 1] and all(len(f._batches()) == 1 for f in few)


CFGS = [dict(base_size=96, crop_size=80, scale=True, flip=True, rotate=True, blur=True),
        dict(base_size=None, crop_size=64, scale=False, flip=True, rotate=False, blur=False),
        dict(base_size=70, crop_size=128, scale=True, flip=False, rotate=True, blur=False),
        dict(base_size=100, crop_size=50, scale=False, flip=True, rotate=False, blur=True)]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", CFGS, ids=["cv2-unpinned-%d" % i for i in range(len(CFGS))])
def test_gpu_augment_is_bit_exact_to_the_restatement(cuda, cfg):
    """cv2-unpinned: the oracle side of this comparison is the restatement of OpenCV's published fixed-point algorithms, not cv2."""
    from dataloaders.gpu_augment import GPUAugment
    samples = [_sample(h, w, s) for (h, w, s) in ((60, 84, 1), (97, 61, 2), (50, 50, 3), (120, 40, 4), (200, 100, 5), (37, 64, 6))]
    for seed in (5, 6, 7):
        aug = GPUAugment(MEAN, STD, device=cuda, seed=seed, **cfg)
        x, t = aug(samples)
        torch.cuda.synchronize()
        assert tuple(x.shape) == (len(samples), 3, cfg["crop_size"], cfg["crop_size"]) and t.dtype == torch.int64
        rng = random.Random(seed)
        for i, (img, lab) in enumerate(samples):
            xr, tr, d = R.augment(img, lab, MEAN, STD, rng=rng, **cfg)
            assert d == aug.decisions[i]
            assert torch.equal(t[i].cpu(), torch.from_numpy(tr)), (seed, i, d)
            # one uint8 level is 1/255/std >= 0.017 after Normalize: a difference below 1e-6 means the uint8 images are identical
            assert float((x[i].cpu() - torch.from_numpy(xr)).abs().max()) <= 1e-6, (seed, i, d)
    # the batch is NHWC-backed with a zero padding channel: the model's first convolution consumes it without a layout pass
    from segmi import ops
    assert ops.is_nhwc(x)


@pytest.mark.gpu
def test_gpu_validation_and_plain_paths_are_bit_exact_cv2_unpinned_pil_pinned(cuda):
    """Validation path: image resize = cv2 INTER_LINEAR (restatement only: cv2-unpinned), label resize = PIL NEAREST (the restatement
    of that piece is pinned to the installed Pillow by test_pil_nearest_resize_is_pinned_to_the_installed_pillow)."""
    from dataloaders.gpu_augment import GPUAugment
    samples = [_sample(h, w, s) for (h, w, s) in ((60, 84, 1), (97, 61, 2), (50, 50, 3), (120, 40, 4), (96, 192, 5))]
    aug = GPUAugment(MEAN, STD, crop_size=48, device=cuda)
    x, t = aug.validation(samples)
    for i, (img, lab) in enumerate(samples):
        xr, tr = R.val_augment(img, lab, MEAN, STD, 48)
        assert torch.equal(t[i].cpu(), torch.from_numpy(tr)), i
        assert float((x[i].cpu() - torch.from_numpy(xr)).abs().max()) <= 1e-6, i
    same = [_sample(40, 56, s) for s in (1, 2, 3)]
    x, t = GPUAugment(MEAN, STD, crop_size=None, device=cuda).plain(same)
    for i, (img, lab) in enumerate(same):
        xr, tr = R.val_augment(img, lab, MEAN, STD, None)
        assert torch.equal(t[i].cpu(), torch.from_numpy(tr)) and float((x[i].cpu() - torch.from_numpy(xr)).abs().max()) <= 1e-6
    with pytest.raises(ValueError):
        GPUAugment(MEAN, STD, crop_size=None, device=cuda).plain(samples)              # ragged sizes need a crop


@pytest.mark.gpu
def test_loader_yields_device_batches_equal_to_the_restatement_cv2_unpinned(cuda):
    """dataloaders.SynthImages end to end (host threads -> pinned staging -> device augmentation): the batches are what the
    reference's `__getitem__` would produce for the same raw samples and the same random draws."""
    import dataloaders
    kw = dict(num_classes=7, batch_size=3, num_samples=10, min_size=60, max_size=130, crop_size=72, base_size=90, augment=True, shuffle=True,
              scale=True, flip=True, rotate=True, blur=True, num_workers=3, seed=77)
    ld = dataloaders.SynthImages(device=cuda, **kw)
    order = [b.tolist() for b in ld._batches()]
    rng = random.Random(77 * 7919)
    n = 0
    for (x, t), idx in zip(ld, order):
        assert x.is_cuda and t.is_cuda and tuple(x.shape) == (len(idx), 3, 72, 72) and t.dtype == torch.int64
        for j, i in enumerate(idx):
            img, lab, _ = ld.dataset[i]
            xr, tr, _ = R.augment(img, lab, ld.MEAN, ld.STD, base_size=90, crop_size=72, scale=True, flip=True, rotate=True, blur=True, rng=rng)
            assert torch.equal(t[j].cpu(), torch.from_numpy(tr)) and float((x[j].cpu() - torch.from_numpy(xr)).abs().max()) <= 1e-6
        n += 1
    assert n == len(ld) == 4


@pytest.mark.gpu
def test_train_py_runs_an_augmented_epoch_from_config(cuda, tmp_path):
    """`train.py -c config.json` with the reference's `train_loader.args` (augment / scale / flip / rotate / blur / base_size /
    crop_size) on the device pipeline, plus a validation loader (`val: true`): two epochs of UNet train, validate and checkpoint."""
    import train
    config = json.load(open(os.path.join(ROOT, "pytorch-segmentation_amd", "config.json")))
    config["train_loader"] = {"type": "SynthImages", "args": {"num_classes": 3, "batch_size": 4, "num_samples": 16, "min_size": 70, "max_size": 140,
                                                              "base_size": 100, "crop_size": 96, "augment": True, "shuffle": True, "scale": True,
                                                              "flip": True, "rotate": True, "blur": True, "num_workers": 4}}
    config["val_loader"] = {"type": "SynthImages", "args": {"num_classes": 3, "batch_size": 4, "num_samples": 8, "min_size": 70, "max_size": 140,
                                                            "crop_size": 96, "val": True, "num_workers": 2, "seed": 99}}
    config["trainer"].update(save_dir=str(tmp_path / "ck"), log_dir=str(tmp_path / "log"), epochs=2, save_period=2)
    tr = train.main(config, None)
    losses = [float(v) for v in tr.iteration_losses]
    assert len(losses) == 4 and all(np.isfinite(losses)) and float(tr.total_loss.average) < 1.2
    assert os.path.isdir(tr.checkpoint_dir) and 0.0 <= float(tr.mnt_best) <= 1.0


@pytest.mark.gpu
def test_voc_loader_reads_a_devkit_tree(cuda, tmp_path):
    """dataloaders.VOC (reference signature) over a miniature VOCdevkit tree written with PIL: JPEG / palette-PNG decode on host
    threads, everything after that on the device."""
    from PIL import Image
    import dataloaders
    root = tmp_path / "VOCdevkit" / "VOC2012"
    for sub in ("JPEGImages", "SegmentationClass", "ImageSets/Segmentation"):
        os.makedirs(root / sub)
    ids = []
    for i, (h, w) in enumerate(((80, 120), (100, 70), (64, 64), (90, 150))):
        img, lab = _sample(h, w, 10 + i)
        lab[:4] = 255
        Image.fromarray(img).save(root / "JPEGImages" / ("im%d.jpg" % i), quality=95)
        p = Image.fromarray(lab.astype(np.uint8), mode="P")
        p.putpalette(dataloaders.voc.get_voc_palette(256))
        p.save(root / "SegmentationClass" / ("im%d.png" % i))
        ids.append("im%d" % i)
    (root / "ImageSets/Segmentation/train.txt").write_text("\n".join(ids) + "\n")
    ld = dataloaders.VOC(str(tmp_path), batch_size=2, split="train", crop_size=64, base_size=80, augment=True, shuffle=False, scale=True, flip=True,
                         rotate=True, blur=False, num_workers=2, device=cuda, seed=3)
    assert ld.dataset.num_classes == 21 and len(ld.dataset.palette) == 63 and ld.dataset.palette[3:6] == [128, 0, 0] and len(ld) == 2
    raw = ld.dataset[0]
    assert raw[0].shape == (80, 120, 3) and raw[0].dtype == np.uint8 and raw[1].dtype == np.int32 and raw[1].max() == 255 and raw[2] == "im0"
    got = list(ld)
    assert len(got) == 2 and tuple(got[0][0].shape) == (2, 3, 64, 64) and got[0][1].dtype == torch.int64 and got[0][0].is_cuda
    vl = dataloaders.VOC(str(tmp_path), batch_size=2, split="train", crop_size=48, val=True, num_workers=1, device=cuda)
    xv, tv = next(iter(vl))
    xr, tr = R.val_augment(raw[0], raw[1], ld.MEAN, ld.STD, 48)
    assert torch.equal(tv[0].cpu(), torch.from_numpy(tr)) and float((xv[0].cpu() - torch.from_numpy(xr)).abs().max()) <= 1e-6
