"""Import the REAL reference (read-only at /root/reference) behind stubs for the four python
packages missing from this image (cv2, torchvision, tensorboard, skimage) — SURVEY.md App. D.

Build-container only: /root/reference does not exist on the GPU box, so nothing in tests -m gpu,
smoke() or bench.py imports this module.  It must run in its own process: the reference uses the
same top-level package names (models, base, utils) as the drop-in.
"""
import os
import sys
import types

REFERENCE = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REFERENCE, "models"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Nop:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None

    def add_scalar(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass


def load():
    """Returns (models, losses) modules of the reference."""
    import torch
    if not available():
        raise RuntimeError("reference checkout not present at %s" % REFERENCE)
    sys.dont_write_bytecode = True  # keep /root/reference pristine
    _stub("cv2", setNumThreads=lambda n: None)
    sk = _stub("skimage")
    sk.filters = _stub("skimage.filters", gaussian=None)
    tv = _stub("torchvision")
    from oracle import tv_resnet     # restated torchvision ResNet-v1.5 factories (un-vendored third-party dependency)
    tv.models = _stub("torchvision.models", **{n: getattr(tv_resnet, n) for n in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152")})
    tv.utils = _stub("torchvision.utils", make_grid=None)
    tv.transforms = _stub("torchvision.transforms", ToTensor=_Nop, Normalize=_Nop, Compose=_Nop, Resize=_Nop, ToPILImage=_Nop)
    torch.utils.tensorboard = _stub("torch.utils.tensorboard", SummaryWriter=_Nop)
    for name in list(sys.modules):
        if name.split(".")[0] in ("models", "base", "utils", "dataloaders", "trainer"):
            del sys.modules[name]
    sys.path.insert(0, REFERENCE)
    import models
    from utils import losses
    return models, losses
