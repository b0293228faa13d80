from .pspnet import PSPNet  # noqa: F401
