from .pspnet import PSPNet  # noqa: F401
from .unet import UNet  # noqa: F401
from .deeplabv3_plus import DeepLab  # noqa: F401
