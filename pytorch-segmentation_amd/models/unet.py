"""U-Net — drop-in for the reference's models.UNet (models/unet.py:12-120).

Five-level encoder/decoder of (3x3 conv -> BN -> ReLU) x 2 blocks, 2x2 max-pool with ceil_mode on the
way down, ConvTranspose2d(k=2, s=2) + skip concat on the way up, 1x1 classifier.  Same constructor
signature, parameter-group accessors and checkpoint key names (`start_conv.0.weight`,
`down1.down_conv.4.running_var`, `up3.up.bias`, `final_conv.weight`, ... 130 keys).  Every operator
runs on libsegmi kernels: the up-convolution is a 1x1 MFMA convolution + depth_to_space
(segmi.ops.conv_transpose2x2), BN+ReLU pairs are fused, concat is two strided row copies.
"""
import torch.nn as nn

from base import BaseModel
from segmi import nn as snn
from segmi import ops


def x2conv(in_channels, out_channels, inner_channels=None):
    inner_channels = out_channels // 2 if inner_channels is None else inner_channels
    return snn.Sequential(
        snn.Conv2d(in_channels, inner_channels, kernel_size=3, padding=1, bias=False),
        snn.BatchNorm2d(inner_channels),
        nn.ReLU(inplace=True),
        snn.Conv2d(inner_channels, out_channels, kernel_size=3, padding=1, bias=False),
        snn.BatchNorm2d(out_channels),
        nn.ReLU(inplace=True))


class encoder(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.down_conv = x2conv(in_channels, out_channels)
        self.pool = snn.MaxPool2d(kernel_size=2, ceil_mode=True)

    def forward(self, x):
        return self.pool(self.down_conv(x))


class decoder(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.up = snn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
        self.up_conv = x2conv(in_channels, out_channels)

    def forward(self, x_copy, x, interpolate=True):
        x = self.up(x)
        if x.size(2) != x_copy.size(2) or x.size(3) != x_copy.size(3):
            if not interpolate:
                raise NotImplementedError("padding instead of interpolation is never taken by UNet.forward (models/unet.py:93-106)")
            x = ops.interpolate_bilinear(x, (x_copy.size(2), x_copy.size(3)), align_corners=True)
        return self.up_conv(ops.cat([x_copy, x]))


class UNet(BaseModel):
    def __init__(self, num_classes, in_channels=3, freeze_bn=False, **_):
        super().__init__()
        self.start_conv = x2conv(in_channels, 64)
        self.down1 = encoder(64, 128)
        self.down2 = encoder(128, 256)
        self.down3 = encoder(256, 512)
        self.down4 = encoder(512, 1024)
        self.middle_conv = x2conv(1024, 1024)
        self.up1 = decoder(1024, 512)
        self.up2 = decoder(512, 256)
        self.up3 = decoder(256, 128)
        self.up4 = decoder(128, 64)
        self.final_conv = snn.Conv2d(64, num_classes, kernel_size=1)
        self._initialize_weights()
        snn.link_conv_bn(self)        # conv -> BN pairs: BN statistics from the convolution's epilogue from the first step on
        if freeze_bn:
            self.freeze_bn()

    def _initialize_weights(self):
        for module in self.modules():
            if isinstance(module, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(module.weight)
                if module.bias is not None:
                    module.bias.data.zero_()
            elif isinstance(module, nn.BatchNorm2d):
                module.weight.data.fill_(1)
                module.bias.data.zero_()

    def forward(self, x):
        x1 = self.start_conv(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x = self.middle_conv(self.down4(x4))
        x = self.up1(x4, x)
        x = self.up2(x3, x)
        x = self.up3(x2, x)
        x = self.up4(x1, x)
        return self.final_conv(x)

    def get_backbone_params(self):
        return []   # no backbone: everything trains from scratch (reference models/unet.py:108-110)

    def get_decoder_params(self):
        return self.parameters()

    def freeze_bn(self):
        for module in self.modules():
            if isinstance(module, nn.BatchNorm2d):
                module.eval()
