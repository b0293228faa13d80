"""PSPNet (dilated ResNet + pyramid pooling) — drop-in for the reference's models.PSPNet.

Constructor signature, `(output, aux)` training-mode return convention, parameter-group accessors
and checkpoint key names follow models/pspnet.py:11-105 of the reference; every operator in
forward/backward is a libsegmi HIP kernel (NHWC, fp32 MFMA implicit-GEMM convolutions, fused
BN+ReLU(+residual), one-launch adaptive pooling and gather-form bilinear resize).
"""
import os
from itertools import chain

import torch
import torch.nn as nn

from base import BaseModel
from models import resnet
from segmi import nn as snn
from segmi import ops
from utils.helpers import initialize_weights, set_trainable


class _PSPModule(nn.Module):
    """Pyramid pooling: bins -> 1x1 conv -> BN -> ReLU -> bilinear(align_corners=True) back to the
    feature size; concat with the features; 3x3 bottleneck conv + BN + ReLU + Dropout2d(0.1)."""

    def __init__(self, in_channels, bin_sizes, norm_layer):
        super().__init__()
        out_channels = in_channels // len(bin_sizes)
        self.stages = nn.ModuleList(
            snn.Sequential(snn.AdaptiveAvgPool2d(output_size=b),
                           snn.Conv2d(in_channels, out_channels, 1, bias=False),
                           norm_layer(out_channels),
                           nn.ReLU(inplace=True))
            for b in bin_sizes)
        self.bottleneck = snn.Sequential(
            snn.Conv2d(in_channels + out_channels * len(bin_sizes), out_channels, 3, padding=1, bias=False),
            norm_layer(out_channels),
            nn.ReLU(inplace=True),
            snn.Dropout2d(0.1))

    def forward(self, features):
        size = (features.size(2), features.size(3))
        bins = [stage[0].output_size if isinstance(stage[0].output_size, int) else stage[0].output_size[0] for stage in self.stages]
        fused = len(bins) <= 4 and max(bins) <= 8 and min(size) >= max(bins)
        # all pyramid levels pooled in ONE pass over the feature map (and one gradient write in backward); the
        # AdaptiveAvgPool2d modules stay in `stages` so checkpoint keys / indices are the reference's
        pooled = ops.pyramid_pool(features, bins) if fused else [stage[0](features) for stage in self.stages]
        if all(len(stage) == 4 and isinstance(stage[2], snn.BatchNorm2d) and isinstance(stage[3], nn.ReLU) for stage in self.stages):
            # conv of every level first, then the four BN+ReLU as one group: SyncBN layers share one all-gather / one all-reduce
            # (ops.sync_batch_norm_group; local BN: exactly the layer-by-layer calls)
            branches = ops.sync_batch_norm_group([stage[1](p) for stage, p in zip(self.stages, pooled)], [stage[2] for stage in self.stages])
        else:
            branches = [snn.run_fused(list(stage)[1:], p) for stage, p in zip(self.stages, pooled)]  # [N, C/4, b, b] each
        conv = self.bottleneck[0]
        if self.factored and self._factorable(conv, features, branches):
            # concat + 3x3 convolution in factored form: the convolution runs over the feature channels only, the (linear)
            # upsample-then-convolve of every pyramid branch is a b*b-pixel GEMM plus a separable interpolation — half of the
            # bottleneck's MACs and the 4096-channel concat buffer disappear (segmi.ops.pyramid_bottleneck_conv)
            y = ops.pyramid_bottleneck_conv(features, branches, conv.weight)
            return snn.run_fused(list(self.bottleneck)[1:], y)
        pyramid = [features] + [ops.interpolate_bilinear(b, size, align_corners=True) for b in branches]
        return self.bottleneck(ops.cat(pyramid))

    factored = os.environ.get("SEGMI_PSP_FACTORED", "1") != "0"      # A/B switch; the unfactored path is the test reference

    @staticmethod
    def _factorable(conv, features, branches):
        return (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
                and conv.bias is None and conv.groups == 1 and len(branches) <= 4 and features.shape[1] % 4 == 0
                and conv.out_channels % 4 == 0 and all(b.shape[1] % 4 == 0 and b.shape[2] == b.shape[3] for b in branches)
                and conv.weight.is_contiguous(memory_format=torch.channels_last))


class PSPNet(BaseModel):
    def __init__(self, num_classes, in_channels=3, backbone="resnet152", pretrained=True, use_aux=True,
                 freeze_bn=False, freeze_backbone=False):
        super().__init__()
        norm_layer = snn.BatchNorm2d
        encoder = getattr(resnet, backbone)(pretrained, norm_layer=norm_layer)
        width = encoder.fc.in_features
        self.use_aux = use_aux

        stem = [encoder.conv1, encoder.bn1, encoder.relu, encoder.maxpool]
        if in_channels != 3:
            stem[0] = snn.Conv2d(in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.initial = snn.Sequential(*stem)
        self.layer1, self.layer2 = encoder.layer1, encoder.layer2
        self.layer3, self.layer4 = encoder.layer3, encoder.layer4

        self.master_branch = snn.Sequential(
            _PSPModule(width, bin_sizes=[1, 2, 3, 6], norm_layer=norm_layer),
            snn.Conv2d(width // 4, num_classes, 1))
        self.auxiliary_branch = snn.Sequential(
            snn.Conv2d(width // 2, width // 4, 3, padding=1, bias=False),
            norm_layer(width // 4),
            nn.ReLU(inplace=True),
            snn.Dropout2d(0.1),
            snn.Conv2d(width // 4, num_classes, 1))

        initialize_weights(self.master_branch, self.auxiliary_branch)
        snn.link_conv_bn(self)        # conv -> BN pairs: BN statistics from the convolution's epilogue from the first step on
        if freeze_bn:
            self.freeze_bn()
        if freeze_backbone:
            set_trainable([self.initial, self.layer1, self.layer2, self.layer3, self.layer4], False)

    def forward(self, x):
        size = (x.size(2), x.size(3))
        x = self.layer2(self.layer1(self.initial(x)))
        x_aux = self.layer3(x)
        x = self.layer4(x_aux)
        output = ops.interpolate_bilinear(self.master_branch(x), size, align_corners=False)
        if self.training and self.use_aux:
            aux = ops.interpolate_bilinear(self.auxiliary_branch(x_aux), size, align_corners=False)
            return output, aux
        return output

    def get_backbone_params(self):
        return chain(self.initial.parameters(), self.layer1.parameters(), self.layer2.parameters(),
                     self.layer3.parameters(), self.layer4.parameters())

    def get_decoder_params(self):
        return chain(self.master_branch.parameters(), self.auxiliary_branch.parameters())

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()
