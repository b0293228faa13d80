"""Dilated ResNet encoders (stride-8 feature maps) on the segmi kernels.

Architecture and checkpoint key names follow the reference's models/resnet.py (PyTorch-Encoding
lineage): deep-base stem of three 3x3 convs (:136-145), Bottleneck with stride/dilation on the 3x3
(:72-121), `dilated=True` turning layer3/layer4 into dilation 2/4 with stride 1 where the first block
of a dilated stage uses half the dilation (:154-163,194-199).  Each Bottleneck runs
conv -> fused BN+ReLU -> conv -> fused BN+ReLU -> conv -> fused BN+residual+ReLU.

The ImageNet classifier head (`avgpool`, `fc`) is kept only so that published checkpoints load with
`strict=True`; classification forward is not part of the segmentation hot path.
"""
import math
import os

import torch
import torch.nn as nn

from segmi import nn as snn

__all__ = ["ResNet", "BasicBlock", "Bottleneck", "resnet18", "resnet34", "resnet50", "resnet101", "resnet152"]


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, previous_dilation=1, norm_layer=None):
        super().__init__()
        self.conv1 = snn.Conv2d(inplanes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = snn.Conv2d(planes, planes, 3, stride=1, padding=previous_dilation, dilation=previous_dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if self.downsample is None:
            out, identity = self.conv1(x, with_skip=True)
        else:
            # (SyncBN: bn2 and the projection's BN share one all-gather / one all-reduce — snn.sync_tail)
            out, identity = self.conv1(x), (None if snn.sync_tail(self.bn2, self.downsample) else self.downsample(x))
        out = self.bn1(out, relu=True)
        return snn.residual_out(self.conv2(out), self.bn2, self.downsample, x, identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, previous_dilation=1, norm_layer=None):
        super().__init__()
        self.conv1 = snn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = snn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = snn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.dilation = dilation
        self.stride = stride

    def forward(self, x):
        if self.downsample is None:
            out, identity = self.conv1(x, with_skip=True)     # the skip gradient is accumulated by conv1's dgrad, no autograd add
        else:
            # conv1 and the projection read the same x: one autograd node, their data gradients summed by the dgrad kernels
            # (SyncBN: bn3 and the projection's BN share one all-gather / one all-reduce — snn.sync_tail)
            out, proj = snn.conv_fan(x, [self.conv1, self.downsample[0]]) if len(self.downsample) == 2 else (self.conv1(x), None)
            if snn.sync_tail(self.bn3, self.downsample):
                identity = None
            else:
                identity = self.downsample[1](proj) if proj is not None else self.downsample(x)
        out = self.bn1(out, relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        return snn.residual_out(self.conv3(out), self.bn3, self.downsample, x, identity, proj if self.downsample is not None else None)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, dilated=True, multi_grid=False, deep_base=True,
                 norm_layer=snn.BatchNorm2d):
        super().__init__()
        self.inplanes = 128 if deep_base else 64
        if deep_base:
            self.conv1 = snn.Sequential(
                snn.Conv2d(3, 64, 3, stride=2, padding=1, bias=False), norm_layer(64), nn.ReLU(inplace=True),
                snn.Conv2d(64, 64, 3, stride=1, padding=1, bias=False), norm_layer(64), nn.ReLU(inplace=True),
                snn.Conv2d(64, 128, 3, stride=1, padding=1, bias=False))
        else:
            self.conv1 = snn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = snn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._stage(block, 64, layers[0], norm_layer=norm_layer)
        self.layer2 = self._stage(block, 128, layers[1], stride=2, norm_layer=norm_layer)
        if dilated:
            self.layer3 = self._stage(block, 256, layers[2], stride=1, dilation=2, norm_layer=norm_layer)
            self.layer4 = self._stage(block, 512, layers[3], stride=1, dilation=4, norm_layer=norm_layer, multi_grid=multi_grid)
        else:
            self.layer3 = self._stage(block, 256, layers[2], stride=2, norm_layer=norm_layer)
            self.layer4 = self._stage(block, 512, layers[3], stride=2, norm_layer=norm_layer)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _stage(self, block, planes, blocks, stride=1, dilation=1, norm_layer=None, multi_grid=False):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = snn.Sequential(
                snn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                norm_layer(planes * block.expansion))
        grid = [4, 8, 16]
        if multi_grid:
            first = grid[0]
        elif dilation in (1, 2):
            first = 1
        elif dilation == 4:
            first = 2
        else:
            raise RuntimeError("=> unknown dilation size: {}".format(dilation))
        seq = [block(self.inplanes, planes, stride, dilation=first, downsample=downsample,
                     previous_dilation=dilation, norm_layer=norm_layer)]
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            d = grid[i] if multi_grid else dilation
            seq.append(block(self.inplanes, planes, dilation=d, previous_dilation=dilation, norm_layer=norm_layer))
        return nn.Sequential(*seq)

    def features(self, x):
        x = self.conv1(x)
        x = self.bn1(x, relu=True)
        x = self.maxpool(x)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    def forward(self, x):
        raise NotImplementedError("ImageNet classification forward is outside the segmentation hot path; "
                                  "use .features(x) or a segmentation model built on this encoder")


def _build(block, layers, name, pretrained, root, **kwargs):
    model = ResNet(block, layers, **kwargs)
    if pretrained:
        # The reference downloads ImageNet weights here (models/resnet.py:256-306).  Containers
        # running this framework have no egress: a checkpoint already on disk is loaded, else fail loudly.
        cands = [f for f in (os.listdir(root) if os.path.isdir(root) else []) if f.startswith(name) and f.endswith((".pth", ".pt"))]
        if not cands:
            raise FileNotFoundError("pretrained=True needs %s*.pth under %r (no network access to download it); "
                                    "pass pretrained=False for random initialisation" % (name, root))
        model.load_state_dict(torch.load(os.path.join(root, sorted(cands)[0]), map_location="cpu"), strict=False)
    return model


def resnet18(pretrained=False, root="./pretrained", **kw):
    return _build(BasicBlock, [2, 2, 2, 2], "resnet18", pretrained, root, deep_base=False, **kw)


def resnet34(pretrained=False, root="./pretrained", **kw):
    return _build(BasicBlock, [3, 4, 6, 3], "resnet34", pretrained, root, deep_base=False, **kw)


def resnet50(pretrained=False, root="./pretrained", **kw):
    return _build(Bottleneck, [3, 4, 6, 3], "resnet50", pretrained, root, **kw)


def resnet101(pretrained=False, root="./pretrained", **kw):
    return _build(Bottleneck, [3, 4, 23, 3], "resnet101", pretrained, root, **kw)


def resnet152(pretrained=False, root="./pretrained", **kw):
    return _build(Bottleneck, [3, 8, 36, 3], "resnet152", pretrained, root, **kw)
