"""Restatement of torchvision 0.3's `torchvision.models.resnet` (ResNet-v1.5: stride on the 3x3 of a Bottleneck) in
plain torch — TEST INFRASTRUCTURE.  The reference's DeepLab calls `torchvision.models.resnet101(pretrained)`
(models/deeplabv3_plus.py:18) and then mutates its modules by attribute name (:33-53); torchvision is pinned in
requirements.txt:2 (`==0.3.0`) but is neither vendored nor installed here, so oracle/reference_harness.py injects this
module as `torchvision.models`.  Restated from the published architecture (He et al. 2015 + the v1.5 stride placement):
conv1 7x7/2, bn1, relu, maxpool 3x3/2, layer1-4 of BasicBlock/Bottleneck, avgpool, fc; attribute names
conv1/bn1/conv2/bn2/conv3/bn3/downsample.0/.1 are load-bearing.  No reference test pins it: "parity unpinned" for the
ResNet backbone of DeepLab.
"""
import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def _factory(block, layers):
    def make(pretrained=False, **kw):
        if pretrained:
            raise RuntimeError("no network: pretrained torchvision weights are unavailable")
        return ResNet(block, layers, **kw)
    return make


resnet18 = _factory(BasicBlock, [2, 2, 2, 2])
resnet34 = _factory(BasicBlock, [3, 4, 6, 3])
resnet50 = _factory(Bottleneck, [3, 4, 6, 3])
resnet101 = _factory(Bottleneck, [3, 4, 23, 3])
resnet152 = _factory(Bottleneck, [3, 8, 36, 3])
