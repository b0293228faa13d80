"""DeepLabV3+ (ResNet or aligned-Xception encoder, ASPP, decoder) — drop-in for the reference's models.DeepLab.

Reference: models/deeplabv3_plus.py — `ResNet` wrapper :15-63, `SeparableConv2d` :70-86, `Block` :89-132,
`Xception` :134-247, `ASSP` :253-297, `Decoder` :303-330, `DeepLab` :336-378.  Constructor signatures, module
attribute names and therefore checkpoint keys (848 for Xception, 680 for ResNet-101) are the reference's.

Behaviour that is easy to miss and is reproduced on purpose:
  * `Block`: the first module of `rep` is an in-place ReLU, so in the reference the skip branch (and the identity
    shortcut of the 16 middle-flow blocks) sees relu(x), not x (:99-101,122-131).  Here the ReLU is applied once and
    both branches read its output.  `block1` is built with `use_1st_relu=False` and has no such ReLU.
  * `Xception.forward` applies no ReLU between `bn2` and `block1` (:205-207).
  * the torchvision ResNet is re-strided/dilated after construction by module name (:33-53): for output_stride 16
    layer3 keeps stride 2 and every `conv2` of layer4 gets dilation 2, stride 1; `layer0` is a fresh 7x7 stem when
    `pretrained=False` (:19-26).
Deviations: `freeze_backbone=True` works (the reference raises NameError: `set_trainable` is not imported, :8,354);
`pretrained=True` loads a checkpoint from disk instead of downloading (no egress).

torchvision is not a dependency: the ResNet-v1.5 encoder (stride on the 3x3, `conv1/bn1/conv2/bn2/conv3/bn3/downsample.0/1`
attribute names, as torchvision 0.3 `resnet.py` defines them) is restated below on segmi modules.
"""
import os
from itertools import chain

import torch
import torch.nn as nn

from base import BaseModel
from segmi import nn as snn
from segmi import ops
from utils.helpers import initialize_weights, set_trainable


# ----------------------------------------------------------------------------- torchvision-style ResNet encoder
class _TVBasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = snn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = snn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = snn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = snn.BatchNorm2d(planes)
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


class _TVBottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = snn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = snn.BatchNorm2d(planes)
        self.conv2 = snn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = snn.BatchNorm2d(planes)
        self.conv3 = snn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = snn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
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


_TV_LAYERS = {"resnet18": (_TVBasicBlock, (2, 2, 2, 2)), "resnet34": (_TVBasicBlock, (3, 4, 6, 3)),
              "resnet50": (_TVBottleneck, (3, 4, 6, 3)), "resnet101": (_TVBottleneck, (3, 4, 23, 3)),
              "resnet152": (_TVBottleneck, (3, 8, 36, 3))}


class _TVResNet(nn.Module):
    """The part of torchvision.models.resnet.ResNet the reference keeps: conv1/bn1/relu/maxpool + layer1-4."""

    def __init__(self, block, layers):
        super().__init__()
        self.inplanes = 64
        self.conv1 = snn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = snn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = snn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = snn.Sequential(snn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                        snn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def _load_checkpoint(module, name, root="./pretrained"):
    cands = [f for f in (os.listdir(root) if os.path.isdir(root) else []) if f.startswith(name) and f.endswith((".pth", ".pt"))]
    if not cands:
        raise FileNotFoundError("pretrained=True needs %s*.pth under %r (no network access to download it); "
                                "pass pretrained=False for random initialisation" % (name, root))
    module.load_state_dict(torch.load(os.path.join(root, sorted(cands)[0]), map_location="cpu"), strict=False)


class ResNet(nn.Module):
    def __init__(self, in_channels=3, output_stride=16, backbone="resnet101", pretrained=True):
        super().__init__()
        block, layers = _TV_LAYERS[backbone]
        model = _TVResNet(block, layers)
        if pretrained:
            _load_checkpoint(model, backbone)
        if not pretrained or in_channels != 3:
            self.layer0 = snn.Sequential(snn.Conv2d(in_channels, 64, 7, stride=2, padding=3, bias=False),
                                         snn.BatchNorm2d(64), nn.ReLU(inplace=True),
                                         snn.MaxPool2d(kernel_size=3, stride=2, padding=1))
            initialize_weights(self.layer0)
        else:
            self.layer0 = snn.Sequential(model.conv1, model.bn1, model.relu, model.maxpool)
        self.layer1, self.layer2, self.layer3, self.layer4 = model.layer1, model.layer2, model.layer3, model.layer4

        if output_stride == 16:
            s3, s4, d3, d4 = (2, 1, 1, 2)
        elif output_stride == 8:
            s3, s4, d3, d4 = (1, 1, 2, 4)
        else:
            raise ValueError("output_stride must be 8 or 16")
        basic = backbone in ("resnet18", "resnet34")
        if output_stride == 8:
            self._restride(self.layer3, s3, d3, basic)
        self._restride(self.layer4, s4, d4, basic)

    @staticmethod
    def _restride(layer, s, d, basic):
        for n, m in layer.named_modules():
            if ("conv1" in n and basic) or "conv2" in n:
                m.dilation, m.padding, m.stride = (d, d), (d, d), (s, s)
            elif "downsample.0" in n:
                m.stride = (s, s)

    def forward(self, x):
        x = self.layer1(self.layer0(x))
        low_level_features = x
        x = self.layer4(self.layer3(self.layer2(x)))
        return x, low_level_features


# ----------------------------------------------------------------------------- aligned Xception encoder
class SeparableConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, bias=False, BatchNorm=nn.BatchNorm2d):
        super().__init__()
        padding = dilation if dilation > kernel_size // 2 else kernel_size // 2
        self.conv1 = snn.Conv2d(in_channels, in_channels, kernel_size, stride, padding=padding, dilation=dilation,
                                groups=in_channels, bias=bias)
        self.bn = snn.BatchNorm2d(in_channels)
        self.pointwise = snn.Conv2d(in_channels, out_channels, 1, 1, bias=bias)

    def forward(self, x):
        return self.pointwise(self.bn(self.conv1(x)))

    def bn_fusable(self, bn):
        """Can this layer take the PRE-normalisation input of `bn` (+ReLU) — segmi.ops.batch_norm_depthwise?"""
        return ops.batch_norm_depthwise_ok(bn, self.conv1)

    def forward_bn(self, z, bn, relu=True):
        """self(relu(bn(z))) with the BatchNorm + ReLU applied inside the depthwise kernels' loads (the reference's
        `BatchNorm2d -> ReLU -> SeparableConv2d` runs of Block.rep and of the exit flow, models/deeplabv3_plus.py:99-119, 225-232):
        the normalised tensor is never written or read — bit-identical to the separate passes."""
        fuse = self.conv1._bn_consumer and torch.is_grad_enabled()
        return self.pointwise(self.bn(ops.batch_norm_depthwise(z, bn, self.conv1, relu=relu, bn_stats=fuse)))


class Block(nn.Module):
    fused_tail = os.environ.get("SEGMI_XCEPTION_FUSED_TAIL", "1") == "1"     # `+ skip` (+ the consumer's ReLU) inside the last BatchNorm's pass

    def __init__(self, in_channels, out_channels, stride=1, dilation=1, exit_flow=False, use_1st_relu=True):
        super().__init__()
        if in_channels != out_channels or stride != 1:
            self.skip = snn.Conv2d(in_channels, out_channels, 1, stride=stride, bias=False)
            self.skipbn = snn.BatchNorm2d(out_channels)
        else:
            self.skip = None
        self.relu = nn.ReLU(inplace=True)
        rep = [self.relu, SeparableConv2d(in_channels, out_channels, 3, stride=1, dilation=dilation), snn.BatchNorm2d(out_channels),
               self.relu, SeparableConv2d(out_channels, out_channels, 3, stride=1, dilation=dilation), snn.BatchNorm2d(out_channels),
               self.relu, SeparableConv2d(out_channels, out_channels, 3, stride=stride, dilation=dilation), snn.BatchNorm2d(out_channels)]
        if exit_flow:
            rep[3:6] = rep[:3]
            rep[:3] = [self.relu, SeparableConv2d(in_channels, in_channels, 3, 1, dilation), snn.BatchNorm2d(in_channels)]
        self.use_1st_relu = use_1st_relu
        if not use_1st_relu:
            rep = rep[1:]
        self.rep = snn.Sequential(*rep)   # same child indices (-> checkpoint keys) as the reference's nn.Sequential

    def forward(self, x, relu_in=True, relu_out=False):
        """relu_in=False: the producer of x already applied this block's first (in-place, hence idempotent) ReLU; relu_out=True: the
        consumer is a Block whose first ReLU is in place — the reference then only ever sees relu(rep(x) + skip)
        (models/deeplabv3_plus.py:121-132, :210-220), so this block's last BatchNorm applies `+ skip` and that ReLU in its own pass
        (the fused apply(+residual)(+ReLU) kernel of the ResNet blocks): no stand-alone add / ReLU kernels and no stand-alone ReLU
        backward between the 20 Xception blocks (round 5).  Same arithmetic in the same order: bit-identical."""
        mods = list(self.rep)
        if self.use_1st_relu:
            if relu_in:
                x = ops.relu(x)           # the reference's in-place ReLU: the skip branch sees relu(x) too
            mods = mods[1:]
        skip = x if self.skip is None else self.skipbn(self.skip(x))
        if not Block.fused_tail:              # A/B and tests: the literal form (stand-alone add, then the consumer's ReLU here)
            out = ops.add(snn.run_fused(mods, x), skip)
            return ops.relu(out) if relu_out else out
        output = snn.run_fused(mods[:-1], x)
        return mods[-1](output, residual=skip, relu=relu_out)      # rep's last module is its BatchNorm2d


class Xception(nn.Module):
    def __init__(self, output_stride=16, in_channels=3, pretrained=True):
        super().__init__()
        if output_stride == 16:
            b3_s, mf_d, ef_d = 2, 1, (1, 2)
        elif output_stride == 8:
            b3_s, mf_d, ef_d = 1, 2, (2, 4)
        else:
            raise ValueError("output_stride must be 8 or 16")
        self.conv1 = snn.Conv2d(in_channels, 32, 3, 2, padding=1, bias=False)
        self.bn1 = snn.BatchNorm2d(32)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = snn.Conv2d(32, 64, 3, 1, padding=1, bias=False)
        self.bn2 = snn.BatchNorm2d(64)
        self.block1 = Block(64, 128, stride=2, dilation=1, use_1st_relu=False)
        self.block2 = Block(128, 256, stride=2, dilation=1)
        self.block3 = Block(256, 728, stride=b3_s, dilation=1)
        for i in range(16):
            setattr(self, "block%d" % (i + 4), Block(728, 728, stride=1, dilation=mf_d))
        self.block20 = Block(728, 1024, stride=1, dilation=ef_d[0], exit_flow=True)
        self.conv3 = SeparableConv2d(1024, 1536, 3, stride=1, dilation=ef_d[1])
        self.bn3 = snn.BatchNorm2d(1536)
        self.conv4 = SeparableConv2d(1536, 1536, 3, stride=1, dilation=ef_d[1])
        self.bn4 = snn.BatchNorm2d(1536)
        self.conv5 = SeparableConv2d(1536, 2048, 3, stride=1, dilation=ef_d[1])
        self.bn5 = snn.BatchNorm2d(2048)
        initialize_weights(self)
        if pretrained:
            _load_checkpoint(self, "xception")

    def forward(self, x):
        x = self.bn1(self.conv1(x), relu=True)
        x = self.bn2(self.conv2(x))                 # no ReLU here in the reference (:205-207)
        x = self.block1(x)
        low_level_features = x                      # (pre-ReLU, :211-213 of the reference)
        x = ops.relu(x)
        # every later block starts with an in-place ReLU on its input: the producing block's last BatchNorm applies it (Block.forward)
        x = self.block3(self.block2(x, relu_in=False, relu_out=True), relu_in=False, relu_out=True)
        for i in range(4, 20):
            x = getattr(self, "block%d" % i)(x, relu_in=False, relu_out=True)
        x = self.block20(x, relu_in=False, relu_out=True)           # + the F.relu that follows block20 (:223-224)
        # exit flow: bn3 / bn4 (+ReLU) feed the next SeparableConv2d only — applied inside its depthwise kernel (SeparableConv2d.forward_bn)
        x = self.conv3(x)
        for bn, conv in ((self.bn3, self.conv4), (self.bn4, self.conv5)):
            x = conv.forward_bn(x, bn, relu=True) if conv.bn_fusable(bn) else conv(bn(x, relu=True))
        x = self.bn5(x, relu=True)
        return x, low_level_features


# ----------------------------------------------------------------------------- ASPP + decoder
def assp_branch(in_channels, out_channles, kernel_size, dilation):
    padding = 0 if kernel_size == 1 else dilation
    return snn.Sequential(snn.Conv2d(in_channels, out_channles, kernel_size, padding=padding, dilation=dilation, bias=False),
                          snn.BatchNorm2d(out_channles), nn.ReLU(inplace=True))


class ASSP(nn.Module):
    def __init__(self, in_channels, output_stride):
        super().__init__()
        assert output_stride in [8, 16], "Only output strides of 8 or 16 are suported"
        dilations = [1, 6, 12, 18] if output_stride == 16 else [1, 12, 24, 36]
        self.aspp1 = assp_branch(in_channels, 256, 1, dilation=dilations[0])
        self.aspp2 = assp_branch(in_channels, 256, 3, dilation=dilations[1])
        self.aspp3 = assp_branch(in_channels, 256, 3, dilation=dilations[2])
        self.aspp4 = assp_branch(in_channels, 256, 3, dilation=dilations[3])
        self.avg_pool = snn.Sequential(snn.AdaptiveAvgPool2d((1, 1)), snn.Conv2d(in_channels, 256, 1, bias=False),
                                       snn.BatchNorm2d(256), nn.ReLU(inplace=True))
        self.conv1 = snn.Conv2d(256 * 5, 256, 1, bias=False)
        self.bn1 = snn.BatchNorm2d(256)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = snn.Dropout(0.5)
        initialize_weights(self)

    def forward(self, x):
        size = (x.size(2), x.size(3))
        x1, x2, x3, x4 = self.aspp1(x), self.aspp2(x), self.aspp3(x), self.aspp4(x)
        x5 = ops.interpolate_bilinear(self.avg_pool(x), size, align_corners=True)
        x = ops.cat([x1, x2, x3, x4, x5])
        return self.dropout(self.bn1(self.conv1(x), relu=True))


class Decoder(nn.Module):
    def __init__(self, low_level_channels, num_classes):
        super().__init__()
        self.conv1 = snn.Conv2d(low_level_channels, 48, 1, bias=False)
        self.bn1 = snn.BatchNorm2d(48)
        self.relu = nn.ReLU(inplace=True)
        self.output = snn.Sequential(
            snn.Conv2d(48 + 256, 256, 3, stride=1, padding=1, bias=False), snn.BatchNorm2d(256), nn.ReLU(inplace=True),
            snn.Conv2d(256, 256, 3, stride=1, padding=1, bias=False), snn.BatchNorm2d(256), nn.ReLU(inplace=True),
            snn.Dropout(0.1),
            snn.Conv2d(256, num_classes, 1, stride=1))
        initialize_weights(self)

    factored = os.environ.get("SEGMI_DECODER_FACTORED", "1") != "0"      # A/B switch; the literal form is the test reference

    def forward(self, x, low_level_features):
        low = self.bn1(self.conv1(low_level_features), relu=True)
        conv = self.output[0]
        if self.factored and low.shape[1] % 4 == 0 and x.shape[1] % 4 == 0 and conv.weight.is_contiguous(memory_format=torch.channels_last):
            # upsample(x) -> cat -> 3x3 convolution in factored form (segmi.ops.pyramid_bottleneck_conv): the
            # convolution proper runs over the 48 low-level channels; the 256 upsampled channels contribute through a 1x1 GEMM on
            # the low-resolution map and a separable interpolation — 84 % of this layer's MACs and the upsampled map disappear
            y = ops.pyramid_bottleneck_conv(low, [x], conv.weight)
            return snn.run_fused(list(self.output)[1:], y)
        x = ops.interpolate_bilinear(x, (low.size(2), low.size(3)), align_corners=True)
        return self.output(ops.cat([low, x]))


class DeepLab(BaseModel):
    def __init__(self, num_classes, in_channels=3, backbone="xception", pretrained=True, output_stride=16,
                 freeze_bn=False, freeze_backbone=False, **_):
        super().__init__()
        if "resnet" in backbone:
            self.backbone = ResNet(in_channels=in_channels, output_stride=output_stride, backbone=backbone, pretrained=pretrained)
            low_level_channels = 256 if backbone not in ("resnet18", "resnet34") else 64
        else:
            self.backbone = Xception(output_stride=output_stride, pretrained=pretrained)
            low_level_channels = 128
        self.ASSP = ASSP(in_channels=2048, output_stride=output_stride)
        self.decoder = Decoder(low_level_channels, num_classes)
        snn.link_conv_bn(self)        # conv -> BN pairs: BN statistics from the convolution's epilogue from the first step on
        if freeze_bn:
            self.freeze_bn()
        if freeze_backbone:
            set_trainable([self.backbone], False)

    def forward(self, x):
        size = (x.size(2), x.size(3))
        x, low_level_features = self.backbone(x)
        x = self.decoder(self.ASSP(x), low_level_features)
        return ops.interpolate_bilinear(x, size, align_corners=True)

    def get_backbone_params(self):
        return self.backbone.parameters()

    def get_decoder_params(self):
        return chain(self.ASSP.parameters(), self.decoder.parameters())

    def freeze_bn(self):
        for module in self.modules():
            if isinstance(module, nn.BatchNorm2d):
                module.eval()
