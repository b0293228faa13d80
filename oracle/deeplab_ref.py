"""Functional torch-CPU restatement of the reference DeepLabV3+ forward (TEST INFRASTRUCTURE).

Follows models/deeplabv3_plus.py: DeepLab.forward :356-362, ResNet wrapper :15-63 (torchvision ResNet-v1.5 re-strided by
module name: output_stride 16 -> layer3 stride 2, layer4 conv2 dilation 2 / stride 1; output_stride 8 -> layer3 conv2
dilation 2, layer4 conv2 dilation 4, both stride 1), Xception :134-247 with Block :89-132 and SeparableConv2d :70-86,
ASSP :253-297, Decoder :303-330.  Weights are looked up by the reference's state_dict key names.

Reference behaviours restated (not "fixed"): a Block whose `rep` starts with the in-place ReLU feeds relu(x) to its
skip / identity branch as well; there is no ReLU between bn2 and block1; the exit-flow block is in->in, in->out, out->out.
Dropout layers are the identity (parity runs neutralise dropout).
"""
import torch
import torch.nn.functional as F

BLOCKS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3), "resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3),
          "resnet152": (3, 8, 36, 3)}


def _bn(sd, key, x, training, momentum=0.1, eps=1e-5):
    if training and (key + ".num_batches_tracked") in sd:
        sd[key + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                        training, momentum, eps)


def _conv(sd, key, x, stride=1, pad=0, dil=1, groups=1):
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride, pad, dil, groups)


# ----------------------------------------------------------------------------- ResNet (bottleneck variants)
def _bottleneck(sd, pre, x, stride, dil, bnt):
    out = F.relu(_bn(sd, pre + ".bn1", _conv(sd, pre + ".conv1", x), bnt))
    out = F.relu(_bn(sd, pre + ".bn2", _conv(sd, pre + ".conv2", out, stride, dil, dil), bnt))
    out = _bn(sd, pre + ".bn3", _conv(sd, pre + ".conv3", out), bnt)
    if (pre + ".downsample.0.weight") in sd:
        x = _bn(sd, pre + ".downsample.1", _conv(sd, pre + ".downsample.0", x, stride), bnt)
    return F.relu(out + x)


def _layer(sd, name, x, blocks, stride, dil, bnt):
    x = _bottleneck(sd, "%s.0" % name, x, stride, dil, bnt)
    for i in range(1, blocks):
        x = _bottleneck(sd, "%s.%d" % (name, i), x, 1, dil, bnt)   # re-striding sets (s, s) on every conv2; blocks > 0 only matter for s == 1
    return x


def resnet_features(sd, x, backbone, output_stride, bnt, pre="backbone."):
    nb = BLOCKS[backbone]
    s3, s4, d3, d4 = (2, 1, 1, 2) if output_stride == 16 else (1, 1, 2, 4)
    y = F.relu(_bn(sd, pre + "layer0.1", _conv(sd, pre + "layer0.0", x, 2, 3), bnt))
    y = F.max_pool2d(y, 3, 2, 1)
    y = _layer(sd, pre + "layer1", y, nb[0], 1, 1, bnt)
    low = y
    y = _layer(sd, pre + "layer2", y, nb[1], 2, 1, bnt)
    if output_stride == 16:
        y = _layer(sd, pre + "layer3", y, nb[2], 2, 1, bnt)            # untouched torchvision layer3
    else:
        y = _layer_strided_all(sd, pre + "layer3", y, nb[2], s3, d3, bnt)
    y = _layer_strided_all(sd, pre + "layer4", y, nb[3], s4, d4, bnt)
    return y, low


def _layer_strided_all(sd, name, x, blocks, s, d, bnt):
    """A re-strided layer: EVERY block's conv2 gets stride (s, s), dilation d, padding d (:39-53)."""
    for i in range(blocks):
        x = _bottleneck(sd, "%s.%d" % (name, i), x, s, d, bnt)
    return x


# ----------------------------------------------------------------------------- Xception
def _sep(sd, pre, x, stride, dil, bnt):
    C = x.shape[1]
    pad = dil if dil > 1 else 1
    x = F.conv2d(x, sd[pre + ".conv1.weight"], None, stride, pad, dil, groups=C)
    x = _bn(sd, pre + ".bn", x, bnt)
    return F.conv2d(x, sd[pre + ".pointwise.weight"])


def _block(sd, pre, x, stride, dil, bnt, first_relu=True):
    base = 0
    if first_relu:
        x = F.relu(x)          # in-place in the reference: the skip branch sees it too
        base = 1
    y = x
    strides = (1, 1, stride)
    for j in range(3):
        if j > 0:
            y = F.relu(y)
        i = base + 3 * j
        y = _bn(sd, "%s.rep.%d" % (pre, i + 1), _sep(sd, "%s.rep.%d" % (pre, i), y, strides[j], dil, bnt), bnt)
    if (pre + ".skip.weight") in sd:
        skip = _bn(sd, pre + ".skipbn", F.conv2d(x, sd[pre + ".skip.weight"], None, stride), bnt)
    else:
        skip = x
    return y + skip


def xception_features(sd, x, output_stride, bnt, pre="backbone."):
    b3_s, mf_d, ef_d = (2, 1, (1, 2)) if output_stride == 16 else (1, 2, (2, 4))
    y = F.relu(_bn(sd, pre + "bn1", _conv(sd, pre + "conv1", x, 2, 1), bnt))
    y = _bn(sd, pre + "bn2", _conv(sd, pre + "conv2", y, 1, 1), bnt)
    y = _block(sd, pre + "block1", y, 2, 1, bnt, first_relu=False)
    low = y
    y = F.relu(y)
    y = _block(sd, pre + "block2", y, 2, 1, bnt)
    y = _block(sd, pre + "block3", y, b3_s, 1, bnt)
    for i in range(4, 20):
        y = _block(sd, pre + "block%d" % i, y, 1, mf_d, bnt)
    y = F.relu(_block(sd, pre + "block20", y, 1, ef_d[0], bnt))
    y = F.relu(_bn(sd, pre + "bn3", _sep(sd, pre + "conv3", y, 1, ef_d[1], bnt), bnt))
    y = F.relu(_bn(sd, pre + "bn4", _sep(sd, pre + "conv4", y, 1, ef_d[1], bnt), bnt))
    y = F.relu(_bn(sd, pre + "bn5", _sep(sd, pre + "conv5", y, 1, ef_d[1], bnt), bnt))
    return y, low


# ----------------------------------------------------------------------------- ASPP, decoder, model
def _aspp(sd, x, output_stride, bnt):
    d = [1, 6, 12, 18] if output_stride == 16 else [1, 12, 24, 36]
    outs = [F.relu(_bn(sd, "ASSP.aspp1.1", _conv(sd, "ASSP.aspp1.0", x), bnt))]
    for i in (2, 3, 4):
        outs.append(F.relu(_bn(sd, "ASSP.aspp%d.1" % i, _conv(sd, "ASSP.aspp%d.0" % i, x, 1, d[i - 1], d[i - 1]), bnt)))
    p = F.relu(_bn(sd, "ASSP.avg_pool.2", _conv(sd, "ASSP.avg_pool.1", F.adaptive_avg_pool2d(x, 1)), bnt))
    outs.append(F.interpolate(p, size=x.shape[2:], mode="bilinear", align_corners=True))
    return F.relu(_bn(sd, "ASSP.bn1", _conv(sd, "ASSP.conv1", torch.cat(outs, dim=1)), bnt))


def _decoder(sd, x, low, bnt):
    low = F.relu(_bn(sd, "decoder.bn1", _conv(sd, "decoder.conv1", low), bnt))
    x = F.interpolate(x, size=low.shape[2:], mode="bilinear", align_corners=True)
    y = torch.cat([low, x], dim=1)
    y = F.relu(_bn(sd, "decoder.output.1", _conv(sd, "decoder.output.0", y, 1, 1), bnt))
    y = F.relu(_bn(sd, "decoder.output.4", _conv(sd, "decoder.output.3", y, 1, 1), bnt))
    return _conv(sd, "decoder.output.7", y)


def deeplab_forward(sd, x, backbone="xception", output_stride=16, training=True, bn_training=None):
    bnt = training if bn_training is None else bn_training
    if "resnet" in backbone:
        feat, low = resnet_features(sd, x, backbone, output_stride, bnt)
    else:
        feat, low = xception_features(sd, x, output_stride, bnt)
    y = _decoder(sd, _aspp(sd, feat, output_stride, bnt), low, bnt)
    return F.interpolate(y, size=x.shape[2:], mode="bilinear", align_corners=True)
