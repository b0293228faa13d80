"""Functional torch-CPU restatement of the reference PSPNet forward (test infrastructure).

Follows models/pspnet.py:77-94 (PSPNet.forward), :32-38 (_PSPModule.forward) and models/resnet.py
(:136-151 deep-base stem + maxpool, :101-121 Bottleneck.forward, :154-163,180-210 stage layout:
layer3 dilation 2 / layer4 dilation 4, stride 1, first block of a dilated stage at half dilation).
Weights are looked up by the reference's state_dict key names, so a state_dict taken from the real
reference model runs here unchanged.  Dropout2d is the identity (parity runs neutralise dropout,
SURVEY.md §7).  BN uses F.batch_norm exactly like nn.BatchNorm2d (running stats updated in place).
"""
import torch
import torch.nn.functional as F

STAGES = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}


def _conv(sd, key, x, stride=1, pad=0, dil=1):
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride, pad, dil)


def _bn(sd, key, x, training, momentum=0.1, eps=1e-5):
    if training:
        nbt = sd.get(key + ".num_batches_tracked")
        if nbt is not None:
            nbt += 1
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                        training, momentum, eps)


def _bottleneck(sd, pre, x, stride, dil, training):
    out = F.relu(_bn(sd, pre + ".bn1", _conv(sd, pre + ".conv1", x), training))
    out = F.relu(_bn(sd, pre + ".bn2", _conv(sd, pre + ".conv2", out, stride, dil, dil), training))
    out = _bn(sd, pre + ".bn3", _conv(sd, pre + ".conv3", out), training)
    if (pre + ".downsample.0.weight") in sd:
        x = _bn(sd, pre + ".downsample.1", _conv(sd, pre + ".downsample.0", x, stride), training)
    return F.relu(out + x)


def _stage(sd, name, x, blocks, stride, dilation, training, checkpoint=False):
    first = 1 if dilation in (1, 2) else 2

    def block(pre, x, stride, dil):
        if not checkpoint:
            return _bottleneck(sd, pre, x, stride, dil, training)
        # recompute the block's interior in backward (same kernels on the same operands => the same bits); only the block inputs
        # stay alive.  The recomputation applies the block's running-statistics update a second time: callers that read the
        # running statistics must not ask for checkpointing (the fp64 gradient pass of gen_golden_fullsize.py does not).
        from torch.utils.checkpoint import checkpoint as ckpt
        return ckpt(lambda t: _bottleneck(sd, pre, t, stride, dil, training), x, use_reentrant=False)

    x = block("%s.0" % name, x, stride, first)
    for i in range(1, blocks):
        x = block("%s.%d" % (name, i), x, 1, dilation)
    return x


def pspnet_forward(sd, x, training=True, backbone="resnet50", use_aux=True, bins=(1, 2, 3, 6), bn_training=None, checkpoint=False):
    """Returns (output, aux) when `training and use_aux`, else output — as models/pspnet.py:89-94.
    `bn_training` overrides the BN mode (freeze_bn() => False while the rest trains).  `checkpoint` recomputes every bottleneck
    in backward (memory of the fp64 pass at 8 x 769^2; gradients are bit-identical to the plain pass)."""
    bnt = training if bn_training is None else bn_training
    blocks = STAGES[backbone]
    H, W = x.shape[2], x.shape[3]
    # deep-base stem: initial.0 = Sequential(conv, bn, relu, conv, bn, relu, conv); initial.1 = bn1
    y = F.relu(_bn(sd, "initial.0.1", _conv(sd, "initial.0.0", x, 2, 1), bnt))
    y = F.relu(_bn(sd, "initial.0.4", _conv(sd, "initial.0.3", y, 1, 1), bnt))
    y = F.relu(_bn(sd, "initial.1", _conv(sd, "initial.0.6", y, 1, 1), bnt))
    y = F.max_pool2d(y, 3, 2, 1)
    y = _stage(sd, "layer1", y, blocks[0], 1, 1, bnt, checkpoint)
    y = _stage(sd, "layer2", y, blocks[1], 2, 1, bnt, checkpoint)
    y_aux = _stage(sd, "layer3", y, blocks[2], 1, 2, bnt, checkpoint)
    y = _stage(sd, "layer4", y_aux, blocks[3], 1, 4, bnt, checkpoint)

    h, w = y.shape[2], y.shape[3]
    pyramid = [y]
    for i, b in enumerate(bins):
        p = "master_branch.0.stages.%d" % i
        s = F.relu(_bn(sd, p + ".2", _conv(sd, p + ".1", F.adaptive_avg_pool2d(y, b)), bnt))
        pyramid.append(F.interpolate(s, size=(h, w), mode="bilinear", align_corners=True))
    z = torch.cat(pyramid, dim=1)
    z = F.relu(_bn(sd, "master_branch.0.bottleneck.1", _conv(sd, "master_branch.0.bottleneck.0", z, 1, 1), bnt))
    out = _conv(sd, "master_branch.1", z)
    out = F.interpolate(out, size=(H, W), mode="bilinear", align_corners=False)
    if training and use_aux:
        a = F.relu(_bn(sd, "auxiliary_branch.1", _conv(sd, "auxiliary_branch.0", y_aux, 1, 1), bnt))
        aux = F.interpolate(_conv(sd, "auxiliary_branch.4", a), size=(H, W), mode="bilinear", align_corners=False)
        return out, aux
    return out


def clone_state(sd, requires_grad=True):
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if requires_grad and t.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            t.requires_grad_(True)
        out[k] = t
    return out
