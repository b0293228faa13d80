"""Functional torch-CPU restatement of the reference U-Net forward (test infrastructure).

Follows models/unet.py:93-106 (UNet.forward), :12-21 (x2conv), :23-32 (encoder: x2conv + MaxPool2d(2, ceil_mode=True)),
:34-58 (decoder: ConvTranspose2d(k=2,s=2), bilinear(align_corners=True) when the skip is larger, cat([skip, up]), x2conv).
Weights are looked up by the reference's state_dict key names.
"""
import torch
import torch.nn.functional as F


def _bn(sd, key, x, training, momentum=0.1, eps=1e-5):
    if training and (key + ".num_batches_tracked") in sd:
        sd[key + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                        training, momentum, eps)


def _x2conv(sd, pre, x, bnt):
    x = F.relu(_bn(sd, pre + ".1", F.conv2d(x, sd[pre + ".0.weight"], None, 1, 1), bnt))
    return F.relu(_bn(sd, pre + ".4", F.conv2d(x, sd[pre + ".3.weight"], None, 1, 1), bnt))


def _down(sd, pre, x, bnt):
    return F.max_pool2d(_x2conv(sd, pre + ".down_conv", x, bnt), 2, ceil_mode=True)


def _up(sd, pre, skip, x, bnt):
    x = F.conv_transpose2d(x, sd[pre + ".up.weight"], sd[pre + ".up.bias"], stride=2)
    if x.shape[2:] != skip.shape[2:]:
        x = F.interpolate(x, size=skip.shape[2:], mode="bilinear", align_corners=True)
    return _x2conv(sd, pre + ".up_conv", torch.cat([skip, x], dim=1), bnt)


def unet_forward(sd, x, training=True, bn_training=None):
    bnt = training if bn_training is None else bn_training
    x1 = _x2conv(sd, "start_conv", x, bnt)
    x2 = _down(sd, "down1", x1, bnt)
    x3 = _down(sd, "down2", x2, bnt)
    x4 = _down(sd, "down3", x3, bnt)
    y = _x2conv(sd, "middle_conv", _down(sd, "down4", x4, bnt), bnt)
    y = _up(sd, "up1", x4, y, bnt)
    y = _up(sd, "up2", x3, y, bnt)
    y = _up(sd, "up3", x2, y, bnt)
    y = _up(sd, "up4", x1, y, bnt)
    return F.conv2d(y, sd["final_conv.weight"], sd["final_conv.bias"])
