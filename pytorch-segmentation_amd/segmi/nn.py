"""nn.Module building blocks whose forward/backward run on libsegmi kernels.

Each class subclasses the torch module of the same name, so `state_dict()` keys, constructor
arguments and `isinstance(m, nn.BatchNorm2d)`-style checks of the reference
(utils/helpers.py:12-22 `initialize_weights`, models/*.py `freeze_bn`,
utils/sync_batchnorm/batchnorm.py:353-394 `convert_model`) keep working unchanged; only `forward`
is replaced.  `Sequential` fuses Conv->BN->ReLU chains into the fused kernels while keeping the
child indices (and therefore the checkpoint key names) of the reference's nn.Sequential containers.
"""
import torch
import torch.nn as nn

from . import ops


class Conv2d(nn.Conv2d):
    """nn.Conv2d on libsegmi: dense (groups=1) -> implicit-GEMM MFMA kernels with the filter held KRSC
    in memory (torch channels_last, no per-step re-layout); depthwise (groups == in == out channels, no
    bias) -> the streaming depthwise kernels."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.depthwise = self.groups != 1
        if self.depthwise and not (self.groups == self.in_channels == self.out_channels and self.bias is None):
            raise ValueError("segmi.nn.Conv2d supports groups=1 or bias-free depthwise (groups == in == out channels)")
        if self.padding_mode != "zeros":
            raise ValueError("segmi.nn.Conv2d supports zero padding only")
        for name in ("stride", "padding", "dilation"):
            v = getattr(self, name)
            if v[0] != v[1]:
                raise ValueError("segmi.nn.Conv2d needs symmetric %s, got %s" % (name, (v,)))
        if not self.depthwise and (self.kernel_size[0] > 1 or self.kernel_size[1] > 1):
            self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
        if self.depthwise:
            # the depthwise kernels read the filter tap-major [R, S, C]: the parameter keeps its logical shape [C, 1, R, S] (state_dict
            # keys, shapes and values unchanged) over memory in that order, so no per-step re-layout launch is needed — 63 forward +
            # 63 gradient transposes per DeepLab-Xception step before (ops._dw_rsc_view)
            self.weight.data = ops.dw_filter_rsc_layout(self.weight.data)
        # set by the BatchNorm2d that receives this module's output (ops._note_bn_consumer): True while a batch-statistics BN
        # consumes it — the convolution then emits the BN statistics partials from its epilogue (ops._BN_FUSE)
        self._bn_consumer = False

    def forward(self, x, with_skip=False):
        fuse = self._bn_consumer and torch.is_grad_enabled()
        if self.depthwise:
            return ops.depthwise_conv2d(x, self.weight, self.stride[0], self.padding[0], self.dilation[0], bn_stats=fuse, producer=self)
        if with_skip:   # (conv(x), x): residual fork whose backward accumulates dgrad onto the skip gradient (ops.conv2d_skip)
            if self.bias is not None:
                raise ValueError("with_skip is for the bias-free first convolution of a residual block")
            return ops.conv2d_skip(x, self.weight, self.stride[0], self.padding[0], self.dilation[0], bn_stats=fuse, producer=self)
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.dilation[0], bn_stats=fuse, producer=self)


class ConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(kernel_size=2, stride=2) — the U-Net up-convolution (models/unet.py:37)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.kernel_size != (2, 2) or self.stride != (2, 2) or self.padding != (0, 0) or self.output_padding != (0, 0) \
                or self.groups != 1 or self.dilation != (1, 1):
            raise ValueError("segmi.nn.ConvTranspose2d implements kernel_size=2, stride=2, padding=0 only")

    def forward(self, x, output_size=None):
        return ops.conv_transpose2x2(x, self.weight, self.bias)


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d with optional fused residual add and ReLU (one apply pass)."""

    sync = None  # set by utils.sync_batchnorm.SynchronizedBatchNorm2d

    def forward(self, x, residual=None, relu=False):
        if self.momentum is None:
            raise NotImplementedError("cumulative moving average (momentum=None) is not supported")
        use_batch_stats = self.training or not self.track_running_stats
        return ops.batch_norm_act(
            x, self.weight, self.bias,
            self.running_mean if self.track_running_stats else None,
            self.running_var if self.track_running_stats else None,
            self.num_batches_tracked if (self.track_running_stats and self.training) else None,
            residual=residual, training=use_batch_stats, momentum=self.momentum, eps=self.eps, relu=relu,
            sync=self.sync if use_batch_stats else None)


class ReLU(nn.ReLU):
    def forward(self, x):
        return ops.relu(x)


class MaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        k = self.kernel_size if isinstance(self.kernel_size, int) else self.kernel_size[0]
        s = self.stride if isinstance(self.stride, int) else self.stride[0]
        p = self.padding if isinstance(self.padding, int) else self.padding[0]
        return ops.max_pool2d(x, k, s, p, self.ceil_mode)


class AdaptiveAvgPool2d(nn.AdaptiveAvgPool2d):
    def forward(self, x):
        return ops.adaptive_avg_pool2d(x, self.output_size)


class Dropout2d(nn.Dropout2d):
    def forward(self, x):
        return ops.dropout(x, self.p, self.training, channelwise=True)


class Dropout(nn.Dropout):
    def forward(self, x):
        return ops.dropout(x, self.p, self.training, channelwise=False)


def conv_fan(x, convs):
    """[conv(x) for conv in convs] for bias-free dense Conv2d modules sharing their input, as ONE autograd node whose backward sums
    the branches' data gradients inside the dgrad kernels (ops.conv2d_fan) — conv1 and the projection shortcut of a residual
    block.  Falls back to separate calls when a branch does not qualify."""
    ok = all(isinstance(c, Conv2d) and not c.depthwise and c.bias is None for c in convs) and convs[0].stride[0] == 1
    if not ok or not FAN_IN_FUSION:
        return [c(x) for c in convs]
    fuse = torch.is_grad_enabled()
    return ops.conv2d_fan(x, [(c.weight, c.stride[0], c.padding[0], c.dilation[0], c._bn_consumer and fuse, c) for c in convs])


FAN_IN_FUSION = __import__("os").environ.get("SEGMI_CONV_FAN", "1") == "1"      # A/B switch


def link_conv_bn(root):
    """Mark every dense Conv2d whose output a BatchNorm2d consumes (`_bn_consumer`), from the module REGISTRATION order: inside
    one parent, a BatchNorm2d child pairs with the last convolution registered before it — `conv1, bn1, conv2, bn2, ...` of the
    residual blocks (models/resnet.py:80-99), `Sequential(conv, bn, relu, ...)` (models/unet.py:12-21, models/pspnet.py:17-30), a
    container whose LAST leaf is a convolution followed by a BatchNorm2d of the parent (the deep-base stem + bn1,
    SeparableConv2d.pointwise + the Block's BatchNorm2d, models/deeplabv3_plus.py:70-132).  A marked convolution emits the BN
    statistics partials from its epilogue from the FIRST training step on (ops._BN_FUSE); a wrong mark only costs an unused
    epilogue (the BN layer takes partials only from the very tensor it receives), a missed pair is found at run time
    (ops._note_bn_consumer) from the second step on.  Returns the number of marked convolutions."""
    marked = 0

    def last_leaf(m):
        kids = list(m.children())
        return last_leaf(kids[-1]) if kids else m

    def walk(parent):
        nonlocal marked
        last = None
        for child in parent.children():
            if isinstance(child, BatchNorm2d):
                if isinstance(last, Conv2d):
                    if not last._bn_consumer:
                        marked += 1
                    last._bn_consumer = True
                last = None
                continue
            if isinstance(child, Conv2d):
                last = child
                continue
            if any(True for _ in child.children()):
                walk(child)
                leaf = last_leaf(child)
                last = leaf if isinstance(leaf, Conv2d) else None
    walk(root)
    return marked


def sync_tail(bn, downsample):
    """True when `bn` and the BN of a projection shortcut `downsample` = Sequential(conv, BN) are SyncBN layers that can share
    their collectives (ops.sync_batch_norm_residual_tail): the block then evaluates relu(bn(.) + BN(conv(x))) as one node."""
    return (downsample is not None and len(downsample) == 2 and isinstance(downsample[1], BatchNorm2d) and isinstance(bn, BatchNorm2d)
            and ops.sync_groupable([bn, downsample[1]]))


def residual_out(x_last, bn, downsample, x, identity, proj=None):
    """Last step of a residual block: relu(bn(x_last) + identity), or — identity is None: see sync_tail — the fused SyncBN tail
    (proj: the projection convolution's output when the block already computed it, see conv_fan)."""
    if identity is None:
        return ops.sync_batch_norm_residual_tail(x_last, bn, proj if proj is not None else downsample[0](x), downsample[1])
    return bn(x_last, residual=identity, relu=True)


def run_fused(modules, x):
    """Run a module chain, fusing BatchNorm2d -> ReLU pairs into one apply pass — or, when the consumer can take the
    PRE-normalisation tensor (a module with `forward_bn`, e.g. SeparableConv2d: its depthwise kernel applies the BatchNorm + ReLU
    on the loaded taps, ops.batch_norm_depthwise), into the consumer itself: the normalised tensor is then never materialised."""
    mods = list(modules)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, BatchNorm2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
            nxt = mods[i + 2] if i + 2 < len(mods) else None
            if nxt is not None and hasattr(nxt, "forward_bn") and nxt.bn_fusable(m):
                x = nxt.forward_bn(x, m, relu=True)
                i += 3
                continue
            x = m(x, relu=True)
            i += 2
            continue
        x = m(x)
        i += 1
    return x


class Sequential(nn.Sequential):
    """nn.Sequential with the same child indices, executing fused BN+ReLU."""

    def forward(self, x):
        return run_fused(self, x)


def interpolate(x, size, mode="bilinear", align_corners=None):
    if mode != "bilinear":
        raise NotImplementedError("only bilinear interpolation is on the hot path")
    return ops.interpolate_bilinear(x, size, bool(align_corners))
