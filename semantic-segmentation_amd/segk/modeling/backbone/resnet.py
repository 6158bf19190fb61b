"""ResNet-50/101 feeder with DeepLab output strides (plain PyTorch, not accelerated).

The feeder is outside the accelerated path: it runs as ordinary PyTorch-ROCm
modules and only hands (x, low_level_feat) to the head.  Its normalisation class
is injected (``BatchNorm`` argument) exactly as in the reference, so passing
``SynchronizedBatchNorm2d`` makes all 104 feeder norms use the HIP SyncBN.

Behaviour follows /root/reference/newmodeling/backbone/resnet101_2.py:7-156
(bottleneck :31-51, strides/dilations per output stride :61-68, multi-grid
layer4 with rates (1,2,4)·dilation :124-143, forward returning layer4 and
layer1 features :145-156, init N(0, sqrt(2/(k·k·out))) :220-230).  Attribute
names are kept so state_dict keys interchange with reference checkpoints.
There is no network in this environment: ``pretrained=True`` is refused.
"""
import math

import torch.nn as nn

_MULTI_GRID = (1, 2, 4)


def _bn_relu(bn, relu, x):
    """relu(bn(x)); batch-norm classes that offer a fused form (`forward_relu`) do it in one pass."""
    fused = getattr(bn, "forward_relu", None)
    return fused(x) if fused is not None else relu(bn(x))


def _maxpool(pool, x):
    """pool(x); the stem's 3 x 3 / stride 2 / padding 1 pooling of a channels_last GPU tensor runs on the HIP kernels
    (one byte of window position per element instead of ATen's 64-bit indices)."""
    if (isinstance(pool, nn.MaxPool2d) and pool.kernel_size in (3, (3, 3)) and pool.stride in (2, (2, 2))
            and pool.padding in (1, (1, 1)) and pool.dilation in (1, (1, 1)) and not pool.ceil_mode
            and not pool.return_indices and x.is_cuda and __package__):
        # (this file is also loaded by path, outside the package, as the feeder of the plain-PyTorch restatement the
        # tests compare against: there it stays on ATen)
        from ... import functional as SF
        y = SF.maxpool3x3s2(x)
        if y is not None:
            return y
    return pool(x)


def _conv(cin, cout, k, stride=1, dilation=1):
    pad = dilation if k == 3 else (3 if k == 7 else 0)
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, dilation=dilation, bias=False)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, BatchNorm=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = BatchNorm(planes)
        self.conv2 = _conv(planes, planes, 3, stride=stride, dilation=dilation)
        self.bn2 = BatchNorm(planes)
        self.conv3 = _conv(planes, planes * self.expansion, 1)
        self.bn3 = BatchNorm(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.relu2 = self.relu   # the ReLU behind bn2 under a name of its own (same module): parity tests swap it
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation

    def forward(self, x):
        # batch-norm classes that can run a convolution, their own statistics and the ReLU as one unit do so
        unit = getattr(self.bn1, "forward_conv_relu", None)
        slot = None
        # Links between units of which the second is the first one's only consumer (batch-norm classes that offer
        # `new_stat_link`): the second unit's data-gradient convolution emits the backward statistics of the first
        # one's batch norm.  link1: conv1's unit -> conv2's; link2: conv2's -> conv3's.
        links = hasattr(self.bn1, "new_stat_link") and x.requires_grad
        link1 = link2 = None
        if unit is not None and self.bn1.can_fuse_conv(self.conv1, x):
            # x feeds conv1 and (without a projection) the shortcut: the unit behind bn3 may leave the shortcut's
            # gradient in `slot`, and conv1's data-gradient convolution adds it (one gradient for x, no separate sum)
            # (with a projection it is the projection's input gradient that goes through the slot)
            if x.requires_grad and hasattr(self.bn1, "new_grad_slot") and (
                    self.downsample is None or hasattr(self.bn1, "gradient_to_slot")):
                slot = self.bn1.new_grad_slot()
            if links:
                link1 = self.bn1.new_stat_link()
                y = unit(self.conv1, x, slot, None, link1)
            else:
                y = unit(self.conv1, x, slot) if slot is not None else unit(self.conv1, x)
        else:
            y = _bn_relu(self.bn1, self.relu, self.conv1(x))
        unit = getattr(self.bn2, "forward_conv_relu", None)
        if unit is not None and self.bn2.can_fuse_conv(self.conv2, y):
            if links:
                link2 = self.bn2.new_stat_link()
                y = unit(self.conv2, y, None, link1, link2)
            else:
                y = unit(self.conv2, y)
        else:
            y = _bn_relu(self.bn2, self.relu2, self.conv2(y))
        unit = getattr(self.bn3, "forward_conv_add_relu", None)
        # the slot takes the gradient of the shortcut's INPUT: the residual's own gradient without a projection, the
        # projection's input gradient with one (the unit behind bn3 then returns the residual's gradient to autograd)
        res_slot = slot if self.downsample is None else None
        if unit is not None:
            # (x receives two gradients, from this path and from conv1: their sum does not depend on the order)
            if self.downsample is None:
                res = x
            else:
                res = self.downsample(self.bn1.gradient_to_slot(x, slot) if slot is not None else x)
            if self.bn3.can_fuse_conv_residual(self.conv3, y, res):
                if links:
                    return unit(self.conv3, y, res, res_slot, link2)
                return unit(self.conv3, y, res, res_slot) if res_slot is not None else unit(self.conv3, y, res)
            y = self.conv3(y)
        else:
            y = self.conv3(y)
            # the projection runs after the main path: autograd then sums the gradients that
            # reach x in the same order as the reference does, which keeps training bit-identical
            res = x if self.downsample is None else self.downsample(x)
        fused = getattr(self.bn3, "forward_add_relu", None)
        if fused is not None and self.bn3.can_fuse_residual(y, res):
            return fused(y, res, res_slot) if res_slot is not None else fused(y, res)
        y = self.bn3(y)
        y += res
        return self.relu(y)


class ResNet(nn.Module):
    def __init__(self, block, layers, output_stride, BatchNorm, pretrained=False):
        super().__init__()
        if pretrained:
            raise RuntimeError("pretrained feeder weights need a download; load a checkpoint explicitly instead")
        if output_stride == 16:
            strides, dilations = (1, 2, 2, 1), (1, 1, 1, 2)
        elif output_stride == 8:
            strides, dilations = (1, 2, 1, 1), (1, 1, 2, 4)
        else:
            raise NotImplementedError
        self.inplanes = 64
        self.conv1 = _conv(3, 64, 7, stride=2)
        self.bn1 = BatchNorm(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        widths = (64, 128, 256)
        for i, w in enumerate(widths):
            rates = [dilations[i]] * layers[i]
            setattr(self, "layer%d" % (i + 1), self._stage(block, w, rates, strides[i], BatchNorm))
        rates = [g * dilations[3] for g in _MULTI_GRID]
        self.layer4 = self._stage(block, 512, rates, strides[3], BatchNorm)
        self._init_weight()

    def _stage(self, block, planes, rates, stride, BatchNorm):
        out_ch = planes * block.expansion
        proj = None
        if stride != 1 or self.inplanes != out_ch:
            proj = nn.Sequential(
                nn.Conv2d(self.inplanes, out_ch, kernel_size=1, stride=stride, bias=False),
                BatchNorm(out_ch))
        blocks = [block(self.inplanes, planes, stride, rates[0], proj, BatchNorm)]
        self.inplanes = out_ch
        for r in rates[1:]:
            blocks.append(block(self.inplanes, planes, 1, r, None, BatchNorm))
        return nn.Sequential(*blocks)

    def forward(self, input):
        x = _maxpool(self.maxpool, _bn_relu(self.bn1, self.relu, self.conv1(input)))
        low_level_feat = self.layer1(x)
        x = self.layer4(self.layer3(self.layer2(low_level_feat)))
        return x, low_level_feat

    def _init_weight(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.modules.batchnorm._BatchNorm):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


def ResNet101(output_stride, BatchNorm, pretrained=False):
    return ResNet(Bottleneck, [3, 4, 23, 3], output_stride, BatchNorm, pretrained=pretrained)


def ResNet50(output_stride, BatchNorm, pretrained=False):
    return ResNet(Bottleneck, [3, 4, 6, 3], output_stride, BatchNorm, pretrained=pretrained)
