"""Oracle (TEST INFRASTRUCTURE, not product): plain-PyTorch CPU restatement of the
dilated-ResNet FCN backbone the reference instantiates at
``dense_correspondence/network/dense_correspondence_network.py:373-375``::

    fcn = getattr(resnet_dilated, "Resnet34_8s")(num_classes=D)

PARITY UNPINNED.  The arithmetic lives in a third-party dependency that is NOT vendored in
/root/reference: git submodule ``external/pytorch-segmentation-detection``
(warmspringwinds/pytorch-segmentation-detection, ``.gitmodules:1-3``; the pinned commit is not
recoverable from the mount) which itself uses that repo's forked torchvision
(``models.resnet34(fully_conv=True, pretrained=True, output_stride=8,
remove_avg_pool_layer=True)``), run under torch 1.1 / torchvision 0.3
(``docker/install_pytorch.sh:6-7``).  There are no golden vectors for it anywhere in the
reference, so this file restates the *published architecture*:

* ``conv1`` 7x7 / stride 2 / pad 3, no bias -> ``bn1`` -> ReLU -> maxpool 3x3 / 2 / pad 1
* ``layer1`` 3 x BasicBlock(64); ``layer2`` 4 x BasicBlock(128), first block stride 2 with a
  1x1/2 conv + BN downsample
* output-stride budget (8) is now spent, so every later ``stride=2`` is converted into
  ``dilation *= 2``: ``layer3`` 6 x BasicBlock(256) with dilation 2 *in every 3x3 conv of the
  layer, first block included*; ``layer4`` 3 x BasicBlock(512) with dilation 4; the downsample
  convs are 1x1 / stride 1 and are never dilated; 3x3 padding == dilation
* the average pool is removed; ``fc`` is replaced by ``Conv2d(512*expansion, D, 1)`` WITH bias,
  weight ~ N(0, 0.01), bias = 0
* output = ``upsample_bilinear`` to the input H x W  (== ``interpolate(mode="bilinear",
  align_corners=True)``)
* Bottleneck variants (ResNet-50/101) put the stride / dilation on the 3x3 ``conv2`` as every
  torchvision release does.

Initialisation: the fork's ``ResNet.__init__`` uses He-normal (fan-out) for convs and (1, 0) for
BN, and then loads ImageNet weights.  The ImageNet checkpoint cannot be downloaded here, so the
oracle (and the product, independently) use the seeded He-normal init only.

State-dict keys match the reference's checkpoints: the wrapper attribute is ``resnet34_8s`` etc.,
so under ``DenseCorrespondenceNetwork`` (attribute ``_fcn``, network.py:43) a key reads
``_fcn.resnet34_8s.layer1.0.conv1.weight``.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv3x3(cin, cout, stride=1, dilation=1):
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride, dilation)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=False)
        self.conv2 = _conv3x3(planes, planes, 1, dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv3x3(planes, planes, stride, dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class DilatedResNet(nn.Module):
    """Fully-convolutional ResNet trunk at a fixed output stride (no avg-pool), ``fc`` = 1x1 conv."""

    def __init__(self, block, layers, num_classes, output_stride=8, base_width=64):
        super().__init__()
        self.output_stride = output_stride
        self.current_stride = 4
        self.current_dilation = 1
        w = base_width
        self.inplanes = w
        self.conv1 = nn.Conv2d(3, w, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(w)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, w, layers[0])
        self.layer2 = self._make_layer(block, 2 * w, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 4 * w, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 8 * w, layers[3], stride=2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        # scoring layer, created last so the seeded RNG stream is: trunk convs in module order, then fc
        self.fc = nn.Conv2d(8 * w * block.expansion, num_classes, 1)
        self.fc.weight.data.normal_(0, 0.01)
        self.fc.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            if self.current_stride == self.output_stride:
                self.current_dilation *= stride
                stride = 1
            else:
                self.current_stride *= stride
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample, dilation=self.current_dilation)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=self.current_dilation))
        return nn.Sequential(*layers)

    use_checkpoint = False   # set_checkpointing(): recompute every block in backward (same arithmetic, ~8x less memory)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        if self.use_checkpoint and torch.is_grad_enabled():
            from torch.utils.checkpoint import checkpoint
            for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
                for blk in layer:
                    x = checkpoint(blk, x, use_reentrant=False, preserve_rng_state=False)
        else:
            x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(x)


_ARCH = {
    "Resnet18_8s": (BasicBlock, [2, 2, 2, 2], "resnet18_8s"),
    "Resnet34_8s": (BasicBlock, [3, 4, 6, 3], "resnet34_8s"),
    "Resnet50_8s": (Bottleneck, [3, 4, 6, 3], "resnet50_8s"),
    "Resnet101_8s": (Bottleneck, [3, 4, 23, 3], "resnet101_8s"),
}


class _Resnet8s(nn.Module):
    arch = None

    def __init__(self, num_classes=1000, base_width=64):
        super().__init__()
        block, layers, attr = _ARCH[self.arch]
        setattr(self, attr, DilatedResNet(block, layers, num_classes, 8, base_width))
        self._attr = attr

    def forward(self, x):
        size = x.shape[2:]
        x = getattr(self, self._attr)(x)
        return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


class Resnet18_8s(_Resnet8s):
    arch = "Resnet18_8s"


class Resnet34_8s(_Resnet8s):
    arch = "Resnet34_8s"


class Resnet50_8s(_Resnet8s):
    arch = "Resnet50_8s"


class Resnet101_8s(_Resnet8s):
    arch = "Resnet101_8s"


def set_checkpointing(model, on=True):
    """Per-block activation checkpointing for the full-size fixture runs that do not fit the authoring container's
    memory (tests/golden/make_backbone_goldens.py).  Outputs and gradients are unchanged; BN running statistics are
    updated once more by the recomputation."""
    getattr(model, model._attr).use_checkpoint = bool(on)


def build(name, num_classes, seed=0, base_width=64):
    """Seeded construction (weights use ``manual_seed(0)``)."""
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    m = globals()[name](num_classes=num_classes, base_width=base_width)
    torch.random.set_rng_state(g)
    return m
