"""A SECOND, independently derived statement of the dilated-ResNet FCN backbone (TEST INFRASTRUCTURE).

``oracle/resnet_dilated_oracle.py`` restates the un-vendored backbone the reference instantiates at
``dense_correspondence/network/dense_correspondence_network.py:373-375`` the way the warmspringwinds fork builds it
(``_make_layer`` with a running stride / dilation budget, ``nn.Module`` blocks).  This file derives the same network the
other way round, so that a slip in either derivation shows up as a difference between the two:

1. ``stock_spec``: the STOCK, stride-32 ImageNet ResNet exactly as published (He et al. 2015, table 1; the layout every
   torchvision release ships): a flat, ordered list of convolution records -- nothing dilated, layers 2-4 enter with stride 2.
2. ``to_output_stride``: the FCN surgery as a pass over that list -- walk the convolutions in network order, keep striding
   until the accumulated stride reaches the target (8), from then on turn every stride s into "rate *= s" and give every
   3x3 convolution from THAT block on (the striding block's own convolutions included, as the fork does; modern
   torchvision's ``replace_stride_with_dilation`` would keep the previous rate in that first block) dilation = padding =
   the current rate; 1x1 convolutions are never dilated.  The average pool is dropped and ``fc`` becomes a 1x1
   convolution with bias.
3. ``run``: a purely functional interpreter of the record list on a plain ``{key: tensor}`` state dict (torch.nn.functional
   only, no ``nn.Module``), train-mode batch norm, bilinear resize with ``align_corners=True`` back to the input size.

Also here: ``torchvision_surgery`` builds the same network from ``torchvision.models`` when that package is importable
(it is not in the authoring image; the test skips).  Nothing in the product imports this file."""
import torch
import torch.nn.functional as F

ARCHS = {"Resnet18_8s": ("basic", (2, 2, 2, 2)), "Resnet34_8s": ("basic", (3, 4, 6, 3)),
         "Resnet50_8s": ("bottleneck", (3, 4, 6, 3)), "Resnet101_8s": ("bottleneck", (3, 4, 23, 3))}


def stock_spec(arch, base_width=64):
    """Ordered records of the stock network: dicts with name / cin / cout / k / stride / role
    (role: stem | conv | proj (1x1 shortcut projection)) and the block boundaries."""
    kind, counts = ARCHS[arch]
    exp = 4 if kind == "bottleneck" else 1
    net = {"stem": dict(name="conv1", bn="bn1", cin=3, cout=base_width, k=7, stride=2, pad=3), "blocks": [], "exp": exp}
    cin = base_width
    for stage, nblocks in enumerate(counts):
        width = base_width * 2 ** stage
        for b in range(nblocks):
            s = 2 if (b == 0 and stage > 0) else 1                      # stages 2-4 halve the resolution on entry
            pre = "layer%d.%d" % (stage + 1, b)
            if kind == "basic":
                convs = [dict(name=pre + ".conv1", bn=pre + ".bn1", cin=cin, cout=width, k=3, stride=s),
                         dict(name=pre + ".conv2", bn=pre + ".bn2", cin=width, cout=width, k=3, stride=1)]
            else:                                                        # (stride on the 3x3, as every torchvision release)
                convs = [dict(name=pre + ".conv1", bn=pre + ".bn1", cin=cin, cout=width, k=1, stride=1),
                         dict(name=pre + ".conv2", bn=pre + ".bn2", cin=width, cout=width, k=3, stride=s),
                         dict(name=pre + ".conv3", bn=pre + ".bn3", cin=width, cout=width * 4, k=1, stride=1)]
            proj = None
            if s != 1 or cin != width * exp:
                proj = dict(name=pre + ".downsample.0", bn=pre + ".downsample.1", cin=cin, cout=width * exp, k=1, stride=s)
            net["blocks"].append(dict(convs=convs, proj=proj))
            cin = width * exp
    net["features"] = cin
    return net


def to_output_stride(net, output_stride=8):
    """The surgery (in place): see the module docstring, step 2."""
    acc = 4                     # stem (2) x max pool (2)
    rate = 1
    for blk in net["blocks"]:
        s = max(c["stride"] for c in blk["convs"])
        if s > 1:
            if acc >= output_stride:
                rate *= s       # the stride is traded for dilation
                for c in blk["convs"] + ([blk["proj"]] if blk["proj"] else []):
                    c["stride"] = 1
            else:
                acc *= s
        for c in blk["convs"]:
            c["dil"] = rate if c["k"] == 3 else 1
            c["pad"] = c["dil"] if c["k"] == 3 else 0
        if blk["proj"]:
            blk["proj"]["dil"], blk["proj"]["pad"] = 1, 0
    return net


def expected_state_dict_layout(arch, num_classes, base_width=64, prefix=None):
    """[(key, shape)] in checkpoint order, derived from the record list (conv.weight, then the batch norm's weight, bias,
    running_mean, running_var, num_batches_tracked; the shortcut projection after the block's own layers; fc last)."""
    net = to_output_stride(stock_spec(arch, base_width))
    prefix = (arch.lower() + ".") if prefix is None else prefix
    out = []

    def conv_bn(c):
        out.append((prefix + c["name"] + ".weight", (c["cout"], c["cin"], c["k"], c["k"])))
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            out.append((prefix + c["bn"] + "." + leaf, (c["cout"],)))
        out.append((prefix + c["bn"] + ".num_batches_tracked", ()))
    conv_bn(net["stem"])
    for blk in net["blocks"]:
        for c in blk["convs"]:
            conv_bn(c)
        if blk["proj"]:
            conv_bn(blk["proj"])
    out.append((prefix + "fc.weight", (num_classes, net["features"], 1, 1)))
    out.append((prefix + "fc.bias", (num_classes,)))
    return out


def run(arch, sd, x, base_width=64, prefix=None, momentum=0.1, eps=1e-5, training=True):
    """Functional forward on the state dict ``sd`` (running statistics are updated IN ``sd`` like nn.BatchNorm2d does)."""
    net = to_output_stride(stock_spec(arch, base_width))
    prefix = (arch.lower() + ".") if prefix is None else prefix

    def conv(c, t):
        return F.conv2d(t, sd[prefix + c["name"] + ".weight"], None, c["stride"], c.get("pad", 0), c.get("dil", 1))

    def bn(c, t):
        k = prefix + c["bn"]
        if training:
            sd[k + ".num_batches_tracked"] += 1
        return F.batch_norm(t, sd[k + ".running_mean"], sd[k + ".running_var"], sd[k + ".weight"], sd[k + ".bias"],
                            training, momentum, eps)
    size = x.shape[2:]
    t = F.max_pool2d(F.relu(bn(net["stem"], conv(net["stem"], x))), 3, 2, 1)
    for blk in net["blocks"]:
        shortcut = t if blk["proj"] is None else bn(blk["proj"], conv(blk["proj"], t))
        u = t
        for i, c in enumerate(blk["convs"]):
            u = bn(c, conv(c, u))
            if i + 1 < len(blk["convs"]):
                u = F.relu(u)
        t = F.relu(u + shortcut)
    t = F.conv2d(t, sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])
    return F.interpolate(t, size=size, mode="bilinear", align_corners=True)


def torchvision_surgery(arch, num_classes):
    """The same network from a stock ``torchvision.models.resnetNN`` (raises ImportError without torchvision): strides of
    layer3 / layer4 -> 1 (first block and its projection), every 3x3 of layer3 dilated 2 and of layer4 dilated 4 with
    padding = dilation -- the first block too --, avgpool dropped, fc -> 1x1 convolution with bias."""
    import torch.nn as nn
    import torchvision
    tv = getattr(torchvision.models, arch.split("_")[0].lower())(weights=None)
    for layer, rate in ((tv.layer3, 2), (tv.layer4, 4)):
        for m in layer.modules():
            if isinstance(m, nn.Conv2d):
                m.stride = (1, 1)
                if m.kernel_size == (3, 3):
                    m.dilation, m.padding = (rate, rate), (rate, rate)
    fc = nn.Conv2d(tv.fc.in_features, num_classes, 1)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.tv, self.fc = tv, fc

        def forward(self, x):
            t = self.tv.maxpool(self.tv.relu(self.tv.bn1(self.tv.conv1(x))))
            t = self.tv.layer4(self.tv.layer3(self.tv.layer2(self.tv.layer1(t))))
            return F.interpolate(self.fc(t), size=x.shape[2:], mode="bilinear", align_corners=True)
    return Net()
