"""``resnet_dilated.Resnet{18,34,50,101}_8s`` -- the names the reference resolves with
``getattr(resnet_dilated, config["backbone"]["resnet_name"])(num_classes=D)``
(dense_correspondence/network/dense_correspondence_network.py:373-375).

Each class is an ``nn.Module`` whose parameters / buffers carry the reference checkpoints' names
(``resnet34_8s.layer1.0.conv1.weight`` ...), so ``optim.Adam(dcn.parameters())``, ``state_dict()``,
``.cuda()``, ``.train()/.eval()`` behave as with the original (training.py:144,245,510).  ``forward`` does not
run any torch operator on the activations: it hands the parameter pointers to the MI355X backbone
engine (dcn_hip / csrc/backbone_engine.hip).  There is no CPU path."""
import math
import warnings

import torch
import torch.nn as nn

from dcn_hip import backbone as _bb


class _Node(nn.Module):
    """Parameter container (never executed on its own)."""

    def forward(self, *a, **k):
        raise RuntimeError("this sub-module only holds parameters; call the top-level Resnet*_8s module "
                           "(the whole network runs as one fused MI355X engine call)")


def _child(parent, name):
    if name not in parent._modules:
        parent.add_module(name, _Node())
    return parent._modules[name]


class _DilatedResnet8s(nn.Module):
    arch = None
    attr = None

    def __init__(self, num_classes=1000, base_width=64):
        super(_DilatedResnet8s, self).__init__()
        self.num_classes = int(num_classes)
        self.base_width = int(base_width)
        probe = _bb.get_plan(self.arch, self.base_width, 1, 32, 32, self.num_classes)  # names/shapes only
        self._param_names = list(probe.param_names)
        self._bn_names = list(probe.bn_names)
        trunk = _Node()
        setattr(self, self.attr, trunk)
        for name, shape in zip(probe.param_names, probe.param_shapes):
            parts = name.split(".")
            node = trunk
            for p in parts[:-1]:
                node = _child(node, p)
            t = torch.empty(shape, dtype=torch.float32)
            if len(shape) == 4:
                t = t.contiguous(memory_format=torch.channels_last)  # the kernels' [Cout][kh][kw][Cin]
            node.register_parameter(parts[-1], nn.Parameter(t))
        for name, ch in zip(probe.bn_names, probe.bn_channels):
            node = trunk
            for p in name.split("."):
                node = _child(node, p)
            node.register_buffer("running_mean", torch.zeros(ch))
            node.register_buffer("running_var", torch.ones(ch))
            node.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.bn_momentum = 0.1
        self.bn_eps = 1e-5
        self._pair_fallback_warned = set()
        self.reset_parameters()

    def reset_parameters(self):
        """He-normal (fan-out) convs, BN (1, 0), scoring layer N(0, 0.01) / bias 0 -- what the original does before
        it loads ImageNet weights (not downloadable offline)."""
        trunk = getattr(self, self.attr)
        bn_set = set(self._bn_names)
        with torch.no_grad():
            for name in self._param_names:
                p = trunk.get_parameter(name)
                owner, leaf = name.rsplit(".", 1)
                if owner == "fc":
                    if leaf == "weight":
                        p.normal_(0, 0.01)
                    else:
                        p.zero_()
                elif owner in bn_set:
                    p.fill_(1.0) if leaf == "weight" else p.zero_()
                else:
                    o, c, kh, kw = p.shape
                    p.normal_(0, math.sqrt(2.0 / (kh * kw * o)))

    def load_imagenet_trunk(self, source, strict=True):
        """Start from an ImageNet-trained torchvision trunk, as the original backbone does (``pretrained=True`` behind
        ``resnet_dilated.Resnet34_8s(num_classes=D)``, dense_correspondence_network.py:373-375; doc/model_zoo.md:6-18).
        ``source``: the path of a stock ``torchvision.models.resnet{18,34,50,101}`` checkpoint (``.pth`` state dict) or the
        state dict itself -- its keys (``conv1.weight``, ``layer1.0.bn1.running_mean`` ...) are this trunk's keys without
        the ``resnetNN_8s.`` prefix, and dilation does not change a single shape.  The ImageNet classifier (``fc.*``,
        1000 x C) is dropped and the scoring layer is re-initialised N(0, 0.01) / bias 0, like the original.
        ``strict``: every trunk tensor must be present with the right shape (missing / unexpected keys raise).
        Returns the names of the tensors that were loaded."""
        sd = torch.load(source, map_location="cpu") if isinstance(source, (str, bytes)) or hasattr(source, "read") else source
        if isinstance(sd, dict) and "state_dict" in sd and not any(k.endswith(".weight") for k in sd):
            sd = sd["state_dict"]
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        own = getattr(self, self.attr).state_dict()
        wanted = [k for k in own if not k.startswith("fc.")]
        missing = [k for k in wanted if k not in sd and not k.endswith("num_batches_tracked")]
        unexpected = [k for k in sd if k not in own and not k.startswith("fc.")]
        bad = [k for k in wanted if k in sd and tuple(sd[k].shape) != tuple(own[k].shape)]
        if bad:
            raise ValueError("load_imagenet_trunk: shape mismatch for %s (expected a torchvision %s checkpoint, base width %d)"
                             % (bad[:4], self.arch.split("_")[0].lower(), self.base_width))
        if strict and (missing or unexpected):
            raise KeyError("load_imagenet_trunk: missing %s, unexpected %s" % (missing[:6], unexpected[:6]))
        loaded = []
        with torch.no_grad():
            for k in wanted:
                if k in sd:
                    own[k].copy_(sd[k].to(own[k].dtype))   # (state_dict() tensors alias the parameters / buffers)
                    loaded.append(k)
            fcw, fcb = own["fc.weight"], own["fc.bias"]
            fcw.normal_(0, 0.01)
            fcb.zero_()
        return loaded

    def __getstate__(self):
        # per-process handles (a CUDA event, pinned host memory, the owner table of _tables): not part of the module's state --
        # copy.deepcopy / pickle of the whole module work after a forward call too
        d = dict(self.__dict__)
        for k in ("_pending_status", "_status_host", "_table_cache", "_last_plan"):
            d.pop(k, None)
        return d

    def _tables(self):
        """(parameters, running statistics, num_batches_tracked) in the engine's order.  Walking the module tree by dotted name
        (110 get_parameter + 36 get_submodule calls) cost 1.5 - 2 ms of host time per forward call -- at the reference's batch
        size of 1 with its two forward calls per step the HOST was the bottleneck (round 5: host 10.9 of 12.2 ms) -- so the
        owners are resolved once and every call only checks that each slot still holds the tensor it held (a parameter or buffer
        re-registered, replaced by .to() under overwrite_module_params_on_conversion, or deleted rebuilds the table)."""
        tab = self.__dict__.get("_table_cache")
        if tab is not None:
            ok = True
            for parent, child, mod in tab[5]:      # every module on the way to an owner is still the one registered there
                if parent._modules.get(child) is not mod:   # (a replaced submodule keeps ITS tensors: the slots alone would pass)
                    ok = False
                    break
            for owner, name, t in (tab[0] if ok else ()):
                if owner._parameters.get(name) is not t:
                    ok = False
                    break
            if ok:
                for owner, name, t in tab[1]:
                    if owner._buffers.get(name) is not t:
                        ok = False
                        break
            if ok:
                return tab[2], tab[3], tab[4]
        trunk = getattr(self, self.attr)
        slots_p, slots_b, params, running, tracked = [], [], [], [], []
        links, seen = [(self, self.attr, trunk)], set()

        def chain(dotted):                      # registers the (parent, child name, child) edges from the trunk to `dotted`
            node = trunk
            for part in dotted.split(".") if dotted else ():
                nxt = node._modules[part]
                if (id(node), part) not in seen:
                    seen.add((id(node), part))
                    links.append((node, part, nxt))
                node = nxt
            return node
        for n in self._param_names:
            owner_name, _, leaf = n.rpartition(".")
            owner = chain(owner_name)
            t = owner._parameters[leaf]
            slots_p.append((owner, leaf, t))
            params.append(t)
        for n in self._bn_names:
            node = chain(n)
            for leaf in ("running_mean", "running_var", "num_batches_tracked"):
                slots_b.append((node, leaf, node._buffers[leaf]))
            running += [node.running_mean, node.running_var]
            tracked.append(node.num_batches_tracked)
        self.__dict__["_table_cache"] = (slots_p, slots_b, params, running, tracked, links)
        return params, running, tracked

    def forward(self, x, normalize=False, groups=1, x_b=None):
        """``groups`` = 2: ``x`` stacks two independent batches (``cat([img_a, img_b])``) -- or ``x`` is the first and ``x_b`` the
        second batch, no concatenated copy -- and a PAIR of outputs is returned;
        the result equals two consecutive forward calls -- batch-norm statistics, their gradients and the running-statistics updates are per batch -- but
        runs as ONE launch sequence (fills the 256 CUs better at small batch).  See ``forward_pair``."""
        n, _, h, w = x.shape
        if x_b is not None:
            if int(groups) != 2 or x_b.shape != x.shape:
                raise ValueError("x_b needs groups=2 and the shape of x")
            n = 2 * n
        plan = _bb.get_plan(self.arch, self.base_width, int(n), int(h), int(w), self.num_classes, int(groups))
        self._check_previous_status()   # (may raise for an EARLIER call: before this call changes any state)
        params, running, tracked = self._tables()
        if self.training:
            torch._foreach_add_(tracked, int(groups))
        owner = getattr(self, "_flat_grad_owner", None)   # dcn_hip.distributed.FlatGradients, if one manages the gradients
        self._last_plan = plan
        out = _bb.backbone_forward(x, plan, params, running, self.training, normalize, self.bn_momentum, self.bn_eps,
                                   grad_sink=owner.flat if owner is not None else None, grad_owner=owner, image_b=x_b)
        self._watch_status(plan)
        return out

    # The status word of a forward call (last_forward_status) is looked at WITHOUT a synchronisation: it is copied to pinned
    # host memory behind the call and read at the start of a later call, once the copy has completed.  A convolution weight
    # outside the range of its fp16 image (bit 1: |w| >= 1023 with the fixed weight scale 64 -- a checkpoint of another
    # training recipe could hold one) makes the split-fp16 products of that layer wrong: the training path must not go on
    # silently, so the NEXT call raises.  A non-finite activation (bit 0) is what the reference would propagate as NaN too: warned once.
    def _watch_status(self, plan):
        rng = getattr(plan, "last_activation_range", None)
        if rng is None or getattr(self, "_pending_status", None) is not None:
            return
        status = rng[1]
        if not status.is_cuda:
            self._pending_status = (status, None)
            return
        if torch.cuda.is_current_stream_capturing():
            return
        host = getattr(self, "_status_host", None)
        if host is None:
            host = self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        host.copy_(status, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending_status = (host, ev)

    def _check_previous_status(self):
        pend = getattr(self, "_pending_status", None)
        if pend is None:
            return
        host, ev = pend
        if ev is not None and (torch.cuda.is_current_stream_capturing() or not ev.query()):
            return                      # (not there yet: looked at again at the next call)
        self._pending_status = None
        st = int(host[0])
        if st & 2:
            raise FloatingPointError(
                "%s: a convolution weight of the previous forward call was outside the range of its fp16 image (|w| >= 1023 with "
                "the fixed weight scale of the split-fp16 arithmetic, or NaN): that call's results are not fp32-accurate.  Use "
                "dcn_hip.backbone.set_conv_mode('fp32') for such weights." % type(self).__name__)
        if (st & 1) and not getattr(self, "_warned_nonfinite", False):
            self._warned_nonfinite = True
            warnings.warn("%s: a non-finite activation (inf / NaN) in a forward call -- the descriptors of that call are NaN, "
                          "as they would be in the reference" % type(self).__name__)

    def last_forward_status(self):
        """(abs-max of every convolution input, status word) of the most recent forward call, as device tensors (reading
        them synchronises).  Status bit 0: an activation (or the image) was not finite -- inf or NaN; every finite fp32 range is
        handled by the split-fp16 kernels' power-of-two operand pre-scales.  Bit 1: a convolution weight was outside the
        range of its fp16 image (|w| >= 1023 with the fixed weight scale 64, or NaN)."""
        plan = getattr(self, "_last_plan", None)
        if plan is None or getattr(plan, "last_activation_range", None) is None:
            raise RuntimeError("no forward call yet")
        return plan.last_activation_range

    def forward_pair(self, x_a, x_b, normalize=False):
        """forward(x_a), forward(x_b) of the reference's training step (training.py:329-333) as one grouped engine call.
        Falls back to two calls when the shapes differ or a batch's rows are not tile-aligned."""
        if x_a.shape == x_b.shape:
            try:
                y = self.forward(x_a, normalize, groups=2, x_b=x_b)   # (two base pointers: no concatenated copy)
            except ValueError as e:
                y = None
                key = tuple(x_a.shape)
                if key not in self._pair_fallback_warned:   # once per shape: the fallback is correct but ~10-15 % slower
                    self._pair_fallback_warned.add(key)
                    warnings.warn("forward_pair: grouped launch unavailable for input %s (%s); running two forward calls"
                                  % (key, e))
            if y is not None:
                return y
        return self.forward(x_a, normalize), self.forward(x_b, normalize)

    def forward_flops(self, n, h, w):
        return _bb.get_plan(self.arch, self.base_width, int(n), int(h), int(w), self.num_classes).forward_flops


class Resnet18_8s(_DilatedResnet8s):
    arch, attr = "Resnet18_8s", "resnet18_8s"


class Resnet34_8s(_DilatedResnet8s):
    arch, attr = "Resnet34_8s", "resnet34_8s"


class Resnet50_8s(_DilatedResnet8s):
    arch, attr = "Resnet50_8s", "resnet50_8s"


class Resnet101_8s(_DilatedResnet8s):
    arch, attr = "Resnet101_8s", "resnet101_8s"
