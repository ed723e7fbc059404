"""Host side of the backbone engine (csrc/backbone_engine.hip): plan cache + autograd glue.

The network arithmetic (convolutions, batch norm, pooling, upsampling, every gradient) runs in the HIP
kernels; this file allocates buffers with the PyTorch caching allocator, builds the pointer tables the
C ABI takes and hooks the two C calls into ``torch.autograd``.
"""
import ctypes
import weakref

import torch

from . import _lib


class Plan(object):
    """dcn_plan for (arch, base_width, N, H, W, D); owns the parameter / batch-norm name tables."""

    def __init__(self, arch, base_width, n, h, w, d, groups=1):
        lib = _lib.get()
        handle = ctypes.c_void_p()
        rc = lib.dcn_plan_create_grouped(arch.encode(), base_width, n, groups, h, w, d, ctypes.byref(handle))
        if rc != 0:
            raise ValueError("dcn_hip: cannot plan %s (base %d) for input [%d,3,%d,%d] in %d group(s), D=%d: %s" %
                             (arch, base_width, n, h, w, groups, d, _lib.ERRORS.get(rc, rc)))
        self.handle = handle
        self.key = (arch, base_width, n, h, w, d, groups)
        self.n, self.h, self.w, self.d, self.groups = n, h, w, d, groups
        self.param_names, self.param_shapes = [], []
        buf = ctypes.create_string_buffer(256)
        shape = (ctypes.c_int64 * 4)()
        ndim = ctypes.c_int()
        for i in range(lib.dcn_plan_num_params(handle)):
            _lib.check(lib.dcn_plan_param_info(handle, i, buf, 256, shape, ctypes.byref(ndim)), "dcn_plan_param_info")
            self.param_names.append(buf.value.decode())
            self.param_shapes.append(tuple(int(shape[k]) for k in range(ndim.value)))
        self.bn_names, self.bn_channels = [], []
        ch = ctypes.c_int64()
        for j in range(lib.dcn_plan_num_bn(handle)):
            _lib.check(lib.dcn_plan_bn_info(handle, j, buf, 256, ctypes.byref(ch)), "dcn_plan_bn_info")
            self.bn_names.append(buf.value.decode())
            self.bn_channels.append(int(ch.value))
        self.saved_bytes = int(lib.dcn_plan_saved_bytes(handle))
        self.workspace_bytes = int(lib.dcn_plan_workspace_bytes(handle))
        self.forward_flops = float(lib.dcn_plan_forward_flops(handle))
        # flat gradient layout (one buffer -> one RCCL all-reduce)
        self.param_numel = [int(torch.Size(s).numel()) for s in self.param_shapes]
        self.grad_offsets = [0]
        for nmel in self.param_numel:
            self.grad_offsets.append((self.grad_offsets[-1] + nmel + 3) // 4 * 4)  # keep every gradient 16-B aligned
        # gradient buckets in the order backward completes them: (first float, one-past-last float) of the flat buffer
        firsts = [lib.dcn_plan_grad_bucket_first_param(handle, k) for k in range(lib.dcn_plan_num_grad_buckets(handle))]
        ends = [len(self.param_numel)] + firsts[:-1]
        self.grad_buckets = [(self.grad_offsets[a], self.grad_offsets[b]) for a, b in zip(firsts, ends)]
        self.num_activation_slots = int(lib.dcn_plan_num_activation_slots(handle))
        self.activation_absmax_offset = int(lib.dcn_plan_activation_absmax_offset(handle))
        # the <= 3 alignment floats behind a gradient tensor whose size is not a multiple of 4 (never written by the engine)
        self._pad_positions = [o + n + k for o, n, e in zip(self.grad_offsets[:-1], self.param_numel, self.grad_offsets[1:])
                               for k in range(e - o - n)]
        self._pad_index = {}

    def grad_pad_index(self, dev):
        """int64 device tensor with the positions of the flat gradient buffer's alignment floats (or None): ONE index_fill_
        zeroes them, instead of one fill launch per odd-sized tensor."""
        if not self._pad_positions:
            return None
        idx = self._pad_index.get(dev)
        if idx is None:
            idx = self._pad_index[dev] = torch.tensor(self._pad_positions, dtype=torch.int64, device=dev)
        return idx

    @property
    def conv_mode(self):
        """"fp32" (fp32 MFMA) or "f16x3" (split-fp16 on the fp16 MFMA pipe, fp32-level accuracy) -- include/dcn_hip.h."""
        return CONV_MODES[_lib.get().dcn_plan_conv_mode(self.handle)]

    def set_conv_mode(self, mode):
        _lib.check(_lib.get().dcn_plan_set_conv_mode(self.handle, CONV_MODES.index(mode)), "dcn_plan_set_conv_mode")

    def stream_wait_grad_bucket(self, k, stream_ptr):
        """Make the stream wait until every gradient of bucket k of the LAST backward pass has been computed."""
        _lib.check(_lib.get().dcn_plan_stream_wait_grad_bucket(self.handle, k, stream_ptr), "dcn_plan_stream_wait_grad_bucket")

    def num_forward_records(self):
        """Forward calls whose saved arena is still alive (one small host-side record each in the engine)."""
        return int(_lib.get().dcn_plan_num_forward_records(self.handle))

    def fused_bn_backward(self):
        """Batch norms of the last backward call whose reduction ran in a dgrad epilogue (0 in fp32 mode / DCN_BN_BWD_FUSED=0)."""
        return int(_lib.get().dcn_plan_fused_bn_backward(self.handle))

    def activation_range(self, saved):
        """(abs-max per activation slot [n] fp32, status int) of the forward call that filled ``saved`` (device tensors are
        sliced, not synchronised: call .item() / .cpu() to look).  Status bit 0: a convolution input was not finite."""
        o, n = self.activation_absmax_offset, self.num_activation_slots
        return saved[o:o + 4 * n].view(torch.float32), saved[o + 4 * n:o + 4 * n + 4].view(torch.int32)

    def profile_begin(self):
        _lib.check(_lib.get().dcn_plan_profile_begin(self.handle), "dcn_plan_profile_begin")

    PROFILE_CATEGORIES = ("conv_gemm", "conv_wgrad", "conv_gemm_hl", "bn_apply", "bn_bwd_reduce", "bn_bwd_apply", "bn_finalize",
                          "resample", "other")   # DCN_PROF_* of include/dcn_hip.h

    def profile_end(self):
        """-> {category: (ms, launches, work)} of EVERY engine launch since profile_begin (synchronises).  work = algorithmic
        FLOPs for "conv_gemm" (forward + dgrad), "conv_wgrad" and "conv_gemm_hl" (the part of "conv_gemm" on the pre-split
        hl32 kernel), algorithmic HBM bytes for the streaming passes ("bn_apply", "bn_bwd_reduce", "bn_bwd_apply", "resample" =
        max pool / upsample / input layout), 0 for "bn_finalize" (latency-bound) and "other" (fills, weight splits ...)."""
        k = len(self.PROFILE_CATEGORIES)
        ms = (ctypes.c_double * k)()
        n = (ctypes.c_int64 * k)()
        fl = (ctypes.c_double * k)()
        _lib.check(_lib.get().dcn_plan_profile_end_all(self.handle, ms, n, fl), "dcn_plan_profile_end_all")
        return {name: (ms[i], int(n[i]), fl[i]) for i, name in enumerate(self.PROFILE_CATEGORIES)}

    def __del__(self):
        try:
            _lib.get().dcn_plan_destroy(self.handle)
        except Exception:
            pass


_PLANS = {}
_lib._reset_hooks.append(_PLANS.clear)
CONV_MODES = ("fp32", "f16x3")
_conv_mode = [None]   # None: the library default (f16x3, or the DCN_CONV_MODE environment variable)


def set_conv_mode(mode):
    """Process-wide convolution arithmetic for all plans, existing and future: "fp32", "f16x3" or None (library default)."""
    if mode is not None and mode not in CONV_MODES:
        raise ValueError("conv mode must be one of %s" % (CONV_MODES,))
    _conv_mode[0] = mode
    if mode is not None:
        for p in _PLANS.values():
            p.set_conv_mode(mode)


def get_plan(arch, base_width, n, h, w, d, groups=1):
    """``groups`` > 1: n images = ``groups`` independent batches stacked along N (batch-norm statistics per batch)."""
    key = (arch, base_width, n, h, w, d, groups)
    p = _PLANS.get(key)
    if p is None:
        p = _PLANS[key] = Plan(*key)
        if _conv_mode[0] is not None:
            p.set_conv_mode(_conv_mode[0])
    return p


def _kernel_layout(p):
    """Conv weights are consumed as [Cout][kh][kw][Cin] == logical OIHW in channels_last memory."""
    if p.dim() == 4:
        return p.contiguous(memory_format=torch.channels_last)
    return p.contiguous()


def _grad_views(flat, plan):
    views = []
    for i, shape in enumerate(plan.param_shapes):
        v = flat[plan.grad_offsets[i]:plan.grad_offsets[i] + plan.param_numel[i]]
        if len(shape) == 4:
            o, c, kh, kw = shape
            v = v.view(o, kh, kw, c).permute(0, 3, 1, 2)
        views.append(v)
    return views



POISON_ARENAS = False   # tests: fill the saved / workspace arenas with 0xFF bytes (NaN as fp32 and as fp16) before every call,
                        # so that a kernel reading a byte nobody wrote in this step shows up in the results


def _arena(nbytes, dev):
    if POISON_ARENAS:
        return torch.full((nbytes,), 255, dtype=torch.uint8, device=dev)
    return torch.empty(nbytes, dtype=torch.uint8, device=dev)


def _forget_saved(plan_ref, lib, ptr):
    """Finalizer of a saved arena: the plan's record of the forward call that filled it dies with the tensor."""
    plan = plan_ref()
    if plan is not None and _lib._lib is lib:     # (the plan and the library it was made by are still alive)
        lib.dcn_plan_forget_saved(plan.handle, ptr)


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, image_b, plan, bn_running, training, normalize, momentum, eps, *params):
        """image_b (grouped plans): the second batch as a tensor of its own -- images [N/2, N) -- instead of a concatenated copy."""
        lib = _lib.get()
        _lib.require_device(image, *params)
        n_here = plan.n // 2 if image_b is not None else plan.n
        for im in (image,) + ((image_b,) if image_b is not None else ()):
            if im.dim() != 4 or im.shape[1] != 3 or im.dtype != torch.float32:
                raise ValueError("expected a float32 [N,3,H,W] image batch, got %s %s" % (tuple(im.shape), im.dtype))
            if tuple(im.shape) != (n_here, 3, plan.h, plan.w):
                raise ValueError("plan %s does not match input %s" % (plan.key, tuple(im.shape)))
        if image_b is not None:
            _lib.require_device(image_b)
            image_b = image_b.contiguous()
        image = image.contiguous()
        dev = image.device
        kparams = [_kernel_layout(p.detach()) for p in params]
        pptr = (ctypes.c_void_p * len(kparams))(*[p.data_ptr() for p in kparams])
        rptr = (ctypes.c_void_p * len(bn_running))(*[b.data_ptr() for b in bn_running])
        desc = torch.empty((plan.n, plan.h, plan.w, plan.d), dtype=torch.float32, device=dev)
        saved = _arena(plan.saved_bytes, dev)
        ws = _arena(plan.workspace_bytes, dev)
        if image_b is not None:
            rc = lib.dcn_backbone_forward_pair(plan.handle, _lib.ptr(image), _lib.ptr(image_b), pptr, rptr, float(momentum),
                                               float(eps), int(bool(training)), int(bool(normalize)), _lib.ptr(desc),
                                               _lib.ptr(saved), _lib.ptr(ws), _lib.stream_ptr())
        else:
            rc = lib.dcn_backbone_forward(plan.handle, _lib.ptr(image), pptr, rptr, float(momentum), float(eps),
                                          int(bool(training)), int(bool(normalize)), _lib.ptr(desc), _lib.ptr(saved),
                                          _lib.ptr(ws), _lib.stream_ptr())
        _lib.check(rc, "dcn_backbone_forward")
        if training:   # the engine keyed a record of this call by the arena's address: tie it to the arena's lifetime
            weakref.finalize(saved, _forget_saved, weakref.ref(plan), lib, saved.data_ptr()).atexit = False
        ctx.plan = plan
        # the n + 1 range scalars are COPIED out (a small device-to-device copy, no sync): views would keep the whole saved
        # arena of this call alive on the long-lived plan -- two arenas per forward, and one pinned for ever after an
        # eval / inference call
        amax, status = plan.activation_range(saved)
        plan.last_activation_range = (amax.clone(), status.clone())
        ctx.grad_sink = getattr(bn_running, "grad_sink", None)
        ctx.grad_owner = getattr(bn_running, "grad_owner", None)
        ctx.grad_probe = (params[0], params[-1])  # to verify at backward time that .grad still aliases the sink
        ctx.saved_arena = saved
        ctx.kparams = kparams
        ctx.trained = bool(training)
        ctx.normalize = bool(normalize)
        # logical [N,D,H,W] over NHWC memory == torch.channels_last
        if plan.groups == 2:
            # one output per image batch: their gradients arrive separately and channels_last (slicing ONE output instead
            # makes autograd assemble the gradient in NCHW: a 2N x D x H x W transpose copy per step)
            half = plan.n // 2
            return desc[:half].permute(0, 3, 1, 2), desc[half:].permute(0, 3, 1, 2)
        return desc.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, *grad_descs):
        lib = _lib.get()
        plan = ctx.plan
        if not ctx.trained:
            raise RuntimeError("dcn_hip: backward through an eval-mode forward is not supported")
        g_b = None
        if plan.groups == 2:
            # the two outputs' gradients are handed to the engine as they arrive (two pointers): NHWC views of
            # channels_last tensors are contiguous, so nothing is copied; a missing one is a zero map
            half = plan.n // 2
            ref = next(gk for gk in grad_descs if gk is not None)
            gs = []
            for gk in grad_descs:
                if gk is None:
                    gs.append(torch.zeros((half, plan.h, plan.w, plan.d), dtype=torch.float32, device=ref.device))
                else:
                    gs.append(gk.permute(0, 2, 3, 1).contiguous())
            g, g_b = gs
        else:
            g = grad_descs[0].permute(0, 2, 3, 1).contiguous()  # NHWC; no copy when grad is channels_last
        dev = g.device
        ws = _arena(plan.workspace_bytes, dev)
        flat = torch.empty(plan.grad_offsets[-1], dtype=torch.float32, device=dev)
        pad = plan.grad_pad_index(dev)   # the <= 3 alignment floats behind a tensor are never written by the engine:
        if pad is not None:              # keep them zero, the buffer is ADDED into the shared sink (one launch for all)
            flat.index_fill_(0, pad, 0.0)
        base = flat.data_ptr()
        gptr = (ctypes.c_void_p * len(plan.param_numel))(*[base + 4 * o for o in plan.grad_offsets[:-1]])
        pptr = (ctypes.c_void_p * len(ctx.kparams))(*[p.data_ptr() for p in ctx.kparams])
        if g_b is not None:
            rc = lib.dcn_backbone_backward_pair(plan.handle, _lib.ptr(g), _lib.ptr(g_b), pptr, _lib.ptr(ctx.saved_arena),
                                                _lib.ptr(ws), gptr, int(ctx.normalize), _lib.stream_ptr())
        else:
            rc = lib.dcn_backbone_backward(plan.handle, _lib.ptr(g), pptr, _lib.ptr(ctx.saved_arena), _lib.ptr(ws), gptr,
                                           int(ctx.normalize), _lib.stream_ptr())
        _lib.check(rc, "dcn_backbone_backward")
        ctx.saved_arena = None
        sink = ctx.grad_sink
        first, last = ctx.grad_probe
        aliased = (sink is not None and sink.numel() == flat.numel() and first.grad is not None and
                   last.grad is not None and first.grad.data_ptr() == sink.data_ptr() and
                   last.grad.data_ptr() == sink.data_ptr() + 4 * plan.grad_offsets[-2])
        if aliased:
            # the parameters' .grad are views of ONE flat buffer with this very layout (dcn_hip.distributed.FlatGradients):
            # accumulate with a single kernel instead of ~110 autograd AccumulateGrad launches -- or, data-parallel, bucket
            # by bucket on the communication stream as the engine's grad-ready events fire, each followed by its all-reduce
            owner = ctx.grad_owner
            if owner is not None and owner.wants_buckets():
                owner.accumulate_and_reduce_buckets(plan, flat)
            else:
                sink.add_(flat)
            return (None,) * (8 + len(plan.param_numel))
        if sink is not None and ctx.grad_owner is not None:
            ctx.grad_owner.note_detached()   # e.g. optimizer.zero_grad(set_to_none=True) dropped the views: see all_reduce_mean
        return (None, None, None, None, None, None, None, None) + tuple(_grad_views(flat, plan))


class _RunningList(list):
    """bn running-statistic tensors + an optional flat gradient sink (non-tensor payload of the autograd call)."""
    grad_sink = None
    grad_owner = None


def backbone_forward(image, plan, params, bn_running, training, normalize=False, momentum=0.1, eps=1e-5, grad_sink=None,
                     grad_owner=None, image_b=None):
    """image [N,3,H,W] -> descriptors, logical [N,D,H,W] in channels_last memory.
    ``image_b`` (plans with two groups): the second batch as its own tensor; ``image`` is then the first batch only.
    ``params`` / ``bn_running`` follow ``plan.param_names`` / ``plan.bn_names`` (running_mean, running_var per BN).
    ``grad_sink``: flat fp32 buffer laid out like ``plan.grad_offsets`` that the parameters' ``.grad`` alias."""
    rl = _RunningList(bn_running)
    rl.grad_sink = grad_sink
    rl.grad_owner = grad_owner   # dcn_hip.distributed.FlatGradients that owns grad_sink (bucketed all-reduce), or None
    return _BackboneFn.apply(image, image_b, plan, rl, training, normalize, momentum, eps, *params)
