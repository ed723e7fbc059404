"""`optimizer.step()` of the training step as one HIP pass (dcn_adam_step, csrc/optim_kernels.hip).

The reference builds `torch.optim.Adam(dcn.parameters(), lr=1e-4, weight_decay=1e-4)` (training.py:133-145) and
calls `optimizer.zero_grad()` / `optimizer.step()` every iteration (training.py:325,346); `adjust_learning_rate`
(training.py:544-558) rewrites `param_group['lr']`, `save_network` (training.py:501-521) stores
`optimizer.state_dict()`.  `Adam` below has the same constructor, the same per-parameter state (`step`, `exp_avg`,
`exp_avg_sq`) and therefore the same `state_dict()` layout -- a `.pth.opt` file of either class loads into the other --
but updates all parameters of a group with one kernel launch per 80 tensors instead of ~10 passes over the 85 MB of
parameters.  There is no CPU fallback: without the HIP library `step()` raises."""
import ctypes

import torch

from . import _lib


def _dense(t):
    """True when the tensor's elements occupy numel() consecutive floats in SOME dimension order."""
    dims = sorted((d for d in range(t.dim()) if t.shape[d] != 1), key=lambda d: t.stride(d))
    expect = 1
    for d in dims:
        if t.stride(d) != expect:
            return False
        expect *= t.shape[d]
    return True


def _like_param(t, p):
    """`t` with the memory order of the parameter `p` (element k of the storage of both is the same logical element)."""
    if t.stride() == p.stride() and t.dtype == p.dtype and t.device == p.device:
        return t
    out = torch.empty_like(p, memory_format=torch.preserve_format)
    if out.stride() != p.stride():  # preserve_format keeps the strides of a dense tensor; anything else is unsupported
        raise ValueError("dcn_hip.optim.Adam: cannot lay out optimizer state like a parameter with strides %s" % (p.stride(),))
    out.copy_(t)
    return out


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (L2 weight decay added to the gradient, bias correction, no amsgrad) for dense fp32
    parameters on the GPU."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        if not 0.0 <= weight_decay:
            raise ValueError("Invalid weight_decay value: {}".format(weight_decay))
        if amsgrad:
            raise NotImplementedError("dcn_hip.optim.Adam: amsgrad is not implemented (the reference does not use it)")
        super(Adam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                                maximize=False))

    def zero_grad(self, set_to_none=True):
        """``optimizer.zero_grad()`` of training.py:325.  When the gradients are views of a flat buffer
        (``dcn_hip.distributed.FlatGradients(...).attach(optimizer)``) they are zeroed IN PLACE with one kernel whatever
        ``set_to_none`` says -- dropping the views would detach ``p.grad`` from the buffer the collective averages."""
        flat = getattr(self, "_flat_gradients", None)
        if flat is not None:
            flat.zero_()
            return
        super(Adam, self).zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.get()
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("dcn_hip.optim.Adam: amsgrad / maximize are not implemented")
            if self._fast_step(lib, gi, group):
                continue
            by_step = {}
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                if p.dtype != torch.float32 or g.dtype != torch.float32:
                    raise TypeError("dcn_hip.optim.Adam: float32 parameters and gradients only")
                if not _dense(p):
                    raise ValueError("dcn_hip.optim.Adam: parameter storage must be dense, got strides %s" % (p.stride(),))
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                # state loaded from a checkpoint written with another memory format: re-lay it once
                st["exp_avg"] = _like_param(st["exp_avg"], p)
                st["exp_avg_sq"] = _like_param(st["exp_avg_sq"], p)
                g = _like_param(g, p)
                _lib.require_device(p, g, st["exp_avg"], st["exp_avg_sq"])
                if not torch.is_tensor(st["step"]):
                    st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
                st["step"] += 1
                by_step.setdefault(int(st["step"].item()), []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            beta1, beta2 = group["betas"]
            for t, items in by_step.items():
                n = len(items)
                cols = [(ctypes.c_void_p * n)(*[x[k].data_ptr() for x in items]) for k in range(4)]
                numel = (ctypes.c_int64 * n)(*[x[0].numel() for x in items])
                rc = lib.dcn_adam_step(n, cols[0], cols[1], cols[2], cols[3], numel, float(group["lr"]), float(beta1),
                                       float(beta2), float(group["eps"]), float(group["weight_decay"]), t,
                                       _lib.stream_ptr())
                _lib.check(rc, "dcn_adam_step")
            if len(by_step) == 1:   # every parameter of the group at the same step: the next call can take the fast path
                (t, items), = by_step.items()
                if len(items) == len(group["params"]):
                    self._fast = getattr(self, "_fast", {})
                    # (pointers and shapes only for the gradients: the cache must not keep last step's gradient tensors alive)
                    self._fast[gi] = {"t": t, "items": [(it[0], None, it[2], it[3]) for it in items], "cols": cols,
                                      "numel": numel, "ptrs": [tuple(x.data_ptr() for x in it) for it in items],
                                      "steps": [self.state[it[0]]["step"] for it in items]}
        return loss

    def _fast_step(self, lib, gi, group):
        """The steady state of a training loop -- the same parameters, gradient buffers and state tensors as in the previous
        call, all at the same step: one launch from the pointer tables of the previous call, after checking that nothing moved
        (110 parameters x the per-parameter checks, state-dict lookups and .item() of the general path were ~2 ms of host time
        per step: what bounds the reference's batch size of 1 is the host).  Anything unexpected: the general path."""
        fast = getattr(self, "_fast", {}).get(gi)
        if fast is None:
            return False
        params = group["params"]
        items = fast["items"]
        if len(params) != len(items):
            self._fast.pop(gi, None)
            return False
        for p, it, ptr, st in zip(params, items, fast["ptrs"], fast["steps"]):
            g = p.grad
            state = self.state.get(p)
            # (pointer, strides and dtype of the gradient are compared on EVERY call: the same tensor object may have been
            # re-pointed -- `p.grad.data = ...`, `set_`, a bucket view reassigned -- and the cached tables hold addresses)
            if (p is not it[0] or g is None or g.data_ptr() != ptr[1] or g.stride() != p.stride() or g.dtype != p.dtype
                    or g.is_sparse or state is None or state.get("exp_avg") is not it[2] or state.get("exp_avg_sq") is not it[3]
                    or it[2].data_ptr() != ptr[2] or it[3].data_ptr() != ptr[3]
                    or state.get("step") is not st or p.data_ptr() != ptr[0]):
                self._fast.pop(gi, None)
                return False
        t = fast["t"] + 1
        torch._foreach_add_(fast["steps"], 1.0)    # the per-parameter step counters (host tensors), one call
        beta1, beta2 = group["betas"]
        cols = fast["cols"]
        rc = lib.dcn_adam_step(len(items), cols[0], cols[1], cols[2], cols[3], fast["numel"], float(group["lr"]), float(beta1),
                               float(beta2), float(group["eps"]), float(group["weight_decay"]), t, _lib.stream_ptr())
        _lib.check(rc, "dcn_adam_step")
        fast["t"] = t
        return True
