"""Image-batch data parallelism for the dense-correspondence step: one process per MI355X, one RCCL all-reduce
(torch.distributed backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests) of the backbone gradients per step.

The reference has no distributed code at all, so this is new: every rank runs the
reference's step on its own image pairs (own index lists, own hard-negative normalisers, BN statistics over
its own micro-batch -- there is no SyncBN in the reference), then gradients are averaged.

All parameter gradients live in ONE persistent flat fp32 buffer (21 286 211 floats = 85.1 MB for Resnet34_8s,
D=3): ``p.grad`` are views into it, autograd accumulates in place, and a step costs exactly one collective over
xGMI (direct reduce-scatter + all-gather moves 1/8 of the buffer per link: ~0.15 ms on an 8-GPU node)."""
import torch
import torch.distributed as dist


class FlatGradients(object):
    """Owns the flat gradient buffer of ``module`` and averages it across ranks."""

    def __init__(self, module, process_group=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev = self.params[0].device
        # every gradient starts 16-byte aligned: the same layout the backbone engine writes (Plan.grad_offsets), so its
        # backward can add its whole flat result into this buffer with one kernel
        offsets = [0]
        for p in self.params:
            offsets.append((offsets[-1] + p.numel() + 3) // 4 * 4)
        self.flat = torch.zeros(offsets[-1], dtype=torch.float32, device=dev)
        self.group = process_group
        for p, off in zip(self.params, offsets):
            n = p.numel()
            chunk = self.flat[off:off + n]
            if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
                o, c, kh, kw = p.shape
                view = chunk.view(o, kh, kw, c).permute(0, 3, 1, 2)  # same strides as the parameter
            else:
                view = chunk.view(p.shape)
            p.grad = view
        # let the fused backbone (a module that owns ALL of these parameters) accumulate straight into the buffer
        owners = [m for m in module.modules() if hasattr(m, "_param_names")]
        if len(owners) == 1 and sum(p.numel() for p in owners[0].parameters()) == sum(p.numel() for p in self.params):
            owners[0]._flat_grad_sink = self.flat

    def zero_(self):
        """optimizer.zero_grad() equivalent that keeps the views alive (use instead of set_to_none=True)."""
        self.flat.zero_()

    def all_reduce_mean(self, async_op=False):
        """Average over ranks.  No-op when torch.distributed is not initialised / world size 1."""
        if not dist.is_available() or not dist.is_initialized():
            return None
        world = dist.get_world_size(self.group)
        if world == 1:
            return None
        self.flat.div_(world)
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)


def broadcast_module(module, src=0, process_group=None):
    """Make every rank start from rank ``src``'s parameters and buffers (DDP's construction-time broadcast)."""
    if not dist.is_available() or not dist.is_initialized():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
