"""Image-batch data parallelism for the dense-correspondence step: one process per MI355X, RCCL all-reduce
(torch.distributed backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests) of the backbone gradients.

The reference has no distributed code at all, so this is new: every rank runs the
reference's step on its own image pairs (own index lists, own hard-negative normalisers, BN statistics over
its own micro-batch -- there is no SyncBN in the reference), then gradients are averaged.

All parameter gradients live in ONE persistent flat fp32 buffer (21 286 211 floats = 85.1 MB for Resnet34_8s,
D=3): ``p.grad`` are views into it and autograd accumulates in place.  Two schedules:

* monolithic -- one collective over the whole buffer after ``loss.backward()`` (``all_reduce_mean``);
* bucketed (default when torch.distributed is initialised with world size > 1 and the model is the fused backbone) --
  the backward pass of the backbone engine completes its gradients in the order fc + layer4 (62 % of the bytes),
  layer3 (32 %), rest, and records an event per bucket (include/dcn_hip.h, dcn_plan_stream_wait_grad_bucket); a
  communication stream waits for bucket k, adds the engine's result into the flat buffer and all-reduces that slice
  while the compute stream is still working on buckets k+1..: only the last, small bucket's collective is exposed.
  ``all_reduce_mean`` then just joins the communication stream.  xGMI is point-to-point (7 links per GPU), a ring
  all-reduce of 85 MB on 8 GPUs moves 2 * 7/8 * 85 MB per GPU over one link pair: ~1 ms unhidden, ~0.05 ms exposed here.
"""
import torch
import torch.distributed as dist


class FlatGradients(object):
    """Owns the flat gradient buffer of ``module`` and averages it across ranks."""

    def __init__(self, module, process_group=None, bucketed=None, single_rank_collectives=False):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev = self.params[0].device
        # every gradient starts 16-byte aligned: the same layout the backbone engine writes (Plan.grad_offsets), so its
        # backward can add its whole flat result into this buffer with one kernel
        offsets = [0]
        for p in self.params:
            offsets.append((offsets[-1] + p.numel() + 3) // 4 * 4)
        self.offsets = offsets
        self.flat = torch.zeros(offsets[-1], dtype=torch.float32, device=dev)
        self.group = process_group
        self.bucketed = bucketed        # None: decide per step (distributed initialised, world > 1)
        # True: issue the collectives even when the group has ONE rank (an all-reduce over one rank is the identity) -- the
        # way to run the communication-stream / RCCL path of the bucketed schedule on a single GPU (bench.py --force-dist)
        self.single_rank_collectives = bool(single_rank_collectives)
        self._views = []
        for p, off in zip(self.params, offsets):
            self._views.append(self._view_for(p, off))
            p.grad = self._views[-1]
        # let the fused backbone (a module that owns ALL of these parameters) accumulate straight into the buffer
        owners = [m for m in module.modules() if hasattr(m, "_param_names")]
        if len(owners) == 1 and sum(p.numel() for p in owners[0].parameters()) == sum(p.numel() for p in self.params):
            owners[0]._flat_grad_owner = self
        self._comm_stream = None
        self._bucket_step = False       # this step's gradients were reduced bucket by bucket during backward
        self._detached = False
        self.stats = {"bucketed_steps": 0, "monolithic_steps": 0, "reinstalled_views": 0, "bucket_collectives": 0}

    def _view_for(self, p, off):
        chunk = self.flat[off:off + p.numel()]
        if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
            o, c, kh, kw = p.shape
            return chunk.view(o, kh, kw, c).permute(0, 3, 1, 2)  # same strides as the parameter
        return chunk.view(p.shape)

    def attach(self, optimizer):
        """Route ``optimizer.zero_grad()`` to ``zero_()`` (dcn_hip.optim.Adam honours it; for any other optimizer pass
        ``set_to_none=False`` or rely on ``ensure_views``).  Returns the optimizer."""
        optimizer._flat_gradients = self
        return optimizer

    # ------------------------------------------------------------------ view hygiene
    def note_detached(self):
        self._detached = True

    def _join_comm_stream(self):
        """A bucketed backward that was never followed by ``all_reduce_mean`` (a step skipped on a non-finite loss, world 1
        with ``bucketed=True``) still has its ``flat[lo:hi].add_`` / all-reduce pending on the communication stream: anything
        that touches the flat buffer on the current stream must be ordered behind it."""
        if self._comm_stream is not None and self.flat.is_cuda:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._comm_stream)

    def ensure_views(self):
        """``optimizer.zero_grad()`` (training.py:325) defaults to ``set_to_none=True`` in today's torch and drops the
        ``p.grad`` views; autograd then allocates fresh gradient tensors and the flat buffer would be stale.  Called
        before every collective: any ``p.grad`` that no longer aliases the buffer is copied in and re-installed
        (``None`` -> that slice is zeroed), so the collective always averages what the optimizer will read."""
        if self._bucket_step:
            self._join_comm_stream()
        fixed = 0
        for p, v in zip(self.params, self._views):
            g = p.grad
            if g is not None and g.data_ptr() == v.data_ptr() and g.stride() == v.stride():
                continue
            with torch.no_grad():
                if g is None:
                    v.zero_()
                else:
                    v.copy_(g)
            p.grad = v
            fixed += 1
        self.stats["reinstalled_views"] += fixed
        self._detached = False
        return fixed

    def zero_(self):
        """optimizer.zero_grad() equivalent that keeps the views alive (preferred over set_to_none=True: no
        re-installation copies, and the fused backbone keeps accumulating with one kernel)."""
        self._join_comm_stream()
        self.ensure_views()
        self.flat.zero_()
        self._bucket_step = False

    # ------------------------------------------------------------------ collectives
    def _world(self):
        if not dist.is_available() or not dist.is_initialized():
            return 1
        return dist.get_world_size(self.group)

    def wants_buckets(self):
        """Asked by the backbone's backward: reduce bucket by bucket on the communication stream?"""
        if self.bucketed is False:
            return False
        if self.flat.is_cuda and torch.cuda.is_current_stream_capturing():
            return False    # (inside a hipGraph capture the collectives stay outside the graph: monolithic)
        return self._world() > 1 or bool(self.bucketed)

    def _collective(self, world):
        return world > 1 or (self.single_rank_collectives and dist.is_available() and dist.is_initialized())

    def accumulate_and_reduce_buckets(self, plan, engine_flat):
        """Called by the backbone's backward once all its launches are enqueued.  For every bucket, in completion order:
        wait for the engine's grad-ready event on the communication stream, average the ENGINE's slice over the ranks
        (pre-divide, sum-all-reduce) and add it into the flat buffer -- all on the communication stream.  Reducing the
        engine's contribution rather than the accumulated buffer keeps ``p.grad`` accumulation semantics: a step with
        several backward calls (forward(img_a), forward(img_b) as two calls) sums correctly averaged contributions."""
        world = self._world()
        if self.flat.is_cuda:
            from . import _lib
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=self.flat.device)
            comm = self._comm_stream
            engine_flat.record_stream(comm)
            for k, (lo, hi) in enumerate(plan.grad_buckets):
                plan.stream_wait_grad_bucket(k, _lib.c_void_p(comm.cuda_stream))
                with torch.cuda.stream(comm):
                    src = engine_flat[lo:hi]
                    if self._collective(world):
                        if world > 1:
                            src.div_(world)
                        dist.all_reduce(src, op=dist.ReduceOp.SUM, group=self.group)   # enqueued; `comm` is ordered behind it
                        self.stats["bucket_collectives"] += 1
                    self.flat[lo:hi].add_(src)
        else:   # host-emulation tests (gloo): same slices, same order, synchronously
            for lo, hi in plan.grad_buckets:
                src = engine_flat[lo:hi]
                if self._collective(world):
                    if world > 1:
                        src.div_(world)
                    dist.all_reduce(src, op=dist.ReduceOp.SUM, group=self.group)
                    self.stats["bucket_collectives"] += 1
                self.flat[lo:hi].add_(src)
        self._bucket_step = True
        self.stats["bucketed_steps"] += 1

    def all_reduce_mean(self, async_op=False):
        """Average over ranks.  After a bucketed backward this only joins the communication stream; otherwise ONE
        collective over the whole buffer.  No-op when torch.distributed is not initialised / world size 1 (apart from
        re-installing detached views)."""
        if self._bucket_step:
            self._bucket_step = False
            if self._comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self._comm_stream)
            if self._detached:      # a second backward wrote detached gradients: they are NOT in the reduced buffer
                raise RuntimeError("FlatGradients: p.grad was detached from the flat buffer during a bucketed step; use "
                                   "FlatGradients.zero_() or zero_grad(set_to_none=False)")
            return None
        self.ensure_views()
        world = self._world()
        if not self._collective(world):
            return None
        self.stats["monolithic_steps"] += 1
        if world > 1:
            self.flat.div_(world)
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)


def broadcast_module(module, src=0, process_group=None):
    """Make every rank start from rank ``src``'s parameters and buffers (DDP's construction-time broadcast)."""
    if not dist.is_available() or not dist.is_initialized():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
