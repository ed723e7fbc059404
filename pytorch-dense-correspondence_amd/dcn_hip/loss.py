"""Host side of kernel K9 (csrc/loss_kernels.hip): ragged pair lists + autograd glue.

The arithmetic is in the HIP kernels; this file only packs index lists, allocates outputs with the
PyTorch caching allocator and wires ``torch.autograd``.  Reference semantics:
dense_correspondence/loss_functions/pixelwise_contrastive_loss.py and loss_composer.py.
"""
import ctypes
import os as _os

import torch

from . import _lib

LIST_MATCH, LIST_MASKED, LIST_BACKGROUND, LIST_BLIND = 0, 1, 2, 3
COMPOSE_WITHIN_SCENE, COMPOSE_DIFFERENT_OBJECT, COMPOSE_ACROSS_SCENE, COMPOSE_RAW_SUMS = 0, 1, 2, 3


def _is_host_sentinel(t):
    """The reference's empty-list marker ``LongTensor([-1])`` (dense_correspondence_dataset_masked.py:209-223).
    Only inspected when the tensor lives on the host; a device-resident sentinel is left in place and skipped
    by the kernel (negative index), which avoids the device->host sync ``is_empty`` costs the reference."""
    return t is None or t.numel() == 0 or (t.device.type == "cpu" and t.numel() == 1 and int(t[0]) == -1)


class PairLists(object):
    """Concatenated int64 pair lists of ``num_pairs`` image pairs:
    ``idx_a/idx_b[offsets[4p+t] : offsets[4p+t+1]]`` is list type t (match, masked, background, blind) of pair p.
    Build once per batch (``from_lists``) -- it is the CSR calling convention of SURVEY.md 8a note B."""

    def __init__(self, idx_a, idx_b, offsets_host):
        self.idx_a = idx_a
        self.idx_b = idx_b
        self.offsets_host = tuple(int(o) for o in offsets_host)
        self.num_pairs = (len(self.offsets_host) - 1) // 4
        self._off_c = (ctypes.c_int64 * len(self.offsets_host))(*self.offsets_host)
        self.offsets_dev = torch.tensor(self.offsets_host, dtype=torch.int64).to(idx_a.device, non_blocking=True)
        self.max_len = max([self.offsets_host[i + 1] - self.offsets_host[i] for i in range(4 * self.num_pairs)] + [0])
        self.total = self.offsets_host[-1]

    @staticmethod
    def from_lists(pairs, device, hw=None):
        """pairs: sequence of 8-tuples (matches_a, matches_b, masked_a, masked_b, background_a, background_b,
        blind_a, blind_b) -- the order loss_composer.get_loss takes them (loss_composer.py:7-12).
        ``hw`` (optional): pixels per image; lists that arrive on the HOST are then range-checked here (the reference's
        index_select raises IndexError on an out-of-range index; device-resident lists are checked by the kernel, which
        skips such pairs and raises the status word -- see loss_composer.get_loss_batched)."""
        chunks_a, chunks_b, offsets = [], [], [0]
        for lists in pairs:
            assert len(lists) == 8
            for t in range(4):
                a, b = lists[2 * t], lists[2 * t + 1]
                if _is_host_sentinel(a) or _is_host_sentinel(b):
                    offsets.append(offsets[-1])
                    continue
                if a.numel() != b.numel():
                    raise ValueError("pair list %d: a has %d entries, b has %d" % (t, a.numel(), b.numel()))
                if hw is not None:
                    for side in (a, b):
                        if side.device.type == "cpu" and side.numel() and (int(side.max()) >= hw or int(side.min()) < 0):
                            raise IndexError("pair list %d holds a pixel index outside [0, %d) (lists built for another "
                                             "image size?)" % (t, hw))
                chunks_a.append(a.reshape(-1))
                chunks_b.append(b.reshape(-1))
                offsets.append(offsets[-1] + a.numel())
        if chunks_a:
            idx_a = torch.cat(chunks_a).to(device=device, dtype=torch.int64)
            idx_b = torch.cat(chunks_b).to(device=device, dtype=torch.int64)
        else:
            idx_a = torch.zeros(1, dtype=torch.int64, device=device)
            idx_b = torch.zeros(1, dtype=torch.int64, device=device)
        return PairLists(idx_a.contiguous(), idx_b.contiguous(), offsets)

    def length(self, pair, t):
        return self.offsets_host[4 * pair + t + 1] - self.offsets_host[4 * pair + t]


def make_config(margins, image_width, match_loss_weight=1.0, non_match_loss_weight=1.0, scale_by_hard_negatives=True,
                compose=COMPOSE_WITHIN_SCENE, invert=(0, 0, 0, 0), pixel_weight=(0, 0, 0, 0), m_pixel=1.0):
    cfg = _lib.LossConfig()
    for i in range(4):
        cfg.margin[i] = float(margins[i])
        cfg.invert[i] = int(invert[i])   # 0 / 1 (inverted hinge) / 2 (legacy hinge on the squared distance)
        cfg.pixel_weight[i] = int(bool(pixel_weight[i]))
    cfg.m_pixel = float(m_pixel)
    cfg.image_width = int(image_width)
    cfg.match_loss_weight = float(match_loss_weight)
    cfg.non_match_loss_weight = float(non_match_loss_weight)
    cfg.scale_by_hard_negatives = int(bool(scale_by_hard_negatives))
    cfg.compose = int(compose)
    return cfg


# Measured on the MI355X at BASELINE configs[2]'s list sizes (profiles/r4e_loss_bench.txt): the backward pass is bound by the
# fp32 atomics of its scatter (7 M descriptor-sized runs in ~420 us), not by its gathers -- records alone: call 912 -> 885 us
# (backward kernel 462 -> 418 us, forward 259 -> 279 us); the early zero-fill competes with the forward gather for HBM and
# buys nothing (897 us alone, 910 us with records) while its two cross-stream waits cost ~13 us on a 41-us call at configs[1]'s
# sizes: off by default.
# Round 6: the backward scatter with INTEGER atomics on 64-bit fixed point (csrc/loss_kernels.hip, loss_bwd_saved_exact_kernel):
# order-independent, so the gradient maps -- and with them the whole training step -- are bit-reproducible run to run.  Costs
# two int64 maps (twice the bytes of the fp32 gradient maps) and a conversion pass: the default wherever that workspace stays
# below EXACT_BACKWARD_MAX_BYTES (BASELINE configs 1 / 2 / 4: 15 - 118 MB); above it (configs 3 / 5: 2.5 / 1.3 GB) the fp32
# atomics.  DCN_LOSS_EXACT=1 / 0 forces either.
EXACT_BACKWARD = {"1": True, "0": False}.get(_os.environ.get("DCN_LOSS_EXACT", ""), None)   # None: by workspace size
EXACT_BACKWARD_MAX_BYTES = 256 << 20
SAVE_PAIR_RECORDS = True   # forward keeps per-pair (difference, factor) records, backward reads them instead of gathering again
PREFILL_GRADIENTS = False  # the dense gradient maps zero-filled on a side stream while the forward kernels run
_fill_streams = {}


def _fill_stream(dev):
    st = _fill_streams.get(dev)
    if st is None:
        st = _fill_streams[dev] = torch.cuda.Stream(device=dev)
    return st


def _run_forward(desc_a, desc_b, lists, cfg, want_per_term, records=None):
    lib = _lib.get()
    _lib.require_device(desc_a, desc_b, lists.idx_a, lists.idx_b)
    P, HW, D = desc_a.shape
    if desc_a.dtype != torch.float32 or desc_b.dtype != torch.float32:
        raise TypeError("dcn_hip loss kernels take float32 descriptor maps, got %s / %s (cast with .float(); the gradient "
                        "then flows back through the cast)" % (desc_a.dtype, desc_b.dtype))
    if desc_b.shape != desc_a.shape or P != lists.num_pairs:
        raise ValueError("descriptor maps %s / %s do not match %d pair lists" %
                         (tuple(desc_a.shape), tuple(desc_b.shape), lists.num_pairs))
    desc_a = desc_a.contiguous()
    desc_b = desc_b.contiguous()
    dev = desc_a.device
    f32 = dict(dtype=torch.float32, device=dev)
    terms = torch.empty(P, 5, **f32)
    sums = torch.empty(P, 4, **f32)
    hard = torch.empty(P, 4, dtype=torch.int32, device=dev)
    loss = torch.empty(1, **f32)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    per_term = torch.empty(max(lists.total, 1), **f32) if want_per_term else None
    ws = torch.empty(lib.dcn_loss_workspace_bytes(P, lists.max_len), dtype=torch.uint8, device=dev)
    if records is not None:
        rc = lib.dcn_contrastive_loss_forward_save(
            _lib.ptr(desc_a), _lib.ptr(desc_b), P, HW, D, _lib.ptr(lists.idx_a), _lib.ptr(lists.idx_b),
            ctypes.cast(lists._off_c, ctypes.c_void_p), _lib.ptr(lists.offsets_dev), ctypes.byref(cfg),
            _lib.ptr(terms), _lib.ptr(sums), _lib.ptr(hard), _lib.ptr(loss), _lib.ptr(per_term), _lib.ptr(status),
            _lib.ptr(ws), _lib.ptr(records), _lib.stream_ptr())
    else:
        rc = lib.dcn_contrastive_loss_forward(
            _lib.ptr(desc_a), _lib.ptr(desc_b), P, HW, D, _lib.ptr(lists.idx_a), _lib.ptr(lists.idx_b),
            ctypes.cast(lists._off_c, ctypes.c_void_p), _lib.ptr(lists.offsets_dev), ctypes.byref(cfg),
            _lib.ptr(terms), _lib.ptr(sums), _lib.ptr(hard), _lib.ptr(loss), _lib.ptr(per_term), _lib.ptr(status),
            _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, "dcn_contrastive_loss_forward")
    return desc_a, desc_b, loss, terms, sums, hard, status, per_term


class _ContrastiveLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, desc_a, desc_b, lists, cfg, want_per_term):
        lib = _lib.get()
        wants_grad = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        records = None
        ctx.grads, ctx.grads_ready = None, None
        if wants_grad and SAVE_PAIR_RECORDS and lists.total > 0:
            records = torch.empty(lib.dcn_loss_saved_floats(lists.total, int(desc_a.shape[2])), dtype=torch.float32,
                                  device=desc_a.device)
        if wants_grad and PREFILL_GRADIENTS and records is not None:
            # the two dense gradient maps of the backward pass, zero-filled NOW on a side stream: the fill (1.26 GB at
            # BASELINE configs[2]'s sizes) streams at HBM speed next to the latency-bound gather of the forward kernels
            # instead of in front of the backward scatter
            g2 = torch.empty((2,) + tuple(desc_a.shape), dtype=torch.float32, device=desc_a.device)
            if g2.is_cuda:
                cur, side = torch.cuda.current_stream(g2.device), _fill_stream(g2.device)
                side.wait_stream(cur)     # (the allocator may hand out memory whose last use is still queued on `cur`)
                _lib.check(lib.dcn_fill_bytes(_lib.ptr(g2), 0, g2.numel() * 4, ctypes.c_void_p(side.cuda_stream)), "dcn_fill_bytes")
                g2.record_stream(side)
                ctx.grads_ready = side.record_event()
            else:
                _lib.check(lib.dcn_fill_bytes(_lib.ptr(g2), 0, g2.numel() * 4, None), "dcn_fill_bytes")
            ctx.grads = g2
        desc_a, desc_b, loss, terms, sums, hard, status, per_term = _run_forward(desc_a, desc_b, lists, cfg,
                                                                                 want_per_term, records)
        ctx.records = records
        if records is not None:
            ctx.save_for_backward(sums, hard)       # (the descriptor maps themselves are not needed again)
            ctx.shape = tuple(desc_a.shape)
        else:
            ctx.save_for_backward(desc_a, desc_b, sums, hard)
        ctx.lists = lists
        ctx.cfg = cfg
        if want_per_term:
            ctx.mark_non_differentiable(terms, sums, hard, status, per_term)
        else:
            ctx.mark_non_differentiable(terms, sums, hard, status)
        return loss.reshape(()), terms, sums, hard, status, per_term

    @staticmethod
    def backward(ctx, grad_loss, *unused):
        lib = _lib.get()
        lists, cfg = ctx.lists, ctx.cfg
        if ctx.records is not None:
            sums, hard = ctx.saved_tensors
            P, HW, D = ctx.shape
            g2, prefilled = ctx.grads, 0
            ctx.grads = None                      # (a second backward through the same graph fills its own maps)
            if g2 is not None:
                prefilled = 1
                if ctx.grads_ready is not None:
                    torch.cuda.current_stream(g2.device).wait_event(ctx.grads_ready)
            else:
                g2 = torch.empty((2, P, HW, D), dtype=torch.float32, device=hard.device)
            gl = grad_loss.reshape(1).to(torch.float32).contiguous()
            exact_bytes = int(lib.dcn_loss_exact_workspace_bytes(P, HW, D))
            per_pair_max = max(lists.offsets_host[4 * p + 4] - lists.offsets_host[4 * p] for p in range(lists.num_pairs))
            exact = (EXACT_BACKWARD if EXACT_BACKWARD is not None else exact_bytes <= EXACT_BACKWARD_MAX_BYTES) and \
                per_pair_max < (1 << 22) and P <= 16383     # (maps zero-filled ahead of time, PREFILL_GRADIENTS, are simply overwritten)
            if exact:
                ws = torch.empty(exact_bytes, dtype=torch.uint8, device=hard.device)
                rc = lib.dcn_contrastive_loss_backward_saved_exact(
                    P, HW, D, _lib.ptr(lists.idx_a), _lib.ptr(lists.idx_b), ctypes.cast(lists._off_c, ctypes.c_void_p),
                    _lib.ptr(lists.offsets_dev), ctypes.byref(cfg), _lib.ptr(hard), _lib.ptr(gl), _lib.ptr(ctx.records),
                    _lib.ptr(ws), _lib.ptr(g2[0]), _lib.ptr(g2[1]), _lib.stream_ptr())
                _lib.check(rc, "dcn_contrastive_loss_backward_saved_exact")
                return g2[0], g2[1], None, None, None
            rc = lib.dcn_contrastive_loss_backward_saved(
                P, HW, D, _lib.ptr(lists.idx_a), _lib.ptr(lists.idx_b), ctypes.cast(lists._off_c, ctypes.c_void_p),
                _lib.ptr(lists.offsets_dev), ctypes.byref(cfg), _lib.ptr(hard), _lib.ptr(gl), _lib.ptr(ctx.records), prefilled,
                _lib.ptr(g2[0]), _lib.ptr(g2[1]), _lib.stream_ptr())
            _lib.check(rc, "dcn_contrastive_loss_backward_saved")
            return g2[0], g2[1], None, None, None
        desc_a, desc_b, sums, hard = ctx.saved_tensors
        P, HW, D = desc_a.shape
        g2 = torch.empty((2,) + tuple(desc_a.shape), dtype=desc_a.dtype, device=desc_a.device)   # one allocation: the kernel side
        ga, gb = g2[0], g2[1]                                                                     # zero-fills both maps in one launch
        gl = grad_loss.reshape(1).to(torch.float32).contiguous()
        rc = lib.dcn_contrastive_loss_backward(
            _lib.ptr(desc_a), _lib.ptr(desc_b), P, HW, D, _lib.ptr(lists.idx_a), _lib.ptr(lists.idx_b),
            ctypes.cast(lists._off_c, ctypes.c_void_p), _lib.ptr(lists.offsets_dev), ctypes.byref(cfg),
            _lib.ptr(sums), _lib.ptr(hard), _lib.ptr(gl), None, _lib.ptr(ga), _lib.ptr(gb), _lib.stream_ptr())
        _lib.check(rc, "dcn_contrastive_loss_backward")
        return ga, gb, None, None, None


class _PerTermFn(torch.autograd.Function):
    """Differentiable per-pair vector (match: ||a-b||^2, non-match: l_j) -- pcl.py:171-213's first return value."""

    @staticmethod
    def forward(ctx, desc_a, desc_b, lists, cfg):
        desc_a, desc_b, loss, terms, sums, hard, status, per_term = _run_forward(desc_a, desc_b, lists, cfg, True)
        ctx.save_for_backward(desc_a, desc_b)
        ctx.lists = lists
        ctx.cfg = cfg
        ctx.mark_non_differentiable(hard)
        return per_term[:lists.total], hard

    @staticmethod
    def backward(ctx, grad_vec, _unused):
        lib = _lib.get()
        desc_a, desc_b = ctx.saved_tensors
        lists, cfg = ctx.lists, ctx.cfg
        P, HW, D = desc_a.shape
        ga = torch.empty_like(desc_a)
        gb = torch.empty_like(desc_b)
        gv = grad_vec.to(torch.float32).contiguous()
        rc = lib.dcn_contrastive_loss_backward(
            _lib.ptr(desc_a), _lib.ptr(desc_b), P, HW, D, _lib.ptr(lists.idx_a), _lib.ptr(lists.idx_b),
            ctypes.cast(lists._off_c, ctypes.c_void_p), _lib.ptr(lists.offsets_dev), ctypes.byref(cfg),
            None, None, None, _lib.ptr(gv), _lib.ptr(ga), _lib.ptr(gb), _lib.stream_ptr())
        _lib.check(rc, "dcn_contrastive_loss_backward(per-term)")
        return ga, gb, None, None


def contrastive_loss(desc_a, desc_b, lists, cfg, want_per_term=False):
    """Fused forward (+ autograd).  desc_*: [num_pairs, HW, D].  Returns
    (loss 0-dim, terms [P,5], sums [P,4], hard_neg int32 [P,4], status int32 [1], per_term or None)."""
    return _ContrastiveLossFn.apply(desc_a, desc_b, lists, cfg, want_per_term)


def per_term_losses(desc_a, desc_b, lists, cfg):
    """(per-pair vector [total], hard_neg int32 [P,4]); the vector is differentiable."""
    return _PerTermFn.apply(desc_a, desc_b, lists, cfg)


class _TripletLossFn(torch.autograd.Function):
    """pixelwise_contrastive_loss.py:104-129 as one fused gather kernel (+ scatter-add backward)."""

    @staticmethod
    def forward(ctx, desc_a, desc_b, non_a, match_b, non_b, alpha):
        lib = _lib.get()
        _lib.require_device(desc_a, desc_b, non_a, match_b, non_b)
        if desc_a.dim() != 3 or desc_a.shape[0] != 1 or desc_a.shape != desc_b.shape:
            raise ValueError("triplet loss: descriptors must be two [1, HW, D] tensors, got %s / %s" %
                             (tuple(desc_a.shape), tuple(desc_b.shape)))
        n, n_match = int(non_a.numel()), int(match_b.numel())
        if int(non_b.numel()) != n or n_match < 1 or n % n_match:
            raise ValueError("triplet loss: %d non-matches for %d matches (must be a whole multiple, pcl.py:113-120)" %
                             (n, n_match))
        desc_a = desc_a.contiguous().float()
        desc_b = desc_b.contiguous().float()
        non_a, match_b, non_b = (t.contiguous().to(torch.int64) for t in (non_a, match_b, non_b))
        _, hw, d = desc_a.shape
        dev = desc_a.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        status = torch.empty(1, dtype=torch.int32, device=dev)
        ws = torch.empty(lib.dcn_triplet_loss_workspace_bytes(n), dtype=torch.uint8, device=dev)
        rc = lib.dcn_triplet_loss_forward(_lib.ptr(desc_a), _lib.ptr(desc_b), hw, d, _lib.ptr(non_a), _lib.ptr(match_b),
                                          _lib.ptr(non_b), n, n_match, float(alpha), _lib.ptr(loss), _lib.ptr(status),
                                          _lib.ptr(ws), _lib.stream_ptr())
        _lib.check(rc, "dcn_triplet_loss_forward")
        ctx.save_for_backward(desc_a, desc_b, non_a, match_b, non_b)
        ctx.alpha = float(alpha)
        ctx.status = status
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _lib.get()
        desc_a, desc_b, non_a, match_b, non_b = ctx.saved_tensors
        _, hw, d = desc_a.shape
        ga = torch.zeros_like(desc_a)
        gb = torch.zeros_like(desc_b)
        g = grad_loss.to(torch.float32).contiguous()
        rc = lib.dcn_triplet_loss_backward(_lib.ptr(desc_a), _lib.ptr(desc_b), hw, d, _lib.ptr(non_a), _lib.ptr(match_b),
                                           _lib.ptr(non_b), int(non_a.numel()), int(match_b.numel()), ctx.alpha,
                                           _lib.ptr(g), _lib.ptr(ga), _lib.ptr(gb), _lib.stream_ptr())
        _lib.check(rc, "dcn_triplet_loss_backward")
        return ga, gb, None, None, None, None


def triplet_loss(desc_a, desc_b, non_matches_a, matches_b, non_matches_b, alpha):
    """0-dim loss of pcl.py:104-129 for one image pair (descriptors [1, HW, D]); differentiable w.r.t. both maps."""
    return _TripletLossFn.apply(desc_a, desc_b, non_matches_a, matches_b, non_matches_b, alpha)
