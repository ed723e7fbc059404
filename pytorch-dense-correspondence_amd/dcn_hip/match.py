"""Host side of the best-match kernel (csrc/match_kernels.hip)."""
import torch

from . import _lib


def find_best_matches(res, queries, mask=None, return_norm_diffs=False):
    """res: [H, W, D] (or [HW, D]) descriptor image, queries: [Q, D] -> (best_flat_idx int64 [Q], best_dist [Q],
    norm_diffs [Q, H, W] or None).  Same arithmetic as dense_correspondence_network.py:541-547 for every query."""
    lib = _lib.get()
    shape = res.shape
    d = int(shape[-1])
    res2 = res.reshape(-1, d).contiguous().float()
    q = queries.reshape(-1, d).contiguous().float()
    m = None if mask is None else (mask.reshape(-1) != 0).to(torch.uint8).contiguous()   # non-zero == object, like the reference
    _lib.require_device(res2, q, m)
    hw, nq = res2.shape[0], q.shape[0]
    dev = res2.device
    idx = torch.empty(nq, dtype=torch.int64, device=dev)
    dist = torch.empty(nq, dtype=torch.float32, device=dev)
    nd = torch.empty(nq, hw, dtype=torch.float32, device=dev) if return_norm_diffs else None
    ws = torch.empty(lib.dcn_find_best_match_workspace(nq), dtype=torch.uint8, device=dev)
    rc = lib.dcn_find_best_match(_lib.ptr(res2), hw, d, _lib.ptr(q), nq, _lib.ptr(m), _lib.ptr(idx), _lib.ptr(dist),
                                 _lib.ptr(nd), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, "dcn_find_best_match")
    if nd is not None and len(shape) == 3:
        nd = nd.view(nq, shape[0], shape[1])
    return idx, dist, nd


def match_statistics(res_b, queries, gt_idx, mask=None):
    """res_b: [H, W, D] descriptor image, queries: [Q, D] (= res_a at the query pixels), gt_idx: int64 [Q] flat index
    (u + W*v) of the ground-truth match in image b, mask: optional [H, W] (non-zero = on the object).
    One pass over res_b (evaluation.py:1046-1100 for every query) -> dict of device tensors, "image" / "masked" pairs as
    [2, Q]: best_idx, best_dist, count (pixels closer than the ground truth), dist_sum (their pixel distance to it), and
    gt_dist [Q]."""
    lib = _lib.get()
    h, w, d = (int(s) for s in res_b.shape)
    res2 = res_b.reshape(-1, d).contiguous().float()
    q = queries.reshape(-1, d).contiguous().float()
    g = gt_idx.reshape(-1).contiguous().to(torch.int64)
    m = None if mask is None else (mask.reshape(-1) != 0).to(torch.uint8).contiguous()
    _lib.require_device(res2, q, g, m)
    nq, dev = q.shape[0], res2.device
    out = {"best_idx": torch.empty(2, nq, dtype=torch.int64, device=dev),
           "best_dist": torch.empty(2, nq, dtype=torch.float32, device=dev),
           "count": torch.empty(2, nq, dtype=torch.int32, device=dev),
           "dist_sum": torch.empty(2, nq, dtype=torch.float32, device=dev),
           "gt_dist": torch.empty(nq, dtype=torch.float32, device=dev)}
    ws = torch.empty(lib.dcn_match_statistics_workspace(nq), dtype=torch.uint8, device=dev)
    rc = lib.dcn_match_statistics(_lib.ptr(res2), h * w, w, d, _lib.ptr(q), _lib.ptr(g), nq, _lib.ptr(m),
                                  _lib.ptr(out["best_idx"]), _lib.ptr(out["best_dist"]), _lib.ptr(out["count"]),
                                  _lib.ptr(out["dist_sum"]), _lib.ptr(out["gt_dist"]), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, "dcn_match_statistics")
    return out
