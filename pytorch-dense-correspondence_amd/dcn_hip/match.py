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
    m = None if mask is None else mask.reshape(-1).to(torch.uint8).contiguous()
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
