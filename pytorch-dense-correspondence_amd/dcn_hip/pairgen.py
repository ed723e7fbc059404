"""Host side of the pair-generation kernels (csrc/pairgen_kernels.hip)."""
import ctypes

import numpy as np
import torch

from . import _lib


def _f32_host(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32).reshape(-1))
    if a.size != n:
        raise ValueError("expected %d matrix entries, got %d" % (n, a.size))
    return a


def invert_rigid(T):
    """Inverse of a 4x4 rigid transform (correspondence_finder.py:52-60)."""
    T = np.asarray(T, dtype=np.float64)
    out = np.eye(4)
    out[:3, :3] = T[:3, :3].T
    out[:3, 3] = -T[:3, :3].T.dot(T[:3, 3])
    return out


def find_correspondences(depth_a, depth_b, K, pose_a, pose_b, cand_u, cand_v):
    """depth_*: [H, W] device tensors (int16 / uint16 millimetres, same bits), poses: 4x4 camera-to-world (host),
    cand_*: int64 device tensors.  -> (u_a, v_a int64, u_b, v_b float32), candidate order kept (one host sync for the count)."""
    lib = _lib.get()
    _lib.require_device(depth_a, depth_b, cand_u, cand_v)
    if depth_a.shape != depth_b.shape or depth_a.dim() != 2 or depth_a.element_size() != 2:
        raise ValueError("depth images must be two [H, W] 16-bit tensors of the same shape")
    h, w = int(depth_a.shape[0]), int(depth_a.shape[1])
    depth_a, depth_b = depth_a.contiguous(), depth_b.contiguous()
    cand_u, cand_v = cand_u.contiguous().long(), cand_v.contiguous().long()
    n = int(cand_u.numel())
    dev = depth_a.device
    Kh = _f32_host(K, 9)
    Kih = _f32_host(np.linalg.inv(np.asarray(K, dtype=np.float64)), 9)
    Ta = _f32_host(pose_a, 16)
    Tbi = _f32_host(invert_rigid(pose_b), 16)
    ua = torch.empty(n, dtype=torch.int64, device=dev)
    va = torch.empty(n, dtype=torch.int64, device=dev)
    ub = torch.empty(n, dtype=torch.float32, device=dev)
    vb = torch.empty(n, dtype=torch.float32, device=dev)
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    ws = torch.empty(lib.dcn_find_correspondences_workspace(n), dtype=torch.uint8, device=dev)
    hp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.dcn_find_correspondences(_lib.ptr(depth_a), _lib.ptr(depth_b), h, w, hp(Kh), hp(Kih), hp(Ta), hp(Tbi),
                                      _lib.ptr(cand_u), _lib.ptr(cand_v), n, _lib.ptr(ua), _lib.ptr(va), _lib.ptr(ub),
                                      _lib.ptr(vb), _lib.ptr(cnt), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, "dcn_find_correspondences")
    c = int(cnt.item())
    return ua[:c], va[:c], ub[:c], vb[:c]


def mask_nonzero(mask):
    """flat indices of the non-zero pixels of a float mask, increasing; -> (list int64 [HW], count int64 [1]) on the device."""
    lib = _lib.get()
    m = mask.reshape(-1).contiguous().float()
    _lib.require_device(m)
    hw = int(m.numel())
    lst = torch.empty(hw, dtype=torch.int64, device=m.device)
    cnt = torch.empty(1, dtype=torch.int64, device=m.device)
    ws = torch.empty(lib.dcn_mask_nonzero_workspace(hw), dtype=torch.uint8, device=m.device)
    _lib.check(lib.dcn_mask_nonzero(_lib.ptr(m), hw, _lib.ptr(lst), _lib.ptr(cnt), _lib.ptr(ws), _lib.stream_ptr()),
               "dcn_mask_nonzero")
    return lst, cnt


def sample_pixels(rand, n, w, h, pixel_list=None, count=None):
    """rand: float32 [2, n] (uniform over the image) or [n] (over ``pixel_list[:count]``) -> (u, v) float32 [n]."""
    lib = _lib.get()
    rand = rand.contiguous().float()
    _lib.require_device(rand)
    u = torch.empty(n, dtype=torch.float32, device=rand.device)
    v = torch.empty(n, dtype=torch.float32, device=rand.device)
    rc = lib.dcn_sample_pixels(_lib.ptr(rand), n, w, h, _lib.ptr(pixel_list), _lib.ptr(count), _lib.ptr(u), _lib.ptr(v),
                               _lib.stream_ptr())
    _lib.check(rc, "dcn_sample_pixels")
    return u, v
