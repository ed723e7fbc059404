"""ctypes loader for libdcn_hip.so.  Declares every symbol of include/dcn_hip.h."""
import atexit
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(HERE, "libdcn_hip.so")

_lib = None
_info = {"path": None, "version": None, "hostemu": False}
_reset_hooks = []  # callables run when a different library is loaded (drops cached plan handles)

c_void_p, c_int, c_int64, c_size_t, c_float, c_char_p = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                         ctypes.c_size_t, ctypes.c_float, ctypes.c_char_p)


class LossConfig(ctypes.Structure):
    """struct dcn_loss_config"""
    _fields_ = [("margin", c_float * 4), ("invert", ctypes.c_int32 * 4), ("pixel_weight", ctypes.c_int32 * 4),
                ("m_pixel", c_float), ("image_width", ctypes.c_int32), ("match_loss_weight", c_float),
                ("non_match_loss_weight", c_float), ("scale_by_hard_negatives", ctypes.c_int32),
                ("compose", ctypes.c_int32)]


class ConvDesc(ctypes.Structure):
    """struct dcn_conv_desc"""
    _fields_ = [(k, ctypes.c_int32) for k in ("n", "hin", "win", "cin", "hout", "wout", "cout", "kh", "kw", "stride",
                                              "pad", "dil", "ldc", "group_rows")]


SYMBOLS = {
    # name: (restype, argtypes)
    "dcn_version": (c_char_p, []),
    "dcn_reload_env": (None, []),
    "dcn_release_pooled_buffers": (None, []),
    "dcn_plan_num_activation_slots": (c_int, [c_void_p]),
    "dcn_plan_fused_bn_backward": (c_int, [c_void_p]),
    "dcn_plan_activation_absmax_offset": (c_size_t, [c_void_p]),
    "dcn_plan_num_grad_buckets": (c_int, [c_void_p]),
    "dcn_plan_grad_bucket_first_param": (c_int, [c_void_p, c_int]),
    "dcn_plan_stream_wait_grad_bucket": (c_int, [c_void_p, c_int, c_void_p]),
    "dcn_bn_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                               c_float, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_bn_backward_workspace": (c_size_t, [c_int64, c_int]),
    "dcn_bn_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    "dcn_maxpool_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dcn_maxpool_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dcn_loss_workspace_bytes": (c_size_t, [c_int, c_int64]),
    "dcn_contrastive_loss_forward": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                             c_void_p, ctypes.POINTER(LossConfig), c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_contrastive_loss_backward": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p, ctypes.POINTER(LossConfig), c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_loss_saved_floats": (c_size_t, [c_int64, c_int]),
    "dcn_contrastive_loss_forward_save": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                                  c_void_p, ctypes.POINTER(LossConfig), c_void_p, c_void_p, c_void_p,
                                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_contrastive_loss_backward_saved": (c_int, [c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                    ctypes.POINTER(LossConfig), c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                                    c_void_p, c_void_p]),
    "dcn_loss_exact_workspace_bytes": (c_size_t, [c_int, c_int64, c_int]),
    "dcn_contrastive_loss_backward_saved_exact": (c_int, [c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                          ctypes.POINTER(LossConfig), c_void_p, c_void_p, c_void_p, c_void_p,
                                                          c_void_p, c_void_p, c_void_p]),
    "dcn_fill_bytes": (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    "dcn_plan_create": (c_int, [c_char_p, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "dcn_plan_destroy": (None, [c_void_p]),
    "dcn_plan_forget_saved": (c_int, [c_void_p, c_void_p]),
    "dcn_plan_num_forward_records": (c_int, [c_void_p]),
    "dcn_plan_num_params": (c_int, [c_void_p]),
    "dcn_plan_num_bn": (c_int, [c_void_p]),
    "dcn_plan_param_info": (c_int, [c_void_p, c_int, c_char_p, c_int, ctypes.POINTER(c_int64), ctypes.POINTER(c_int)]),
    "dcn_plan_bn_info": (c_int, [c_void_p, c_int, c_char_p, c_int, ctypes.POINTER(c_int64)]),
    "dcn_plan_saved_bytes": (c_size_t, [c_void_p]),
    "dcn_plan_workspace_bytes": (c_size_t, [c_void_p]),
    "dcn_plan_forward_flops": (ctypes.c_double, [c_void_p]),
    "dcn_plan_profile_begin": (c_int, [c_void_p]),
    "dcn_plan_profile_end": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64),
                                     ctypes.POINTER(ctypes.c_double)]),
    "dcn_plan_profile_end3": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64),
                                      ctypes.POINTER(ctypes.c_double)]),
    "dcn_plan_profile_end_all": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64),
                                         ctypes.POINTER(ctypes.c_double)]),
    "dcn_backbone_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_int, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "dcn_backbone_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dcn_backbone_forward_pair": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_backbone_backward_pair": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                           c_void_p]),
    "dcn_conv_forward": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p]),
    "dcn_conv_gemm_workspace": (c_size_t, [ctypes.POINTER(ConvDesc), c_int]),
    "dcn_conv_num_mtiles": (c_int, [ctypes.POINTER(ConvDesc)]),
    "dcn_conv_dgrad": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_conv_wgrad": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_conv_wgrad_workspace": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "dcn_transpose_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dcn_f16_kpad": (c_int, [c_int]),
    "dcn_split_rows_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "dcn_conv_num_mtiles_f16": (c_int, [ctypes.POINTER(ConvDesc)]),
    "dcn_conv_gemm_workspace_f16": (c_size_t, [ctypes.POINTER(ConvDesc), c_int]),
    "dcn_conv_forward_f16": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_conv_dgrad_f16": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "dcn_split_stem_weights_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "dcn_conv_stem_forward_f16": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                          c_void_p, c_void_p]),
    "dcn_conv_dgrad_bn_num_mtiles_f16": (c_int, [ctypes.POINTER(ConvDesc)]),
    "dcn_conv_dgrad_bn_f16": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_bn_backward_from_partial": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_triplet_loss_workspace_bytes": (c_size_t, [c_int64]),
    "dcn_triplet_loss_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                         c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_triplet_loss_backward": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                          c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_find_correspondences_workspace": (c_size_t, [c_int64]),
    "dcn_find_correspondences": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p]),
    "dcn_mask_nonzero_workspace": (c_size_t, [c_int64]),
    "dcn_mask_nonzero": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_sample_pixels": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_adam_step": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_double, ctypes.c_double,
                              ctypes.c_double, ctypes.c_double, ctypes.c_double, c_int64, c_void_p]),
    "dcn_plan_create_grouped": (c_int, [c_char_p, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "dcn_plan_set_conv_mode": (c_int, [c_void_p, c_int]),
    "dcn_plan_conv_mode": (c_int, [c_void_p]),
    "dcn_conv_wgrad_f16": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "dcn_split_weights_f16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_float, c_void_p]),
    "dcn_split_weights_scaled_f16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_int, c_float, c_void_p]),
    "dcn_conv_forward_fused_f16": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                           c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_split_act_f16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "dcn_grad_blocked_bytes": (c_size_t, [c_int, c_int]),
    "dcn_split_grad_blocked_f16": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dcn_conv_wgrad_workspace_f16": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "dcn_match_statistics_workspace": (c_size_t, [c_int]),
    "dcn_match_statistics": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_find_best_match": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "dcn_find_best_match_workspace": (c_size_t, [c_int]),
    "dcn_upsample_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p]),
    "dcn_upsample_backward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                      c_void_p]),
    "dcn_upsample_backward_tmp_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dcn_split_weights_checked_f16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_float, c_void_p, c_void_p]),
    "dcn_conv_wgrad_hl_eligible": (c_int, [ctypes.POINTER(ConvDesc)]),
    "dcn_conv_wgrad_hl_kind": (c_int, [ctypes.POINTER(ConvDesc)]),
    "dcn_conv_wgrad_workspace_hl": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "dcn_conv_wgrad_hl": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dcn_conv_hl_eligible": (c_int, [ctypes.POINTER(ConvDesc), c_int]),
    "dcn_conv_num_mtiles_hl": (c_int, [ctypes.POINTER(ConvDesc)]),
    "dcn_conv_tile_rows_hl": (c_int, [ctypes.POINTER(ConvDesc), c_int]),
    "dcn_conv_gemm_workspace_hl": (c_size_t, [ctypes.POINTER(ConvDesc), c_int]),
    "dcn_conv_hl_shape_info": (c_int, [ctypes.POINTER(ConvDesc), c_int, ctypes.POINTER(c_int)]),
    "dcn_split_act_hl32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "dcn_split_weights_hl32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                       c_void_p]),
    "dcn_conv_forward_hl": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p]),
    "dcn_conv_dgrad_hl": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
}

ERRORS = {-1: "DCN_E_INVALID (bad argument)", -2: "DCN_E_LAUNCH (kernel launch failed)",
          -3: "DCN_E_UNSUPPORTED"}


def load(path=None):
    """Load the shared library (idempotent).  ``path`` is only given by tests (host-emulation build)."""
    global _lib
    if _lib is not None and (path is None or os.path.abspath(path) == _info["path"]):
        return _lib
    p = os.path.abspath(path or os.environ.get("DCN_HIP_LIBRARY", DEFAULT_PATH))
    if not os.path.exists(p):
        raise RuntimeError(
            "dcn_hip: %s not found.  Build the gfx950 kernels first: `python __graft_entry__.py` "
            "(or dcn_hip.build.build_library()).  There is no CPU / PyTorch fallback for this path." % p)
    lib = ctypes.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here == the library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    ver = lib.dcn_version().decode()
    for hook in _reset_hooks:
        hook()
    atexit.register(lib.dcn_release_pooled_buffers)   # (the per-stream arrival-word buffers, csrc/conv_hlx_kernels.hip)
    _lib = lib
    _info.update(path=p, version=ver, hostemu=("hostemu" in ver))
    return _lib


def get():
    return _lib if _lib is not None else load()


def library_info():
    get()
    return dict(_info)


def is_hostemu():
    get()
    return _info["hostemu"]


def check(rc, what):
    if rc != 0:
        raise RuntimeError("dcn_hip: %s failed: %s" % (what, ERRORS.get(rc, rc)))


def require_device(*tensors):
    """The shipped library only takes GPU tensors; the host-emulation build (tests) only CPU tensors."""
    emu = is_hostemu()
    for t in tensors:
        if t is None:
            continue
        if emu and t.is_cuda:
            raise RuntimeError("dcn_hip (host-emulation test build) got a GPU tensor")
        if not emu and not t.is_cuda:
            raise RuntimeError("dcn_hip: tensor is on %s -- the dense-correspondence hot path runs on the MI355X only "
                               "(no CPU fallback); move the module and its inputs with .cuda()" % t.device)


def stream_ptr():
    if is_hostemu():
        return None
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())
