"""Shared helpers for the test-suite (not a test module)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pytorch-dense-correspondence_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def use_emulation_library():
    """CPU tests: load the host-emulation build of the kernels (tests/hostemu) into dcn_hip."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hostemu"))
    import build_emu
    from dcn_hip import _lib
    path = build_emu.build()
    _lib.load(path)
    assert _lib.is_hostemu()
    return _lib


def use_gfx950_library():
    """GPU tests: the shipped libdcn_hip.so (must already be built: __graft_entry__.build())."""
    from dcn_hip import _lib
    _lib.load(_lib.DEFAULT_PATH)
    assert not _lib.is_hostemu()
    return _lib


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def load_golden_loss(path):
    z = np.load(path, allow_pickle=False)
    cfg = {}
    for k, v in zip(z["cfg_keys"], z["cfg_vals"]):
        k = str(k)
        cfg[k] = bool(v) if k.startswith(("use_", "scale_")) else float(v)
    return z, cfg


def lists_from_golden(z, device="cpu"):
    t = lambda k: torch.tensor(z[k]).to(device)
    return (t("matches_a"), t("matches_b"), t("masked_a"), t("masked_b"), t("background_a"), t("background_b"),
            t("blind_a"), t("blind_b"))
