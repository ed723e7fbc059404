"""No kernel of the shipped gfx950 library may use scratch (the 256-row stream-K instantiation of
conv_gemm_hl_kernel spilled 416 bytes per lane, the 192-row one 16-24).  Reads the ``amdhsa.kernels`` notes of the code objects
inside libdcn_hip.so (tools/kernel_resources.py) -- the binary that travels to the GPU box, not a separate compile.  CPU only."""
import os
import sys

import pytest

from helpers import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def rows():
    from dcn_hip import build
    import kernel_resources
    return kernel_resources.kernels(build.build_library())


def test_every_kernel_family_is_present(rows):
    names = " ".join(r["name"] for r in rows)
    for family in ("conv_gemm_hl_kernel", "conv_gemm_hlx_kernel", "conv_wgrad_hl_kernel", "conv_wgrad_hlrp_kernel",
                   "conv_gemm_f16_kernel", "conv_wgrad_f16_kernel", "conv_gemm_kernel", "conv_wgrad_kernel",
                   "loss_fwd", "bn_apply", "adam"):
        assert family in names, family
    assert len(rows) >= 150
    # all ten instantiations of the wide-layer kernel: <TR, SK, MT> -- 256- / 192-row with and without stream-K, 320-row without
    assert sum("19conv_gemm_hl_kernelI" in r["name"] for r in rows) == 10


def test_no_kernel_uses_scratch(rows):
    bad = [(r["name"], r["private_segment_fixed_size"], r["vgpr_spill_count"]) for r in rows
           if r["private_segment_fixed_size"] != 0 or r["vgpr_spill_count"] != 0 or str(r["uses_dynamic_stack"]).lower() == "true"]
    assert not bad, bad


def test_matrix_kernels_fit_their_occupancy(rows):
    """The LDS-DMA GEMMs are written for ONE 512-work-item workgroup per CU: <= 256 VGPRs (+ AGPRs 0) and <= 160 KB of LDS."""
    for r in rows:
        if "conv_gemm_hl" in r["name"] or "conv_wgrad_hl" in r["name"]:
            assert r["vgpr_count"] <= 256 and r["group_segment_fixed_size"] <= 160 * 1024, r
