// Small-tile variants of the hl32 gather-GEMM (conv_hlx_kernels.hip): shape selection and launcher, used by the launcher of
// the big tiles (conv_hl_kernels.hip), which owns the C ABI of the hl32 path.
#pragma once
#include "conv_shared.h"

namespace dcnconv {

struct HlxShape {
    bool ok = false;
    int kg = 0, bn = 0;          // K groups inside the workgroup (1: 160 x 256 tile, 2: 160 x 128), tile width
    int mtiles = 0, ntiles = 0, nk = 0;
    int splits = 1;              // workgroups per tile along K (partials completed inside the launch, fixed order)
    double cost = 0.0;           // in hl_shape's unit x stages: rows x (columns / 256) x 32-K stages of one workgroup, per round
    size_t ws_bytes = 0, cnt_off = 0;
};

constexpr int kHlxRows = 160;

// group_rows: rows per statistics group of the forward epilogue (0: none; a group is made of whole tiles); taps: filter taps;
// cs: source channels
HlxShape hlx_shape(int M, int cd, int K, int group_rows, int taps, int cs);
int launch_gemm_hlx(GemmConv& p, const HlxShape& g, void* workspace, hipStream_t st);

}  // namespace dcnconv
