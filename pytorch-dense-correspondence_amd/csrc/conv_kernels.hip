// K1/K3/K4/K5/K6: NHWC fp32 convolutions as implicit GEMMs on the gfx950 matrix cores
// (v_mfma_f32_32x32x2_f32: exact-fp32 MFMA, 64 FLOP/clk/SIMD -- the only matrix path that can hold the
// 1e-4 parity bound against the fp32 CPU reference; gfx950 has no xf32).
//
//   forward : out[m][n]  = sum_{tap,c} in[pix(m,tap)][c] * W[n][tap][c]            (+ bias[n])
//   dgrad   : din[m][c]  = sum_{tap,n} dout[pixT(m,tap)][n] * Wt[c][tap][n]        (+ add[m][c])
//   wgrad   : dW[n][tap][c] = sum_m dout[m][n] * in[pix(m,tap)][c]
//
// forward and dgrad are the same "gather-GEMM" kernel (M = N*H*W destination pixels, N = destination
// channels, K = taps * source channels); dgrad gathers with the transposed index map and a channel-
// transposed weight copy, so stride-2 convolutions need no separate kernel.  wgrad reduces over pixels
// (split over the grid, partial slabs + a deterministic second pass).
//
// Tiling (wave64): 256 work-items = 4 wavefronts in a 2x2 arrangement, workgroup tile 128 x 128, each
// wavefront 64 x 64 = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).  K advances 16 at a time through a
// double-buffered LDS tile (one barrier per K step); global->LDS goes through registers because the
// gather needs per-row zero fill (padding / dilation halo / ragged M) which LDS-DMA cannot express.
//   gather-GEMM LDS image: [row][16 k + 4 pad] so that one ds_read_b128 per lane fetches 4 consecutive k
//   for row (lane & 31) at k-offset 4*(lane >> 5): MFMA step j then contracts k = {j, 4 + j} -- any
//   k permutation is legal as long as A and B use the same one.  The 80-byte row pitch makes the
//   16-lane ds_read_b128 groups hit 16 distinct 16-byte slots (conflict-free).
//   wgrad LDS image: [pixel][128 channels], fragments by ds_read_b64 (lane i holds channels 2i, 2i+1 of
//   pixel 2q + (lane >> 5)): 64 consecutive dwords per half-wave, conflict-free without padding.
#include "conv_shared.h"
#include "dcn_tuning.h"

namespace {

using namespace dcnconv;

// One (tile, K range) segment: K tiles [k0, k1) of output tile `tile`.  A full range ends in the epilogue, a partial
// one (stream-K) parks the raw accumulators in `slot` (fragment order: fully coalesced) for the fix-up kernel.
// All gather arithmetic is branch-free (selects + clamped addresses); the main loop is software-pipelined.
template <int WM, int TM, int TN, int BK, bool TR>
__device__ __forceinline__ void gemm_segment(const GemmConv& p, float* lds, int tile, int k0, int k1, int nk, float* slot) {
    using G = GemmGeo<WM, TM, TN, BK>;
    constexpr int BM = G::BM, BN = G::BN, LDK = G::LDK, KQ = G::KQ, ROWS = G::ROWS, PA = G::PA, PB = G::PB,
                  kStage = G::kStage;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm_ = WM == 2 ? (wv >> 1) : 0, wn_ = WM == 2 ? (wv & 1) : wv;
    const int mt = fdiv(tile, p.div_nt), nt = tile - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-thread gather coordinates: this thread stages rows r0 + ROWS*j at k-quad kq
    const int kq = tid % KQ, r0 = tid / KQ;
    int by[PA], bx[PA], pixbase[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int m = m0 + r0 + ROWS * j;
        const int mm = m < p.M ? m : 0;  // (v_cndmask, not a branch)
        const int img = fdiv(mm, p.div_hw), rem = mm - img * p.div_hw.d;
        const int y = fdiv(rem, p.div_w), x = rem - y * p.div_w.d;
        by[j] = TR ? y + p.pad : y * p.stride - p.pad;
        bx[j] = TR ? x + p.pad : x * p.stride - p.pad;
        if (m >= p.M || r0 + ROWS * j >= BM) by[j] = -(1 << 28);
        pixbase[j] = img * p.hs * p.ws;
    }
    int wrow[PB];      // element offset of this thread's weight row (0 when the row is out of range)
    unsigned wokm = 0;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int n = n0 + r0 + ROWS * j;
        const bool ok = (n < p.cd) & ((r0 + ROWS * j) < BN);
        wokm |= (ok ? 1u : 0u) << j;
        wrow[j] = (ok ? n : 0) * p.K + kq * 4;
    }
    const int smask = p.stride - 1;
    float4 ra[PA], rb[PB];
    unsigned okm = 0;  // validity bits of the tile held in ra / rb (applied when it is written to LDS, AFTER the MFMAs:
                       // selecting on freshly loaded data here would put the load latency in front of the MFMA block)

    auto load_tile = [&](int kt) {
        const int k = kt * BK + kq * 4;
        const bool kval = k < p.K;
        const int kk = kval ? k : 0;
        const int tap = fdiv(kk, p.div_cs), c = kk - tap * p.cs;
        const int r = fdiv(tap, p.div_kw), s = tap - r * p.kw;
        const int dy = r * p.dil, dx = s * p.dil;
        okm = 0;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            int sy, sx;
            bool ok = kval;
            if (TR) {
                const int ny = by[j] - dy, nx = bx[j] - dx;
                ok = ok & ((ny | nx) >= 0) & (((ny | nx) & smask) == 0);
                sy = ny >> p.sshift; sx = nx >> p.sshift;
            } else {
                sy = by[j] + dy; sx = bx[j] + dx;
            }
            ok = ok & ((unsigned)sy < (unsigned)p.hs) & ((unsigned)sx < (unsigned)p.ws);
            int off = (pixbase[j] + sy * p.ws + sx) * p.cs + c;
            off = ok ? off : 0;
            ra[j] = *reinterpret_cast<const float4*>(p.src + off);
            okm |= (ok ? 1u : 0u) << j;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const bool ok = kval & (((wokm >> j) & 1u) != 0);
            rb[j] = *reinterpret_cast<const float4*>(p.wm + (ok ? wrow[j] + kt * BK : 0));
            okm |= (ok ? 1u : 0u) << (16 + j);
        }
    };
    auto store_tile = [&](int stage) {
        float* a = lds + stage * kStage;
        float* b = a + BM * LDK;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const bool ok = (okm >> j) & 1u;
            const float4 v = ra[j];
            if (BM % ROWS == 0 || r0 + ROWS * j < BM)
                *reinterpret_cast<float4*>(a + (r0 + ROWS * j) * LDK + kq * 4) =
                    make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const bool ok = (okm >> (16 + j)) & 1u;
            const float4 v = rb[j];
            if (BN % ROWS == 0 || r0 + ROWS * j < BN)
                *reinterpret_cast<float4*>(b + (r0 + ROWS * j) * LDK + kq * 4) =
                    make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- software-pipelined main loop (one barrier per K tile, two LDS stages, global loads one full tile ahead)
    //   phase A: MFMAs on fragment set F0 | in their shadow: ds_read F1 (second half of this tile), zero-fill selects
    //            + ds_write of tile kt+1 (its global loads were issued a whole iteration ago) into the other stage
    //   barrier: everybody has written stage nxt / finished reading stage cur's first half
    //   phase B: MFMAs on F1 | in their shadow: address math + global loads of tile kt+2, ds_read of tile kt+1's F0
    // so that a lone wavefront per SIMD (few workgroups per CU at small batch) still keeps the matrix pipe fed.
    // The loop body is branch-free (tile indices are clamped instead of guarded: the extra tile loaded / staged in
    // the last iterations is never consumed) so that each phase is ONE scheduling region, and
    // sched_group_barrier pins the interleave "1 MFMA, a few VALU/LDS/VMEM instructions" inside it.
    static_assert(BK == 16, "the pipelined loop is written for two 8-deep fragment groups per K tile");
    const int fi = lane & 31, fh = lane >> 5;
    const int a_off = (wm_ * 32 * TM + fi) * LDK + 4 * fh;
    const int b_off = BM * LDK + (wn_ * 32 * TN + fi) * LDK + 4 * fh;
    float fa[2][TM][4], fb[2][TN][4];
    auto read_frags = [&](int stage, int ks, int set) {
        const float* a = lds + stage * kStage + a_off + ks * 8;
        const float* b = lds + stage * kStage + b_off + ks * 8;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(a + t * 32 * LDK);
            fa[set][t][0] = v.x; fa[set][t][1] = v.y; fa[set][t][2] = v.z; fa[set][t][3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(b + t * 32 * LDK);
            fb[set][t][0] = v.x; fb[set][t][1] = v.y; fb[set][t][2] = v.z; fb[set][t][3] = v.w;
        }
    };
    auto mfma_steps = [&](int set) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][tm][j], fb[set][tn][j], acc[tm][tn], 0, 0, 0);
    };
    constexpr int kMfmaPerPhase = 4 * TM * TN;
    load_tile(k0);
    store_tile(0);
    load_tile(k0 + 1 < k1 ? k0 + 1 : k0);
    __syncthreads();
    read_frags(0, 0, 0);
    for (int kt = k0; kt < k1; ++kt) {
        const int cur = (kt - k0) & 1;
        // ---- phase A
        read_frags(cur, 1, 1);
        store_tile(cur ^ 1);
        mfma_steps(0);
#pragma unroll
        for (int i = 0; i < kMfmaPerPhase; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x306, 48 / kMfmaPerPhase + 1, 0);  // VALU | SALU | DS
        }
        __syncthreads();
        // ---- phase B
        load_tile(kt + 2 < k1 ? kt + 2 : k1 - 1);
        read_frags(cur ^ 1, 0, 0);
        mfma_steps(1);
#pragma unroll
        for (int i = 0; i < kMfmaPerPhase; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x126, 112 / kMfmaPerPhase + 1, 0);  // VALU | SALU | VMEM read | DS read
        }
    }
    __syncthreads();  // LDS is reused below (reduction scratch) / by the next segment

    if (k0 == 0 && k1 == nk) {
        gemm_epilogue<WM, TM, TN, BK>(p, acc, mt, nt, lds);
    } else {
        float* o = slot + wv * (TM * TN * 16 * 64) + lane;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[((tm * TN + tn) * 16 + r) * 64] = acc[tm][tn][r];
    }
}

// Data-parallel form: one workgroup per output tile.
// Stream-K form (SK): the tiles x K-tiles iteration space is cut into equal contiguous ranges, one per workgroup, so
// that all 256 CUs finish together even when the layer has too few tiles to fill them evenly (small batch); tiles that
// end up split across workgroups are completed by conv_gemm_fixup_kernel.
template <int WM, int TM, int TN, int BK, bool TR, bool SK>
__global__ void __launch_bounds__(NT, (TM * TN == 4 ? 3 : 1))   // 64x64 wave tile: cap at 168 registers -> 3 waves / SIMD
conv_gemm_kernel(GemmConv p) {
    using G = GemmGeo<WM, TM, TN, BK>;
    __shared__ __attribute__((aligned(16))) float lds[2 * G::kStage];
    const int nk = (p.K + BK - 1) / BK;
    if (!SK) {
        gemm_segment<WM, TM, TN, BK, TR>(p, lds, xcd_remap(blockIdx.x, p.mtiles * p.ntiles), 0, nk, nk, nullptr);
    } else {
        const int g = xcd_remap(blockIdx.x, gridDim.x);  // neighbouring unit ranges on the same XCD (shared L2 lines)
        int u = g * p.sk_units;
        const int total = p.mtiles * p.ntiles * nk;
        const int u_end = min(total, u + p.sk_units);
        bool first = true;
        while (u < u_end) {
            const int tile = fdiv(u, p.div_nk), k0 = u - tile * nk;
            const int k1 = min(nk, k0 + (u_end - u));
            gemm_segment<WM, TM, TN, BK, TR>(p, lds, tile, k0, k1, nk,
                                             p.sk_partial + (int64_t)(2 * g + (first ? 0 : 1)) * G::kSlotFloats);
            u += k1 - k0;
            first = false;
            __syncthreads();  // the epilogue's reduction scratch lives in the LDS tile the next segment overwrites
        }
    }
}

// Completes the tiles that stream-K split: adds the parked accumulators in workgroup order (deterministic) and runs
// the normal epilogue.  One workgroup per tile; tiles computed whole by a single workgroup return at once.
template <int WM, int TM, int TN, int BK>
__global__ void __launch_bounds__(NT)
conv_gemm_fixup_kernel(GemmConv p) {
    using G = GemmGeo<WM, TM, TN, BK>;
    __shared__ float red[6 * G::BN];
    const int nk = (p.K + BK - 1) / BK;
    const int tile = blockIdx.x;
    const int ua = tile * nk, ub = ua + nk - 1;
    const int ga = ua / p.sk_units, gb = ub / p.sk_units;
    if (ga == gb) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int g = ga; g <= gb; ++g) {
        const int first_tile = (g * p.sk_units) / nk;  // the tile this workgroup's FIRST segment belongs to
        const float* o = p.sk_partial + (int64_t)(2 * g + (first_tile == tile ? 0 : 1)) * G::kSlotFloats +
                         wv * (TM * TN * 16 * 64) + lane;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] += o[((tm * TN + tn) * 16 + r) * 64];
    }
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    gemm_epilogue<WM, TM, TN, BK>(p, acc, mt, nt, red);
}

// Tile-shape choice (fp32 MFMA is slow enough that load balance across the 256 CUs matters more than tile reuse):
//   N tile  64 when the layer has at most 64 output channels (no MFMA work on padding), else 128
//   M tile  128, or 64 when 128-row tiles would give the chip fewer than ~3 workgroups per CU (tail effect)
int gemm_tile_m(int M, int cd, int align) {
    int bm;
    const int ntiles = dcn::ceil_div(cd, cd <= 64 ? 64 : 128);
    bm = dcn::ceil_div(M, 128) * ntiles < 3 * 256 ? 64 : 128;
    if (const int v = dcn::tuning().gemm_tile_m) {  // tuning / test override (dcn_tuning.h)
        if (v == 32 || v == 64 || v == 128) bm = (v == 32 && cd <= 64) ? 64 : v;
    }
    if (align > 0 && (align % bm) != 0) bm = 64;   // rows per batch-norm group: a multiple of 64 (checked by the launcher)
    return bm;
}

// Stream-K decision.  With one workgroup per tile the chip processes ceil(tiles / 256) "rounds"; when the last round
// is mostly empty (small batch: a few hundred tiles) the equal-work split recovers the idle CUs.
struct GemmShape {
    int bm, bn, mtiles, ntiles, nk, sk_wgs, sk_units;
    bool narrow, sk;
    size_t ws_bytes;
};
GemmShape gemm_shape(int M, int cd, int K, int align = 0) {
    GemmShape g;
    g.narrow = cd <= 64;
    g.bm = gemm_tile_m(M, cd, align);
    g.bn = g.bm == 32 ? 128 : (g.narrow ? 64 : 128);
    g.mtiles = dcn::ceil_div(M, g.bm);
    g.ntiles = dcn::ceil_div(cd, g.bn);
    g.nk = dcn::ceil_div(K, 16);
    const int tiles = g.mtiles * g.ntiles;
    const double rounds = tiles / 256.0;
    const double waste = 1.0 - rounds / (double)(int)(rounds + 0.999999);
    g.sk = tiles < 8 * 256 && waste > 0.08 && g.nk >= 32;  // short K loops (1x1 convs) do not amortise the fix-up pass
    int wgs = 512;
    if (dcn::tuning().gemm_sk >= 0) {  // tuning / test override: 0 = off, 1 = as decided, N > 1 = force N workgroups
        const int v = dcn::tuning().gemm_sk;
        if (v == 0) g.sk = false;
        if (v > 1) { g.sk = g.nk >= 2; wgs = v; }
    }
    g.sk_wgs = 0; g.sk_units = 0; g.ws_bytes = 0;
    if (g.sk) {
        const int64_t total = (int64_t)tiles * g.nk;
        if (total >= ((int64_t)1 << 30)) { g.sk = false; return g; }
        if (wgs > total / 2) wgs = (int)(total / 2) > 0 ? (int)(total / 2) : 1;  // at least two K tiles per workgroup
        g.sk_units = (int)((total + wgs - 1) / wgs);
        g.sk_wgs = (int)((total + g.sk_units - 1) / g.sk_units);
        g.ws_bytes = (size_t)2 * g.sk_wgs * g.bm * g.bn * sizeof(float);
    }
    return g;
}

int launch_gemm(GemmConv& p, void* workspace, hipStream_t st, int align = 0) {
    if (align > 0 && (align % 64) != 0) return DCN_E_UNSUPPORTED;
    if (p.stride != 1 && p.stride != 2 && p.stride != 4) return DCN_E_UNSUPPORTED;
    p.sshift = p.stride == 1 ? 0 : (p.stride == 2 ? 1 : 2);
    p.div_hw = make_fastdiv(p.hd * p.wd);
    p.div_w = make_fastdiv(p.wd);
    p.div_cs = make_fastdiv(p.cs);
    p.div_kw = make_fastdiv(p.kw);
    // 32-bit element offsets inside the kernel
    if ((int64_t)p.M / (p.hd * p.wd) * p.hs * p.ws * p.cs >= ((int64_t)1 << 31) || (int64_t)p.cd * p.K >= ((int64_t)1 << 31))
        return DCN_E_UNSUPPORTED;
    const GemmShape g = gemm_shape(p.M, p.cd, p.K, align);
    const bool sk = g.sk && workspace != nullptr;
    p.mtiles = g.mtiles;
    p.ntiles = g.ntiles;
    p.div_nt = make_fastdiv(g.ntiles);
    p.div_nk = make_fastdiv(g.nk);
    p.sk_units = sk ? g.sk_units : 0;
    p.sk_partial = sk ? (float*)workspace : nullptr;
    const dim3 grid(sk ? g.sk_wgs : g.mtiles * g.ntiles), fgrid(g.mtiles * g.ntiles), block(NT);
#define DCN_GEMM(WM, TM, TN)                                                                                          \
    do {                                                                                                              \
        if (sk) {                                                                                                     \
            if (p.transposed) hipLaunchKernelGGL((conv_gemm_kernel<WM, TM, TN, 16, true, true>), grid, block, 0, st, p); \
            else hipLaunchKernelGGL((conv_gemm_kernel<WM, TM, TN, 16, false, true>), grid, block, 0, st, p);           \
            hipLaunchKernelGGL((conv_gemm_fixup_kernel<WM, TM, TN, 16>), fgrid, block, 0, st, p);                      \
        } else {                                                                                                      \
            if (p.transposed) hipLaunchKernelGGL((conv_gemm_kernel<WM, TM, TN, 16, true, false>), grid, block, 0, st, p); \
            else hipLaunchKernelGGL((conv_gemm_kernel<WM, TM, TN, 16, false, false>), grid, block, 0, st, p);          \
        }                                                                                                             \
    } while (0)
    if (g.bm == 32) DCN_GEMM(1, 1, 1);                                     //  32 x 128, waves 1x4
    else if (g.bm == 64) { if (g.narrow) DCN_GEMM(2, 1, 1); else DCN_GEMM(2, 1, 2); }   //  64 x 64 | 64 x 128
    else { if (g.narrow) DCN_GEMM(2, 2, 1); else DCN_GEMM(2, 2, 2); }                   // 128 x 64 | 128 x 128
#undef DCN_GEMM
    return dcn::check_launch();
}

// ------------------------------------------------------------------------------------------- wgrad
constexpr int WK = 16;  // pixels per reduction step

struct WgradConv {
    const float* in;    // [n, hin, win, cin]
    const float* dout;  // [M][ldo]
    float* slab;        // [splits][cout][K]
    int hin, win, cin, hout, wout, cout, kh, kw, stride, pad, dil, ldo, M, K, splits, rows_per_split, ntiles_n,
        ntiles_k;
    FastDiv div_hw, div_w, div_cin, div_kw;
};

// WM = 2: workgroup tile 128 output channels x 128 K columns (2x2 wavefronts);
// WM = 1:                  64 output channels x 256 K columns (1x4) -- layers with <= 64 output channels (stem, layer1)
//                          would otherwise spend half of their MFMAs on zero rows.
template <int WM>
__global__ void __launch_bounds__(NT)
conv_wgrad_kernel(WgradConv p) {
    constexpr int DW = 64 * WM, XW = 64 * (4 / WM), DQ = DW / 4, XQ = XW / 4, DR = NT / DQ, XR = NT / XQ, PD = WK / DR,
                  PX = WK / XR, kStage = WK * (DW + XW);
    __shared__ __attribute__((aligned(16))) float lds[2 * kStage];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm_ = WM == 2 ? (wv >> 1) : 0, wn_ = WM == 2 ? (wv & 1) : wv;
    const int tiles = p.ntiles_n * p.ntiles_k;
    const int bid = xcd_remap(blockIdx.x, tiles * p.splits);
    const int split = bid / tiles, tile = bid - split * tiles;
    const int tn_ = tile / p.ntiles_k, tk_ = tile - tn_ * p.ntiles_k;
    const int n0 = tn_ * DW, j0 = tk_ * XW;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    // this thread stages channel quad dc4 of dout pixel rows dp0 + DR*j, and K-column quad xc4 of input rows xp0 + XR*j
    const int dc4 = tid % DQ, dp0 = tid / DQ, xc4 = tid % XQ, xp0 = tid / XQ;
    const int ncol = n0 + dc4 * 4;
    const bool nval = ncol < p.ldo;          // dout rows are padded to ldo (multiple of 4) with zeros
    const int kcol = j0 + xc4 * 4;
    const bool kval = kcol < p.K;
    const int kc0 = kval ? kcol : 0;
    const int tap = fdiv(kc0, p.div_cin), cc = kc0 - tap * p.cin;
    const int tr = fdiv(tap, p.div_kw), ts = tap - tr * p.kw;
    const int oy = tr * p.dil - p.pad, ox = ts * p.dil - p.pad;
    float4 rd[PD], rx[PX];
    unsigned okm = 0;

    // branch-free (clamped addresses; the zero-fill selects are applied when the tile is written to LDS, after the
    // MFMA block) so the loads are issued early and complete in the shadow of the MFMAs
    auto load_tile = [&](int m_base) {
        okm = 0;
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const int m = m_base + dp0 + DR * j;
            const bool dok = (m < m_end) & nval;
            rd[j] = *reinterpret_cast<const float4*>(p.dout + (dok ? m * p.ldo + ncol : 0));
            okm |= (dok ? 1u : 0u) << j;
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int m = m_base + xp0 + XR * j;
            const bool mval = m < m_end;
            const int mm = mval ? m : 0;
            const int img = fdiv(mm, p.div_hw), rem = mm - img * p.div_hw.d;
            const int y = fdiv(rem, p.div_w), x = rem - y * p.div_w.d;
            const int sy = y * p.stride + oy, sx = x * p.stride + ox;
            const bool ok = mval & kval & ((unsigned)sy < (unsigned)p.hin) & ((unsigned)sx < (unsigned)p.win);
            int off = ((img * p.hin + sy) * p.win + sx) * p.cin + cc;
            off = ok ? off : 0;
            rx[j] = *reinterpret_cast<const float4*>(p.in + off);
            okm |= (ok ? 1u : 0u) << (8 + j);
        }
    };
    auto store_tile = [&](int stage) {
        float* d = lds + stage * kStage;
        float* x = d + WK * DW;
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const bool dok = (okm >> j) & 1u;
            const float4 dv = rd[j];
            *reinterpret_cast<float4*>(d + (dp0 + DR * j) * DW + dc4 * 4) =
                make_float4(dok ? dv.x : 0.f, dok ? dv.y : 0.f, dok ? dv.z : 0.f, dok ? dv.w : 0.f);
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const bool ok = (okm >> (8 + j)) & 1u;
            const float4 xv = rx[j];
            *reinterpret_cast<float4*>(x + (xp0 + XR * j) * XW + xc4 * 4) =
                make_float4(ok ? xv.x : 0.f, ok ? xv.y : 0.f, ok ? xv.z : 0.f, ok ? xv.w : 0.f);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fi = lane & 31, fh = lane >> 5;
    const int nsteps = (m_end - m_begin + WK - 1) / WK;
    // Software-pipelined like the gather-GEMM kernel: two fragment sets (pixel pairs 0-3 / 4-7 of a 16-pixel step),
    // one barrier per step, global loads a full step ahead, branch-free body + sched_group_barrier interleave.
    const int d_off = fh * DW + wm_ * 64 + 2 * fi;
    const int x_off = WK * DW + fh * XW + wn_ * 64 + 2 * fi;
    float2 fd[2][4], fx[2][4];
    auto read_frags = [&](int stage, int half, int set) {
        const float* d = lds + stage * kStage + d_off + half * 8 * DW;
        const float* x = lds + stage * kStage + x_off + half * 8 * XW;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fd[set][q] = *reinterpret_cast<const float2*>(d + 2 * q * DW);
            fx[set][q] = *reinterpret_cast<const float2*>(x + 2 * q * XW);
        }
    };
    auto mfma_steps = [&](int set) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd[set][q].x, fx[set][q].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd[set][q].x, fx[set][q].y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd[set][q].y, fx[set][q].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fd[set][q].y, fx[set][q].y, acc[1][1], 0, 0, 0);
        }
    };
    if (nsteps > 0) {
        load_tile(m_begin);
        store_tile(0);
        load_tile(m_begin + (nsteps > 1 ? WK : 0));
        __syncthreads();
        read_frags(0, 0, 0);
        for (int st = 0; st < nsteps; ++st) {
            const int cur = st & 1;
            read_frags(cur, 1, 1);
            store_tile(cur ^ 1);
            mfma_steps(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x306, 3, 0);
            }
            __syncthreads();
            load_tile(m_begin + (st + 2 < nsteps ? st + 2 : nsteps - 1) * WK);
            read_frags(cur ^ 1, 0, 0);
            mfma_steps(1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x126, 6, 0);
            }
        }
    }
    // fragment (tm, r) <-> output channel n0 + wm*64 + 2*row_r + tm ; (tn, lane) <-> k column j0 + wn*64 + 2*fi + tn
    float* out = p.slab + (int64_t)split * p.cout * p.K;
    const int kc = j0 + wn_ * 64 + 2 * fi;
    if (kc < p.K) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm_ * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * fh) + tm;
                if (n < p.cout)
                    *reinterpret_cast<float2*>(out + (int64_t)n * p.K + kc) = make_float2(acc[tm][0][r], acc[tm][1][r]);
            }
        }
    }
}

// dw[i] = sum_s slab[s][i]   (fixed order: deterministic)
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int64_t n4, int splits) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4* src = reinterpret_cast<const float4*>(slab) + i;
    float4 a = src[0];
    int s = 1;
    for (; s + 4 <= splits; s += 4) {  // four loads in flight; the additions keep the fixed order s = 1, 2, 3, ...
        const float4 b0 = src[(int64_t)s * n4], b1 = src[(int64_t)(s + 1) * n4], b2 = src[(int64_t)(s + 2) * n4],
                     b3 = src[(int64_t)(s + 3) * n4];
        a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
        a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
        a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
        a.x += b3.x; a.y += b3.y; a.z += b3.z; a.w += b3.w;
    }
    for (; s < splits; ++s) {
        const float4 b = src[(int64_t)s * n4];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(dw)[i] = a;
}

// The same sum for MANY splits of a SMALL tensor (the narrow layers' weight gradients: 64 x 576 values in 250 slabs -- one
// work-item per float4 walking 250 slabs four loads at a time took 30 us, three times the GEMM it completes): 16 float4 columns x
// 16 split lanes per workgroup, lane sl sums the slabs sl, sl + 16, ... in order, the 16 lane sums are added in order 0..15.
// Fixed order again (another one than the kernel above: a launch uses one or the other by its split count alone).
__global__ void __launch_bounds__(256)
wgrad_reduce_wide_kernel(const float* __restrict__ slab, float* __restrict__ dw, int64_t n4, int splits) {
    __shared__ float4 part[16][16];
    const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + c;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        const float4* src = reinterpret_cast<const float4*>(slab) + i;
        int s = sl;
        for (; s + 48 < splits; s += 64) {
            const float4 b0 = src[(int64_t)s * n4], b1 = src[(int64_t)(s + 16) * n4], b2 = src[(int64_t)(s + 32) * n4],
                         b3 = src[(int64_t)(s + 48) * n4];
            a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
            a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
            a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
            a.x += b3.x; a.y += b3.y; a.z += b3.z; a.w += b3.w;
        }
        for (; s < splits; s += 16) {
            const float4 b = src[(int64_t)s * n4];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
    }
    part[sl][c] = a;
    __syncthreads();
    if (sl == 0 && i < n4) {
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const float4 b = part[k][c];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(dw)[i] = a;
    }
}

// wt[c][tap][n] = w[n][tap][c]
__global__ void __launch_bounds__(256)
transpose_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int taps, int cin, int ldn) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, c = c0 + tx;
        tile[i][tx] = (n < cout && c < cin) ? w[((int64_t)n * taps + tap) * cin + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, n = n0 + tx;
        if (c < cin && n < ldn) wt[((int64_t)c * taps + tap) * ldn + n] = tile[tx][i];  // n >= cout: zero pad
    }
}

int wgrad_splits(const dcn_conv_desc* c, int* rows_per_split) {
    const int M = c->n * c->hout * c->wout, K = c->kh * c->kw * c->cin;
    const bool narrow = c->cout <= 64;
    const int tiles = dcn::ceil_div(c->cout, narrow ? 64 : 128) * dcn::ceil_div(K, narrow ? 256 : 128);
    // aim for ~8 workgroups per CU, at least 8 reduction steps per workgroup
    int splits = dcn::ceil_div(2048, tiles);
    const int max_by_rows = (M / (8 * WK)) > 1 ? (M / (8 * WK)) : 1;
    if (splits > max_by_rows) splits = max_by_rows;
    // the partial slabs are written and read back once; narrow layers (few, small tiles) may split further
    const int cap = narrow ? 256 : 64;
    if (splits > cap) splits = cap;
    if (splits < 1) splits = 1;
    int rps = dcn::ceil_div(dcn::ceil_div(M, splits), WK) * WK;
    splits = dcn::ceil_div(M, rps);
    *rows_per_split = rps;
    return splits;
}

bool valid_desc(const dcn_conv_desc* c) {
    return c && c->n > 0 && c->hin > 0 && c->win > 0 && c->cin > 0 && (c->cin % 4) == 0 && c->hout > 0 && c->wout > 0 &&
           c->cout > 0 && c->kh > 0 && c->kw > 0 && c->stride > 0 && c->dil > 0 && c->pad >= 0 && c->ldc >= c->cout;
}

}  // namespace

namespace dcnconv {
void launch_wgrad_reduce(const float* slabs, float* dw, int64_t n4, int splits, hipStream_t st) {
    if (splits >= 32) {   // (a count no launch of the wide layers reaches at eight images: their summation order stays what it was)
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)dcn::ceil_div64(n4, 16)), dim3(256), 0, st, slabs, dw, n4, splits);
        return;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)dcn::ceil_div64(n4, 256)), dim3(256), 0, st, slabs, dw, n4,
                       splits);
}
}  // namespace dcnconv

extern "C" int dcn_conv_num_mtiles(const dcn_conv_desc* c) {
    if (!valid_desc(c)) return DCN_E_INVALID;
    return gemm_shape(c->n * c->hout * c->wout, c->cout, c->kh * c->kw * c->cin, c->group_rows).mtiles;
}

extern "C" size_t dcn_conv_gemm_workspace(const dcn_conv_desc* c, int dgrad) {
    if (!valid_desc(c)) return 0;
    if (dgrad) return gemm_shape(c->n * c->hin * c->win, c->cin, c->kh * c->kw * c->ldc).ws_bytes;
    return gemm_shape(c->n * c->hout * c->wout, c->cout, c->kh * c->kw * c->cin, c->group_rows).ws_bytes;
}

extern "C" int dcn_conv_forward(const dcn_conv_desc* c, const float* in, const float* w, const float* bias, float* out,
                                float* bn_partial, void* workspace, void* stream) {
    if (!valid_desc(c) || !in || !w || !out) return DCN_E_INVALID;
    GemmConv p;
    p.src = in; p.wm = w; p.bias = bias; p.add = nullptr; p.dst = out; p.bn_partial = bn_partial; p.out_absmax = nullptr;
    p.hs = c->hin; p.ws = c->win; p.cs = c->cin; p.hd = c->hout; p.wd = c->wout; p.cd = c->cout;
    p.kh = c->kh; p.kw = c->kw; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.ldc = c->ldc;
    p.M = c->n * c->hout * c->wout; p.K = c->kh * c->kw * c->cin; p.transposed = 0; p.relu = 0;
    return launch_gemm(p, workspace, (hipStream_t)stream, c->group_rows);
}

// The description is the FORWARD convolution's; dout is [n,hout,wout,ld = c->ldc], din is [n,hin,win,cin].
extern "C" int dcn_conv_dgrad(const dcn_conv_desc* c, const float* dout, const float* wt, const float* add, float* din,
                              void* workspace, void* stream) {
    if (!valid_desc(c) || !dout || !wt || !din || (c->ldc % 4) != 0) return DCN_E_INVALID;
    GemmConv p;
    p.src = dout; p.wm = wt; p.bias = nullptr; p.add = add; p.dst = din; p.bn_partial = nullptr; p.out_absmax = nullptr;
    p.hs = c->hout; p.ws = c->wout; p.cs = c->ldc;   // source channels = (padded) forward output channels
    p.hd = c->hin; p.wd = c->win; p.cd = c->cin;
    p.kh = c->kh; p.kw = c->kw; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.ldc = c->cin;
    p.M = c->n * c->hin * c->win; p.K = c->kh * c->kw * c->ldc; p.transposed = 1; p.relu = 0;
    return launch_gemm(p, workspace, (hipStream_t)stream);
}

extern "C" size_t dcn_conv_wgrad_workspace(const dcn_conv_desc* c) {
    if (!valid_desc(c)) return 0;
    int rps;
    const int splits = wgrad_splits(c, &rps);
    return (size_t)splits * c->cout * c->kh * c->kw * c->cin * sizeof(float);
}

extern "C" int dcn_conv_wgrad(const dcn_conv_desc* c, const float* in, const float* dout, float* dw, void* slabs,
                              void* stream) {
    if (!valid_desc(c) || !in || !dout || !dw || !slabs || (c->ldc % 4) != 0) return DCN_E_INVALID;
    if ((int64_t)c->n * c->hin * c->win * c->cin >= ((int64_t)1 << 31) ||
        (int64_t)c->n * c->hout * c->wout * c->ldc >= ((int64_t)1 << 31))
        return DCN_E_UNSUPPORTED;  // 32-bit element offsets inside the kernel
    WgradConv p;
    p.in = in; p.dout = dout;
    p.hin = c->hin; p.win = c->win; p.cin = c->cin; p.hout = c->hout; p.wout = c->wout; p.cout = c->cout;
    p.kh = c->kh; p.kw = c->kw; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.ldo = c->ldc;
    p.M = c->n * c->hout * c->wout; p.K = c->kh * c->kw * c->cin;
    p.splits = wgrad_splits(c, &p.rows_per_split);
    const bool narrow = c->cout <= 64;
    p.ntiles_n = dcn::ceil_div(c->cout, narrow ? 64 : 128); p.ntiles_k = dcn::ceil_div(p.K, narrow ? 256 : 128);
    p.div_hw = make_fastdiv(c->hout * c->wout); p.div_w = make_fastdiv(c->wout);
    p.div_cin = make_fastdiv(c->cin); p.div_kw = make_fastdiv(c->kw);
    p.slab = p.splits == 1 ? dw : (float*)slabs;
    hipStream_t st = (hipStream_t)stream;
    if (narrow) hipLaunchKernelGGL(conv_wgrad_kernel<1>, dim3(p.ntiles_n * p.ntiles_k * p.splits), dim3(NT), 0, st, p);
    else hipLaunchKernelGGL(conv_wgrad_kernel<2>, dim3(p.ntiles_n * p.ntiles_k * p.splits), dim3(NT), 0, st, p);
    if (p.splits > 1) launch_wgrad_reduce((const float*)slabs, dw, (int64_t)c->cout * p.K / 4, p.splits, st);
    return dcn::check_launch();
}

extern "C" int dcn_transpose_weight(const float* w, float* wt, int cout, int taps, int cin, int ldn, void* stream) {
    if (!w || !wt || cout < 1 || taps < 1 || cin < 1 || ldn < cout) return DCN_E_INVALID;
    hipLaunchKernelGGL(transpose_weight_kernel, dim3(dcn::ceil_div(cin, 32), dcn::ceil_div(ldn, 32), taps), dim3(256),
                       0, (hipStream_t)stream, w, wt, cout, taps, cin, ldn);
    return dcn::check_launch();
}
