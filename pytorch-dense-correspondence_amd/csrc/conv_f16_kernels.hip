// Split-fp16 ("f16x3") convolution kernels for gfx950: fp32 tensors in HBM, fp32 accumulation, but the products run on
// the fp16 matrix pipe (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA rate) by splitting every operand element
//     s*x = hi + lo,   hi = fp16(s*x),  lo = fp16(s*x - hi)          (s: a power of two, exact)
// and accumulating  lo_a*hi_b + hi_a*lo_b + hi_a*hi_b  (the dropped lo*lo term is ~2^-22 relative).  fp16 products are
// exact in fp32, so the result carries ~22 mantissa bits per operand: on the reference network (36 conv layers with
// train-mode BN, 640x480) the descriptor map is 1.3e-5 from a float64 run -- the fp32 CPU reference itself is 1.5e-5 --
// and every parameter gradient deviates from float64 exactly as much as the fp32 reference does (3.1e-2 max / 7.0e-3 L2,
// ill-conditioning, not precision).  bf16 cannot do this (a 2-way bf16 split is 2.3e-4 off, above the 1e-4 parity bar).
//   forward : activations s = 1, weights s = 64 (all weight tensors pre-split by one launch per call)
//   dgrad   : the incoming gradient tensor is scaled by 2^e with e from an upper bound of its abs-max (a device scalar
//             written by the kernels that produce the tensor) so that tiny gradients stay inside fp16's range; weights s = 64
//   wgrad   : dy (same scale) x activations (s = 1), reduction over pixels; both operands pre-split once per tensor
//             (the gradient "pixel-blocked" by the BN backward pass itself), transposed into k-major LDS rows
// MFMA work per K drops 5.3x (3 x 32 cycles per 16 K vs 8 x 64 cycles), so the kernels are designed around the
// operand path: 128x128 tiles, 32-K stages, fp32->(hi,lo) conversion once per element at staging time (v_cvt_pk_f16_f32),
// conflict-free 80-byte-pitch fp16 LDS images, one ds_read_b128 per 32x16 fragment.
#include <algorithm>
#include <atomic>

#include "conv_shared.h"
#include "dcn_tuning.h"
#include "f16_split.h"

namespace {

using namespace dcnconv;
using namespace dcnsplit;

constexpr int HBK = 32;            // K elements per stage
constexpr int LDH = HBK + 8;       // LDS row pitch in halves (80 bytes: conflict-free ds_read_b128 fragments)

// rows x K fp32 -> hi / lo fp16 [rows][kp] (kp = K rounded up to 8, zero padded), scaled by s
__global__ void __launch_bounds__(256)
split_rows_kernel(const float* __restrict__ w, _Float16* __restrict__ hi, _Float16* __restrict__ lo, int64_t rows, int K,
                  int kp, float s) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over rows * kp / 4
    const int q = kp / 4;
    if (i >= rows * q) return;
    const int64_t r = i / q;
    const int k = (int)(i - r * q) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K) v = *reinterpret_cast<const float4*>(w + r * K + k);  // K % 4 == 0
    h4 a, b;
    split4(v, s, a, b);
    *reinterpret_cast<h4*>(hi + r * kp + k) = a;
    *reinterpret_cast<h4*>(lo + r * kp + k) = b;
}

// All weight tensors of a network in ONE launch (the per-tensor kernels cost ~5 us each, 110 launches per step):
// entry e splits w[rows][K] (forward image), or -- transposed -- produces the dgrad image wt[cin][taps][ldn] of
// w[cout][taps][cin] (zero for n >= cout), both as hi / lo fp16 [rows][kp].  The table travels as a kernel argument.
struct SplitEntry {
    const float* w;
    const float* row_scale;   // forward image only, optional: row r is multiplied by row_scale[r] (folded eval-mode BN)
    _Float16* hi;
    _Float16* lo;
    int rows, K, kp;          // forward: rows = cout, K = taps * cin.   transposed: rows = cin, K = taps * ldn
    int cout, taps, cin, ldn;
    int first_block;          // prefix sum of workgroups
    int hl;                   // 1: ONE "hl32" image at `hi` -- [rows][K / 32][hi x32 | lo x32] (K % 32 == 0; conv_hl_kernels.hip)
};
constexpr int kSplitBatch = 48;   // 48 x 64 bytes + header < 4 KB of kernel arguments
struct SplitTable {
    SplitEntry e[kSplitBatch];
    int n;
    float scale;
    int* status;   // optional device word: bit 1 is raised when a weight leaves the fp16 range of its image (|scale * w| > 65504
                   // or NaN: the fixed power-of-two scale 64 covers |w| < 1023)
};

__device__ __forceinline__ void flag_weight_range(int* status, float4 v, float s) {
    const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) * fabsf(s);
    if (status && (!(m <= 65504.f) || v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w)) atomicOr(status, 2);
}

__device__ __forceinline__ int split_entry_of(const SplitTable& t, int block) {
    int e = 0;
    while (e + 1 < t.n && block >= t.e[e + 1].first_block) ++e;   // (uniform; at most 55 steps)
    return e;
}

__global__ void __launch_bounds__(256)
split_rows_batched_kernel(SplitTable t) {
    const int ei = split_entry_of(t, blockIdx.x);
    const SplitEntry& E = t.e[ei];
    const int64_t i = (int64_t)(blockIdx.x - E.first_block) * 256 + threadIdx.x;
    const int q = E.kp / 4;
    if (i >= (int64_t)E.rows * q) return;
    const int64_t r = i / q;
    const int k = (int)(i - r * q) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < E.K) v = *reinterpret_cast<const float4*>(E.w + r * E.K + k);
    h4 a, b;
    const float sc = E.row_scale ? t.scale * E.row_scale[r] : t.scale;
    flag_weight_range(t.status, v, sc);
    split4(v, sc, a, b);
    if (E.hl) {   // (4 consecutive k never straddle a 32-k chunk)
        _Float16* line = E.hi + r * 2 * E.kp + (k >> 5) * 64 + (k & 31);
        *reinterpret_cast<h4*>(line) = a;
        *reinterpret_cast<h4*>(line + 32) = b;
        return;
    }
    *reinterpret_cast<h4*>(E.hi + r * E.kp + k) = a;
    *reinterpret_cast<h4*>(E.lo + r * E.kp + k) = b;
}

// workgroup = one 32 (n) x 32 (c) tile of one tap: coalesced reads along c, LDS transpose, each work-item then owns
// 4 consecutive n of one c = one 8-byte run of the transposed row
__global__ void __launch_bounds__(256)
transpose_split_batched_kernel(SplitTable t) {
    __shared__ float tile[32][33];
    const int ei = split_entry_of(t, blockIdx.x);
    const SplitEntry& E = t.e[ei];
    const int ct = (E.cin + 31) / 32, nt = (E.ldn + 31) / 32;
    int b = blockIdx.x - E.first_block;
    const int tap = b / (ct * nt);
    b -= tap * ct * nt;
    const int n0 = (b / ct) * 32, c0 = (b % ct) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, cc = c0 + tx;
        tile[i][tx] = (n < E.cout && cc < E.cin) ? E.w[((int64_t)n * E.taps + tap) * E.cin + cc] : 0.f;
    }
    __syncthreads();
    const int cl = threadIdx.x >> 3, nq = threadIdx.x & 7;   // channel 0..31, n quad 0..7
    const int cc = c0 + cl, n = n0 + 4 * nq;
    if (cc >= E.cin || n >= E.ldn) return;                   // (ldn % 4 == 0)
    h4 a, bb;
    split4(make_float4(tile[4 * nq][cl], tile[4 * nq + 1][cl], tile[4 * nq + 2][cl], tile[4 * nq + 3][cl]), t.scale, a, bb);
    if (E.hl) {   // (K = taps * ldn, % 32 == 0: no padding to zero)
        const int k = tap * E.ldn + n;
        _Float16* line = E.hi + (int64_t)cc * 2 * E.kp + (k >> 5) * 64 + (k & 31);
        *reinterpret_cast<h4*>(line) = a;
        *reinterpret_cast<h4*>(line + 32) = bb;
        return;
    }
    const int64_t o = (int64_t)cc * E.kp + (int64_t)tap * E.ldn + n;
    *reinterpret_cast<h4*>(E.hi + o) = a;
    *reinterpret_cast<h4*>(E.lo + o) = bb;
    if (tap == E.taps - 1 && n + 4 >= E.ldn) {               // zero the K padding [taps * ldn, kp) of this row
        h4 z;
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = (_Float16)0.f;
        for (int k = E.K; k < E.kp; k += 4) {
            *reinterpret_cast<h4*>(E.hi + (int64_t)cc * E.kp + k) = z;
            *reinterpret_cast<h4*>(E.lo + (int64_t)cc * E.kp + k) = z;
        }
    }
}

// stem weights w4[cout][7][7][4] (3 real channels + a zero) -> hi / lo [cout][7][8][4]: filter rows of 8 x 4 = 32 K, the 8th
// column zero (the layout of the uniform-tap stem path, gemm_segment_f16)
__global__ void __launch_bounds__(256)
split_stem8_kernel(const float* __restrict__ w4, _Float16* __restrict__ hi, _Float16* __restrict__ lo, int cout, float s) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // over cout * 7 * 8
    if (i >= cout * 56) return;
    const int kx = i & 7, t = i >> 3;               // t = n * 7 + ky
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kx < 7) v = *reinterpret_cast<const float4*>(w4 + ((int64_t)t * 7 + kx) * 4);
    h4 a, b;
    split4(v, s, a, b);
    *reinterpret_cast<h4*>(hi + (int64_t)i * 4) = a;
    *reinterpret_cast<h4*>(lo + (int64_t)i * 4) = b;
}

// ----------------------------------------------------------------------------------------------- gather-GEMM, f16x3
// WR: wavefront rows of the workgroup (WR x 2 wavefronts, 128 WR work-items): 2 -> tiles of 64 TM x 64 TN, two workgroups
// per CU; 4 -> 128 TM x 64 TN on 8 wavefronts, ONE workgroup per CU (same 8 wavefronts per CU, but a quarter fewer
// operand bytes per MFMA through the vector-memory path and the LDS store path, profiles/EXPERIMENTS.md).
// WC: wavefront columns (WR x WC wavefronts; 2 everywhere -- the 128 x 256 shape WR = 2, WC = 4 measured as a wash in round 2
// and was removed in round 3: the wide layers now run on conv_hl_kernels.hip).
template <int TM, int TN, int WR = 2, int WC = 2> struct F16Geo {
    static constexpr int NTH = 64 * WR * WC, BM = 32 * TM * WR, BN = 32 * TN * WC, RA = NTH / 8, RB = NTH / 4, PA = BM / RA,
                         PB = BN / RB, kStageHalves = 2 * (BM + BN) * LDH;   // A hi, A lo, B hi, B lo
    static_assert(BM % RA == 0 && BN % RB == 0, "tile rows must be whole staging passes (WR = 4, WC = 2 needs TN = 2)");
};


constexpr int kOob = (int)0x80000000;   // voffset that fails the bounds check of any buffer <= 2 GiB: the load returns 0

// UNI: cs % 32 == 0, so a 32-K stage lies inside ONE filter tap and everything about the tap is wave-uniform: the
// gather is `buffer_load_dwordx4 voffset[row] + soffset(channel chunk)` with per-row byte offsets that only change when
// the tap does; out-of-image taps, rows past M and weight rows past cd get an out-of-range voffset and come back as
// zeros from the bounds check (no address clamps, no zero-fill selects in the K loop).
template <int TM, int TN, bool TR, bool UNI, int WR = 2, int WC = 2>
__device__ __forceinline__ void gemm_segment_f16(const GemmConv& p, _Float16* lds, int tile, int k0, int k1, int nk,
                                                 float* slot) {
    using G = F16Geo<TM, TN, WR, WC>;
    constexpr int BM = G::BM, BN = G::BN, PA = G::PA, PB = G::PB, kStage = G::kStageHalves;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm_ = wv / WC, wn_ = wv % WC;
    const int mt = fdiv(tile, p.div_nt), nt = tile - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // A staging: float4 (4 k) at k-quad kq of rows ra0 + RA j;  B staging: 8 halves at k-octet ko of rows rb0 + RB j
    // Row order within a store lane-group: the 16 lanes of a ds_write_b64 group (8 of a ds_write_b128 group) cover two
    // rows; rows r and r + 4 are 320 bytes = 16 banks apart, so the two 64-byte runs tile all 32 banks exactly (rows r and
    // r + 1 would overlap on 4 banks: measured as SQ_LDS_BANK_CONFLICT = 50 % extra LDS cycles).
    const int kq = tid & 7, ra0 = ((tid >> 4) & 3) + 4 * ((tid >> 3) & 1) + 8 * (tid >> 6);
    const int ko = tid & 3, rb0 = ((tid >> 3) & 3) + 4 * ((tid >> 2) & 1) + 8 * (tid >> 5);
    int by[PA], bx[PA], pixbase[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int m = m0 + ra0 + G::RA * j;
        const int mm = m < p.M ? m : 0;
        const int img = fdiv(mm, p.div_hw), rem = mm - img * p.div_hw.d;
        const int y = fdiv(rem, p.div_w), x = rem - y * p.div_w.d;
        by[j] = TR ? y + p.pad : y * p.stride - p.pad;
        bx[j] = TR ? x + p.pad : x * p.stride - p.pad;
        if (m >= p.M) by[j] = -(1 << 28);
        pixbase[j] = img * p.hs * p.ws;
    }
    int wrow[PB];
    unsigned wokm = 0;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int n = n0 + rb0 + G::RB * j;
        const bool ok = n < p.cd;
        wokm |= (ok ? 1u : 0u) << j;
        wrow[j] = (ok ? n : 0) * p.kp + ko * 8;
    }
    const int smask = p.stride - 1;
    // power-of-two pre-scale of the gathered operand from an upper bound of its abs-max (a device scalar written by the
    // kernel that produced the tensor): gradients AND activations, so that neither tiny nor huge tensors leave fp16's range
    const float sa = p.a_absmax ? pow2_scale(*p.a_absmax) : 1.f;
    // two register sets: global loads run TWO 32-K stages ahead of the MFMAs (a stage of fp16 MFMA work is ~0.3 us,
    // shorter than a trip to L2 / HBM under load)
    float4 ra[2][PA];
    h8 rbh[2][PB], rbl[2][PB];
    unsigned okm[2] = {0u, 0u};

    // ---- UNI state
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, (int)p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.wh), 0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.wl), 0, (int)p.w_bytes, 0x00020000);
    // STEM (p.stem8, forward only): the 7x7 / stride-2 stem on 4 (3 + 1) input channels as a uniform-tap convolution -- a
    // "tap" is a filter ROW and its 32-K chunk the 8 pixels x 4 channels = 128 contiguous bytes that start at the row's
    // first column (8th pixel: zero weights, not loaded); a work-item's k quad kq is then a PIXEL offset, so the validity
    // of a load depends on (row, kq) -- still one offset per register, no per-element tap decode (K = 7 x 32 = 224)
    const bool st8 = UNI && !TR && p.stem8 != 0;
    const int cpt = st8 ? 1 : p.cs / HBK;                           // 32-K chunks per tap
    const int cbs = (4 * p.cs) >> (TR ? p.sshift : 0);             // bytes per unit of (by, bx)
    // K traversal of the UNI path: channel-chunk groups outermost (kcg chunks = up to 128 channels), then the filter
    // taps, then the chunks of the group -- a tap change (new voff[]) only every kcg stages, and the SAME input pixels
    // come back for the next tap after kcg stages, while they are still in this XCD's L2 (tap-major order re-reads them
    // cs / 32 stages later: measured 48 % L2 hit rate on the 512-channel layers).
    const int kcg = (cpt & 3) == 0 ? 4 : ((cpt & 1) == 0 ? 2 : 1);
    const int taps = st8 ? p.kh : p.kh * p.kw;
    int rowoff[PA], voff[PA], voffb[PB];
    int u_tap = 0, u_grp = 0, u_c = 0, u_kt = k0;
    auto set_tap = [&](int tap) {
        const int r = st8 ? tap : fdiv(tap, p.div_kw), s = st8 ? 0 : tap - r * p.kw;
        const int dy = r * p.dil, dx = s * p.dil;
        const int delta = (dy * p.ws + dx) * cbs;
        const int dxq = st8 ? kq : 0;                               // (stem: this work-item's column inside the 8-pixel chunk)
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            bool ok;
            if (TR) {
                const int ny = by[j] - dy, nx = bx[j] - dx;
                ok = ((ny | nx) >= 0) & (((ny | nx) & smask) == 0) & ((ny >> p.sshift) < p.hs) & ((nx >> p.sshift) < p.ws);
            } else {
                ok = ((unsigned)(by[j] + dy) < (unsigned)p.hs) & ((unsigned)(bx[j] + dx + dxq) < (unsigned)p.ws) & (dxq < 7);
            }
            voff[j] = ok ? (TR ? rowoff[j] - delta : rowoff[j] + delta) : kOob;
        }
    };
    if (UNI) {
#pragma unroll
        for (int j = 0; j < PA; ++j)   // (rows past M carry by = -2^28: never inside the image)
            rowoff[j] = 4 * (pixbase[j] * p.cs + kq * 4) + (by[j] > -(1 << 27) ? (by[j] * p.ws + bx[j]) * cbs : 0);
#pragma unroll
        for (int j = 0; j < PB; ++j) voffb[j] = ((wokm >> j) & 1u) ? 2 * wrow[j] : kOob;
        const int per_grp = taps * kcg;       // stage index -> (group, tap, chunk in group)
        u_grp = k0 / per_grp;
        const int rem = k0 - u_grp * per_grp;
        u_tap = rem / kcg;
        u_c = rem - u_tap * kcg;
        set_tap(u_tap);
    }
    // issues the loads of stage u_kt, then steps to the next stage of the segment (staying on the last one)
    auto load_next = [&](int set) {
        const int chunk = u_grp * kcg + u_c;
        const int soff = chunk * (HBK * 4);
#pragma unroll
        for (int j = 0; j < PA; ++j)
            ra[set][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, voff[j], soff, 0));
        const int soffb = (u_tap * (st8 ? HBK : p.cs) + chunk * HBK) * 2;
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            rbh[set][j] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs_h, voffb[j], soffb, 0));
            rbl[set][j] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs_l, voffb[j], soffb, 0));
        }
    };
    // (kept apart from the loads: the tap change is a uniform branch, and placed after the MFMAs of the phase it leaves
    // "loads + fragment reads + MFMAs" as ONE scheduling region for the interleave below)
    // (measured: a branch-free variant -- scalar selects and set_tap every stage -- scheduled INTO the MFMA phase is 1-3 % slower)
    auto advance = [&]() {
        if (!UNI) return;
        if (u_kt + 1 < k1) {
            ++u_kt;
            if (++u_c == kcg) {
                u_c = 0;
                if (++u_tap == taps) { u_tap = 0; ++u_grp; }
                set_tap(u_tap);
            }
        }
    };

    auto load_tile = [&](int kt, int set) {
        if (UNI) { load_next(set); return; }
        kt = kt < k1 ? kt : k1 - 1;
        const int k = kt * HBK + kq * 4;
        const bool kval = k < p.K;
        const int kk = kval ? k : 0;
        const int tap = fdiv(kk, p.div_cs), c = kk - tap * p.cs;
        const int r = fdiv(tap, p.div_kw), s = tap - r * p.kw;
        const int dy = r * p.dil, dx = s * p.dil;
        unsigned om = 0;
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
            ra[set][j] = *reinterpret_cast<const float4*>(p.src + off);
            om |= (ok ? 1u : 0u) << j;
        }
        const int kb = kt * HBK + ko * 8;
        const bool kbval = kb < p.kp;
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const bool ok = kbval & (((wokm >> j) & 1u) != 0);
            const int off = ok ? wrow[j] + kt * HBK : 0;
            rbh[set][j] = *reinterpret_cast<const h8*>(p.wh + off);
            rbl[set][j] = *reinterpret_cast<const h8*>(p.wl + off);
            om |= (ok ? 1u : 0u) << (16 + j);
        }
        okm[set] = om;
    };
    auto store_tile = [&](int stage, int set) {
        _Float16* ah = lds + stage * kStage;
        _Float16* al = ah + BM * LDH;
        _Float16* bh = al + BM * LDH;
        _Float16* bl = bh + BN * LDH;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const bool ok = UNI | (((okm[set] >> j) & 1u) != 0);
            float4 v = ra[set][j];
            v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            h4 hi, lo;
            split4(v, sa, hi, lo);
            *reinterpret_cast<h4*>(ah + (ra0 + G::RA * j) * LDH + kq * 4) = hi;
            *reinterpret_cast<h4*>(al + (ra0 + G::RA * j) * LDH + kq * 4) = lo;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const bool ok = UNI | (((okm[set] >> (16 + j)) & 1u) != 0);
            h8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.f;
            *reinterpret_cast<h8*>(bh + (rb0 + G::RB * j) * LDH + ko * 8) = ok ? rbh[set][j] : z;
            *reinterpret_cast<h8*>(bl + (rb0 + G::RB * j) * LDH + ko * 8) = ok ? rbl[set][j] : z;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragments: lane (fi = lane & 31, fh = lane >> 5) holds 8 consecutive k at offset 16*ks + 8*fh of row fi
    const int fi = lane & 31, fh = lane >> 5;
    const int a_off = (wm_ * 32 * TM + fi) * LDH + 8 * fh;
    const int b_off = 2 * BM * LDH + (wn_ * 32 * TN + fi) * LDH + 8 * fh;
    h8 fah[2][TM], fal[2][TM], fbh[2][TN], fbl[2][TN];
    auto read_frags = [&](int stage, int ks, int set) {
        const _Float16* a = lds + stage * kStage + a_off + ks * 16;
        const _Float16* b = lds + stage * kStage + b_off + ks * 16;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            fah[set][t] = *reinterpret_cast<const h8*>(a + t * 32 * LDH);
            fal[set][t] = *reinterpret_cast<const h8*>(a + BM * LDH + t * 32 * LDH);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            fbh[set][t] = *reinterpret_cast<const h8*>(b + t * 32 * LDH);
            fbl[set][t] = *reinterpret_cast<const h8*>(b + BN * LDH + t * 32 * LDH);
        }
    };
    // product-type outermost: consecutive MFMAs go to different accumulators (no back-to-back read-after-write); the
    // small cross terms are accumulated before the large hi*hi term
    auto mfma_steps = [&](int set) {
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pt == 0 ? fal[set][tm] : fah[set][tm],
                                                                         pt == 1 ? fbl[set][tn] : fbh[set][tn],
                                                                         acc[tm][tn], 0, 0, 0);
    };

    // two-phase software pipeline, one barrier per 32-K stage; LDS stage (kt - k0) & 1 holds stage kt, register set
    // (kt - k0) & 1 carries stage kt on its way to LDS, and the loads of stage kt + 3 are issued as soon as set
    // (kt + 1) & 1 has been stored.  The loop is unrolled by two so that the set / stage indices are literals.
    // sched_group_barrier pins "1 MFMA, then a few of the other instructions" inside each phase, so that the conversion
    // VALU, the LDS traffic and the global loads issue in the shadow of the 32-cycle MFMAs instead of in front of them.
    constexpr int kMfma = 3 * TM * TN;   // per phase
    auto interleave_a = [&]() {
#pragma unroll
        for (int i = 0; i < kMfma; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x306, 84 / kMfma + 1, 0);      // VALU | SALU | DS
        }
    };
    auto interleave_b = [&]() {
#pragma unroll
        for (int i = 0; i < kMfma; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x126, 24 / kMfma + 1, 0);      // VALU | SALU | VMEM read | DS read
        }
    };
    // (sched_barrier: the loads must be ISSUED in stage order -- the s_waitcnt of the loop counts them -- and the
    // scheduler is otherwise free to sink a stage-1 load below the stage-2 loads of this straight-line prologue)
    load_tile(k0, 0); advance();
    __builtin_amdgcn_sched_barrier(0);
    load_tile(k0 + 1, 1); advance();
    __builtin_amdgcn_sched_barrier(0);
    store_tile(0, 0);
    load_tile(k0 + 2, 0); advance();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_frags(0, 0, 0);
    // (No `break` in the middle of the pair: the structurizer routes such an exit through a block that statically falls
    // back into the loop header, s_waitcnt insertion then has to assume the register sets may have been loaded in either
    // order, and every wait of one half drains ALL 16 loads in flight -- vmcnt(7..0) instead of vmcnt(15..8): the
    // two-stage prefetch distance was only real in every other stage.  Measured effect of the fix: 0-2 % per layer,
    // nothing on the step -- the loop is not load-latency-bound)
    int kt = k0;
    for (; kt + 1 < k1; kt += 2) {
        read_frags(0, 1, 1);
        store_tile(1, 1);          // stage kt + 1
        mfma_steps(0);
        interleave_a();
        __syncthreads();
        load_tile(kt + 3, 1);
        read_frags(1, 0, 0);
        mfma_steps(1);
        interleave_b();
        advance();
        read_frags(1, 1, 1);
        store_tile(0, 0);          // stage kt + 2
        mfma_steps(0);
        interleave_a();
        __syncthreads();
        load_tile(kt + 4, 0);
        read_frags(0, 0, 0);
        mfma_steps(1);
        interleave_b();
        advance();
    }
    if (kt < k1) {                 // odd number of stages: the last one (nothing left to stage or load)
        read_frags(0, 1, 1);
        mfma_steps(0);
        mfma_steps(1);
    }
    __syncthreads();

    const float inv = p.b_inv_scale / sa;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] *= inv;
    if (k0 == 0 && k1 == nk) {
        gemm_epilogue<WR, TM, TN, 16, WC>(p, acc, mt, nt, reinterpret_cast<float*>(lds));
    } else {
        // slot layout [wavefront][tm][tn][r / 4][lane][r % 4]: one 16-byte store per lane, 1 KB per wave-instruction
        bool inl = false;
        // (the 128 x 128 / 4-wavefront shape sits exactly at its 256-register budget: it keeps the separate fix-up kernel)
        if constexpr (!(WR == 2 && WC == 2 && TM == 2 && TN == 2)) inl = p.sk_count != nullptr;
        if (!inl) {
            float4* o = reinterpret_cast<float4*>(slot + wv * (TM * TN * 16 * 64)) + lane;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[((tm * TN + tn) * 4 + q) * 64] = make_float4(acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1],
                                                                       acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]);
        }
        if constexpr (!(WR == 2 && WC == 2 && TM == 2 && TN == 2)) if (inl) {
            // Completed inside the launch.  The parked partials travel between workgroups on different XCDs (separate
            // L2s), so they are written and read with device-coherent accesses (sc1: write-through / L2-bypassing) and ordered
            // against the arrival word by counter waits only -- NO device-scope fence: a release / acquire fence writes back
            // and invalidates the whole L2 of the XCD under the 31 other CUs that are in the middle of their K loops
            // (measured: +50 us per launch, profiles/r2b_ab_fence_variant.txt).
            __shared__ int s_last;
            const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(p.sk_partial, 0, (int)p.sk_bytes, 0x00020000);
            constexpr int kSc1 = 16;                           // cache-policy bit of buffer loads / stores: device scope
            const int lane_off = (wv * (TM * TN * 16 * 64) + lane * 4) * 4;
            {
                const int so = (int)((slot - p.sk_partial) * 4);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 v = make_float4(acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2],
                                                         acc[tm][tn][4 * q + 3]);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_p,
                                                                   lane_off + ((tm * TN + tn) * 4 + q) * 1024, so, kSc1);
                        }
            }
            const int rel = tile - p.sk_dp;                    // (index among the stream-K'd tiles)
            const int ua = rel * nk, ub = ua + nk - 1;         // its unit range relative to the start of the stream-K pass
            const int ga = ua / p.sk_units, gb = ub / p.sk_units;
            __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this work-item's partial has been written through
            __syncthreads();
            if (tid == 0) s_last = sk_arrive_is_last(p.sk_count + rel, p.sk_id, gb - ga + 1) ? 1 : 0;
            __syncthreads();
            if (s_last) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
                // (each contributor's partial in batches of kBatch 16-byte loads issued back to back -- the staging registers
                // of the K loop are free here -- and only then added: left to itself the compiler waits for every 4 loads,
                // i.e. 16 device-coherent round trips per contributor on the critical path of the launch)
                constexpr int kPieces = TM * TN * 4, kBatch = kPieces < 32 ? kPieces : 32;
                for (int g = ga; g <= gb; ++g) {               // fixed order, own partial included: deterministic
                    const int first_tile = (g * p.sk_units) / nk;
                    const int so = (2 * g + (first_tile == rel ? 0 : 1)) * (BM * BN * 4);
#pragma unroll
                    for (int b0 = 0; b0 < kPieces; b0 += kBatch) {
                        u32x4 t[kBatch];
#pragma unroll
                        for (int j = 0; j < kBatch; ++j)
                            t[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, lane_off + (b0 + j) * 1024, so, kSc1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < kBatch; ++j) {
                            const int pc = b0 + j, blk = pc >> 2, q = pc & 3;   // piece -> (tm, tn) block, row quad
                            const float4 v = __builtin_bit_cast(float4, t[j]);
                            acc[blk / TN][blk % TN][4 * q] += v.x; acc[blk / TN][blk % TN][4 * q + 1] += v.y;
                            acc[blk / TN][blk % TN][4 * q + 2] += v.z; acc[blk / TN][blk % TN][4 * q + 3] += v.w;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                gemm_epilogue<WR, TM, TN, 16, WC>(p, acc, mt, nt, reinterpret_cast<float*>(lds));
                if (tid == 0) atomicExch(p.sk_count + rel, 0ull);
            }
        }
    }
}

template <int TM, int TN, bool TR, bool SK, bool UNI, int WR = 2, int WC = 2>
__global__ void __launch_bounds__(64 * WR * WC, WR * WC == 8 ? 1 : 2)
conv_gemm_f16_kernel(GemmConv p) {
    using G = F16Geo<TM, TN, WR, WC>;
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * G::kStageHalves];
    const int nk = (p.K + HBK - 1) / HBK;
    if (!SK) {
        gemm_segment_f16<TM, TN, TR, UNI, WR, WC>(p, lds, xcd_remap(blockIdx.x, p.mtiles * p.ntiles), 0, nk, nk, nullptr);
    } else {
        // hybrid schedule: whole rounds of tiles data-parallel (all workgroups of an XCD walk K in step and share their
        // operands through L2), then ONE stream-K pass that splits the K stages of the leftover tiles evenly
        const int g = xcd_remap(blockIdx.x, gridDim.x);
        for (int tile = g; tile < p.sk_dp; tile += gridDim.x) {
            gemm_segment_f16<TM, TN, TR, UNI, WR, WC>(p, lds, tile, 0, nk, nk, nullptr);
            __syncthreads();
        }
        int u = p.sk_dp * nk + g * p.sk_units;
        const int total = p.mtiles * p.ntiles * nk;
        const int u_end = min(total, u + p.sk_units);
        bool first = true;
        while (u < u_end) {
            const int tile = fdiv(u, p.div_nk), k0 = u - tile * nk;
            const int k1 = min(nk, k0 + (u_end - u));
            gemm_segment_f16<TM, TN, TR, UNI, WR, WC>(p, lds, tile, k0, k1, nk,
                                         p.sk_partial + (int64_t)(2 * g + (first ? 0 : 1)) * (G::BM * G::BN));
            u += k1 - k0;
            first = false;
            __syncthreads();
        }
    }
}

// completes stream-K tiles (32-K stages).  WC workgroups per tile, one per wavefront column (the WR wavefronts that own
// one slice of the tile's columns): batch-norm partial sums are per column, so the slices are independent, and the pass
// -- a few dozen leftover tiles -- spreads over more CUs.
template <int TM, int TN, int WR = 2, int WC = 2>
__global__ void __launch_bounds__(64 * WR)
conv_gemm_f16_fixup_kernel(GemmConv p) {
    using G = F16Geo<TM, TN, WR, WC>;
    __shared__ float red[4 * WR * G::BN / WC];           // 4 WR x (BN / WC)
    const int nk = (p.K + HBK - 1) / HBK;
    const int rel = blockIdx.x / WC, part = blockIdx.x % WC;
    const int tile = p.sk_dp + rel;                      // only the leftover tiles were stream-K'd
    const int ua = rel * nk, ub = ua + nk - 1;           // unit range relative to the start of the stream-K pass
    const int ga = ua / p.sk_units, gb = ub / p.sk_units;
    if (ga == gb) return;
    const int lane = threadIdx.x & 63, wv = WC * (threadIdx.x >> 6) + part;   // wavefront index inside the GEMM workgroup
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int g = ga; g <= gb; ++g) {
        const int first_tile = (g * p.sk_units) / nk;
        const float4* o = reinterpret_cast<const float4*>(
                              p.sk_partial + (int64_t)(2 * g + (first_tile == rel ? 0 : 1)) * (G::BM * G::BN) +
                              wv * (TM * TN * 16 * 64)) + lane;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = o[((tm * TN + tn) * 4 + q) * 64];
                    acc[tm][tn][4 * q] += v.x; acc[tm][tn][4 * q + 1] += v.y;
                    acc[tm][tn][4 * q + 2] += v.z; acc[tm][tn][4 * q + 3] += v.w;
                }
    }
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    gemm_epilogue<WR, TM, TN, 16, 1>(p, acc, mt, WC * nt + part, red);   // a (32 TM WR) x (32 TN) "tile" of one wavefront column
}

struct F16Shape {
    int tm, tn, wr, wc, mtiles, ntiles, nk, sk_wgs, sk_units, sk_dp;   // tile = (32 tm wr) x (32 tn wc), 64 wr wc work-items
    bool sk;
    size_t ws_bytes, sk_count_off;
};
// align: 0, or the M tile must divide it (rows per batch-norm group)
F16Shape f16_shape(int M, int cd, int K, int align = 0) {
    F16Shape g;
    g.tn = cd <= 64 ? 1 : 2;
    const int ntiles128 = dcn::ceil_div(cd, 64 * g.tn);
    g.tm = dcn::ceil_div(M, 128) * ntiles128 < 2 * 256 ? 1 : 2;
    // 256 x 128 on 8 wavefronts wherever the channel count fills a 128-wide tile: measured +4-5 % on every such layer of
    // configs 1-3 against the best 64 / 128-row choice (a quarter fewer operand bytes per MFMA at the same 8 wavefronts per CU)
    g.wr = 2; g.wc = 2;
    if (g.tn == 2 && M >= 4096) { g.tm = 2; g.wr = 4; }
    const dcn::Tuning& tune = dcn::tuning();
    if (tune.gemm_tile_m) {
        const int v = tune.gemm_tile_m;
        if (v == 64) { g.tm = 1; g.wr = 2; g.wc = 2; }
        if (v == 128 && g.wc == 2) { g.tm = 2; g.wr = 2; }
        if (v == 256 && g.tn == 2) { g.tm = 2; g.wr = 4; g.wc = 2; }
    }
    if (align > 0 && (align % (32 * g.tm * g.wr)) != 0) { g.wr = 2; g.wc = 2; if ((align % (64 * g.tm)) != 0) g.tm = 1; }
    const int bm = 32 * g.tm * g.wr, bn = 32 * g.tn * g.wc;
    g.mtiles = dcn::ceil_div(M, bm);
    g.ntiles = dcn::ceil_div(cd, bn);
    g.nk = dcn::ceil_div(K, HBK);
    const int tiles = g.mtiles * g.ntiles;
    const int resident = g.wr * g.wc == 8 ? 256 : 512;   // workgroups the chip holds at once
    const double rounds = tiles / 256.0;
    const double waste = 1.0 - rounds / (double)(int)(rounds + 0.999999);
    // Stream-K removes the idle part of the last round -- (ceil(rounds) - rounds) * nk stage times -- and costs the
    // partial-tile writes plus the fix-up pass (25-30 us ~ 20 stage times).  Measured at N = 8: 128-channel layer (150
    // tiles of 36 stages, 15 stage times to gain) 50 us data-parallel vs 62 us stream-K; 256-channel layer (300 tiles of 72
    // stages, 60 to gain) 189 vs 156 us.
    const double sk_min_gain = tune.gemm_sk_min_gain;
    const double gain = ((double)(int)(rounds + 0.999999) - rounds) * g.nk;
    g.sk = tiles < 8 * 256 && waste > 0.08 && g.nk >= 8 && gain >= sk_min_gain;
    int wgs = resident;
    if (tune.gemm_sk >= 0) {
        const int v = tune.gemm_sk;
        if (v == 0) g.sk = false;
        if (v > 1) { g.sk = g.nk >= 2; wgs = v; }
    }
    g.sk_wgs = 0; g.sk_units = 0; g.sk_dp = 0; g.ws_bytes = 0; g.sk_count_off = 0;
    if (g.sk) {
        if ((int64_t)tiles * g.nk >= ((int64_t)1 << 30)) { g.sk = false; return g; }
        g.sk_dp = tiles / wgs * wgs;                       // whole rounds stay data-parallel
        const int64_t total = (int64_t)(tiles - g.sk_dp) * g.nk;
        if (total == 0) { g.sk = false; g.sk_dp = 0; return g; }
        if (g.sk_dp == 0 && wgs > total / 2) wgs = (int)(total / 2) > 0 ? (int)(total / 2) : 1;
        g.sk_units = (int)((total + wgs - 1) / wgs);
        g.sk_wgs = g.sk_dp > 0 ? wgs : (int)((total + g.sk_units - 1) / g.sk_units);
        g.ws_bytes = (size_t)2 * g.sk_wgs * bm * bn * sizeof(float);
        g.sk_count_off = g.ws_bytes;                                   // arrival words of the stream-K'd tiles
        g.ws_bytes += (size_t)(tiles - g.sk_dp) * sizeof(unsigned long long);
    }
    return g;
}

int launch_gemm_f16(GemmConv& p, void* workspace, hipStream_t st, int align = 0) {
    if (align > 0 && (align % 64) != 0) return DCN_E_UNSUPPORTED;
    if (p.stride != 1 && p.stride != 2 && p.stride != 4) return DCN_E_UNSUPPORTED;
    p.sshift = p.stride == 1 ? 0 : (p.stride == 2 ? 1 : 2);
    p.div_hw = make_fastdiv(p.hd * p.wd);
    p.div_w = make_fastdiv(p.wd);
    p.div_cs = make_fastdiv(p.cs);
    p.div_kw = make_fastdiv(p.kw);
    if ((int64_t)p.M / (p.hd * p.wd) * p.hs * p.ws * p.cs >= ((int64_t)1 << 31) || (int64_t)p.cd * p.kp >= ((int64_t)1 << 31))
        return DCN_E_UNSUPPORTED;
    const F16Shape g = f16_shape(p.M, p.cd, p.K, align);
    const bool sk = g.sk && workspace != nullptr;
    p.mtiles = g.mtiles;
    p.ntiles = g.ntiles;
    p.div_nt = make_fastdiv(g.ntiles);
    p.div_nk = make_fastdiv(g.nk);
    p.sk_units = sk ? g.sk_units : 0;
    p.sk_dp = sk ? g.sk_dp : 0;
    p.sk_partial = sk ? (float*)workspace : nullptr;
    const bool sk_inline = sk && dcn::tuning().gemm_sk_inline != 0 && !(g.wr == 2 && g.wc == 2 && g.tm == 2 && g.tn == 2);
    p.sk_count = sk_inline ? (unsigned long long*)((char*)workspace + g.sk_count_off) : nullptr;
    p.sk_bytes = sk_inline ? (unsigned)g.sk_count_off : 0u;
    if (sk_inline) {
        p.sk_id = next_sk_launch_id();
        // the arrival words share the scratch with other launches' partials: cleared on the stream in front of every launch,
        // so that the count never starts from a stale word that happens to carry this launch's id (ids wrap after 2^32 launches)
        if (dcn::fill_bytes_async(p.sk_count, 0, (size_t)(g.mtiles * g.ntiles - g.sk_dp) * sizeof(unsigned long long), st) != DCN_OK)
            return DCN_E_LAUNCH;
    }
    // uniform-tap fast path: whole 32-K stages inside one filter tap, tensors addressable through 2 GiB buffer resources
    const int64_t src_bytes = (int64_t)p.M / (p.hd * p.wd) * p.hs * p.ws * p.cs * 4, w_bytes = (int64_t)p.cd * p.kp * 2;
    bool uni = ((p.cs % HBK) == 0 || p.stem8) && src_bytes <= ((int64_t)1 << 31) && w_bytes <= ((int64_t)1 << 31);
    if (p.stem8 && !(uni && dcn::tuning().gemm_uni != 0)) return DCN_E_UNSUPPORTED;
    uni = uni && dcn::tuning().gemm_uni != 0;
    p.src_bytes = uni ? (unsigned)src_bytes : 0u;
    p.w_bytes = uni ? (unsigned)w_bytes : 0u;
    const dim3 grid(sk ? g.sk_wgs : g.mtiles * g.ntiles), fgrid(g.wc * (g.mtiles * g.ntiles - g.sk_dp)), block(64 * g.wr * g.wc),
               fblock(64 * g.wr);
#define DCN_GEMM16_K(TM, TN, WR, WC, TR, SK)                                                                        \
    do {                                                                                                            \
        if (uni) hipLaunchKernelGGL((conv_gemm_f16_kernel<TM, TN, TR, SK, true, WR, WC>), grid, block, 0, st, p);   \
        else hipLaunchKernelGGL((conv_gemm_f16_kernel<TM, TN, TR, SK, false, WR, WC>), grid, block, 0, st, p);      \
    } while (0)
#define DCN_GEMM16(TM, TN, WR, WC)                                                                     \
    do {                                                                                               \
        if (sk) {                                                                                      \
            if (p.transposed) DCN_GEMM16_K(TM, TN, WR, WC, true, true);                                \
            else DCN_GEMM16_K(TM, TN, WR, WC, false, true);                                            \
            if (!sk_inline) hipLaunchKernelGGL((conv_gemm_f16_fixup_kernel<TM, TN, WR, WC>), fgrid, fblock, 0, st, p);  \
        } else {                                                                                       \
            if (p.transposed) DCN_GEMM16_K(TM, TN, WR, WC, true, false);                               \
            else DCN_GEMM16_K(TM, TN, WR, WC, false, false);                                           \
        }                                                                                              \
    } while (0)
    if (g.wr == 4) DCN_GEMM16(2, 2, 4, 2);
    else if (g.tm == 1) { if (g.tn == 1) DCN_GEMM16(1, 1, 2, 2); else DCN_GEMM16(1, 2, 2, 2); }
    else { if (g.tn == 1) DCN_GEMM16(2, 1, 2, 2); else DCN_GEMM16(2, 2, 2, 2); }
#undef DCN_GEMM16_K
#undef DCN_GEMM16
    return dcn::check_launch();
}

// ----------------------------------------------------------------------------------------------- wgrad, f16x3
// dW[n][kcol] = sum_m dout[m][n] * in[pix(m, tap(kcol))][c(kcol)]  as a GEMM whose reduction index is the pixel m.
// Every operand element takes part in many tiles here (an input element in up to taps x Cout/128 of them, a gradient
// element in K/128), so the fp32 -> (hi, lo) split is done ONCE per tensor by the two kernels below and the GEMM
// kernel's operand path is pure data movement:
//   * gradient tensor, "pixel-blocked": dq[m >> 2][sub][n >> 2][8 halves], sub = 0: hi of channels (4q, 4q+1) x pixels
//     4(m>>2)..+3, 1: hi of (4q+2, 4q+3), 2 / 3: the lo parts.  A work-item's 4-pixel x 4-channel micro-tile is four 16-byte
//     loads (8 lanes = 128 contiguous bytes each), and every channel's 4 consecutive pixels (= 4 consecutive reduction
//     indices) are one 8-byte run that goes to LDS as it is;
//   * activation tensor, channel-contiguous like the fp32 tensor (the gather shifts pixels by the filter tap, so
//     pixel-blocking is impossible): xs[pixel][c >> 2][hi x4 | lo x4] -- one 16-byte load per (pixel, channel quad),
//     transposed 4x4 in registers with byte-permutes.
// Both have the byte size of the fp32 tensor.  LDS image, fragment reads and MFMA loop are those of the gather-GEMM kernel.
__global__ void __launch_bounds__(256)
split_act_kernel(const float* __restrict__ src, u32x4* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        h4 a, b;
        split4_unscaled(reinterpret_cast<const float4*>(src)[i], a, b);
        const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
        dst[i] = u32x4{ua[0], ua[1], ub[0], ub[1]};
    }
}

// dy [M][ld] fp32 -> dq (layout above); one work-item per (pixel quad, channel quad)
__global__ void __launch_bounds__(256)
split_grad_blocked_kernel(const float* __restrict__ dy, int M, int ld, const float* __restrict__ absmax,
                          u32x4* __restrict__ dq) {
    const float s = absmax ? pow2_scale(*absmax) : 1.f;
    const int c4n = ld >> 2;
    const int64_t total = (int64_t)((M + 3) >> 2) * c4n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t q = i / c4n;
        const int cq = (int)(i - q * c4n);
        float v[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t m = q * 4 + r;
            const float4 x = m < M ? *reinterpret_cast<const float4*>(dy + m * ld + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[r][0] = x.x; v[r][1] = x.y; v[r][2] = x.z; v[r][3] = x.w;
        }
        store_blocked_quad(dq, q, cq, c4n, v, s);
    }
}

struct WgradF16 {
    const void* xs;       // split activations [n, hin, win, cin/4][hi x4 | lo x4]
    const void* dq;       // split gradient, pixel-blocked (layout above), pre-scaled by pow2_scale(*d_absmax)
    float* slab;          // [splits][cout][K]
    const float* d_absmax;
    const float* x_absmax;   // abs-max bound of the fp32 activation operand (null, or ignored with a pre-split operand: scale 1)
    unsigned x_bytes, d_bytes;
    int hin, win, cin, hout, wout, cout, kh, kw, stride, pad, dil, ldo, M, K, splits, rows_per_split, ntiles_n, ntiles_k;
    FastDiv div_hw, div_w, div_cin, div_kw;
};

// FAST: wout % 4 == 0 and wout >= 32 (every real layer): a work-item's pixel quad lies in one image row and its
// (image, y, x) position is carried from stage to stage with a few selects instead of being re-derived by division.
// XPRE: the activation operand is the pre-split tensor xs; otherwise it is the fp32 tensor itself and is split on the fly
// (same addresses, 4 bytes per element either way) -- cheaper when an element is only used by a few tiles (1x1 convs).
// TM = 1, 2: 256 work-items, two workgroups per CU.  TM = 4: 256 output channels x 128 K columns on 8 wavefronts (4 x 2),
// ONE workgroup per CU: the gradient operand -- stored pre-split, a pure copy -- is the doubled one, so per MFMA the
// activation operand's loads, conversions, 4x4 transposes and LDS stores halve (wavefronts 0-3 stage it, all 8 stage dy).
// ROLES (TM = 4): wavefronts 0-3 stage ONLY the activation operand (loads, conversions, transposes) and wavefronts 4-7 the
// whole gradient tile (two micro-tiles each, pure copies) -- instead of the gradient copy being spread over all eight on top
// of the activation work of the first four: the two halves then reach the stage barrier closer together.
template <int TM, bool FAST, bool XPRE, bool DEEP = false, bool ROLES = false>   // 64*TM output channels x 128 K columns per workgroup
__global__ void __launch_bounds__(TM == 4 ? 512 : NT, TM == 4 ? 1 : 2)
conv_wgrad_f16_kernel(WgradF16 p) {
    static_assert(!ROLES || TM == 4, "ROLES is a variant of the 8-wavefront tile");
    constexpr int BM = 64 * TM, BN = 128, kStage = 2 * (BM + BN) * LDH;
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * kStage];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm_ = wv >> 1, wn_ = wv & 1;
    const int tiles = p.ntiles_n * p.ntiles_k;
    const int bid = xcd_remap(blockIdx.x, tiles * p.splits);
    const int split = bid / tiles, tile = bid - split * tiles;
    const int tn_ = tile / p.ntiles_k, tk_ = tile - tn_ * p.ntiles_k;
    const int n0 = tn_ * BM, j0 = tk_ * BN;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    // (ROLES: the second load slot of a work-item reads the activations in wavefronts 0-3 and the gradient's second micro-tile
    // in wavefronts 4-7: one wave-uniform resource, selected from scalars)
    const bool hiw = ROLES && __builtin_amdgcn_readfirstlane(tid >> 8) != 0;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(hiw ? p.dq : p.xs), 0,
                                                                          (int)(hiw ? p.d_bytes : p.x_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dq), 0, (int)p.d_bytes, 0x00020000);

    // micro-tile of this work-item: pixels 4*pq .. 4*pq+3 of the stage, channel quad cq (dout: n0 + 4*cq, in: j0 + 4*cq)
    const int pq = tid & 7;
    const int cq = ROLES ? ((tid >> 3) & 31) : (tid >> 3);    // 8 pixel quads x 32 (TM = 4 without ROLES: 64) channel quads
    // (ROLES: wavefronts 4-7 own gradient channel quads cq and cq + 32, wavefronts 0-3 none)
    const bool dact = ROLES ? hiw : 4 * cq < BM;              // (BM = 64: only half of the work-items stage dout)
    const bool xact = TM != 4 || tid < 256;                   // (TM = 4: wavefronts 0-3 stage the activation operand; wave-uniform)
    const int ncol = n0 + cq * 4;
    const bool nval = dact & (ncol < p.ldo);
    const bool nval2 = ROLES && dact && (ncol + 128 < p.ldo);
    const int kcol = j0 + (cq & 31) * 4;
    const bool kval = xact & (kcol < p.K);
    const int kc0 = kval ? kcol : 0;
    const int tap = fdiv(kc0, p.div_cin), cc = kc0 - tap * p.cin;
    const int tr = fdiv(tap, p.div_kw), ts = tap - tr * p.kw;
    const int oy = tr * p.dil - p.pad, ox = ts * p.dil - p.pad;
    // DEEP: two register sets, so that the global loads of a stage are issued 1.5 iterations before the stage is stored to
    // LDS instead of half an iteration (SQ counters of the single-set loop on the layer-4 convolution: matrix pipe busy 36 %
    // of the cycles at 2.0 GHz -- not power-limited --, wavefronts parked at s_waitcnt / the barrier 48 % of theirs,
    // profiles/r2l_wgrad_sq_counters_*.txt).  Where the register budget allows it.
    constexpr int NSET = DEEP ? 2 : 1;
    u32x4 rd[NSET][4];       // sub-planes: hi (ch 0,1), hi (ch 2,3), lo (ch 0,1), lo (ch 2,3), each x 4 pixels
    u32x4 rx[NSET][4];       // per pixel: hi x4 | lo x4
    const int d_sub = (p.ldo >> 2) * 16;   // bytes between sub-planes of one pixel quad

    // Addresses of a stage are prepared ahead (prep: VALU only, placed before the barrier) so that the loads of the next
    // stage issue back to back right behind the barrier; everything is 32-bit unsigned arithmetic and branch-free,
    // invalid pieces get the out-of-range offset and load zeros.
    int doff, xoff[4];
    int px = 0, py = 0, pimg = 0;   // FAST: position of pixel 4*pq of the stage being prepared
    if (FAST) {
        const int m = m_begin + 4 * pq;
        const int mm = m < p.M ? m : 0;
        pimg = fdiv(mm, p.div_hw);
        const int rem = mm - pimg * p.div_hw.d;
        py = fdiv(rem, p.div_w);
        px = rem - py * p.div_w.d;
    }
    const unsigned cin4 = (unsigned)p.cin * 4u;
    auto prep = [&](int m_base) {
        const int mq = (m_base >> 2) + pq;                    // (m_base is a multiple of 32)
        const bool dok = nval & (4 * mq < m_end);
        doff = dok ? (int)((unsigned)mq * 4u * (unsigned)d_sub + (unsigned)(ncol >> 2) * 16u) : kOob;
        if (ROLES && hiw) {   // second gradient micro-tile (channel quad cq + 32) through the activation slot
            const bool dok2 = nval2 & (4 * mq < m_end);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xoff[i] = dok2 ? (int)((unsigned)mq * 4u * (unsigned)d_sub + (unsigned)i * (unsigned)d_sub + (unsigned)((ncol >> 2) + 32) * 16u) : kOob;
            return;
        }
        if (FAST) {
            const bool qval = kval & (m_base + 4 * pq < m_end);
            const int sy = py * p.stride + oy;
            const bool oky = qval & ((unsigned)sy < (unsigned)p.hin);
            const unsigned rowb = ((unsigned)(pimg * p.hin + sy) * (unsigned)p.win) * cin4 + (unsigned)cc * 4u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sx = (px + i) * p.stride + ox;
                const bool ok = oky & ((unsigned)sx < (unsigned)p.win);
                xoff[i] = ok ? (int)(rowb + (unsigned)sx * cin4) : kOob;
            }
            px += HBK;                                        // wout >= 32: at most one row wrap per stage
            const bool wx = px >= p.wout;
            px -= wx ? p.wout : 0;
            py += wx ? 1 : 0;
            const bool wy = py >= p.hout;
            py = wy ? 0 : py;
            pimg += wy ? 1 : 0;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m_base + 4 * pq + i;
                const bool mval = m < m_end;
                const int mm = mval ? m : 0;
                const int img = fdiv(mm, p.div_hw), rem = mm - img * p.div_hw.d;
                const int y = fdiv(rem, p.div_w), x = rem - y * p.div_w.d;
                const int sy = y * p.stride + oy, sx = x * p.stride + ox;
                const bool ok = mval & kval & ((unsigned)sy < (unsigned)p.hin) & ((unsigned)sx < (unsigned)p.win);
                const unsigned off = ((unsigned)((img * p.hin + sy) * p.win + sx)) * cin4 + (unsigned)cc * 4u;
                xoff[i] = ok ? (int)off : kOob;
            }
        }
    };
    auto issue_loads = [&](int set) {
#pragma unroll
        for (int h = 0; h < 4; ++h) rd[set][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, doff, h * d_sub, 0);
        // (wavefronts that do not stage the activation operand carry out-of-range offsets: their loads touch no memory; no
        // branch here, so that the compiler's vmcnt bookkeeping knows exactly how many loads a stage has in flight)
#pragma unroll
        for (int i = 0; i < 4; ++i) rx[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff[i], 0, 0);
    };
    const float sx = (!XPRE && p.x_absmax) ? pow2_scale(*p.x_absmax) : 1.f;
    auto store_tile = [&](int stage, int set) {
        _Float16* dh = lds + stage * kStage;
        _Float16* dl = dh + BM * LDH;
        _Float16* xh = dl + BM * LDH;
        _Float16* xl = xh + BN * LDH;
        if (!XPRE && xact) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h4 a, b;
                split4(__builtin_bit_cast(float4, rx[set][i]), sx, a, b);
                const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
                rx[set][i] = u32x4{ua[0], ua[1], ub[0], ub[1]};
            }
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            _Float16* xd = pl ? xl : xh;
            // 4x4 transpose of halves: channel e of pixels 0..3 = {lo/hi half of dword e>>1 of each pixel}
            if (xact)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                const int w = 2 * pl + (e >> 1);
                u32x2 o;
                if ((e & 1) == 0) {
                    o[0] = (rx[set][0][w] & 0xffffu) | (rx[set][1][w] << 16);
                    o[1] = (rx[set][2][w] & 0xffffu) | (rx[set][3][w] << 16);
                } else {
                    o[0] = (rx[set][0][w] >> 16) | (rx[set][1][w] & 0xffff0000u);
                    o[1] = (rx[set][2][w] >> 16) | (rx[set][3][w] & 0xffff0000u);
                }
                *reinterpret_cast<u32x2*>(xd + (4 * (cq & 31) + e) * LDH + 4 * pq) = o;
            }
            if (dact) {
                _Float16* dd = pl ? dl : dh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u32x2 o;
                    o[0] = rd[set][2 * pl + (e >> 1)][2 * (e & 1)];
                    o[1] = rd[set][2 * pl + (e >> 1)][2 * (e & 1) + 1];
                    *reinterpret_cast<u32x2*>(dd + (4 * cq + e) * LDH + 4 * pq) = o;
                    if (ROLES) {   // second micro-tile (channel quad cq + 32), loaded through the activation slot
                        o[0] = rx[set][2 * pl + (e >> 1)][2 * (e & 1)];
                        o[1] = rx[set][2 * pl + (e >> 1)][2 * (e & 1) + 1];
                        *reinterpret_cast<u32x2*>(dd + (128 + 4 * cq + e) * LDH + 4 * pq) = o;
                    }
                }
            }
        }
    };

    // wavefront tiling: TM == 2 / 4: 2x2 / 4x2 waves, each 64 channels x 64 columns; TM == 1: 1x4 waves, each 64 channels x 32 columns
    constexpr int CT = TM >= 2 ? 2 : 1;
    f32x16 acc[2][CT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fi = lane & 31, fh = lane >> 5;
    const int wrow0 = TM >= 2 ? wm_ * 64 : 0, wcol0 = TM >= 2 ? wn_ * 64 : wv * 32;
    const int a_off = (wrow0 + fi) * LDH + 8 * fh;
    const int b_off = 2 * BM * LDH + (wcol0 + fi) * LDH + 8 * fh;
    h8 fah[2][2], fal[2][2], fbh[2][CT], fbl[2][CT];
    auto read_frags = [&](int stage, int ks, int set) {
        const _Float16* a = lds + stage * kStage + a_off + ks * 16;
        const _Float16* b = lds + stage * kStage + b_off + ks * 16;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fah[set][t] = *reinterpret_cast<const h8*>(a + t * 32 * LDH);
            fal[set][t] = *reinterpret_cast<const h8*>(a + BM * LDH + t * 32 * LDH);
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            fbh[set][t] = *reinterpret_cast<const h8*>(b + t * 32 * LDH);
            fbl[set][t] = *reinterpret_cast<const h8*>(b + BN * LDH + t * 32 * LDH);
        }
    };
    auto mfma_steps = [&](int set) {
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < CT; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pt == 0 ? fal[set][tm] : fah[set][tm],
                                                                         pt == 1 ? fbl[set][tn] : fbh[set][tn],
                                                                         acc[tm][tn], 0, 0, 0);
    };

    const int nsteps = (m_end - m_begin + HBK - 1) / HBK;
    if (nsteps > 0) {
        // (stages past the end of the split prepare out-of-range addresses: their loads return zeros and are never used)
        if constexpr (!DEEP) {
            prep(m_begin);
            issue_loads(0);
            prep(m_begin + HBK);
            store_tile(0, 0);
            issue_loads(0);
            prep(m_begin + 2 * HBK);
            __syncthreads();
            read_frags(0, 0, 0);
            for (int st = 0; st < nsteps; ++st) {
                const int cur = st & 1;
                read_frags(cur, 1, 1);
                store_tile(cur ^ 1, 0);
                mfma_steps(0);
                __syncthreads();
                issue_loads(0);                                  // stage st + 2
                read_frags(cur ^ 1, 0, 0);
                mfma_steps(1);
                prep(m_begin + (st + 3) * HBK);
            }
        } else {
            // stage k travels in register set k & 1; its loads are issued in iteration k - 3 (behind the barrier) and it
            // is stored to LDS at the top of iteration k - 1.  Unrolled by two: set and LDS-stage indices are literals.
            prep(m_begin);
            issue_loads(0);
            __builtin_amdgcn_sched_barrier(0);   // (issue order = stage order: the compiler's vmcnt waits count in it)
            prep(m_begin + HBK);
            issue_loads(1);
            __builtin_amdgcn_sched_barrier(0);
            prep(m_begin + 2 * HBK);
            store_tile(0, 0);
            issue_loads(0);
            __builtin_amdgcn_sched_barrier(0);
            prep(m_begin + 3 * HBK);
            __syncthreads();
            read_frags(0, 0, 0);
            auto iter = [&](int st, int cur, int set1) {
                read_frags(cur, 1, 1);
                store_tile(cur ^ 1, set1);                       // stage st + 1
                mfma_steps(0);
                __syncthreads();
                issue_loads(set1);                               // stage st + 3
                read_frags(cur ^ 1, 0, 0);
                mfma_steps(1);
                prep(m_begin + (st + 4) * HBK);
            };
            int st = 0;
            for (; st + 1 < nsteps; st += 2) {
                iter(st, 0, 1);
                iter(st + 1, 1, 0);
            }
            if (st < nsteps) iter(st, 0, 1);
        }
    }
    // C fragment: row (r) <-> output channel, column (lane & 31) <-> K column: 128-byte coalesced rows
    const float inv = 1.f / (p.d_absmax ? pow2_scale(*p.d_absmax) : 1.f) / sx;
    float* out = p.slab + (int64_t)split * p.cout * p.K;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < CT; ++tn) {
            const int kc = j0 + wcol0 + tn * 32 + fi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wrow0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (n < p.cout && kc < p.K) out[(int64_t)n * p.K + kc] = acc[tm][tn][r] * inv;
            }
        }
}

// Pixel-range splits: enough workgroups to fill the chip (2 resident per CU), chosen so that the LAST round of
// workgroups is nearly full -- e.g. 144 tiles x 4 splits = 576 workgroups = 1.125 rounds of 512 wastes almost half of the
// machine, 144 x 7 = 1.97 rounds does not.  Fewer splits win ties (less slab traffic for the reduce pass).
// the 256-channel / 8-wavefront tile: layers whose output channels fill it (DCN_WGRAD_TILE=128 keeps the 128-channel tile).
// Measured per layer at N = 8 / 16 (profiles/r2d_wgrad_tile_ab.txt): 256-channel layer-3 convolutions +13 % / +5 %, layer 4
// +7-11 % / +1-7 %; only the 1x1 128 -> 256 downsample (a single K tile) loses 6 %: K >= 256 required.
bool wgrad_wide_f16(const dcn_conv_desc* c) {
    return c->cout >= 256 && (c->cout % 256) == 0 && c->kh * c->kw * c->cin >= 256 && dcn::tuning().wgrad_tile != 128;
}

int wgrad_splits_f16(const dcn_conv_desc* c, int* rows_per_split, bool wide) {
    const int M = c->n * c->hout * c->wout, K = c->kh * c->kw * c->cin;
    const bool narrow = c->cout <= 64;
    const int tiles = dcn::ceil_div(c->cout, narrow ? 64 : (wide ? 256 : 128)) * dcn::ceil_div(K, 128);
    const int slots = wide ? 256 : 512;   // workgroups the chip holds at once
    const int max_by_rows = (M / (8 * HBK)) > 1 ? (M / (8 * HBK)) : 1;
    int cap = narrow ? 256 : (tiles < 16 ? 128 : 64);   // (slab traffic grows with the split count; few-tile layers need more splits to fill the chip)
    if (cap > max_by_rows) cap = max_by_rows;
    // search from half a round to three rounds of workgroups; the score prefers FULL rounds and, among those, few splits:
    // measured (N = 8): one full round beats two -- layer 1: 100 splits 107 us vs 200 splits 120 us, layer 3: 14 splits
    // 223 us vs 28 splits 243 us -- every workgroup pays its prologue and its 64 KB slab once
    const int target = dcn::ceil_div(slots, tiles);
    int lo = target / 2 > 1 ? target / 2 : 1, hi = 3 * target;
    if (hi > cap) hi = cap;
    if (lo > hi / 2) lo = hi / 2 > 1 ? hi / 2 : 1;   // capped: still search below the cap (e.g. 9 tiles x 64 splits = 1.125 rounds, x 55 = one)
    int best = hi;
    double best_score = -1.0;
    for (int s = lo; s <= hi; ++s) {
        const int rps = dcn::ceil_div(dcn::ceil_div(M, s), HBK) * HBK;
        const int real = dcn::ceil_div(M, rps);                       // splits actually launched with 32-row granularity
        const int wgs = real * tiles;
        const double eff = (double)wgs / (double)(dcn::ceil_div(wgs, slots) * slots);
        const double score = eff - 0.004 * real - (wgs < slots ? 0.5 * (1.0 - (double)wgs / slots) : 0.0);
        if (score > best_score) { best_score = score; best = s; }
    }
    if (const int v = dcn::tuning().wgrad_splits) { if (v >= 1 && v <= max_by_rows) best = v; }
    int rps = dcn::ceil_div(dcn::ceil_div(M, best), HBK) * HBK;
    *rows_per_split = rps;
    return dcn::ceil_div(M, rps);
}

bool valid_desc16(const dcn_conv_desc* c) {
    return c && c->n > 0 && c->hin > 0 && c->win > 0 && c->cin > 0 && (c->cin % 4) == 0 && c->hout > 0 && c->wout > 0 &&
           c->cout > 0 && c->kh > 0 && c->kw > 0 && c->stride > 0 && c->dil > 0 && c->pad >= 0 && c->ldc >= c->cout;
}

// rows of the dgrad result per batch-norm statistics group (the descriptor's group_rows counts rows of the conv OUTPUT)
int dgrad_group_rows(const dcn_conv_desc* c) {
    if (c->group_rows <= 0) return 0;
    const int groups = c->n * c->hout * c->wout / c->group_rows;
    return groups > 1 ? c->n / groups * c->hin * c->win : 0;
}

}  // namespace

unsigned dcnconv::next_sk_launch_id() {
    static std::atomic<unsigned> next_id{1u};
    unsigned id;
    do { id = next_id.fetch_add(1u, std::memory_order_relaxed); } while (id == 0u);
    return id;
}

extern "C" int dcn_f16_kpad(int k) { return (k + 7) / 8 * 8; }

extern "C" int dcn_split_rows_f16(const float* w, void* hi, void* lo, int64_t rows, int k, float scale, void* stream) {
    if (!w || !hi || !lo || rows < 1 || k < 4 || (k % 4) != 0) return DCN_E_INVALID;
    const int kp = dcn_f16_kpad(k);
    const int64_t n = rows * (kp / 4);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)dcn::ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16*)hi, (_Float16*)lo, rows, k, kp, scale);
    return dcn::check_launch();
}

// Batched forms of dcn_split_rows_f16 (transposed == 0) and dcn_transpose_weight + dcn_split_rows_f16 (transposed != 0)
// over n weight tensors: w[i] is [cout[i]][taps[i]][cin[i]]; hi[i] / lo[i] receive [cout][kpad(taps*cin)] or, transposed,
// [cin][kpad(taps*ldn[i])] (ldn = the output gradient's leading dimension, >= cout, % 4 == 0).  All arrays are HOST arrays.
extern "C" int dcn_split_weights_f16(int n, const float* const* w, void* const* hi, void* const* lo, const int* cout,
                                     const int* taps, const int* cin, const int* ldn, int transposed, float scale,
                                     void* stream) {
    return dcn_split_weights_scaled_f16(n, w, nullptr, hi, lo, cout, taps, cin, ldn, transposed, scale, stream);
}

// row_scale (optional, forward images only): row_scale[i] is NULL or a device vector of cout[i] per-output-channel factors
namespace {
int split_weights_impl(int n, const float* const* w, const float* const* row_scale, void* const* hi, void* const* lo,
                       const int* cout, const int* taps, const int* cin, const int* ldn, int transposed, float scale,
                       void* stream, bool hl, int* status = nullptr);
}

extern "C" int dcn_split_weights_scaled_f16(int n, const float* const* w, const float* const* row_scale, void* const* hi,
                                            void* const* lo, const int* cout, const int* taps, const int* cin,
                                            const int* ldn, int transposed, float scale, void* stream) {
    if (!lo) return DCN_E_INVALID;
    return split_weights_impl(n, w, row_scale, hi, lo, cout, taps, cin, ldn, transposed, scale, stream, false);
}

// dcn_split_weights_scaled_f16 (forward images) that also checks the weights' range: bit 1 of *status (device int, not
// cleared here) is raised when some |scale * row_scale * w| exceeds fp16's range or is NaN
extern "C" int dcn_split_weights_checked_f16(int n, const float* const* w, const float* const* row_scale, void* const* hi,
                                             void* const* lo, const int* cout, const int* taps, const int* cin, float scale,
                                             int* status, void* stream) {
    if (!lo) return DCN_E_INVALID;
    return split_weights_impl(n, w, row_scale, hi, lo, cout, taps, cin, nullptr, 0, scale, stream, false, status);
}

// hl32 images (conv_hl_kernels.hip): out[i] receives [cout][taps * cin / 32][hi x32 | lo x32] or, transposed,
// [cin][taps * ldn / 32][hi x32 | lo x32]; the K of every entry must be a multiple of 32.
extern "C" int dcn_split_weights_hl32(int n, const float* const* w, void* const* out, const int* cout, const int* taps,
                                      const int* cin, const int* ldn, int transposed, float scale, void* stream) {
    return split_weights_impl(n, w, nullptr, out, nullptr, cout, taps, cin, ldn, transposed, scale, stream, true);
}

namespace {
int split_weights_impl(int n, const float* const* w, const float* const* row_scale, void* const* hi, void* const* lo,
                       const int* cout, const int* taps, const int* cin, const int* ldn, int transposed, float scale,
                       void* stream, bool hl, int* status) {
    if (n < 1 || !w || !hi || (!hl && !lo) || !cout || !taps || !cin || (transposed && (!ldn || row_scale)) || !(scale > 0.f))
        return DCN_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += kSplitBatch) {
        SplitTable t;
        t.n = n - base < kSplitBatch ? n - base : kSplitBatch;
        t.scale = scale;
        t.status = status;
        int blocks = 0;
        for (int j = 0; j < t.n; ++j) {
            const int i = base + j;
            SplitEntry& E = t.e[j];
            if (!w[i] || !hi[i] || (!hl && !lo[i]) || cout[i] < 1 || taps[i] < 1 || cin[i] < 4 || (cin[i] % 4)) return DCN_E_INVALID;
            E.w = w[i]; E.hi = (_Float16*)hi[i]; E.lo = hl ? nullptr : (_Float16*)lo[i]; E.hl = hl ? 1 : 0;
            E.row_scale = row_scale ? row_scale[i] : nullptr;
            E.cout = cout[i]; E.taps = taps[i]; E.cin = cin[i]; E.ldn = transposed ? ldn[i] : cout[i];
            E.first_block = blocks;
            if (transposed) {
                if (E.ldn < E.cout || (E.ldn % 4)) return DCN_E_INVALID;
                E.rows = E.cin; E.K = E.taps * E.ldn; E.kp = dcn_f16_kpad(E.K);
                blocks += E.taps * ((E.cin + 31) / 32) * ((E.ldn + 31) / 32);
            } else {
                E.rows = E.cout; E.K = E.taps * E.cin; E.kp = dcn_f16_kpad(E.K);
                blocks += (int)dcn::ceil_div64((int64_t)E.rows * (E.kp / 4), 256);
            }
            if (hl && (E.K % 32) != 0) return DCN_E_INVALID;
        }
        if (transposed) hipLaunchKernelGGL(transpose_split_batched_kernel, dim3(blocks), dim3(256), 0, st, t);
        else hipLaunchKernelGGL(split_rows_batched_kernel, dim3(blocks), dim3(256), 0, st, t);
    }
    return dcn::check_launch();
}
}  // namespace

extern "C" int dcn_conv_num_mtiles_f16(const dcn_conv_desc* c) {
    if (!valid_desc16(c)) return DCN_E_INVALID;
    return f16_shape(c->n * c->hout * c->wout, c->cout, c->kh * c->kw * c->cin, c->group_rows).mtiles;
}

extern "C" size_t dcn_conv_gemm_workspace_f16(const dcn_conv_desc* c, int dgrad) {
    if (!valid_desc16(c)) return 0;
    if (dgrad)   // (either tiling: plain, or aligned to the statistics groups for dcn_conv_dgrad_bn_f16)
        return std::max(f16_shape(c->n * c->hin * c->win, c->cin, c->kh * c->kw * c->ldc).ws_bytes,
                        f16_shape(c->n * c->hin * c->win, c->cin, c->kh * c->kw * c->ldc, dgrad_group_rows(c)).ws_bytes);
    return f16_shape(c->n * c->hout * c->wout, c->cout, c->kh * c->kw * c->cin, c->group_rows).ws_bytes;
}

// w_hi / w_lo: [cout][kpad(K)] fp16 from dcn_split_rows_f16(w, ..., scale = w_scale)
extern "C" int dcn_conv_forward_f16(const dcn_conv_desc* c, const float* in, const float* in_absmax, const void* w_hi,
                                    const void* w_lo, float w_scale, const float* bias, float* out, float* bn_partial,
                                    void* workspace, void* stream) {
    if (!valid_desc16(c) || !in || !w_hi || !w_lo || !out || !(w_scale > 0.f)) return DCN_E_INVALID;
    GemmConv p;
    p.src = in; p.wm = nullptr; p.bias = bias; p.add = nullptr; p.dst = out; p.bn_partial = bn_partial; p.out_absmax = nullptr;
    p.wh = (const _Float16*)w_hi; p.wl = (const _Float16*)w_lo; p.a_absmax = in_absmax; p.b_inv_scale = 1.f / w_scale;
    p.hs = c->hin; p.ws = c->win; p.cs = c->cin; p.hd = c->hout; p.wd = c->wout; p.cd = c->cout;
    p.kh = c->kh; p.kw = c->kw; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.ldc = c->ldc;
    p.M = c->n * c->hout * c->wout; p.K = c->kh * c->kw * c->cin; p.kp = dcn_f16_kpad(p.K); p.transposed = 0; p.relu = 0;
    return launch_gemm_f16(p, workspace, (hipStream_t)stream, c->group_rows);
}

// The 7x7 / stride-2 / pad-3 stem on 4-channel input (3 + a zero channel) through the uniform-tap path: w_hi / w_lo from
// dcn_split_stem_weights_f16 ([cout][7][8][4]).  Same result as dcn_conv_forward_f16 on the same descriptor.
extern "C" int dcn_split_stem_weights_f16(const float* w4, void* hi, void* lo, int cout, float scale, void* stream) {
    if (!w4 || !hi || !lo || cout < 1 || !(scale > 0.f)) return DCN_E_INVALID;
    hipLaunchKernelGGL(split_stem8_kernel, dim3((unsigned)dcn::ceil_div(cout * 56, 256)), dim3(256), 0, (hipStream_t)stream, w4,
                       (_Float16*)hi, (_Float16*)lo, cout, scale);
    return dcn::check_launch();
}

extern "C" int dcn_conv_stem_forward_f16(const dcn_conv_desc* c, const float* in, const float* in_absmax, const void* w_hi,
                                         const void* w_lo, float w_scale, float* out, float* bn_partial, void* stream) {
    if (!valid_desc16(c) || !in || !w_hi || !w_lo || !out || !(w_scale > 0.f)) return DCN_E_INVALID;
    if (c->kh != 7 || c->kw != 7 || c->cin != 4 || c->stride != 2 || c->pad != 3 || c->dil != 1 || c->win < 8) return DCN_E_UNSUPPORTED;
    GemmConv p;
    p.src = in; p.wm = nullptr; p.bias = nullptr; p.add = nullptr; p.dst = out; p.bn_partial = bn_partial; p.out_absmax = nullptr;
    p.wh = (const _Float16*)w_hi; p.wl = (const _Float16*)w_lo; p.a_absmax = in_absmax; p.b_inv_scale = 1.f / w_scale;
    p.hs = c->hin; p.ws = c->win; p.cs = c->cin; p.hd = c->hout; p.wd = c->wout; p.cd = c->cout;
    p.kh = c->kh; p.kw = c->kw; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.ldc = c->ldc;
    p.M = c->n * c->hout * c->wout; p.K = 7 * HBK; p.kp = 7 * HBK; p.transposed = 0; p.relu = 0; p.stem8 = 1;
    return launch_gemm_f16(p, nullptr, (hipStream_t)stream, c->group_rows);
}

// Inference form: out = [relu](conv(in, w) + bias [+ add]) in one pass (w / bias with the eval-mode batch norm folded in:
// dcn_split_weights_scaled_f16 with row_scale = gamma / sqrt(var + eps), bias = beta - mean * that).  add: [M][ldc] or NULL.
extern "C" int dcn_conv_forward_fused_f16(const dcn_conv_desc* c, const float* in, const float* in_absmax, const void* w_hi,
                                          const void* w_lo, float w_scale, const float* bias, const float* add, int relu,
                                          float* out, float* out_absmax, void* workspace, void* stream) {
    if (!valid_desc16(c) || !in || !w_hi || !w_lo || !out || !(w_scale > 0.f)) return DCN_E_INVALID;
    GemmConv p;
    p.src = in; p.wm = nullptr; p.bias = bias; p.add = add; p.dst = out; p.bn_partial = nullptr; p.out_absmax = out_absmax;
    p.wh = (const _Float16*)w_hi; p.wl = (const _Float16*)w_lo; p.a_absmax = in_absmax; p.b_inv_scale = 1.f / w_scale;
    p.hs = c->hin; p.ws = c->win; p.cs = c->cin; p.hd = c->hout; p.wd = c->wout; p.cd = c->cout;
    p.kh = c->kh; p.kw = c->kw; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.ldc = c->ldc;
    p.M = c->n * c->hout * c->wout; p.K = c->kh * c->kw * c->cin; p.kp = dcn_f16_kpad(p.K); p.transposed = 0;
    p.relu = relu ? 1 : 0;
    return launch_gemm_f16(p, workspace, (hipStream_t)stream, c->group_rows);
}

// wt_hi / wt_lo: split of the channel-transposed weights [cin][taps][ldc] (dcn_transpose_weight, then dcn_split_rows_f16);
// dout_absmax: device scalar >= max |dout| (selects the power-of-two pre-scale), or NULL for scale 1
extern "C" int dcn_conv_dgrad_f16(const dcn_conv_desc* c, const float* dout, const void* wt_hi, const void* wt_lo,
                                  float w_scale, const float* dout_absmax, const float* add, float* din, void* workspace,
                                  void* stream) {
    if (!valid_desc16(c) || !dout || !wt_hi || !wt_lo || !din || (c->ldc % 4) != 0 || !(w_scale > 0.f)) return DCN_E_INVALID;
    GemmConv p;
    p.src = dout; p.wm = nullptr; p.bias = nullptr; p.add = add; p.dst = din; p.bn_partial = nullptr; p.out_absmax = nullptr;
    p.wh = (const _Float16*)wt_hi; p.wl = (const _Float16*)wt_lo; p.a_absmax = dout_absmax; p.b_inv_scale = 1.f / w_scale;
    p.hs = c->hout; p.ws = c->wout; p.cs = c->ldc;
    p.hd = c->hin; p.wd = c->win; p.cd = c->cin;
    p.kh = c->kh; p.kw = c->kw; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.ldc = c->cin;
    p.M = c->n * c->hin * c->win; p.K = c->kh * c->kw * c->ldc; p.kp = dcn_f16_kpad(p.K); p.transposed = 1; p.relu = 0;
    return launch_gemm_f16(p, workspace, (hipStream_t)stream);
}

extern "C" int dcn_conv_dgrad_bn_num_mtiles_f16(const dcn_conv_desc* c) {
    if (!valid_desc16(c)) return DCN_E_INVALID;
    return f16_shape(c->n * c->hin * c->win, c->cin, c->kh * c->kw * c->ldc, dgrad_group_rows(c)).mtiles;
}

// dcn_conv_dgrad_f16 whose result din is the upstream gradient of a batch norm (the one that produced this convolution's
// input): din receives the ReLU-masked gradient (relu_mask: the bytes dcn_bn_apply wrote for that batch norm's output, or
// NULL: no ReLU) and bn_partial[dcn_conv_dgrad_bn_num_mtiles_f16(c)][cin][4] the per-tile sums of the backward reduction
// (consumed by dcn_bn_backward_from_partial).  bn_x: that batch norm's input [n, hin, win, cin]; bn_stats: its
// [groups][4][cin] statistics block (scale, shift, mean, invstd).
extern "C" int dcn_conv_dgrad_bn_f16(const dcn_conv_desc* c, const float* dout, const void* wt_hi, const void* wt_lo,
                                     float w_scale, const float* dout_absmax, const float* add, float* din,
                                     const float* bn_x, const unsigned char* relu_mask, const float* bn_stats,
                                     float* bn_partial, void* workspace, void* stream) {
    if (!valid_desc16(c) || !dout || !wt_hi || !wt_lo || !din || (c->ldc % 4) != 0 || !(w_scale > 0.f) || !bn_x || !bn_stats ||
        !bn_partial || (c->cin % 4) != 0)
        return DCN_E_INVALID;
    GemmConv p;
    p.src = dout; p.wm = nullptr; p.bias = nullptr; p.add = add; p.dst = din; p.bn_partial = nullptr; p.out_absmax = nullptr;
    p.wh = (const _Float16*)wt_hi; p.wl = (const _Float16*)wt_lo; p.a_absmax = dout_absmax; p.b_inv_scale = 1.f / w_scale;
    p.hs = c->hout; p.ws = c->wout; p.cs = c->ldc;
    p.hd = c->hin; p.wd = c->win; p.cd = c->cin;
    p.kh = c->kh; p.kw = c->kw; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.ldc = c->cin;
    p.M = c->n * c->hin * c->win; p.K = c->kh * c->kw * c->ldc; p.kp = dcn_f16_kpad(p.K); p.transposed = 1; p.relu = 0;
    p.bnb_x = bn_x; p.bnb_mask = relu_mask; p.bnb_mean = bn_stats + 2 * c->cin; p.bnb_invstd = bn_stats + 3 * c->cin;
    p.bnb_partial = bn_partial; p.bnb_gstride = 4 * c->cin; p.bnb_group_rows = dgrad_group_rows(c);
    return launch_gemm_f16(p, workspace, (hipStream_t)stream, p.bnb_group_rows);
}

extern "C" int dcn_split_act_f16(const float* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst || n < 4 || (n % 4) != 0) return DCN_E_INVALID;
    const int64_t n4 = n / 4;
    const unsigned blocks = (unsigned)(dcn::ceil_div64(n4, 256) > 8192 ? 8192 : dcn::ceil_div64(n4, 256));
    hipLaunchKernelGGL(split_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (u32x4*)dst, n4);
    return dcn::check_launch();
}

extern "C" size_t dcn_grad_blocked_bytes(int m, int ld) { return (size_t)((m + 3) / 4) * 4 * (size_t)ld * 4; }

extern "C" int dcn_split_grad_blocked_f16(const float* dy, int m, int ld, const float* absmax, void* dq, void* stream) {
    if (!dy || !dq || m < 1 || ld < 4 || (ld % 4) != 0) return DCN_E_INVALID;
    const int64_t total = (int64_t)((m + 3) / 4) * (ld / 4);
    const unsigned blocks = (unsigned)(dcn::ceil_div64(total, 256) > 8192 ? 8192 : dcn::ceil_div64(total, 256));
    hipLaunchKernelGGL(split_grad_blocked_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, m, ld, absmax,
                       (u32x4*)dq);
    return dcn::check_launch();
}

// slab bytes only; the split operands are the caller's (dcn_split_act_f16 / dcn_split_grad_blocked_f16)
extern "C" size_t dcn_conv_wgrad_workspace_f16(const dcn_conv_desc* c) {
    if (!valid_desc16(c)) return 0;
    int rps;   // (either tile: the tuning table may change between sizing and launch)
    const int splits = std::max(wgrad_splits_f16(c, &rps, false), wgrad_splits_f16(c, &rps, c->cout >= 256 && (c->cout % 256) == 0));
    return (size_t)splits * c->cout * c->kh * c->kw * c->cin * sizeof(float);
}

extern "C" int dcn_conv_wgrad_f16(const dcn_conv_desc* c, const void* xs, int xs_is_fp32, const float* x_absmax,
                                  const void* dq, const float* dout_absmax, float* dw, void* slabs, void* stream) {
    if (!valid_desc16(c) || !xs || !dq || !dw || !slabs || (c->ldc % 4) != 0) return DCN_E_INVALID;
    const int64_t x_bytes = (int64_t)c->n * c->hin * c->win * c->cin * 4;
    const int64_t d_bytes = (int64_t)dcn_grad_blocked_bytes(c->n * c->hout * c->wout, c->ldc);
    if (x_bytes > ((int64_t)1 << 31) || d_bytes > ((int64_t)1 << 31)) return DCN_E_UNSUPPORTED;
    WgradF16 p;
    p.xs = xs; p.dq = dq;
    p.d_absmax = dout_absmax; p.x_absmax = xs_is_fp32 ? x_absmax : nullptr; p.x_bytes = (unsigned)x_bytes; p.d_bytes = (unsigned)d_bytes;
    p.hin = c->hin; p.win = c->win; p.cin = c->cin; p.hout = c->hout; p.wout = c->wout; p.cout = c->cout;
    p.kh = c->kh; p.kw = c->kw; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.ldo = c->ldc;
    p.M = c->n * c->hout * c->wout; p.K = c->kh * c->kw * c->cin;
    const bool narrow = c->cout <= 64, wide = wgrad_wide_f16(c);
    p.splits = wgrad_splits_f16(c, &p.rows_per_split, wide);
    p.ntiles_n = dcn::ceil_div(c->cout, narrow ? 64 : (wide ? 256 : 128)); p.ntiles_k = dcn::ceil_div(p.K, 128);
    p.div_hw = make_fastdiv(c->hout * c->wout); p.div_w = make_fastdiv(c->wout);
    p.div_cin = make_fastdiv(c->cin); p.div_kw = make_fastdiv(c->kw);
    p.slab = p.splits == 1 ? dw : (float*)slabs;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(p.ntiles_n * p.ntiles_k * p.splits), block(wide ? 512 : NT);
    const bool fast = (c->wout % 4) == 0 && c->wout >= HBK;
    // deep prefetch (two register sets): DCN_WGRAD_DEEP is a mask over the tile variants (1: 64-channel, 2: 128, 4: 256)
    const int deep_mask = dcn::tuning().wgrad_deep;
#define DCN_WGRAD16_D(TM, DEEP)                                                                                        \
    do {                                                                                                               \
        if (fast) {                                                                                                    \
            if (xs_is_fp32) hipLaunchKernelGGL((conv_wgrad_f16_kernel<TM, true, false, DEEP>), grid, block, 0, st, p); \
            else hipLaunchKernelGGL((conv_wgrad_f16_kernel<TM, true, true, DEEP>), grid, block, 0, st, p);            \
        } else {                                                                                                       \
            if (xs_is_fp32) hipLaunchKernelGGL((conv_wgrad_f16_kernel<TM, false, false, DEEP>), grid, block, 0, st, p); \
            else hipLaunchKernelGGL((conv_wgrad_f16_kernel<TM, false, true, DEEP>), grid, block, 0, st, p);           \
        }                                                                                                              \
    } while (0)
#define DCN_WGRAD16(TM)                                      \
    do {                                                     \
        if (deep_mask & (TM)) DCN_WGRAD16_D(TM, true);       \
        else DCN_WGRAD16_D(TM, false);                       \
    } while (0)
    if (narrow) DCN_WGRAD16(1);
    else if (wide && dcn::tuning().wgrad_roles && fast && xs_is_fp32) {
        if (deep_mask & 4) hipLaunchKernelGGL((conv_wgrad_f16_kernel<4, true, false, true, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_wgrad_f16_kernel<4, true, false, false, true>), grid, block, 0, st, p);
    }
    else if (wide) DCN_WGRAD16(4);
    else DCN_WGRAD16(2);
#undef DCN_WGRAD16_D
#undef DCN_WGRAD16
    if (p.splits > 1) launch_wgrad_reduce((const float*)slabs, dw, (int64_t)c->cout * p.K / 4, p.splits, st);
    return dcn::check_launch();
}
