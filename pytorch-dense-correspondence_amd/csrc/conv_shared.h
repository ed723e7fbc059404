// Device / host pieces shared by the fp32 (conv_kernels.hip) and the split-fp16 (conv_f16_kernels.hip) convolution
// kernels: launch-constant fast division, the gather-GEMM parameter block, tile geometry, the XCD-aware tile remap
// and the common epilogue (bias / residual add / batch-norm partial sums).
#pragma once
#include <cstdlib>

#include "dcn_common.h"

namespace dcnconv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;

// Division by a launch-invariant positive integer without the ~25-instruction software divide:
// q = (umulhi(n, mul) + n) >> shr, exact for 0 <= n < 2^31 (round-up method; mul = floor(2^32 (2^shr - d) / d) + 1).
struct FastDiv {
    uint32_t mul, shr;
    int d;
};
inline FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.d = d;
    uint32_t s = 0;
    while ((1u << s) < (uint32_t)d) ++s;
    f.shr = s;
    f.mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << s) - (uint64_t)d)) / (uint64_t)d) + 1u;
    return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) { return (int)((__umulhi((uint32_t)n, f.mul) + (uint32_t)n) >> f.shr); }

struct GemmConv {
    const float* src;   // [n, hs, ws, cs] NHWC
    const float* wm;    // [cd][taps][cs]
    const float* bias;  // [cd] or null
    const float* add;   // [M][ldc] or null
    float* dst;         // [M][ldc]
    float* bn_partial;  // [mtiles][3][cd] (sum, sum of squares, max |x| per channel of the M tile) or null
    float* sk_partial;  // stream-K: [2 * workgroups][BM*BN] parked accumulators (fragment order)
    unsigned long long* sk_count = nullptr;   // split-fp16 kernel, optional: arrival word per stream-K tile -- (launch id << 32) | arrivals;
                                    // the last contributor of a tile completes it inside the launch (no fix-up kernel)
    unsigned sk_id = 0;             // id of this launch (never 0)
    int stem8 = 0;                  // split-fp16 forward: the 7x7 stem as a uniform-tap convolution over filter rows (conv_f16_kernels.hip)
    unsigned sk_bytes = 0;          // extent of sk_partial (buffer-resource bound of the device-coherent partial accesses)
    // Optional (dgrad whose result is the upstream gradient dy of a batch norm): the backward REDUCTION of that batch norm in
    // this epilogue -- dst receives the ReLU-masked gradient g (mask bit of the BN's output taken from bnb_mask, one byte
    // per float4, when given) and bnb_partial[mtile][cd][4] = (sum g, sum g xhat, max |g|, max |xhat|) over the tile's
    // rows, xhat = (bnb_x - mean) * invstd: what bn_bwd_reduce_kernel computes per row chunk, without its pass over dy.
    // M tiles must not straddle a statistics group (rows >= bnb_group_rows use the second group's mean / invstd).
    const float* bnb_x = nullptr;
    const unsigned char* bnb_mask = nullptr;
    const float* bnb_mean = nullptr;
    const float* bnb_invstd = nullptr;
    float* bnb_partial = nullptr;
    int bnb_gstride = 0, bnb_group_rows = 0;
    float* out_absmax;  // null, or device scalar raised to max |dst| (fused inference path: the next layer's operand pre-scale)
    // split-fp16 path only: pre-split weights [cd][kp] (hi, lo), device scalar with max|src| (or null), 1 / weight scale
    const _Float16* wh;
    const _Float16* wl;
    const float* a_absmax;
    float b_inv_scale;
    int kp;
    unsigned src_bytes, w_bytes;   // extents of src and of each weight image (buffer-resource bounds; 0: tensor too large)
    int hs, ws, cs, hd, wd, cd, kh, kw, stride, sshift, pad, dil, ldc, M, K, transposed, mtiles, ntiles, sk_units, sk_dp, relu;
    FastDiv div_hw, div_w, div_cs, div_kw, div_nt, div_nk;
    // small-tile hl32 kernel (conv_hlx_kernels.hip): workgroups per tile along K (contiguous stage ranges), tiles per launch
    int ksplit = 0;
    int hl_setprio = 1;    // s_setprio 1 around the MFMA bursts of the big-tile hl32 kernel (DCN_HL_SETPRIO)
    int hlx_stagger = 1;   // wavefronts 4-7 issue the next stage's LDS-DMA between the two parts of their compute slot (0: in front)
    FastDiv div_tiles = {0u, 0u, 1};
};

// bijective XCD-aware remap: consecutive logical tiles (sharing an M tile) land on the same XCD / L2
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// Stream-K tiles completed INSIDE the GEMM launch: every workgroup that parks a partial accumulator of a tile then counts
// itself into the tile's arrival word; the one that finds itself last (all the others' partials are then visible: they
// are written through and read past the per-XCD L2s with device-coherent accesses, see gemm_segment_f16) sums the parked
// partials in the fixed order of the contributing workgroups and runs the epilogue -- bit-identical to the separate fix-up
// kernel, without its launch, and overlapped with the tiles still being computed.  The arrival word is
// (launch id << 32) | arrivals: a word left behind by any other launch (or never initialised) counts as zero, so the
// workspace needs no clearing; the last arriver resets it to 0 so that a replay of the same captured launch starts clean.
__device__ __forceinline__ bool sk_arrive_is_last(unsigned long long* cnt, unsigned id, int contributors) {
    unsigned long long old = __atomic_load_n(cnt, __ATOMIC_RELAXED);
    for (;;) {
        const unsigned long long nv = (unsigned)(old >> 32) == id ? old + 1ull : (((unsigned long long)id << 32) | 1ull);
        const unsigned long long seen = atomicCAS(cnt, old, nv);
        if (seen == old) return (int)(unsigned)(nv & 0xffffffffull) == contributors;
        old = seen;
    }
}

// id of a stream-K launch with in-launch completion (process-wide, never 0; defined in conv_f16_kernels.hip)
unsigned next_sk_launch_id();

// ---- geometry shared by the gather-GEMM kernel and its stream-K fix-up kernel
// WM: wavefronts along M (2 -> 2x2 wave grid, 1 -> 1x4).  TM / TN: 32x32 MFMA tiles per wavefront.
// Workgroup tile = (32*TM*WM) x (32*TN*(4/WM)).  BK: K elements staged per barrier.
template <int WM, int TM, int TN, int BK> struct GemmGeo {
    static constexpr int WN = 4 / WM, BM = 32 * TM * WM, BN = 32 * TN * WN, LDK = BK + 4, KQ = BK / 4, ROWS = NT / KQ,
                         PA = (BM + ROWS - 1) / ROWS, PB = (BN + ROWS - 1) / ROWS, kStage = (BM + BN) * LDK,
                         kSlotFloats = BM * BN;
};

// Epilogue: C/D fragments -> NHWC rows (32 consecutive channels per half-wave = 128 B segments), + bias, + residual
// gradient, + per-M-tile batch-norm partial statistics (sum, sum of squares, max |x|; fixed order), or the backward
// reduction of the batch norm that consumes dst (GemmConv::bnb_*).  `red` = at least 4*WM*BN floats of LDS, free to use.
// WN: wavefronts along N (WM * WN wavefronts per workgroup; 4 for every kernel but the 8-wavefront f16x3 tile).
template <int WM, int TM, int TN, int BK, int WN = 4 / WM>
__device__ __forceinline__ void gemm_epilogue(const GemmConv& p, f32x16 (&acc)[TM][TN], int mt, int nt, float* red) {
    struct G { enum { BM = 32 * TM * WM, BN = 32 * TN * WN }; };
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm_ = wv / WN, wn_ = wv % WN;
    const int fi = lane & 31, fh = lane >> 5;
    const int m0 = mt * G::BM, n0 = nt * G::BN;
    float csum[TN], csq[TN], cmax[TN];
    float vmax = 0.f;
    if (p.bnb_partial) {   // (never together with bn_partial / relu / out_absmax)
        const int goff = (p.bnb_group_rows > 0 && m0 >= p.bnb_group_rows) ? p.bnb_gstride : 0;
        float cgx[TN], cxm[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            csum[tn] = 0.f; cgx[tn] = 0.f; cmax[tn] = 0.f; cxm[tn] = 0.f;
            const int col = n0 + wn_ * 32 * TN + tn * 32 + fi;
            const bool cok = col < p.cd;
            const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
            const float mu = cok ? p.bnb_mean[goff + col] : 0.f, is = cok ? p.bnb_invstd[goff + col] : 0.f;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm_ * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (cok && row < p.M) {
                        const int64_t o = (int64_t)row * p.ldc + col;
                        float v = acc[tm][tn][r] + bv;
                        if (p.add) v += p.add[o];
                        if (p.bnb_mask) v = ((p.bnb_mask[o >> 2] >> (col & 3)) & 1) ? v : 0.f;
                        p.dst[o] = v;
                        const float xh = (p.bnb_x[o] - mu) * is;
                        csum[tn] += v;
                        cgx[tn] = fmaf(v, xh, cgx[tn]);
                        cmax[tn] = fmaxf(cmax[tn], fabsf(v));
                        cxm[tn] = fmaxf(cxm[tn], fabsf(xh));
                    }
                }
            }
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            csum[tn] += __shfl_xor(csum[tn], 32, 64);
            cgx[tn] += __shfl_xor(cgx[tn], 32, 64);
            cmax[tn] = fmaxf(cmax[tn], __shfl_xor(cmax[tn], 32, 64));
            cxm[tn] = fmaxf(cxm[tn], __shfl_xor(cxm[tn], 32, 64));
            if (fh == 0) {
                const int cl = wn_ * 32 * TN + tn * 32 + fi;
                red[(wm_ * 4 + 0) * G::BN + cl] = csum[tn];
                red[(wm_ * 4 + 1) * G::BN + cl] = cgx[tn];
                red[(wm_ * 4 + 2) * G::BN + cl] = cmax[tn];
                red[(wm_ * 4 + 3) * G::BN + cl] = cxm[tn];
            }
        }
        __syncthreads();
        if (tid < G::BN && n0 + tid < p.cd) {
            float a = red[0 * G::BN + tid], b = red[1 * G::BN + tid], mg = red[2 * G::BN + tid], mx = red[3 * G::BN + tid];
#pragma unroll
            for (int w = 1; w < WM; ++w) {   // fixed order
                a += red[(w * 4 + 0) * G::BN + tid];
                b += red[(w * 4 + 1) * G::BN + tid];
                mg = fmaxf(mg, red[(w * 4 + 2) * G::BN + tid]);
                mx = fmaxf(mx, red[(w * 4 + 3) * G::BN + tid]);
            }
            reinterpret_cast<float4*>(p.bnb_partial)[(int64_t)mt * p.cd + n0 + tid] = make_float4(a, b, mg, mx);
        }
        return;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        csum[tn] = 0.f; csq[tn] = 0.f; cmax[tn] = 0.f;
        const int col = n0 + wn_ * 32 * TN + tn * 32 + fi;
        const bool cok = col < p.cd;
        const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm_ * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                float v = acc[tm][tn][r] + bv;
                if (cok && row < p.M) {
                    const int64_t o = (int64_t)row * p.ldc + col;
                    if (p.add) v += p.add[o];
                    if (p.relu) v = fmaxf(v, 0.f);   // (inference: conv + folded BN + residual + ReLU in one pass)
                    p.dst[o] = v;
                    vmax = fmaxf(vmax, fabsf(v));
                }
                csum[tn] += acc[tm][tn][r];
                csq[tn] = fmaf(acc[tm][tn][r], acc[tm][tn][r], csq[tn]);
                cmax[tn] = fmaxf(cmax[tn], fabsf(acc[tm][tn][r]));
            }
        }
    }
    if (p.out_absmax) {   // ONE atomic per workgroup at most (same-address atomics serialise in L2: thousands of them cost
                          // tens of microseconds); look before the atomic: the value only grows
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        if (lane == 0) red[wv] = vmax;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < WM * WN; ++w) vmax = fmaxf(vmax, red[w]);
            const unsigned bits = __float_as_uint(vmax);
            if (vmax > 0.f && bits > __atomic_load_n(reinterpret_cast<unsigned*>(p.out_absmax), __ATOMIC_RELAXED))
                atomicMax(reinterpret_cast<unsigned*>(p.out_absmax), bits);
        }
        __syncthreads();
    }
    if (p.bn_partial) {
        // rows >= M and columns >= cd are exactly zero in acc (zero-filled fragments), so no masking is needed
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            csum[tn] += __shfl_xor(csum[tn], 32, 64);
            csq[tn] += __shfl_xor(csq[tn], 32, 64);
            cmax[tn] = fmaxf(cmax[tn], __shfl_xor(cmax[tn], 32, 64));
            if (fh == 0) {
                red[(wm_ * 3 + 0) * G::BN + wn_ * 32 * TN + tn * 32 + fi] = csum[tn];
                red[(wm_ * 3 + 1) * G::BN + wn_ * 32 * TN + tn * 32 + fi] = csq[tn];
                red[(wm_ * 3 + 2) * G::BN + wn_ * 32 * TN + tn * 32 + fi] = cmax[tn];
            }
        }
        __syncthreads();
        if (tid < G::BN && n0 + tid < p.cd) {
            float s = red[0 * G::BN + tid], q = red[1 * G::BN + tid], mx = red[2 * G::BN + tid];
#pragma unroll
            for (int w = 1; w < WM; ++w) {   // fixed order
                s += red[(w * 3 + 0) * G::BN + tid];
                q += red[(w * 3 + 1) * G::BN + tid];
                mx = fmaxf(mx, red[(w * 3 + 2) * G::BN + tid]);
            }
            p.bn_partial[((int64_t)mt * 3 + 0) * p.cd + n0 + tid] = s;
            p.bn_partial[((int64_t)mt * 3 + 1) * p.cd + n0 + tid] = q;
            p.bn_partial[((int64_t)mt * 3 + 2) * p.cd + n0 + tid] = mx;   // -> bound of max |BN output| (bn_finalize_kernel)
        }
    }
}


// The same epilogue for accumulators made of 16 x 16 tiles (v_mfma_f32_16x16x32_f16: column = lane & 15, row = 4 (lane >> 4) +
// register) -- the hl32 kernels since round 5 (tools/hl_gemm_probe3.hip: that shape sustains 10-15 % more than 32 x 32 x 16
// in the same schedule -- half the accumulator-register traffic per FLOP under the power limit).  One wavefront's block:
// TM16 x TN16 tiles at tile-relative row `row0` / column `col0` of the BM x BN workgroup tile (mt, nt); the workgroup has TWO
// wavefronts along M (`wm` = 0 / 1: the fixed order of the statistics).  + bias, + residual gradient, per-M-tile batch-norm
// partial statistics (sum, sum of squares, max |x| of the accumulators).  `red`: >= 6 BN floats of LDS, free to use.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int TM16, int TN16, int BM, int BN>
__device__ __forceinline__ void gemm_epilogue16(const GemmConv& p, f32x4_t (&out)[TM16][TN16], int mt, int nt, int wm, int row0, int col0,
                                                float* red) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int fc = lane & 15, fq = lane >> 4;
    const int m0 = mt * BM + row0, n0 = nt * BN;
    float csum[TN16], csq[TN16], cmax[TN16];
#pragma unroll
    for (int tn = 0; tn < TN16; ++tn) {
        csum[tn] = 0.f; csq[tn] = 0.f; cmax[tn] = 0.f;
        const int col = n0 + col0 + tn * 16 + fc;
        const bool cok = col < p.cd;
        const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM16; ++tm) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + tm * 16 + 4 * fq + r;
                const float a = out[tm][tn][r];
                if (cok && row < p.M) {
                    const int64_t o = (int64_t)row * p.ldc + col;
                    float v = a + bv;
                    if (p.add) v += p.add[o];
                    p.dst[o] = v;
                }
                csum[tn] += a;
                csq[tn] = fmaf(a, a, csq[tn]);
                cmax[tn] = fmaxf(cmax[tn], fabsf(a));
            }
        }
    }
    if (p.bn_partial) {
        // rows >= M and columns >= cd are exactly zero in the accumulators (zero-filled fragments): no masking needed
#pragma unroll
        for (int tn = 0; tn < TN16; ++tn) {
            csum[tn] += __shfl_xor(csum[tn], 16, 64);
            csq[tn] += __shfl_xor(csq[tn], 16, 64);
            cmax[tn] = fmaxf(cmax[tn], __shfl_xor(cmax[tn], 16, 64));
            csum[tn] += __shfl_xor(csum[tn], 32, 64);
            csq[tn] += __shfl_xor(csq[tn], 32, 64);
            cmax[tn] = fmaxf(cmax[tn], __shfl_xor(cmax[tn], 32, 64));
            if (fq == 0) {
                const int cl = col0 + tn * 16 + fc;
                red[(wm * 3 + 0) * BN + cl] = csum[tn];
                red[(wm * 3 + 1) * BN + cl] = csq[tn];
                red[(wm * 3 + 2) * BN + cl] = cmax[tn];
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.cd) {   // the two row halves in fixed order
            const float s = red[0 * BN + tid] + red[3 * BN + tid];
            const float q = red[1 * BN + tid] + red[4 * BN + tid];
            const float mx = fmaxf(red[2 * BN + tid], red[5 * BN + tid]);
            p.bn_partial[((int64_t)mt * 3 + 0) * p.cd + n0 + tid] = s;
            p.bn_partial[((int64_t)mt * 3 + 1) * p.cd + n0 + tid] = q;
            p.bn_partial[((int64_t)mt * 3 + 2) * p.cd + n0 + tid] = mx;
        }
    }
}

// dw[i] = sum over splits of slab[s][i] in fixed order (defined in conv_kernels.hip)
void launch_wgrad_reduce(const float* slabs, float* dw, int64_t n4, int splits, hipStream_t st);

}  // namespace dcnconv
