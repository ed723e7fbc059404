// K2 / K7 / K8 / K10 and layout helpers: the HBM-bound kernels around the convolutions (gfx950).
// All activations are NHWC fp32 with C % 4 == 0, so every work-item moves float4 (16 B/lane, 1 KiB per
// wave-instruction).  Reductions are two-stage and fixed-order (run-to-run deterministic), accumulated
// in fp64 in the tiny finalize kernels.
//
//   batch norm (train):  mean/var over N*H*W of one forward call, eps 1e-5, momentum 0.1, unbiased running var
//                        == torch.nn.BatchNorm2d as the [NOT IN TREE] backbone uses it
//   max pool 3x3/2 pad 1 (first maximum in window scan order wins, like ATen's CPU kernel)
//   bilinear upsample, align_corners=True  == F.upsample_bilinear (called from Resnet34_8s.forward [NOT IN TREE])
//   optional per-pixel L2 normalisation of the descriptor (dense_correspondence_network.py:256-259)
#include "dcn_tuning.h"
#include "elementwise_kernels.h"
#include "f16_split.h"

namespace dcn {

// max |.| over a wavefront, then one atomic per wavefront (non-negative floats order like their bit patterns).
// The abs-max of every tensor that feeds a split-fp16 convolution (activations in forward, gradients in backward) selects
// that operand's power-of-two pre-scale; a NaN anywhere propagates through the scaled products exactly as it would
// through the fp32 ones.  EVERY lane of the wavefront must call this (shuffles).
__device__ __forceinline__ void wave_atomic_absmax(float amax, float* absmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    // thousands of wavefronts target one address: look first (plain load, served by L2) and only the few wavefronts that
    // would actually raise the value issue the atomic -- the value only ever grows, so a stale look is merely conservative
    if ((threadIdx.x & 63) == 0 && amax > 0.f) {
        const unsigned bits = __float_as_uint(amax);
        if (bits > __atomic_load_n(reinterpret_cast<unsigned*>(absmax), __ATOMIC_RELAXED))
            atomicMax(reinterpret_cast<unsigned*>(absmax), bits);
    }
}

// ---------------------------------------------------------------------------------------------- layout
__global__ void __launch_bounds__(256)
nchw3_to_nhwc4_kernel(const float* __restrict__ img, float* __restrict__ out, int hw, int64_t total,
                      float* __restrict__ absmax) {
    __shared__ float s_m[4];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {  // over n*hw
        const int64_t n = i / hw, p = i - n * hw;
        const float* s = img + n * 3 * hw + p;
        const float4 v = make_float4(s[0], s[hw], s[2 * (int64_t)hw], 0.f);
        reinterpret_cast<float4*>(out)[i] = v;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fabsf(v.z)));
        if (v.x != v.x || v.y != v.y || v.z != v.z) m = __builtin_huge_valf();   // (fmaxf drops a NaN: report it as inf)
    }
    if (absmax) {   // one atomic per workgroup (same-address atomics serialise in L2)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
            const unsigned bits = __float_as_uint(m);
            if (m > 0.f && bits > __atomic_load_n(reinterpret_cast<unsigned*>(absmax), __ATOMIC_RELAXED))
                atomicMax(reinterpret_cast<unsigned*>(absmax), bits);
        }
    }
}

// wp[o][tap][4] = (w[o][tap][0..2], 0)   and back (gradient): w[o][tap][c] = wp[o][tap][c], c < 3
__global__ void __launch_bounds__(256)
pad_c3_to_c4_kernel(const float* __restrict__ w, float* __restrict__ wp, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    reinterpret_cast<float4*>(wp)[i] = make_float4(w[3 * i], w[3 * i + 1], w[3 * i + 2], 0.f);
}
__global__ void __launch_bounds__(256)
unpad_c4_to_c3_kernel(const float* __restrict__ wp, float* __restrict__ w, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    const float4 v = reinterpret_cast<const float4*>(wp)[i];
    w[3 * i] = v.x; w[3 * i + 1] = v.y; w[3 * i + 2] = v.z;
}

// pads rows of [rows][d] to [rows][ld] with zeros (fc weight -> ld = round_up(d, 4))
__global__ void __launch_bounds__(256)
pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows, int d, int ld) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ld) return;
    const int64_t r = i / ld;
    const int c = (int)(i - r * ld);
    dst[i] = c < d ? src[r * d + c] : 0.f;
}

// ---------------------------------------------------------------------------------------------- batch norm
// partial[tile][3][C] (sum, sum of squares, max |x| per channel of a convolution M tile) -> per-channel sums (fp64), then
// scale/shift + saved statistics + running update, and -- when out_bound is given -- an upper bound of max |y| of the tensor
// the apply pass is about to produce:  |x scale + shift (+ residual)| <= |scale| max|x| + |shift| (+ *res_bound),  raised
// into *out_bound with ONE atomic per workgroup (no atomics in the big kernels; a ReLU only lowers it).  That bound is the
// power-of-two pre-scale of the split-fp16 convolution that consumes y (f16_split.h).
// GROUPS: the rows of one launch may hold several independent batches ("groups": forward(img_a) and forward(img_b) of a
// training step run as ONE launch sequence over 2B images); batch statistics are per group -- group g owns tiles
// [g*tiles, (g+1)*tiles) and the statistics block at offset g*gstride -- and the running statistics are updated once per
// group, in order, exactly as two consecutive nn.BatchNorm2d calls would.
// Work split of both finalize kernels: 4 channels x 64 row-parts per workgroup (C / 4 workgroups; a work-item walks
// tiles / 64 partial rows), parts reduced by a fixed xor-shuffle tree inside each wavefront, then across the 4 wavefronts
// through LDS in fixed order -- deterministic, and a 64-channel layer with 2 x 1024 partial rows finishes in microseconds.
template <class T> __device__ __forceinline__ T part_tree_sum(T v) {
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float part_tree_max(float v) {
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__global__ void __launch_bounds__(256)
bn_finalize_kernel(const float* __restrict__ partial, int tiles, int groups, int C, double count,
                   const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean,
                   float* __restrict__ running_var, float momentum, float eps, int training, float* __restrict__ scale,
                   float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_invstd, int gstride,
                   float* __restrict__ out_bound, const float* __restrict__ res_bound) {
    __shared__ double s_sum[4][4];
    __shared__ double s_sq[4][4];
    __shared__ float s_mx[4][4];
    __shared__ float s_bound[4];
    const int cl = threadIdx.x & 3, part = threadIdx.x >> 2, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + cl;
    // (round 6) what the four finishing work-items need besides the partials -- the affine parameters, the running statistics,
    // the bound words -- is requested NOW, in the shadow of the partial loads, instead of behind the reduction: the kernel is a
    // chain of dependent memory round trips (8 us for a few KB), 36 of them per forward pass with nothing else on the GPU
    const bool lead = threadIdx.x < 4 && c < C;
    const float pre_gamma = lead ? gamma[c] : 0.f, pre_beta = lead ? beta[c] : 0.f;
    const float pre_rm = (lead && running_mean) ? running_mean[c] : 0.f, pre_rv = (lead && running_mean) ? running_var[c] : 0.f;
    const float pre_res = (threadIdx.x == 0 && res_bound) ? *res_bound : 0.f;
    const unsigned pre_bound = (threadIdx.x == 0 && out_bound && training) ? __atomic_load_n(reinterpret_cast<unsigned*>(out_bound), __ATOMIC_RELAXED) : 0u;
    float bound = 0.f, rm_cur = pre_rm, rv_cur = pre_rv;
    for (int g = 0; g < groups; ++g) {
        double a = 0.0, b = 0.0;
        float mx = 0.f;
        if (training && c < C) {
            // (round 6) the partial rows of this work-item eight at a time as straight-line loads (a row past the end re-reads
            // the last one and is not added), then added in the order they always were: the `#pragma unroll 4` loop ran its
            // first (tiles / 64) mod 4 iterations one memory round trip each -- three of them for 200 tiles
            const int t_end = (g + 1) * tiles;
            for (int t0 = g * tiles + part; t0 < t_end; t0 += 64 * 8) {
                float bs[8], bq[8], bm[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = min(t0 + 64 * u, t_end - 1);
                    bs[u] = partial[((int64_t)t * 3 + 0) * C + c];
                    bq[u] = partial[((int64_t)t * 3 + 1) * C + c];
                    bm[u] = partial[((int64_t)t * 3 + 2) * C + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (t0 + 64 * u < t_end) {
                        a += (double)bs[u];
                        b += (double)bq[u];
                        mx = fmaxf(mx, bm[u]);
                    }
                }
            }
        }
        a = part_tree_sum(a);
        b = part_tree_sum(b);
        mx = part_tree_max(mx);
        __syncthreads();
        if ((threadIdx.x & 63) < 4) { s_sum[wv][cl] = a; s_sq[wv][cl] = b; s_mx[wv][cl] = mx; }
        __syncthreads();
        if (threadIdx.x >= 4 || c >= C) continue;
        if (g == 0) {   // (the prefetched words become visible to the optimiser HERE: no conversion of them hoisted above the partial loads)
            DCN_OPAQUE_INT(rm_cur); DCN_OPAQUE_INT(rv_cur);
        }
        float gam = pre_gamma, bet = pre_beta;
        DCN_OPAQUE_INT(gam); DCN_OPAQUE_INT(bet);
        double mean, var;
        if (training) {
            a = (s_sum[0][cl] + s_sum[1][cl]) + (s_sum[2][cl] + s_sum[3][cl]);
            b = (s_sq[0][cl] + s_sq[1][cl]) + (s_sq[2][cl] + s_sq[3][cl]);
            mean = a / count;
            var = b / count - mean * mean;
            if (var < 0.0) var = 0.0;
            if (running_mean) {
                // (a second group of the same call -- forward_pair -- updates what group 0 has just stored: two consecutive
                // nn.BatchNorm2d calls; the value is carried in a register)
                const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
                rm_cur = (float)((1.0 - momentum) * (double)rm_cur + momentum * mean);
                rv_cur = (float)((1.0 - momentum) * (double)rv_cur + momentum * unbiased);
                running_mean[c] = rm_cur;
                running_var[c] = rv_cur;
            }
        } else {
            mean = (double)rm_cur;
            var = (double)rv_cur;
        }
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gam * invstd;
        const float sh = bet - (float)mean * sc;
        scale[g * gstride + c] = sc;
        shift[g * gstride + c] = sh;
        if (save_mean) { save_mean[g * gstride + c] = (float)mean; save_invstd[g * gstride + c] = invstd; }
        mx = fmaxf(fmaxf(s_mx[0][cl], s_mx[1][cl]), fmaxf(s_mx[2][cl], s_mx[3][cl]));
        // (fmaxf drops a NaN: a non-finite statistic -- NaN sums from NaN activations, a NaN / inf parameter -- is reported as an
        // infinite bound, which the status word behind the abs-max slots flags)
        const float bnd = fabsf(sc) * mx + fabsf(sh);
        bound = (bnd != bnd) ? __builtin_huge_valf() : fmaxf(bound, bnd);
    }
    if (out_bound && training) {
        __syncthreads();   // (every work-item reaches this: the loop above has no early exit)
        if (threadIdx.x < 4) s_bound[threadIdx.x] = c < C ? bound : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(s_bound[0], s_bound[1]), fmaxf(s_bound[2], s_bound[3])) + pre_res;
            const unsigned bits = __float_as_uint(m);
            if (m > 0.f && bits > pre_bound)   // (a stale word only costs a redundant atomicMax)
                atomicMax(reinterpret_cast<unsigned*>(out_bound), bits);
        }
    }
}

// float4 load that will not be needed again soon (DCN_BN_NT: streamed past the caches instead of displacing what the next
// kernel is about to read)
__device__ __forceinline__ float4 ld4(const float* p, int64_t i4, bool nt) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v* q = reinterpret_cast<const f4v*>(p) + i4;
    const f4v v = nt ? __builtin_nontemporal_load(q) : *q;
    return make_float4(v[0], v[1], v[2], v[3]);
}

// y = [relu]( x*s1 + b1 + (res ? (s2 ? res*s2 + b2 : res) : 0) )
__global__ void __launch_bounds__(256)
bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ s1, const float* __restrict__ b1,
                const float* __restrict__ res, const float* __restrict__ s2, const float* __restrict__ b2, int relu,
                float* __restrict__ y, unsigned char* __restrict__ relu_mask, int c4n, int64_t total4, int64_t group4,
                int gstride, dcnsplit::u32x2* __restrict__ hl, const float* __restrict__ hl_absmax, int rev, int nt) {
    // hl (optional, c4n % 8 == 0): y also as the "hl32" image the pre-split convolution kernel reads (conv_hl_kernels.hip) --
    // per 32-channel chunk one 128-byte line [hi x32 | lo x32] fp16 of s y, s = the power of two chosen from *hl_absmax (the
    // bound of max |y| that bn_finalize_kernel stored before this pass): the consumer's operand split costs 4 B / element
    // of extra stores here instead of a pass of its own
    const float hs = hl ? dcnsplit::pow2_scale(*hl_absmax) : 1.f;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < total4; j += (int64_t)gridDim.x * 256) {
        const int64_t i = rev ? total4 - 1 - j : j;   // (rev: back to front, DCN_BN_REVERSE)
        const int c = (int)(i % c4n) * 4 + (i >= group4 ? gstride : 0);   // (at most two groups: second group's statistics)
        const float4 v = ld4(x, i, nt != 0);
        const float4 s = *reinterpret_cast<const float4*>(s1 + c);
        const float4 b = *reinterpret_cast<const float4*>(b1 + c);
        float4 o = make_float4(v.x * s.x + b.x, v.y * s.y + b.y, v.z * s.z + b.z, v.w * s.w + b.w);
        if (res) {
            float4 r = reinterpret_cast<const float4*>(res)[i];
            if (s2) {
                const float4 t = *reinterpret_cast<const float4*>(s2 + c);
                const float4 u = *reinterpret_cast<const float4*>(b2 + c);
                r = make_float4(r.x * t.x + u.x, r.y * t.y + u.y, r.z * t.z + u.z, r.w * t.w + u.w);
            }
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        if (y) reinterpret_cast<float4*>(y)[i] = o;   // (null: every reader of this tensor takes the hl32 image below)
        // which of the four outputs are positive: ONE byte per float4, so that the backward passes read 1 byte instead
        // of 16 to rebuild the ReLU mask
        if (relu_mask)
            relu_mask[i] = (unsigned char)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
        if (hl) {   // float4 i = channel quad (i & 7) of 32-channel chunk (i >> 3): 8 bytes of the hi half-line, 8 of the lo one
            dcnsplit::h4 a, b;
            dcnsplit::split4(o, hs, a, b);
            dcnsplit::u32x2* line = hl + (i >> 3) * 16 + (i & 7);
            line[0] = __builtin_bit_cast(dcnsplit::u32x2, a);
            line[8] = __builtin_bit_cast(dcnsplit::u32x2, b);
        }
    }
}

// relu-masked upstream gradient: from the one-byte-per-float4 mask written by the forward apply pass when there is one,
// otherwise from the activation itself
__device__ __forceinline__ float4 relu_masked(float4 g, const float* relu_out, const unsigned char* relu_mask, int64_t i4) {
    if (relu_mask) {
        const unsigned m = relu_mask[i4];
        g.x = (m & 1u) ? g.x : 0.f; g.y = (m & 2u) ? g.y : 0.f; g.z = (m & 4u) ? g.z : 0.f; g.w = (m & 8u) ? g.w : 0.f;
    } else if (relu_out) {
        const float4 y = reinterpret_cast<const float4*>(relu_out)[i4];
        g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f; g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
    }
    return g;
}

// upstream gradient given as the sum of two tensors (dy2 nullable): the residual branch's gradient is added HERE, in the
// streaming passes, instead of in the epilogue of the dgrad that produced dy (measured: the one-workgroup-per-CU GEMM
// exposes the epilogue's scalar loads -- +22 % on a layer-4 dgrad, +43 % on layer 3, profiles/r2h_kernel_stats_before_add_move.txt)
// (dy2 may be the very buffer the apply pass writes g_out to -- same index, read before written: no __restrict__ on it)
__device__ __forceinline__ float4 load_dy(const float* __restrict__ dy, const float* dy2, int64_t i4) {
    float4 g = reinterpret_cast<const float4*>(dy)[i4];
    if (dy2) {
        const float4 h = reinterpret_cast<const float4*>(dy2)[i4];
        g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
    }
    return g;
}

// Backward reduction.  g = relu_out ? (relu_out > 0 ? dy : 0) : dy.
// partial[chunk][C][4] = ( sum g , sum g * xhat , max |g| , max |xhat| ) over the chunk's rows, xhat = (x - mean) * invstd
// (the four statistics of a channel are one float4: the finalize kernel reads them with a single load).
// (The two maxima bound |dx| per channel without another pass or any atomics in the big kernels: bn_bwd_finalize.)
// work-item (c4 = tid % CQ, rl = tid / CQ): CQ channel quads x 256 / CQ row lanes; grid = (C / (4 CQ) groups, chunks).
// CQ = 16 (64 channels per workgroup) is the general shape; the wider ones (round 4) make a workgroup read WHOLE rows of up
// to 512 channels -- one contiguous stream per tensor instead of 256-byte pieces strided by the row pitch.
template <int CQ>
__global__ void __launch_bounds__(256)
bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* dy2, const float* __restrict__ relu_out,
                     const unsigned char* __restrict__ relu_mask, const float* __restrict__ x,
                     const float* __restrict__ mean, const float* __restrict__ invstd, int C, int64_t rows_per_group,
                     int chunks_per_group, int gstride, int rows_per_chunk, float* __restrict__ partial) {
    constexpr int RL = 256 / CQ;
    __shared__ float4 s_g[RL][CQ];
    __shared__ float4 s_gx[RL][CQ];
    __shared__ float4 s_mg[RL][CQ];
    __shared__ float4 s_mx[RL][CQ];
    const int c4 = threadIdx.x % CQ, rl = threadIdx.x / CQ;
    const int c = blockIdx.x * (4 * CQ) + c4 * 4;
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), agx = ag, mg = ag, mx = ag;
    if (c < C) {
        const int g = (int)blockIdx.y / chunks_per_group;            // chunks never straddle a group boundary
        const float4 mu = *reinterpret_cast<const float4*>(mean + g * gstride + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + g * gstride + c);
        const int64_t r0 = g * rows_per_group + (int64_t)((int)blockIdx.y - g * chunks_per_group) * rows_per_chunk;
        int64_t r1 = r0 + rows_per_chunk;
        if (r1 > (g + 1) * rows_per_group) r1 = (g + 1) * rows_per_group;
#pragma unroll 4   // (8 loads in flight per work-item instead of 2: 27 -> 23 us per launch)
        for (int64_t r = r0 + rl; r < r1; r += RL) {
            const int64_t o = r * C + c;
            const float4 g = relu_masked(load_dy(dy, dy2, o >> 2), relu_out, relu_mask, o >> 2);
            const float4 v = *reinterpret_cast<const float4*>(x + o);
            const float4 xh = make_float4((v.x - mu.x) * is.x, (v.y - mu.y) * is.y, (v.z - mu.z) * is.z, (v.w - mu.w) * is.w);
            ag.x += g.x; ag.y += g.y; ag.z += g.z; ag.w += g.w;
            agx.x = fmaf(g.x, xh.x, agx.x); agx.y = fmaf(g.y, xh.y, agx.y);
            agx.z = fmaf(g.z, xh.z, agx.z); agx.w = fmaf(g.w, xh.w, agx.w);
            mg.x = fmaxf(mg.x, fabsf(g.x)); mg.y = fmaxf(mg.y, fabsf(g.y));
            mg.z = fmaxf(mg.z, fabsf(g.z)); mg.w = fmaxf(mg.w, fabsf(g.w));
            mx.x = fmaxf(mx.x, fabsf(xh.x)); mx.y = fmaxf(mx.y, fabsf(xh.y));
            mx.z = fmaxf(mx.z, fabsf(xh.z)); mx.w = fmaxf(mx.w, fabsf(xh.w));
        }
    }
    s_g[rl][c4] = ag;
    s_gx[rl][c4] = agx;
    s_mg[rl][c4] = mg;
    s_mx[rl][c4] = mx;
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int i = 1; i < RL; ++i) {
            const float4 a = s_g[i][c4], b = s_gx[i][c4], m1 = s_mg[i][c4], m2 = s_mx[i][c4];
            ag.x += a.x; ag.y += a.y; ag.z += a.z; ag.w += a.w;
            agx.x += b.x; agx.y += b.y; agx.z += b.z; agx.w += b.w;
            mg.x = fmaxf(mg.x, m1.x); mg.y = fmaxf(mg.y, m1.y); mg.z = fmaxf(mg.z, m1.z); mg.w = fmaxf(mg.w, m1.w);
            mx.x = fmaxf(mx.x, m2.x); mx.y = fmaxf(mx.y, m2.y); mx.z = fmaxf(mx.z, m2.z); mx.w = fmaxf(mx.w, m2.w);
        }
        float4* p = reinterpret_cast<float4*>(partial) + (int64_t)blockIdx.y * C + c;
        p[0] = make_float4(ag.x, agx.x, mg.x, mx.x);
        p[1] = make_float4(ag.y, agx.y, mg.y, mx.y);
        p[2] = make_float4(ag.z, agx.z, mg.z, mx.z);
        p[3] = make_float4(ag.w, agx.w, mg.w, mx.w);
    }
}

// partial[chunk][C][4] -> dgamma, dbeta and the coefficients of the apply pass:
//   dx = k1 * (g - k2 - xhat * k3),  k1 = gamma*invstd, k2 = sum_g / M, k3 = sum_gx / M
// and, when absmax is given, raises absmax[0] to  max_c |k1| (max|g| + |k2| + max|xhat| |k3|)  >=  max |dx|  (the
// pre-scale of the split-fp16 convolutions only needs an upper bound within a small factor of the true abs-max).
__global__ void __launch_bounds__(256)
bn_bwd_finalize_kernel(const float* __restrict__ partial, int chunks, int groups, int C, double count,
                       const float* __restrict__ gamma, const float* __restrict__ invstd, int gstride,
                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ k123,
                       float* __restrict__ absmax) {
    // 4 channels x 64 chunk-parts per workgroup (see bn_finalize_kernel).  Per group g (chunks [g*chunks, (g+1)*chunks)):
    // k123[g][0..2][C]; dgamma / dbeta are the sums over the groups (the parameters are shared).
    __shared__ double s_a[4][4];
    __shared__ double s_b[4][4];
    __shared__ float s_mg[4][4];
    __shared__ float s_mx[4][4];
    const int cl = threadIdx.x & 3, part = threadIdx.x >> 2, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + cl;
    const bool lead = threadIdx.x < 4 && c < C;
    // (round 6: requested in the shadow of the partial loads, see bn_finalize_kernel; at most two groups)
    const float pre_gamma = lead ? gamma[c] : 0.f;
    const float pre_is0 = lead ? invstd[c] : 0.f, pre_is1 = (lead && groups > 1) ? invstd[gstride + c] : 0.f;
    const unsigned pre_bound = (threadIdx.x == 0 && absmax) ? __atomic_load_n(reinterpret_cast<unsigned*>(absmax), __ATOMIC_RELAXED) : 0u;
    double tot_a = 0.0, tot_b = 0.0;
    float bound = 0.f;
    for (int g = 0; g < groups; ++g) {
        double a = 0.0, b = 0.0;
        float mg = 0.f, mx = 0.f;
        if (c < C) {
            // (round 6) the partial rows of this work-item EIGHT loads at a time, then added in the order they always were: the
            // compiler had turned the `#pragma unroll 4` loop into load | wait | add | load | wait ... -- five dependent memory
            // round trips for the five chunks a work-item owns at eight images: that WAS the kernel's 8 us
            const int t_end = (g + 1) * chunks;
            for (int t0 = g * chunks + part; t0 < t_end; t0 += 64 * 8) {
                float4 buf[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = min(t0 + 64 * u, t_end - 1);   // (straight-line loads: a row past the end re-reads the last one ...)
                    buf[u] = reinterpret_cast<const float4*>(partial)[(int64_t)t * C + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (t0 + 64 * u < t_end) {                  // (... and is not added)
                        a += (double)buf[u].x;
                        b += (double)buf[u].y;
                        mg = fmaxf(mg, buf[u].z);
                        mx = fmaxf(mx, buf[u].w);
                    }
                }
            }
        }
        a = part_tree_sum(a);
        b = part_tree_sum(b);
        mg = part_tree_max(mg);
        mx = part_tree_max(mx);
        __syncthreads();
        if ((threadIdx.x & 63) < 4) { s_a[wv][cl] = a; s_b[wv][cl] = b; s_mg[wv][cl] = mg; s_mx[wv][cl] = mx; }
        __syncthreads();
        if (lead) {
            a = (s_a[0][cl] + s_a[1][cl]) + (s_a[2][cl] + s_a[3][cl]);
            b = (s_b[0][cl] + s_b[1][cl]) + (s_b[2][cl] + s_b[3][cl]);
            mg = fmaxf(fmaxf(s_mg[0][cl], s_mg[1][cl]), fmaxf(s_mg[2][cl], s_mg[3][cl]));
            mx = fmaxf(fmaxf(s_mx[0][cl], s_mx[1][cl]), fmaxf(s_mx[2][cl], s_mx[3][cl]));
            float is_g = g == 0 ? pre_is0 : (g == 1 ? pre_is1 : invstd[g * gstride + c]), gam = pre_gamma;
            DCN_OPAQUE_INT(is_g); DCN_OPAQUE_INT(gam);
            const float c1 = gam * is_g, c2 = (float)(a / count), c3 = (float)(b / count);
            float* k = k123 + (int64_t)g * 3 * C;
            k[c] = c1;
            k[C + c] = c2;
            k[2 * C + c] = c3;
            tot_a += a;
            tot_b += b;
            bound = fmaxf(bound, fabsf(c1) * (mg + fabsf(c2) + mx * fabsf(c3)));
        }
    }
    if (lead) {
        dbeta[c] = (float)tot_a;
        dgamma[c] = (float)tot_b;
    }
    if (absmax && threadIdx.x < 64) {   // the 4 leaders sit in lanes 0..3 of wavefront 0
        bound = fmaxf(bound, __shfl_xor(bound, 1));
        bound = fmaxf(bound, __shfl_xor(bound, 2));
        if (threadIdx.x == 0) {
            const unsigned bits = __float_as_uint(bound);
            if (bound > 0.f && bits > pre_bound)   // (a stale word only costs a redundant atomicMax)
                atomicMax(reinterpret_cast<unsigned*>(absmax), bits);
        }
    }
}

// dx = k1*(g - k2 - (x-mean)*invstd*k3); optionally also writes g (the relu-masked upstream gradient) for the
// residual branch.
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const float* __restrict__ dy, const float* dy2, const float* __restrict__ relu_out,
                    const unsigned char* __restrict__ relu_mask, const float* __restrict__ x,
                    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ k1,
                    const float* __restrict__ k2, const float* __restrict__ k3, float* __restrict__ dx,
                    float* g_out, int c4n, int64_t total4, int64_t group4, int gstride, int kstride, int rev) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < total4; j += (int64_t)gridDim.x * 256) {
        const int64_t i = rev ? total4 - 1 - j : j;
        const int c0 = (int)(i % c4n) * 4;
        const bool second = i >= group4;                          // (at most two groups)
        const int c = c0 + (second ? gstride : 0), ck = c0 + (second ? kstride : 0);
        const float4 g = relu_masked(load_dy(dy, dy2, i), relu_out, relu_mask, i);
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        const float4 a = *reinterpret_cast<const float4*>(k1 + ck);
        const float4 b = *reinterpret_cast<const float4*>(k2 + ck);
        const float4 d = *reinterpret_cast<const float4*>(k3 + ck);
        float4 o;
        o.x = a.x * (g.x - b.x - (v.x - mu.x) * is.x * d.x);
        o.y = a.y * (g.y - b.y - (v.y - mu.y) * is.y * d.y);
        o.z = a.z * (g.z - b.z - (v.z - mu.z) * is.z * d.z);
        o.w = a.w * (g.w - b.w - (v.w - mu.w) * is.w * d.w);
        reinterpret_cast<float4*>(dx)[i] = o;
        if (g_out) reinterpret_cast<float4*>(g_out)[i] = g;
    }
}

// The same pass for the split-fp16 convolution mode: one work-item per (pixel quad, channel quad) so that it can ALSO emit
// dx as the pixel-blocked split tensor wgrad consumes (f16_split.h), scaled by the power of two chosen from the bound the
// finalize kernel has just stored in *absmax -- no separate split pass over dx.  rows_per_group % 4 == 0 when grouped.
__global__ void __launch_bounds__(256)
bn_bwd_apply_blocked_kernel(const float* __restrict__ dy, const float* dy2, const float* __restrict__ relu_out,
                            const unsigned char* __restrict__ relu_mask, const float* __restrict__ x,
                            const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ k1,
                            const float* __restrict__ k2, const float* __restrict__ k3, float* __restrict__ dx,
                            float* g_out, dcnsplit::u32x4* __restrict__ dq, const float* __restrict__ absmax,
                            int c4n, int64_t rows, int64_t rows_per_group, int gstride, int kstride,
                            dcnsplit::u32x2* __restrict__ hl, int rev, int nt) {
    // hl (optional, c4n % 8 == 0): dx as the hl32 image the pre-split dgrad reads (same scale as dq); dx itself may then be
    // null -- nobody else reads the fp32 tensor
    const float s = dcnsplit::pow2_scale(*absmax);
    const int64_t total = ((rows + 3) >> 2) * c4n;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < total; j += (int64_t)gridDim.x * 256) {
        const int64_t i = rev ? total - 1 - j : j;
        const int64_t q = i / c4n;
        const int cq = (int)(i - q * c4n);
        const bool second = q * 4 >= rows_per_group;
        const int c = cq * 4 + (second ? gstride : 0), ck = cq * 4 + (second ? kstride : 0);
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        const float4 a = *reinterpret_cast<const float4*>(k1 + ck);
        const float4 b = *reinterpret_cast<const float4*>(k2 + ck);
        const float4 d = *reinterpret_cast<const float4*>(k3 + ck);
        float o[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t m = q * 4 + r;
            if (m < rows) {
                const int64_t e = m * c4n + cq;
                float4 g0 = ld4(dy, e, nt != 0);
                if (dy2) {
                    const float4 h = reinterpret_cast<const float4*>(dy2)[e];
                    g0.x += h.x; g0.y += h.y; g0.z += h.z; g0.w += h.w;
                }
                const float4 g = relu_masked(g0, relu_out, relu_mask, e);
                const float4 v = ld4(x, e, nt != 0);
                o[r][0] = a.x * (g.x - b.x - (v.x - mu.x) * is.x * d.x);
                o[r][1] = a.y * (g.y - b.y - (v.y - mu.y) * is.y * d.y);
                o[r][2] = a.z * (g.z - b.z - (v.z - mu.z) * is.z * d.z);
                o[r][3] = a.w * (g.w - b.w - (v.w - mu.w) * is.w * d.w);
                if (dx) reinterpret_cast<float4*>(dx)[e] = make_float4(o[r][0], o[r][1], o[r][2], o[r][3]);
                if (g_out) reinterpret_cast<float4*>(g_out)[e] = g;
                if (hl) {
                    dcnsplit::h4 ha, hb;
                    dcnsplit::split4(make_float4(o[r][0], o[r][1], o[r][2], o[r][3]), s, ha, hb);
                    dcnsplit::u32x2* line = hl + (e >> 3) * 16 + (e & 7);
                    line[0] = __builtin_bit_cast(dcnsplit::u32x2, ha);
                    line[8] = __builtin_bit_cast(dcnsplit::u32x2, hb);
                }
            } else {
                o[r][0] = o[r][1] = o[r][2] = o[r][3] = 0.f;
            }
        }
        if (dq) dcnsplit::store_blocked_quad(dq, q, cq, c4n, o, s);   // (null: the weight gradient reads the hl32 image)
    }
}

// absmax[0] = max(absmax[0], max |v[i]|)
__global__ void __launch_bounds__(256)
absmax_kernel(const float* __restrict__ v, int64_t total, float* __restrict__ absmax) {
    float amax = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        amax = fmaxf(amax, fabsf(v[i]));
    wave_atomic_absmax(amax, absmax);
}

// out = a + b
__global__ void __launch_bounds__(256)
add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t total4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}

// ---------------------------------------------------------------------------------------------- max pool 3x3 / 2 / pad 1
// scale / shift (optional): the input is a convolution output whose batch norm + ReLU is applied HERE, on the way into the
// window maximum -- y = relu(in * scale + shift) is never stored (the stem: 157 MB per 8 images read and written by an apply
// pass of its own otherwise).  relu_mask then receives the one-byte-per-float4 sign mask of y that the batch norm's backward
// pass reads (every input belongs to at least one window; windows overlap: the same byte is written up to four times).
__global__ void __launch_bounds__(256)
maxpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, unsigned char* __restrict__ argmax, int hin,
                   int win, int hout, int wout, int c4n, int64_t total4, const float* __restrict__ scale,
                   const float* __restrict__ shift, int gstride, int64_t group_n, unsigned char* __restrict__ relu_mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over n*hout*wout*c4n
    if (i >= total4) return;
    const int c4 = (int)(i % c4n);
    int64_t pix = i / c4n;
    const int ox = (int)(pix % wout); pix /= wout;
    const int oy = (int)(pix % hout);
    const int64_t n = pix / hout;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale) {
        const int c = c4 * 4 + (n >= group_n ? gstride : 0);   // (at most two statistics groups)
        sc = *reinterpret_cast<const float4*>(scale + c);
        sh = *reinterpret_cast<const float4*>(shift + c);
    }
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int bx = -1, by = -1, bz = -1, bw = -1;
    for (int r = 0; r < 3; ++r) {
        const int iy = oy * 2 - 1 + r;
        if (iy < 0 || iy >= hin) continue;
        for (int s = 0; s < 3; ++s) {
            const int ix = ox * 2 - 1 + s;
            if (ix < 0 || ix >= win) continue;
            const int64_t src = ((n * hin + iy) * win + ix) * c4n + c4;
            float4 v = reinterpret_cast<const float4*>(in)[src];
            if (scale) {   // (the expressions of bn_apply_kernel: same bits)
                v = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                if (relu_mask)
                    relu_mask[src] = (unsigned char)((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0));
            }
            const int t = r * 3 + s;
            if (v.x > best.x || bx < 0 || v.x != v.x) { best.x = v.x; bx = t; }
            if (v.y > best.y || by < 0 || v.y != v.y) { best.y = v.y; by = t; }
            if (v.z > best.z || bz < 0 || v.z != v.z) { best.z = v.z; bz = t; }
            if (v.w > best.w || bw < 0 || v.w != v.w) { best.w = v.w; bw = t; }
        }
    }
    reinterpret_cast<float4*>(out)[i] = best;
    if (argmax) {
        unsigned char* a = argmax + i * 4;
        a[0] = (unsigned char)bx; a[1] = (unsigned char)by; a[2] = (unsigned char)bz; a[3] = (unsigned char)bw;
    }
}

// gather form: every input element collects from the (<= 4) windows that contain it
__global__ void __launch_bounds__(256)
maxpool_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ gout2, const unsigned char* __restrict__ argmax,
                   float* __restrict__ gin, int hin, int win, int hout, int wout, int c4n, int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over n*hin*win*c4n
    if (i >= total4) return;
    const int c4 = (int)(i % c4n);
    int64_t pix = i / c4n;
    const int ix = (int)(pix % win); pix /= win;
    const int iy = (int)(pix % hin);
    const int64_t n = pix / hin;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
        if (oy >= hout) continue;
        const int r = iy - (oy * 2 - 1);
        if (r < 0 || r > 2) continue;
        for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
            if (ox >= wout) continue;
            const int s = ix - (ox * 2 - 1);
            if (s < 0 || s > 2) continue;
            const int t = r * 3 + s;
            const int64_t o = ((n * hout + oy) * wout + ox) * c4n + c4;
            float4 g = reinterpret_cast<const float4*>(gout)[o];
            if (gout2) {   // the pooled tensor's gradient arrives as a sum of two (first block's dgrad + its identity gradient)
                const float4 h = reinterpret_cast<const float4*>(gout2)[o];
                g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
            }
            const unsigned char* a = argmax + o * 4;
            if (a[0] == t) acc.x += g.x;
            if (a[1] == t) acc.y += g.y;
            if (a[2] == t) acc.z += g.z;
            if (a[3] == t) acc.w += g.w;
        }
    }
    reinterpret_cast<float4*>(gin)[i] = acc;
}

// ---------------------------------------------------------------------------------------------- bilinear upsample
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    const float s = scale * (float)dst;  // align_corners=True: scale = (in-1)/(out-1)
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 < in_size - 1 ? i0 + 1 : i0;
    l1 = s - (float)i0;
}

// one work-item per output pixel; low-res rows have pitch ldl (>= d)
__global__ void __launch_bounds__(256)
upsample_fwd_kernel(const float* __restrict__ low, int hl, int wl, int ldl, int d, int h, int w, float sh, float sw,
                    int normalize, float* __restrict__ out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over n*h*w
    if (i >= total) return;
    const int x = (int)(i % w);
    const int64_t t = i / w;
    const int y = (int)(t % h);
    const int64_t n = t / h;
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(y, sh, hl, y0, y1, ly);
    src_index(x, sw, wl, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p00 = low + ((n * hl + y0) * wl + x0) * ldl;
    const float* p01 = low + ((n * hl + y0) * wl + x1) * ldl;
    const float* p10 = low + ((n * hl + y1) * wl + x0) * ldl;
    const float* p11 = low + ((n * hl + y1) * wl + x1) * ldl;
    float* o = out + i * d;
    if (!normalize) {
        for (int k = 0; k < d; ++k) o[k] = hy * (hx * p00[k] + lx * p01[k]) + ly * (hx * p10[k] + lx * p11[k]);
    } else {
        float ss = 0.f;
        for (int k = 0; k < d; ++k) {
            const float v = hy * (hx * p00[k] + lx * p01[k]) + ly * (hx * p10[k] + lx * p11[k]);
            ss = fmaf(v, v, ss);
        }
        const float nrm = sqrtf(ss);
        for (int k = 0; k < d; ++k)
            o[k] = (hy * (hx * p00[k] + lx * p01[k]) + ly * (hx * p10[k] + lx * p11[k])) / nrm;
    }
}

// backward of the fused L2 normalisation y = v / ||v||:  g' = (g - y (y . g)) / ||v||, v re-interpolated from `low`
__global__ void __launch_bounds__(256)
normalize_bwd_kernel(const float* __restrict__ low, int hl, int wl, int ldl, int d, int h, int w, float sh, float sw,
                     const float* __restrict__ gout, float* __restrict__ gv, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over n*h*w
    if (i >= total) return;
    const int x = (int)(i % w);
    const int64_t t = i / w;
    const int y = (int)(t % h);
    const int64_t n = t / h;
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(y, sh, hl, y0, y1, ly);
    src_index(x, sw, wl, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p00 = low + ((n * hl + y0) * wl + x0) * ldl;
    const float* p01 = low + ((n * hl + y0) * wl + x1) * ldl;
    const float* p10 = low + ((n * hl + y1) * wl + x0) * ldl;
    const float* p11 = low + ((n * hl + y1) * wl + x1) * ldl;
    const float* g = gout + i * d;
    float ss = 0.f, dot = 0.f;
    for (int k = 0; k < d; ++k) {
        const float v = hy * (hx * p00[k] + lx * p01[k]) + ly * (hx * p10[k] + lx * p11[k]);
        ss = fmaf(v, v, ss);
        dot = fmaf(v, g[k], dot);
    }
    const float nrm = sqrtf(ss), inv = 1.f / nrm;
    dot *= inv;  // y . g
    for (int k = 0; k < d; ++k) {
        const float v = hy * (hx * p00[k] + lx * p01[k]) + ly * (hx * p10[k] + lx * p11[k]);
        gv[i * d + k] = (g[k] - v * inv * dot) * inv;
    }
}

// backward, separable and gather-form (deterministic).  pass 1: tmp[n][iy][x][d] = sum_y wy(y, iy) * gout[n][y][x][d]
__global__ void __launch_bounds__(256)
upsample_bwd_rows_kernel(const float* __restrict__ gout, int hl, int d, int h, int w, float sh, float* __restrict__ tmp,
                         int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over n*hl*w*d
    if (i >= total) return;
    const int64_t wd = (int64_t)w * d;
    const int64_t xd = i % wd;
    const int64_t t = i / wd;
    const int iy = (int)(t % hl);
    const int64_t n = t / hl;
    const float inv = sh > 0.f ? 1.f / sh : 0.f;
    int ylo = (int)floorf(((float)iy - 1.f) * inv) - 1, yhi = (int)ceilf(((float)iy + 1.f) * inv) + 1;
    if (ylo < 0) ylo = 0;
    if (yhi > h - 1 || sh <= 0.f) yhi = h - 1;
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
        int y0, y1;
        float ly;
        src_index(y, sh, hl, y0, y1, ly);
        float wgt = 0.f;
        if (y0 == iy) wgt += 1.f - ly;
        if (y1 == iy) wgt += ly;
        if (wgt != 0.f) acc = fmaf(wgt, gout[(n * h + y) * wd + xd], acc);
    }
    tmp[i] = acc;
}
// pass 2: glow[n][iy][ix][0..ldl) = sum_x wx(x, ix) * tmp[n][iy][x][k]  (k >= d -> 0)
__global__ void __launch_bounds__(256)
upsample_bwd_cols_kernel(const float* __restrict__ tmp, int hl, int wl, int ldl, int d, int w, float sw,
                         float* __restrict__ glow, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over n*hl*wl*ldl
    if (i >= total) return;
    const int k = (int)(i % ldl);
    const int64_t t = i / ldl;
    const int ix = (int)(t % wl);
    const int64_t row = t / wl;  // n*hl + iy
    if (k >= d) { glow[i] = 0.f; return; }
    const float inv = sw > 0.f ? 1.f / sw : 0.f;
    int xlo = (int)floorf(((float)ix - 1.f) * inv) - 1, xhi = (int)ceilf(((float)ix + 1.f) * inv) + 1;
    if (xlo < 0) xlo = 0;
    if (xhi > w - 1 || sw <= 0.f) xhi = w - 1;
    float acc = 0.f;
    for (int x = xlo; x <= xhi; ++x) {
        int x0, x1;
        float lx;
        src_index(x, sw, wl, x0, x1, lx);
        float wgt = 0.f;
        if (x0 == ix) wgt += 1.f - lx;
        if (x1 == ix) wgt += lx;
        if (wgt != 0.f) acc = fmaf(wgt, tmp[(row * w + x) * d + k], acc);
    }
    glow[i] = acc;
}

// ---------------------------------------------------------------------------------------------- host launchers
static inline unsigned blocks_for(int64_t n, int cap = 0) {
    int64_t b = ceil_div64(n, 256);
    if (cap > 0 && b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}
constexpr int kGridCap = 256 * 8;  // grid-stride kernels: 8 workgroups per CU

thread_local const LaunchObserver* launch_observer = nullptr;

void launch_nchw3_to_nhwc4(const float* img, float* out, int n, int hw, float* absmax, hipStream_t st) {
    const int64_t total = (int64_t)n * hw;
    ObservedLaunch obs(DCN_PROF_RESAMPLE, 28.0 * (double)total, st);   // 3 floats in, 4 out per pixel
    hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(blocks_for(total, kGridCap)), dim3(256), 0, st, img, out, hw, total, absmax);
}
void launch_pad_c3_to_c4(const float* w, float* wp, int64_t rows, hipStream_t st) {
    ObservedLaunch obs(DCN_PROF_OTHER, 28.0 * (double)rows, st);
    hipLaunchKernelGGL(pad_c3_to_c4_kernel, dim3(blocks_for(rows)), dim3(256), 0, st, w, wp, rows);
}
void launch_unpad_c4_to_c3(const float* wp, float* w, int64_t rows, hipStream_t st) {
    ObservedLaunch obs(DCN_PROF_OTHER, 28.0 * (double)rows, st);
    hipLaunchKernelGGL(unpad_c4_to_c3_kernel, dim3(blocks_for(rows)), dim3(256), 0, st, wp, w, rows);
}
void launch_pad_rows(const float* src, float* dst, int64_t rows, int d, int ld, hipStream_t st) {
    ObservedLaunch obs(DCN_PROF_OTHER, 4.0 * (double)rows * (d + ld), st);
    hipLaunchKernelGGL(pad_rows_kernel, dim3(blocks_for(rows * ld)), dim3(256), 0, st, src, dst, rows, d, ld);
}
void launch_bn_finalize(const float* partial, int tiles_per_group, int groups, int C, double count_per_group,
                        const float* gamma, const float* beta, float* rmean, float* rvar, float momentum, float eps,
                        int training, float* stats, float* out_bound, const float* res_bound, hipStream_t st) {
    ObservedLaunch obs(DCN_PROF_BN_FINALIZE, training ? 12.0 * (double)tiles_per_group * groups * C : 0.0, st);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, partial, tiles_per_group, groups, C,
                       count_per_group, gamma, beta, rmean, rvar, momentum, eps, training, stats, stats + C, stats + 2 * C,
                       stats + 3 * C, 4 * C, out_bound, res_bound);
}
void launch_bn_apply(const float* x, const float* stats1, const float* res, const float* stats2, int relu, float* y,
                     unsigned char* relu_mask, int C, int64_t rows, int groups, hipStream_t st, void* hl_out,
                     const float* hl_absmax) {
    const int64_t total4 = rows * (C / 4);
    if ((C % 32) != 0 || !hl_absmax) hl_out = nullptr;
    // bytes per element: x (+ the residual) in; y, the hl32 image (the size of y) and a quarter byte of ReLU mask out
    ObservedLaunch obs(DCN_PROF_BN_APPLY, (double)rows * C * (4.0 + (res ? 4.0 : 0.0) + (y ? 4.0 : 0.0) + (hl_out ? 4.0 : 0.0) +
                                                             (relu_mask ? 0.25 : 0.0)), st);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks_for(total4, kGridCap)), dim3(256), 0, st, x, stats1, stats1 + C, res,
                       stats2, stats2 ? stats2 + C : nullptr, relu, y, relu_mask, C / 4, total4, total4 / groups, 4 * C,
                       (dcnsplit::u32x2*)hl_out, hl_absmax, tuning().bn_reverse & 1, tuning().bn_nt & 1);
}
int bn_bwd_chunks(int64_t rows_per_group) {
    // enough row chunks that even a 64-channel layer launches >= ~1000 workgroups (HBM-bound pass: fill all 256 CUs)
    // (measured at 8 images: 64 / 128 / 256 / 512 rows -> 24.69 / 24.60 / 24.62 / 25.00 ms per step).  Round 5: at one or two
    // images 128-row chunks are 38-75 workgroups per 128 channels -- the reduce pass of a 10 MB tensor took 20 us, latency-bound
    // (profiles/r5a_bench_default.json: 0.74 ms per config-1 step for 36 launches): smaller chunks below 16 384 rows
    const int rows = rows_per_group >= 16384 ? 128 : (rows_per_group >= 8192 ? 64 : 32);
    int64_t chunks = ceil_div64(rows_per_group, rows);
    if (chunks > 1024) chunks = 1024;
    if (chunks < 1) chunks = 1;
    return (int)chunks;
}
void launch_bn_bwd(const float* dy, const float* relu_out, const unsigned char* relu_mask, const float* x, const float* stats,
                   const float* gamma, int C,
                   int64_t rows, int groups, float* partial, float* dgamma, float* dbeta, float* k123, float* dx,
                   float* g_out, float* absmax, void* dq, hipStream_t st, int reduced_tiles_per_group, const float* dy2,
                   void* hl_dx, int keep_dx) {
    const int64_t rpg = rows / groups;
    const float* mean = stats + 2 * C;
    const float* invstd = stats + 3 * C;
    int chunks = reduced_tiles_per_group;
    // bytes per element read by both streaming passes: dy (+ dy2), x, the ReLU mask byte per float4 (or the activation)
    const double in_bytes = 8.0 + (dy2 ? 4.0 : 0.0) + (relu_mask ? 0.25 : (relu_out ? 4.0 : 0.0));
    if (chunks <= 0) {
        chunks = bn_bwd_chunks(rpg);   // per group
        const int rpc = (int)ceil_div64(rpg, chunks);
        ObservedLaunch obs(DCN_PROF_BN_BWD_REDUCE, (double)rows * C * in_bytes, st);
        const int cap = tuning().bn_reduce_wide;   // DCN_BN_REDUCE_WIDE: 16 / 32 / 64 / 128 channel quads per workgroup at most
        const int cq = ((C % 512) == 0 && cap >= 128) ? 128 : (((C % 256) == 0 && cap >= 64) ? 64 : (((C % 128) == 0 && cap >= 32) ? 32 : 16));
#define DCN_BN_RED(CQ)                                                                                                        \
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<CQ>, dim3(ceil_div(C, 4 * CQ), chunks * groups), dim3(256), 0, st, dy, dy2, relu_out, \
                           relu_mask, x, mean, invstd, C, rpg, chunks, 4 * C, rpc, partial)
        if (cq == 128) DCN_BN_RED(128);
        else if (cq == 64) DCN_BN_RED(64);
        else if (cq == 32) DCN_BN_RED(32);
        else DCN_BN_RED(16);
#undef DCN_BN_RED
    }
    {
        ObservedLaunch obs(DCN_PROF_BN_FINALIZE, 16.0 * (double)chunks * groups * C, st);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const float*)partial, chunks,
                           groups, C, (double)rpg, gamma, invstd, 4 * C, dgamma, dbeta, k123, absmax);
    }
    const int64_t total4 = rows * (C / 4);
    if ((dq || (hl_dx && (C % 32) == 0)) && absmax) {
        const bool hl = hl_dx && (C % 32) == 0;
        ObservedLaunch obs(DCN_PROF_BN_BWD_APPLY, (double)rows * C * (in_bytes + ((hl && !keep_dx) ? 0.0 : 4.0) + (g_out ? 4.0 : 0.0) +
                                                                      (dq ? 4.0 : 0.0) + (hl ? 4.0 : 0.0)), st);
        hipLaunchKernelGGL(bn_bwd_apply_blocked_kernel, dim3(blocks_for(((rows + 3) / 4) * (C / 4), kGridCap)), dim3(256), 0, st,
                           dy, dy2, relu_out, relu_mask, x, mean, invstd, (const float*)k123, (const float*)(k123 + C),
                           (const float*)(k123 + 2 * C), (hl_dx && (C % 32) == 0 && !keep_dx) ? nullptr : dx, g_out,
                           (dcnsplit::u32x4*)dq, (const float*)absmax, C / 4, rows, rpg, 4 * C, 3 * C,
                           (C % 32) == 0 ? (dcnsplit::u32x2*)hl_dx : nullptr, (tuning().bn_reverse >> 1) & 1, (tuning().bn_nt >> 1) & 1);
        return;
    }
    ObservedLaunch obs(DCN_PROF_BN_BWD_APPLY, (double)rows * C * (in_bytes + 4.0 + (g_out ? 4.0 : 0.0)), st);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(total4, kGridCap)), dim3(256), 0, st, dy, dy2, relu_out, relu_mask,
                       x, mean, invstd, (const float*)k123, (const float*)(k123 + C), (const float*)(k123 + 2 * C), dx, g_out,
                       C / 4, total4, total4 / groups, 4 * C, 3 * C, (tuning().bn_reverse >> 1) & 1);
}
void launch_add(const float* a, const float* b, float* out, int64_t n, hipStream_t st) {
    ObservedLaunch obs(DCN_PROF_OTHER, 12.0 * (double)n, st);
    hipLaunchKernelGGL(add_kernel, dim3(blocks_for(n / 4, kGridCap)), dim3(256), 0, st, a, b, out, n / 4);
}
void launch_maxpool_fwd(const float* in, float* out, unsigned char* argmax, int n, int hin, int win, int hout,
                        int wout, int C, hipStream_t st, const float* bn_stats, int groups, unsigned char* relu_mask) {
    const int64_t total4 = (int64_t)n * hout * wout * (C / 4);
    // input read once (the 3x3 / 2 windows' overlap is served by the caches), pooled output + argmax bytes (+ the sign mask) out
    ObservedLaunch obs(DCN_PROF_RESAMPLE, (double)n * C * ((double)hin * win * (4.0 + (bn_stats && relu_mask ? 0.25 : 0.0)) +
                                                          (double)hout * wout * (4.0 + (argmax ? 1.0 : 0.0))), st);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks_for(total4)), dim3(256), 0, st, in, out, argmax, hin, win, hout,
                       wout, C / 4, total4, bn_stats, bn_stats ? bn_stats + C : nullptr, 4 * C,
                       (int64_t)(groups > 1 ? n / groups : n), bn_stats ? relu_mask : nullptr);
}
void launch_maxpool_bwd(const float* gout, const unsigned char* argmax, float* gin, int n, int hin, int win, int hout,
                        int wout, int C, hipStream_t st, const float* gout2) {
    const int64_t total4 = (int64_t)n * hin * win * (C / 4);
    ObservedLaunch obs(DCN_PROF_RESAMPLE, (double)n * C * ((double)hin * win * 4.0 + (double)hout * wout * (5.0 + (gout2 ? 4.0 : 0.0))), st);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks_for(total4)), dim3(256), 0, st, gout, gout2, argmax, gin, hin, win,
                       hout, wout, C / 4, total4);
}
static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }
void launch_upsample_fwd(const float* low, int n, int hl, int wl, int ldl, int d, int h, int w, int normalize,
                         float* out, hipStream_t st) {
    const int64_t total = (int64_t)n * h * w;
    ObservedLaunch obs(DCN_PROF_RESAMPLE, 4.0 * ((double)n * hl * wl * ldl + (double)total * d), st);
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, st, low, hl, wl, ldl, d, h, w,
                       ac_scale(hl, h), ac_scale(wl, w), normalize, out, total);
}
size_t upsample_bwd_tmp_bytes(int n, int hl, int w, int d) { return (size_t)n * hl * w * d * sizeof(float); }
void launch_normalize_bwd(const float* low, int n, int hl, int wl, int ldl, int d, int h, int w, const float* gout,
                          float* gv, hipStream_t st) {
    const int64_t total = (int64_t)n * h * w;
    ObservedLaunch obs(DCN_PROF_RESAMPLE, 4.0 * ((double)n * hl * wl * ldl + 2.0 * (double)total * d), st);
    hipLaunchKernelGGL(normalize_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, st, low, hl, wl, ldl, d, h, w,
                       ac_scale(hl, h), ac_scale(wl, w), gout, gv, total);
}
void launch_upsample_bwd(const float* gout, int n, int hl, int wl, int ldl, int d, int h, int w, float* tmp,
                         float* glow, float* absmax, hipStream_t st, const float* gout_b) {
    const int64_t t1 = (int64_t)n * hl * w * d;
    if (gout_b) {   // two halves, two launches of the row pass (the column pass reads the common intermediate)
        const int64_t th = t1 / 2;
        for (int k = 0; k < 2; ++k) {
            ObservedLaunch obs(DCN_PROF_RESAMPLE, 4.0 * ((double)(n / 2) * h * w * d + (double)th), st);
            hipLaunchKernelGGL(upsample_bwd_rows_kernel, dim3(blocks_for(th)), dim3(256), 0, st, k ? gout_b : gout, hl, d, h, w,
                               ac_scale(hl, h), tmp + k * th, th);
        }
    } else {
        ObservedLaunch obs(DCN_PROF_RESAMPLE, 4.0 * ((double)n * h * w * d + (double)t1), st);   // the dense gradient map in
        hipLaunchKernelGGL(upsample_bwd_rows_kernel, dim3(blocks_for(t1)), dim3(256), 0, st, gout, hl, d, h, w,
                           ac_scale(hl, h), tmp, t1);
    }
    const int64_t t2 = (int64_t)n * hl * wl * ldl;
    {
        ObservedLaunch obs(DCN_PROF_RESAMPLE, 4.0 * ((double)t1 + (double)t2), st);
        hipLaunchKernelGGL(upsample_bwd_cols_kernel, dim3(blocks_for(t2)), dim3(256), 0, st, (const float*)tmp, hl, wl, ldl,
                           d, w, ac_scale(wl, w), glow, t2);
    }
    if (absmax) {
        ObservedLaunch obs(DCN_PROF_OTHER, 4.0 * (double)t2, st);
        hipLaunchKernelGGL(absmax_kernel, dim3(blocks_for(t2, 256)), dim3(256), 0, st, (const float*)glow, t2, absmax);
    }
}

}  // namespace dcn

extern "C" int dcn_upsample_forward(const float* low, int n, int hl, int wl, int ldl, int d, int h, int w, int normalize,
                                    float* out, void* stream) {
    if (!low || !out || n < 1 || hl < 1 || wl < 1 || d < 1 || ldl < d || h < 1 || w < 1) return DCN_E_INVALID;
    dcn::launch_upsample_fwd(low, n, hl, wl, ldl, d, h, w, normalize, out, (hipStream_t)stream);
    return dcn::check_launch();
}

extern "C" size_t dcn_upsample_backward_tmp_bytes(int n, int hl, int w, int d) {
    return dcn::upsample_bwd_tmp_bytes(n, hl, w, d);
}

extern "C" int dcn_upsample_backward(const float* gout, int n, int hl, int wl, int ldl, int d, int h, int w, float* glow,
                                     float* tmp, void* stream) {
    if (!gout || !glow || !tmp || n < 1 || hl < 1 || wl < 1 || d < 1 || ldl < d || h < 1 || w < 1) return DCN_E_INVALID;
    dcn::launch_upsample_bwd(gout, n, hl, wl, ldl, d, h, w, tmp, glow, nullptr, (hipStream_t)stream);
    return dcn::check_launch();
}

// ---- batch norm / max pool as stand-alone calls (unit tests; the engine uses the launchers above directly)
extern "C" int dcn_bn_forward(const float* x, const float* bn_partial, int mtiles, int c, int64_t rows, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                              int training, const float* res, int relu, float* y, unsigned char* relu_mask, float* stats,
                              void* stream) {
    if (!x || !gamma || !beta || !y || !stats || c < 4 || (c % 4) != 0 || rows < 1) return DCN_E_INVALID;
    if (training ? (!bn_partial || mtiles < 1) : (!running_mean || !running_var)) return DCN_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    dcn::launch_bn_finalize(bn_partial, mtiles, 1, c, (double)rows, gamma, beta, running_mean, running_var, momentum, eps,
                            training, stats, nullptr, nullptr, st);
    dcn::launch_bn_apply(x, stats, res, nullptr, relu, y, relu_mask, c, rows, 1, st);
    return dcn::check_launch();
}

extern "C" size_t dcn_bn_backward_workspace(int64_t rows, int c) {
    if (rows < 1 || c < 4) return 0;
    return ((size_t)dcn::bn_bwd_chunks(rows) * 4 * (size_t)c + 3 * (size_t)c) * sizeof(float);
}

extern "C" int dcn_bn_backward(const float* dy, const unsigned char* relu_mask, const float* x, const float* stats,
                               const float* gamma, int c, int64_t rows, float* dgamma, float* dbeta, float* dx, float* g_out,
                               void* workspace, void* stream) {
    if (!dy || !x || !stats || !gamma || !dgamma || !dbeta || !dx || !workspace || c < 4 || (c % 4) != 0 || rows < 1)
        return DCN_E_INVALID;
    float* partial = (float*)workspace;
    float* k123 = partial + (size_t)dcn::bn_bwd_chunks(rows) * 4 * c;
    dcn::launch_bn_bwd(dy, nullptr, relu_mask, x, stats, gamma, c, rows, 1, partial, dgamma, dbeta, k123, dx, g_out, nullptr,
                       nullptr, (hipStream_t)stream);
    return dcn::check_launch();
}

// the same with the reduction already done by the epilogue of the convolution that produced dy (dcn_conv_dgrad_bn_f16:
// dy is ReLU-masked, bn_partial holds `mtiles` rows of per-tile sums); workspace: 3 * c floats
extern "C" int dcn_bn_backward_from_partial(const float* dy, const float* bn_partial, int mtiles, const float* x,
                                            const float* stats, const float* gamma, int c, int64_t rows, float* dgamma,
                                            float* dbeta, float* dx, void* workspace, void* stream) {
    if (!dy || !bn_partial || mtiles < 1 || !x || !stats || !gamma || !dgamma || !dbeta || !dx || !workspace || c < 4 ||
        (c % 4) != 0 || rows < 1)
        return DCN_E_INVALID;
    dcn::launch_bn_bwd(dy, nullptr, nullptr, x, stats, gamma, c, rows, 1, const_cast<float*>(bn_partial), dgamma, dbeta,
                       (float*)workspace, dx, nullptr, nullptr, nullptr, (hipStream_t)stream, mtiles);
    return dcn::check_launch();
}

extern "C" int dcn_maxpool_forward(const float* in, int n, int hin, int win, int c, float* out, unsigned char* argmax,
                                   void* stream) {
    if (!in || !out || n < 1 || hin < 1 || win < 1 || c < 4 || (c % 4) != 0) return DCN_E_INVALID;
    dcn::launch_maxpool_fwd(in, out, argmax, n, hin, win, (hin + 2 - 3) / 2 + 1, (win + 2 - 3) / 2 + 1, c, (hipStream_t)stream);
    return dcn::check_launch();
}

extern "C" int dcn_maxpool_backward(const float* gout, const unsigned char* argmax, int n, int hin, int win, int c,
                                    float* gin, void* stream) {
    if (!gout || !argmax || !gin || n < 1 || hin < 1 || win < 1 || c < 4 || (c % 4) != 0) return DCN_E_INVALID;
    dcn::launch_maxpool_bwd(gout, argmax, gin, n, hin, win, (hin + 2 - 3) / 2 + 1, (win + 2 - 3) / 2 + 1, c, (hipStream_t)stream);
    return dcn::check_launch();
}
