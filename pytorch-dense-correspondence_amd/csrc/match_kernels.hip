// Best-match search over a dense descriptor image ("next" row 1) -- the evaluation-side hot loop
//   norm_diffs = sqrt(sum(square(res_b - descriptor), axis=2));  argmin      (dense_correspondence_network.py:517-523, :541-547)
// which the reference runs in numpy on the CPU once per query pixel (100 full-image scans per image pair,
// evaluation.py:932-950).  Here ALL queries of an image are answered by one pass over the descriptor image:
// HBM-bound, algorithmic traffic = HW * D * 4 bytes once (+ Q * HW * 4 when the distance images are requested).
// One work-item per pixel keeps its descriptor in registers and loops over the queries (LDS broadcast reads);
// per query: wave64 shuffle min-reduction of the packed key (dist2 bits << 32 | pixel index), across the workgroup's
// wavefronts through LDS, then at most one 64-bit atomicMin per WORKGROUP -- and only when the key beats the value
// currently in memory (a plain load first: thousands of workgroups target the same Q words, and after the first few
// almost every key loses; without the look the kernel is serialised on those atomics: 12 us -> 1 us per query).
// The key order makes ties resolve to the smallest index, exactly like np.argmin.
#include "dcn_common.h"

namespace {

constexpr int kMT = 256;    // work-items (pixels) per workgroup
constexpr int kQT = 32;     // queries staged in LDS at a time
constexpr int kMaxD = 64;

template <int DT>
__global__ void __launch_bounds__(kMT)
best_match_kernel(const float* __restrict__ res, int64_t hw, int d_rt, const float* __restrict__ queries, int nq,
                  const unsigned char* __restrict__ mask, unsigned long long* __restrict__ best,
                  float* __restrict__ norm_diffs) {
    __shared__ float sq[kQT * kMaxD];
    __shared__ unsigned long long skey[kQT][kMT / 64];
    const int D = DT > 0 ? DT : d_rt;
    const int64_t pix = (int64_t)blockIdx.x * kMT + threadIdx.x;
    const bool in = pix < hw;
    const bool cand = in && (!mask || mask[pix] != 0);
    float v[DT > 0 ? DT : kMaxD];
#pragma unroll
    for (int k = 0; k < (DT > 0 ? DT : kMaxD); ++k) v[k] = (in && k < D) ? res[pix * D + k] : 0.f;
    for (int q0 = 0; q0 < nq; q0 += kQT) {
        const int qn = min(kQT, nq - q0);
        __syncthreads();
        for (int i = threadIdx.x; i < qn * D; i += kMT) sq[i] = queries[(int64_t)q0 * D + i];
        __syncthreads();
        for (int q = 0; q < qn; ++q) {
            float d2 = 0.f;
#pragma unroll
            for (int k = 0; k < (DT > 0 ? DT : kMaxD); ++k) {
                if (k < D) { const float t = v[k] - sq[q * D + k]; d2 = fmaf(t, t, d2); }
            }
            if (norm_diffs && in) norm_diffs[(int64_t)(q0 + q) * hw + pix] = sqrtf(d2);
            unsigned long long key = cand ? (((unsigned long long)__float_as_uint(d2)) << 32) | (unsigned long long)(unsigned)pix
                                          : ~0ull;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_down(key, off, 64);
                key = o < key ? o : key;
            }
            if ((threadIdx.x & 63) == 0) skey[q][threadIdx.x >> 6] = key;
        }
        __syncthreads();
        if ((int)threadIdx.x < qn) {
            unsigned long long key = skey[threadIdx.x][0];
#pragma unroll
            for (int w = 1; w < kMT / 64; ++w) key = skey[threadIdx.x][w] < key ? skey[threadIdx.x][w] : key;
            unsigned long long* slot = best + q0 + threadIdx.x;
            if (key != ~0ull && key < __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMin(slot, key);
        }
    }
}

__global__ void __launch_bounds__(256)
best_match_unpack_kernel(const unsigned long long* __restrict__ best, int nq, int64_t* __restrict__ idx,
                         float* __restrict__ dist) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const unsigned long long k = best[q];
    if (k == ~0ull) { idx[q] = -1; dist[q] = INFINITY; return; }  // empty mask
    idx[q] = (int64_t)(k & 0xffffffffull);
    dist[q] = sqrtf(__uint_as_float((unsigned)(k >> 32)));
}

// ------------------------------------------------------------------------------------------------ match statistics
// evaluation.py:1046-1100 for all query matches of an image pair in one pass over res_b.  Per query q (descriptor
// queries[q] = res_a[uv_a], ground-truth pixel gt[q] in image b):
//   best match over the image and over the mask (argmin of d, resp. of d + (1 - mask) * 1e6, first index on ties),
//   t = ||queries[q] - res_b[gt[q]]||, the number of pixels with d < t (image / mask) and the sum of their pixel distances to
//   gt[q] (for "average_l2_distance_for_false_positives").  d = sqrt(sum_k (res_b - query)^2) in fp32.
struct MatchStats {
    unsigned long long* best;     // [2][Q] packed (d2 bits << 32 | pixel): image, masked (masked: d2 -> bits of (d + 1e6) outside)
    int* count;                   // [2][Q]
    float* dist_sum;              // [2][Q]
    float* gt_d;                  // [Q]
};

template <int DT>
__global__ void __launch_bounds__(kMT)
match_stats_kernel(const float* __restrict__ res, int64_t hw, int w, int d_rt, const float* __restrict__ queries,
                   const int64_t* __restrict__ gt, int nq, const unsigned char* __restrict__ mask, MatchStats o) {
    __shared__ float sq[kQT * kMaxD];
    __shared__ float st[kQT];          // squared ground-truth distance
    __shared__ int sgu[kQT], sgv[kQT];
    __shared__ unsigned long long skey[2][kQT][kMT / 64];
    __shared__ int scnt[2][kQT][kMT / 64];
    __shared__ float ssum[2][kQT][kMT / 64];
    const int D = DT > 0 ? DT : d_rt;
    const int64_t pix = (int64_t)blockIdx.x * kMT + threadIdx.x;
    const bool in = pix < hw;
    const bool onm = in && mask && mask[pix] != 0;
    const int pu = in ? (int)(pix % w) : 0, pv = in ? (int)(pix / w) : 0;
    float v[DT > 0 ? DT : kMaxD];
#pragma unroll
    for (int k = 0; k < (DT > 0 ? DT : kMaxD); ++k) v[k] = (in && k < D) ? res[pix * D + k] : 0.f;
    const int wv = threadIdx.x >> 6;
    for (int q0 = 0; q0 < nq; q0 += kQT) {
        const int qn = min(kQT, nq - q0);
        __syncthreads();
        for (int i = threadIdx.x; i < qn * D; i += kMT) sq[i] = queries[(int64_t)q0 * D + i];
        if ((int)threadIdx.x < qn) {
            const int64_t g = gt[q0 + threadIdx.x];
            float t2 = 0.f;
            for (int k = 0; k < D; ++k) { const float t = res[g * D + k] - queries[(int64_t)(q0 + threadIdx.x) * D + k]; t2 = fmaf(t, t, t2); }
            st[threadIdx.x] = t2;
            sgu[threadIdx.x] = (int)(g % w);
            sgv[threadIdx.x] = (int)(g / w);
            if (blockIdx.x == 0) o.gt_d[q0 + threadIdx.x] = sqrtf(t2);
        }
        __syncthreads();
        for (int q = 0; q < qn; ++q) {
            float d2 = 0.f;
#pragma unroll
            for (int k = 0; k < (DT > 0 ? DT : kMaxD); ++k) {
                if (k < D) { const float t = v[k] - sq[q * D + k]; d2 = fmaf(t, t, d2); }
            }
            const float dd = sqrtf(d2), tt = sqrtf(st[q]);
            const float dm = mask ? (onm ? dd : dd + 1e6f) : dd;           // masked_norm_diffs (no mask given: the image itself)
            unsigned long long k0 = in ? (((unsigned long long)__float_as_uint(dd)) << 32) | (unsigned)pix : ~0ull;
            unsigned long long k1 = in ? (((unsigned long long)__float_as_uint(dm)) << 32) | (unsigned)pix : ~0ull;
            const bool c0 = in && dd < tt, c1 = in && dm < tt;
            const float du = (float)(pu - sgu[q]), dv = (float)(pv - sgv[q]);
            const float pd = sqrtf(du * du + dv * dv);
            int n0 = c0 ? 1 : 0, n1 = c1 ? 1 : 0;
            float s0 = c0 ? pd : 0.f, s1 = c1 ? pd : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long a = __shfl_down(k0, off, 64), b = __shfl_down(k1, off, 64);
                k0 = a < k0 ? a : k0;
                k1 = b < k1 ? b : k1;
                n0 += __shfl_down(n0, off, 64);
                n1 += __shfl_down(n1, off, 64);
                s0 += __shfl_down(s0, off, 64);
                s1 += __shfl_down(s1, off, 64);
            }
            if ((threadIdx.x & 63) == 0) {
                skey[0][q][wv] = k0; skey[1][q][wv] = k1;
                scnt[0][q][wv] = n0; scnt[1][q][wv] = n1;
                ssum[0][q][wv] = s0; ssum[1][q][wv] = s1;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < 2 * qn) {
            const int which = threadIdx.x / qn, q = threadIdx.x - which * qn;
            unsigned long long key = skey[which][q][0];
            int n = scnt[which][q][0];
            float s = ssum[which][q][0];
#pragma unroll
            for (int x = 1; x < kMT / 64; ++x) {
                key = skey[which][q][x] < key ? skey[which][q][x] : key;
                n += scnt[which][q][x];
                s += ssum[which][q][x];
            }
            unsigned long long* slot = o.best + (int64_t)which * nq + q0 + q;
            if (key != ~0ull && key < __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMin(slot, key);
            if (n) {
                atomicAdd(o.count + (int64_t)which * nq + q0 + q, n);
                unsafeAtomicAdd(o.dist_sum + (int64_t)which * nq + q0 + q, s);
            }
        }
    }
}

__global__ void __launch_bounds__(256)
match_stats_unpack_kernel(const unsigned long long* __restrict__ best, int nq, int64_t* __restrict__ idx,
                          float* __restrict__ dist) {
    const int q = blockIdx.x * 256 + threadIdx.x;   // over 2 * nq
    if (q >= 2 * nq) return;
    const unsigned long long k = best[q];
    idx[q] = (int64_t)(k & 0xffffffffull);
    dist[q] = __uint_as_float((unsigned)(k >> 32));   // (already the norm, not its square)
}

}  // namespace

extern "C" size_t dcn_match_statistics_workspace(int q) { return (size_t)(q > 0 ? q : 1) * 2 * sizeof(unsigned long long); }

// best_idx / best_dist: [2][Q] (image, masked); count: [2][Q] int32; dist_sum: [2][Q]; gt_dist: [Q].  mask may be NULL
// (then the "masked" half equals the image half).
extern "C" int dcn_match_statistics(const float* res, int64_t hw, int w, int d, const float* queries, const int64_t* gt_idx,
                                    int q, const unsigned char* mask, int64_t* best_idx, float* best_dist, int32_t* count,
                                    float* dist_sum, float* gt_dist, void* workspace, void* stream) {
    if (!res || !queries || !gt_idx || !best_idx || !best_dist || !count || !dist_sum || !gt_dist || !workspace || hw < 1 ||
        hw >= ((int64_t)1 << 32) || w < 1 || d < 1 || d > kMaxD || q < 1)
        return DCN_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    MatchStats o;
    o.best = (unsigned long long*)workspace; o.count = count; o.dist_sum = dist_sum; o.gt_d = gt_dist;
    if (dcn::fill_bytes_async(o.best, 0xFF, (size_t)q * 2 * sizeof(unsigned long long), st) != DCN_OK) return DCN_E_LAUNCH;
    if (dcn::fill_bytes_async(count, 0, (size_t)q * 2 * sizeof(int32_t), st) != DCN_OK) return DCN_E_LAUNCH;
    if (dcn::fill_bytes_async(dist_sum, 0, (size_t)q * 2 * sizeof(float), st) != DCN_OK) return DCN_E_LAUNCH;
    const dim3 grid((unsigned)dcn::ceil_div64(hw, kMT)), block(kMT);
#define DCN_MS(DT) \
    hipLaunchKernelGGL((match_stats_kernel<DT>), grid, block, 0, st, res, hw, w, d, queries, gt_idx, q, mask, o)
    switch (d) {
        case 3: DCN_MS(3); break;
        case 4: DCN_MS(4); break;
        case 8: DCN_MS(8); break;
        case 16: DCN_MS(16); break;
        case 32: DCN_MS(32); break;
        default: DCN_MS(0); break;
    }
#undef DCN_MS
    hipLaunchKernelGGL(match_stats_unpack_kernel, dim3(dcn::ceil_div(2 * q, 256)), dim3(256), 0, st,
                       (const unsigned long long*)o.best, q, best_idx, best_dist);
    return dcn::check_launch();
}

extern "C" size_t dcn_find_best_match_workspace(int q) { return (size_t)(q > 0 ? q : 1) * sizeof(unsigned long long); }

extern "C" int dcn_find_best_match(const float* res, int64_t hw, int d, const float* queries, int q,
                                   const unsigned char* mask, int64_t* best_idx, float* best_dist, float* norm_diffs,
                                   void* workspace, void* stream) {
    if (!res || !queries || !best_idx || !best_dist || !workspace || hw < 1 || hw >= ((int64_t)1 << 32) || d < 1 ||
        d > kMaxD || q < 1)
        return DCN_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* best = (unsigned long long*)workspace;
    if (dcn::fill_bytes_async(best, 0xFF, (size_t)q * sizeof(unsigned long long), st) != DCN_OK) return DCN_E_LAUNCH;
    const dim3 grid((unsigned)dcn::ceil_div64(hw, kMT)), block(kMT);
#define DCN_BM(DT) \
    hipLaunchKernelGGL((best_match_kernel<DT>), grid, block, 0, st, res, hw, d, queries, q, mask, best, norm_diffs)
    switch (d) {
        case 3: DCN_BM(3); break;
        case 4: DCN_BM(4); break;
        case 8: DCN_BM(8); break;
        case 16: DCN_BM(16); break;
        case 32: DCN_BM(32); break;
        default: DCN_BM(0); break;
    }
#undef DCN_BM
    hipLaunchKernelGGL(best_match_unpack_kernel, dim3(dcn::ceil_div(q, 256)), dim3(256), 0, st,
                       (const unsigned long long*)best, q, best_idx, best_dist);
    return dcn::check_launch();
}
