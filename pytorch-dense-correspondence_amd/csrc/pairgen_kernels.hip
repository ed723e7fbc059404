// Pair generation on the device: the reference samples its pixel correspondences on the
// CPU with a 5-worker loader (dense_correspondence/correspondence_tools/correspondence_finder.py), which cannot feed a
// step rate of hundreds of images per second.  Same arithmetic, one work-item per candidate pixel:
//   batch_find_pixel_correspondences (:486-619)  depth lookup in image a -> unproject (K^-1) -> world (pose a) -> camera b
//       (pose b ^-1) -> project (K) -> prune: no depth return / outside the field of view (an exactly-zero coordinate is
//       dropped as well, as `torch.nonzero` does in the reference) / occluded or no return in image b (3 mm margin)
//   create_non_correspondences (:276-405)        uniform or mask-restricted pixel samples from given uniform numbers (the
//       reference's "perturb samples too close to a match" step is inert: its indicator is built from zeros_like, :343)
// plus an ORDER-PRESERVING compaction (the reference's torch.nonzero / index_select chains keep candidate order).
// All float math is fp32 in the reference's evaluation order (mm rows as a0*x0 + a1*x1 + a2*x2 [+ a3]); HBM/latency
// bound and tiny next to the network: 10 000 candidates + 1.5 M non-match samples per image pair.
#include "dcn_common.h"

namespace {

struct Cameras {
    float K[9], Kinv[9], Ta[16], Tbinv[16];
};

__device__ __forceinline__ float row3(const float* m, float x, float y, float z) { return m[0] * x + m[1] * y + m[2] * z; }
__device__ __forceinline__ float row4(const float* m, float x, float y, float z) {
    return m[0] * x + m[1] * y + m[2] * z + m[3] * 1.f;
}

__global__ void __launch_bounds__(256)
project_kernel(const unsigned short* __restrict__ depth_a, const unsigned short* __restrict__ depth_b, int h, int w,
               Cameras cam, const int64_t* __restrict__ cand_u, const int64_t* __restrict__ cand_v, int64_t n,
               unsigned char* __restrict__ flag, float* __restrict__ u2o, float* __restrict__ v2o) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t u = cand_u[i], v = cand_v[i];
    unsigned char ok = 0;
    float u2 = 0.f, v2 = 0.f;
    if (u >= 0 && u < w && v >= 0 && v < h) {
        const float d = (float)depth_a[v * w + u] * 1.0f / 1000.0f;
        if (d != 0.f) {
            const float fx = (float)u * d, fy = (float)v * d, fz = d;
            const float cx = row3(cam.Kinv, fx, fy, fz), cy = row3(cam.Kinv + 3, fx, fy, fz), cz = row3(cam.Kinv + 6, fx, fy, fz);
            const float wx = row4(cam.Ta, cx, cy, cz), wy = row4(cam.Ta + 4, cx, cy, cz), wz = row4(cam.Ta + 8, cx, cy, cz);
            const float bx = row4(cam.Tbinv, wx, wy, wz), by = row4(cam.Tbinv + 4, wx, wy, wz), bz = row4(cam.Tbinv + 8, wx, wy, wz);
            const float px = row3(cam.K, bx, by, bz), py = row3(cam.K + 3, bx, by, bz), pz = row3(cam.K + 6, bx, by, bz);
            u2 = px / pz;
            v2 = py / pz;
            const float ub = (float)w * 1.0f - 1e-3f, vb = (float)h * 1.0f - 1e-3f;
            // (u2 != 0) & in range, written so that NaN coordinates are rejected
            if (u2 > 0.f && u2 <= ub && v2 > 0.f && v2 <= vb) {
                const int64_t fb = (int64_t)v2 * w + (int64_t)u2;            // truncation, as `.type(LongTensor)`
                const float d2 = (float)depth_b[fb] * 1.0f / 1000.f;
                const float z2 = pz - 0.003f;
                ok = (d2 > 0.f && !(d2 < z2)) ? 1 : 0;
            }
        }
    }
    flag[i] = ok;
    u2o[i] = u2;
    v2o[i] = v2;
}

__global__ void __launch_bounds__(256)
flags_from_float_kernel(const float* __restrict__ v, int64_t n, unsigned char* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = v[i] != 0.f ? 1 : 0;
}

// sel[0 .. count) = indices of the set flags, in increasing order.  ONE workgroup walks the array in 1024-element chunks
// with an LDS scan per chunk (the inputs are 10^4 candidates or the 3*10^5 pixels of a mask: microseconds).
constexpr int kScan = 1024;
__global__ void __launch_bounds__(kScan)
ordered_select_kernel(const unsigned char* __restrict__ flag, int64_t n, int64_t* __restrict__ sel,
                      int64_t* __restrict__ count) {
    __shared__ int s[2][kScan];
    __shared__ int64_t base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < n; c0 += kScan) {
        const int64_t i = c0 + threadIdx.x;
        const int f = i < n ? (int)flag[i] : 0;
        int cur = 0;
        s[0][threadIdx.x] = f;
        __syncthreads();
        for (int off = 1; off < kScan; off <<= 1) {            // inclusive Hillis-Steele scan
            const int v = s[cur][threadIdx.x] + ((int)threadIdx.x >= off ? s[cur][threadIdx.x - off] : 0);
            s[cur ^ 1][threadIdx.x] = v;
            cur ^= 1;
            __syncthreads();
        }
        const int incl = s[cur][threadIdx.x];
        if (f) sel[base + incl - 1] = i;
        __syncthreads();
        if (threadIdx.x == kScan - 1) base += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}

__global__ void __launch_bounds__(256)
gather_matches_kernel(const int64_t* __restrict__ sel, const int64_t* __restrict__ count, const int64_t* __restrict__ cand_u,
                      const int64_t* __restrict__ cand_v, const float* __restrict__ u2, const float* __restrict__ v2,
                      int64_t n, int64_t* __restrict__ ua, int64_t* __restrict__ va, float* __restrict__ ub,
                      float* __restrict__ vb) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n || j >= *count) return;
    const int64_t i = sel[j];
    ua[j] = cand_u[i];
    va[j] = cand_v[i];
    ub[j] = u2[i];
    vb[j] = v2[i];
}

// uniform: (floor(r0 * w), floor(r1 * h)) -- pytorch_rand_select_pixel (:29-34); list != NULL: pixel list[floor(r * count)]
__global__ void __launch_bounds__(256)
sample_pixels_kernel(const float* __restrict__ rand, int64_t n, int w, int h, const int64_t* __restrict__ list,
                     const int64_t* __restrict__ count, float* __restrict__ u, float* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (list) {
        const int64_t c = *count;
        int64_t j = (int64_t)floorf(rand[i] * (float)c);
        j = j < 0 ? 0 : (j >= c ? c - 1 : j);      // (rand < 1, so only a rounding of r * c up to c can reach the clamp)
        const int64_t p = c > 0 ? list[j] : 0;
        u[i] = (float)(p % w);
        v[i] = (float)(p / w);
    } else {
        u[i] = floorf(rand[i] * (float)w);
        v[i] = floorf(rand[n + i] * (float)h);
    }
}

inline unsigned blocks256(int64_t n) { return (unsigned)dcn::ceil_div64(n, 256); }

}  // namespace

extern "C" size_t dcn_find_correspondences_workspace(int64_t n) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    return ((nn + 63) / 64 * 64) + 2 * nn * sizeof(float) + nn * sizeof(int64_t) + 64;
}

extern "C" int dcn_find_correspondences(const uint16_t* depth_a, const uint16_t* depth_b, int h, int w, const float* K,
                                        const float* K_inv, const float* pose_a, const float* pose_b_inv,
                                        const int64_t* cand_u, const int64_t* cand_v, int64_t n, int64_t* out_ua,
                                        int64_t* out_va, float* out_ub, float* out_vb, int64_t* out_count,
                                        void* workspace, void* stream) {
    if (!depth_a || !depth_b || !K || !K_inv || !pose_a || !pose_b_inv || !cand_u || !cand_v || !out_ua || !out_va ||
        !out_ub || !out_vb || !out_count || !workspace || h < 1 || w < 1 || n < 1)
        return DCN_E_INVALID;
    Cameras cam;
    for (int i = 0; i < 9; ++i) { cam.K[i] = K[i]; cam.Kinv[i] = K_inv[i]; }
    for (int i = 0; i < 16; ++i) { cam.Ta[i] = pose_a[i]; cam.Tbinv[i] = pose_b_inv[i]; }
    hipStream_t st = (hipStream_t)stream;
    unsigned char* flag = (unsigned char*)workspace;
    float* u2 = (float*)(flag + ((size_t)n + 63) / 64 * 64);
    float* v2 = u2 + n;
    int64_t* sel = (int64_t*)(((uintptr_t)(v2 + n) + 7) & ~(uintptr_t)7);
    hipLaunchKernelGGL(project_kernel, dim3(blocks256(n)), dim3(256), 0, st, depth_a, depth_b, h, w, cam, cand_u, cand_v, n,
                       flag, u2, v2);
    hipLaunchKernelGGL(ordered_select_kernel, dim3(1), dim3(kScan), 0, st, (const unsigned char*)flag, n, sel, out_count);
    hipLaunchKernelGGL(gather_matches_kernel, dim3(blocks256(n)), dim3(256), 0, st, (const int64_t*)sel,
                       (const int64_t*)out_count, cand_u, cand_v, (const float*)u2, (const float*)v2, n, out_ua, out_va,
                       out_ub, out_vb);
    return dcn::check_launch();
}

extern "C" size_t dcn_mask_nonzero_workspace(int64_t hw) { return (size_t)(hw > 0 ? hw : 1) + 64; }

extern "C" int dcn_mask_nonzero(const float* mask, int64_t hw, int64_t* list, int64_t* count, void* workspace, void* stream) {
    if (!mask || !list || !count || !workspace || hw < 1) return DCN_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* flag = (unsigned char*)workspace;
    hipLaunchKernelGGL(flags_from_float_kernel, dim3(blocks256(hw)), dim3(256), 0, st, mask, hw, flag);
    hipLaunchKernelGGL(ordered_select_kernel, dim3(1), dim3(kScan), 0, st, (const unsigned char*)flag, hw, list, count);
    return dcn::check_launch();
}

extern "C" int dcn_sample_pixels(const float* rand, int64_t n, int w, int h, const int64_t* list, const int64_t* count,
                                 float* u, float* v, void* stream) {
    if (!rand || !u || !v || n < 1 || w < 1 || h < 1 || ((list == nullptr) != (count == nullptr))) return DCN_E_INVALID;
    hipLaunchKernelGGL(sample_pixels_kernel, dim3(blocks256(n)), dim3(256), 0, (hipStream_t)stream, rand, n, w, h, list, count,
                       u, v);
    return dcn::check_launch();
}
