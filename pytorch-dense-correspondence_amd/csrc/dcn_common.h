// Shared device helpers for the gfx950 kernels (wave64 reductions, launch checking).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dcn_hip.h"

namespace dcn {

constexpr int kWave = 64;  // CDNA wavefront

template <class T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;  // valid in lane 0
}

// Sum over a workgroup of NT threads (NT multiple of 64).  Result valid in thread 0.  `scratch` holds NT/64 values.
template <int NT, class T> __device__ __forceinline__ T block_sum(T v, T* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    __syncthreads();  // scratch may still be read from a previous reduction
    if (lane == 0) scratch[wv] = v;
    __syncthreads();
    T r = scratch[0];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < NT / kWave; ++i) r += scratch[i];
    }
    return r;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? DCN_OK : DCN_E_LAUNCH; }

__host__ __device__ inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace dcn
