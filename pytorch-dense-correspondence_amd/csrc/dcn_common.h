// Shared device helpers for the gfx950 kernels (wave64 reductions, launch checking).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dcn_hip.h"

// Counted waits of the LDS-DMA pipelines (conv_hl_kernels.hip): inline asm, so that the counts stay exactly where they are
// written (the compiler's own s_waitcnt insertion does not see them).  The test-only host emulation (tests/hostemu) defines
// its own versions before this header is read.
#ifndef DCN_WAIT_VMCNT
#define DCN_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define DCN_WAIT_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define DCN_OPAQUE_INT(v) asm volatile("" : "+v"(v))   // the optimiser forgets what it knew about v (no loop-invariant hoisting)
#endif

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

// Stream-ordered fill of n bytes (n % 4 == 0) with a byte value, as a KERNEL: unlike hipMemsetAsync it is an ordinary
// kernel node when the caller's stream is being captured into a hipGraph.
static __global__ void __launch_bounds__(256) fill_words_kernel(unsigned* __restrict__ p, unsigned v, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = v;
}
inline int fill_bytes_async(void* p, int byte_value, size_t n, hipStream_t st) {
    if (n == 0) return DCN_OK;
    if (n % 4) return DCN_E_INVALID;
    const unsigned b = (unsigned)byte_value & 0xffu, v = b | (b << 8) | (b << 16) | (b << 24);
    const size_t words = n / 4;
    size_t blocks = (words + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned*)p, v, words);
    return hipGetLastError() == hipSuccess ? DCN_OK : DCN_E_LAUNCH;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? DCN_OK : DCN_E_LAUNCH; }

__host__ __device__ inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace dcn
