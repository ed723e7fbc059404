// Micro-benchmark: sustained fp32 MFMA (v_mfma_f32_32x32x2_f32) rate of the whole chip, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f + 1.f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out;
    const int blocks = 256 * 2;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int iters : {2000, 20000, 100000}) {
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 /*waves*/ * iters * 4 * 2.0 * 32 * 32 * 2;
        printf("iters %6d: %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3; implied clock %.2f GHz)\n", iters, ms, flops / ms / 1e9,
               flops / ms / 1e9 / 1.573, flops / ms / 1e9 / 157.3 * 2.4);
    }
    return 0;
}
