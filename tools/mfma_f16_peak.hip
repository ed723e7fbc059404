// Micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate of the whole chip with no memory traffic, as a function of the
// OPERAND DATA (all-zero registers vs random fp16 values).  MI355X runs this instruction against a power limit, not a
// fixed clock: the rate with realistic operand bit patterns is the practical ceiling of the split-fp16 convolutions.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_peak.hip -o /tmp/mfma_f16_peak && /tmp/mfma_f16_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, int mode, int waves_active) {
    if ((int)(threadIdx.x >> 6) >= waves_active) return;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    h8 x, y, z, w;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int e = 0; e < 8; ++e) {
        s = s * 1664525u + 1013904223u; const float r0 = (float)(s >> 8) * (1.f / 16777216.f) - 0.5f;
        s = s * 1664525u + 1013904223u; const float r1 = (float)(s >> 8) * (1.f / 16777216.f) - 0.5f;
        s = s * 1664525u + 1013904223u; const float r2 = (float)(s >> 8) * (1.f / 16777216.f) - 0.5f;
        s = s * 1664525u + 1013904223u; const float r3 = (float)(s >> 8) * (1.f / 16777216.f) - 0.5f;
        x[e] = mode ? (_Float16)r0 : (_Float16)0.f; y[e] = mode ? (_Float16)r1 : (_Float16)0.f;
        z[e] = mode ? (_Float16)(r2 * 1e-3f) : (_Float16)0.f; w[e] = mode ? (_Float16)(r3 * 1e-3f) : (_Float16)0.f;
    }
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(z, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a3, 0, 0, 0);
    }
    float t = 0;
    for (int r = 0; r < 16; ++r) t += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
int main() {
    float* out;
    const int blocks = 256 * 2;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves : {4, 2})
        for (int mode : {0, 1})
            for (int iters : {20000, 200000}) {
                hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, 100, mode, waves);
                hipEventRecord(e0);
                hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters, mode, waves);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double flops = (double)blocks * waves * iters * 4 * 2.0 * 32 * 32 * 16;
                printf("waves/WG %d (x2 WG per CU) data %-6s iters %6d: %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 2516.6; implied clock %.2f GHz at full issue)\n",
                       waves, mode ? "random" : "zero", iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.166,
                       flops / ms / 1e9 / 2516.6 * 2.4 * (4.0 / waves > 2 ? 2 : 1) * (waves == 2 ? 1 : 1));
            }
    return 0;
}
