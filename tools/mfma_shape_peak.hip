// Micro-benchmark: sustained fp16 MFMA rate of the whole chip, no memory traffic, random operands -- v_mfma_f32_32x32x16_f16
// against v_mfma_f32_16x16x32_f16 (the small-tile kernel of conv_hlx_kernels.hip is built from the latter: 16-row granularity).
// Both move the same FLOPs per issue cycle on paper; the 16 x 16 shape reads twice the operand registers per FLOP.  The chip
// runs these loops against a power limit, so the question is what each shape SUSTAINS.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_peak.hip -o tools/mfma_shape_peak.bin && tools/mfma_shape_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void make_operands(h8& x, h8& y, h8& z, h8& w) {
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int e = 0; e < 8; ++e) {
        s = s * 1664525u + 1013904223u; x[e] = (_Float16)((float)(s >> 8) * (1.f / 16777216.f) - 0.5f);
        s = s * 1664525u + 1013904223u; y[e] = (_Float16)((float)(s >> 8) * (1.f / 16777216.f) - 0.5f);
        s = s * 1664525u + 1013904223u; z[e] = (_Float16)(((float)(s >> 8) * (1.f / 16777216.f) - 0.5f) * 1e-3f);
        s = s * 1664525u + 1013904223u; w[e] = (_Float16)(((float)(s >> 8) * (1.f / 16777216.f) - 0.5f) * 1e-3f);
    }
}
// 8 accumulators of 32 x 32 (128 registers): the wavefront tile of the big kernels
__global__ void __launch_bounds__(512) loop32(float* out, int iters) {
    f32x16 a[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) a[i][r] = 0.f;
    h8 x, y, z, w;
    make_operands(x, y, z, w);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16((i & 1) ? z : x, (i & 2) ? w : y, a[i], 0, 0, 0);
    }
    float t = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) t += a[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}
// 20 accumulators of 16 x 16 (80 registers): the wavefront tile of the small-tile kernel
__global__ void __launch_bounds__(512) loop16(float* out, int iters) {
    f32x4 a[20];
    for (int i = 0; i < 20; ++i) for (int r = 0; r < 4; ++r) a[i][r] = 0.f;
    h8 x, y, z, w;
    make_operands(x, y, z, w);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 20; ++i) a[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16((i & 1) ? z : x, (i & 2) ? w : y, a[i], 0, 0, 0);
    }
    float t = 0;
    for (int i = 0; i < 20; ++i) for (int r = 0; r < 4; ++r) t += a[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}
int main() {
    float* out;
    const int blocks = 256;   // one 8-wavefront workgroup per CU, like the convolution kernels
    hipMalloc(&out, blocks * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int shape : {32, 16}) {
            const int iters = 40000;
            const double per_iter = shape == 32 ? 8 * 2.0 * 32 * 32 * 16 : 20 * 2.0 * 16 * 16 * 32;
            if (shape == 32) hipLaunchKernelGGL(loop32, dim3(blocks), dim3(512), 0, 0, out, 200);
            else hipLaunchKernelGGL(loop16, dim3(blocks), dim3(512), 0, 0, out, 200);
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(loop32, dim3(blocks), dim3(512), 0, 0, out, iters);
            else hipLaunchKernelGGL(loop16, dim3(blocks), dim3(512), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 8 * iters * per_iter;
            printf("v_mfma_f32_%s_f16, random operands, 8 wavefronts per CU: %8.3f ms  %7.1f TFLOP/s (%.1f %% of 2516.6)\n",
                   shape == 32 ? "32x32x16" : "16x16x32", ms, flops / ms / 1e9, flops / ms / 1e9 / 25.166);
        }
    return 0;
}
