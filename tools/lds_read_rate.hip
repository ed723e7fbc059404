// LDS read throughput per CU by instruction (gfx950): ds_read_b64_tr_b16 (the transposing read of the weight-gradient kernels),
// ds_read_b64, ds_read_b128 -- 512 work-items per workgroup, one workgroup per CU, conflict-free addresses, nothing else in the loop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_read_rate.bin tools/lds_read_rate.hip && tools/lds_read_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short s4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(512, 1) rate_kernel(int iters, int* out, long long* cycles) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += 512) reinterpret_cast<int*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    int acc = 0;
    // conflict-free per 32-lane half: consecutive lanes, consecutive 8 / 16-byte slots; 8 independent addresses per iteration
    const int unit = MODE == 2 ? 16 : 8;
    const unsigned char* base = lds + wv * 4096 + lane * unit;
    // MODE 3: the weight-gradient kernels' pattern -- lane (gq = lane >> 4, s16 = lane & 15) reads pixel row 8 gq + (s16 >> 2) (pitch
    // 256 B), 32-byte unit (0 ^ key(row)), 8-byte run s16 & 3
    const int gq = lane >> 4, s16 = lane & 15, rq = s16 >> 2;
    const unsigned char* kbase = lds + (wv & 1) * 8192 + (8 * gq + rq) * 256 + (((rq | ((gq & 1) << 2))) * 32) + (s16 & 3) * 8;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        int o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {   // (opaque offsets: every read is issued, none is merged with the previous iteration's)
            o[k] = MODE == 3 ? ((k & 3) * 32 + (k & 4) * 256) : (k & 3) * 1024;
            asm volatile("" : "+v"(o[k]));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 3 || MODE == 0) {
                s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)((MODE == 3 ? kbase : base) + o[k]));
                acc ^= v[0] ^ v[3];
            } else if (MODE == 1) {
                i2v v = *reinterpret_cast<const i2v*>(base + o[k]);
                acc ^= v[0] ^ v[1];
            } else {
                i4v v = *reinterpret_cast<const i4v*>(base + o[k]);
                acc ^= v[0] ^ v[3];
            }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 512 + tid] = acc;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int bytes_per_lane) {
    int* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(512), 0, 0, iters, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        const double instr = (double)iters * 8 * 8;   // wave-instructions per CU
        const double ns_per = ms * 1e6 / instr;
        printf("%-22s rep %d: %.3f ms, %.2f ns per wavefront instruction per CU = %.1f bytes/ns/CU (s_memtime ticks per instr: %.2f)\n", name, rep, ms, ns_per,
               64.0 * bytes_per_lane / ns_per, (double)h[0] / instr);
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("ds_read_b64_tr_b16", 8);
    run<1>("ds_read_b64", 8);
    run<2>("ds_read_b128", 16);
    run<3>("tr_b16, kernel pattern", 8);
    return 0;
}
