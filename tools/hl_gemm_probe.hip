// Stand-alone probe / prototype of the LDS-DMA operand path for the split-fp16 ("f16x3") GEMM core on gfx950:
//   * checks the semantics the convolution kernel relies on: `buffer_load_dwordx4 ... lds` (LDS-DMA) places lane l's 16
//     bytes at M0 base + 16 l, an out-of-range voffset writes ZEROS to LDS, soffset is added without a range check;
//   * times C[M][N] = A[M][K] B[N][K]^T with both operands pre-split as "hl32" tensors ([row][K/32][hi x32 | lo x32] fp16),
//     256 x 256 (and 256 x 128) tiles on 8 wavefronts, both operands filled by LDS-DMA into an XOR-swizzled lane-linear
//     image, 3 MFMAs (lo*hi + hi*lo + hi*hi) per product -- the ceiling of the gather-GEMM's inner loop without any
//     convolution indexing.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/hl_probe tools/hl_gemm_probe.hip && /tmp/hl_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kOob = (int)0x80000000;

// LDS-DMA: 64 lanes x 16 bytes, lane l's bytes land at lds_dst + 16 l; the source address is per lane (voffset) + soffset.
// (Kept in a NON-template function: inside a template the builtin silently breaks the host-side instantiation of the kernel
// stub with this compiler, and the kernel then fails to link.)
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, void* lds_dst, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_dst, 16, voffset, soffset, 0, 0);
}

// ---------------------------------------------------------------------------------------------- semantics probe
__global__ void semantics_kernel(const unsigned* src, int nbytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00020000);
    // instruction 0: lanes read 16 B at byte offset 16 * (63 - lane)  (reversed), into lds[0..255]
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds, 16, 16 * (63 - (int)threadIdx.x), 0, 0, 0);
    // instruction 1: odd lanes out of range, soffset 1024, into lds[256..511]
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 256), 16, (threadIdx.x & 1) ? kOob : 16 * (int)threadIdx.x, 1024, 0, 0);
    // instruction 2: voffset just inside, soffset pushes past the end (is soffset range-checked?), into lds[512..767]
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + 512), 16, 16 * (int)threadIdx.x, nbytes - 512, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += 64) out[i] = lds[i];
}

// ---------------------------------------------------------------------------------------------- fp32 -> hl32
// hl32: [row][K/32][2][32] halves: per 32-k chunk the 32 hi halves (fp16(s x)) then the 32 lo halves (fp16(s x - hi))
__global__ void split_hl32_kernel(const float* __restrict__ x, _Float16* __restrict__ o, long rows, int K, float s) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over rows * K
    if (i >= rows * K) return;
    const long r = i / K;
    const int k = (int)(i - r * K);
    const float v = x[i] * s;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    _Float16* c = o + (r * (K / 32) + k / 32) * 64;
    c[k % 32] = hi;
    c[32 + k % 32] = lo;
}

// ---------------------------------------------------------------------------------------------- GEMM core
// 8 wavefronts as WR (rows) x WC (cols); wavefront tile (32 TM) x (32 TN); workgroup tile BM = 32 TM WR, BN = 32 TN WC.
// PIPE 0: issue the LDS-DMA of the next stage, compute this stage (both k-steps), barrier.
// PIPE 3: explicit ping-pong of two wavefront groups (see the kernel body).
template <int TM, int TN, int WR, int WC, int PIPE>
__global__ void __launch_bounds__(512, 1)
gemm_hl_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ B, float* __restrict__ C, int M, int N, int K,
               float inv_scale) {
    constexpr int BM = 32 * TM * WR, BN = 32 * TN * WC;
    constexpr int kStageBytes = (BM + BN) * 128;          // per 32-K stage: rows x [hi 64 B | lo 64 B]
    constexpr int IA = BM * 128 / 1024 / 8, IB = BN * 128 / 1024 / 8;   // LDS-DMA instructions per wavefront per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm_ = wv / WC, wn_ = wv % WC;
    const int ntn = N / BN;
    const int mt = blockIdx.x / ntn, nt = blockIdx.x - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = K / 32;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(A), 0, (int)((long)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(B), 0, (int)((long)N * K * 4), 0x00020000);
    // LDS-DMA piece i of this wavefront covers image rows (wv * I + i) * 8 + (lane >> 3); lane's PHYSICAL 16-byte slot is
    // lane & 7, its LOGICAL slot (which 16 bytes of the row's 128-byte chunk it fetches) slot ^ ((row >> 1) & 7)
    int voa[IA], vob[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int row = (wv * IA + i) * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        voa[i] = m0 + row < M ? (m0 + row) * (K * 4) + slot * 16 : kOob;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int row = (wv * IB + i) * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        vob[i] = n0 + row < N ? (n0 + row) * (K * 4) + slot * 16 : kOob;
    }
    auto issue = [&](int kt, int buf) {
        unsigned char* base = lds + buf * kStageBytes;
        const int soff = kt * 128;
#pragma unroll
        for (int i = 0; i < IA; ++i)
            glds16(rs_a, base + (wv * IA + i) * 1024, voa[i], soff);
#pragma unroll
        for (int i = 0; i < IB; ++i)
            glds16(rs_b, base + BM * 128 + (wv * IB + i) * 1024, vob[i], soff);
    };
    // fragments: lane (fi = lane & 31, fh = lane >> 5) holds 8 consecutive k (k-octet 2 ks + fh) of row fi of a 32-row tile
    const int fi = lane & 31, fh = lane >> 5, swz = (fi >> 1) & 7;
    int foff[2][2];   // [plane][ks] byte offset of the lane's slot inside its row
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) foff[pl][ks] = ((pl * 4 + ks * 2 + fh) ^ swz) * 16;
    const int a_row = (wm_ * 32 * TM + fi) * 128, b_row = BM * 128 + (wn_ * 32 * TN + fi) * 128;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    h8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
    auto read_frags = [&](int buf, int ks, int set) {
        const unsigned char* st = lds + buf * kStageBytes;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            ah[set][t] = *reinterpret_cast<const h8*>(st + a_row + t * 4096 + foff[0][ks]);
            al[set][t] = *reinterpret_cast<const h8*>(st + a_row + t * 4096 + foff[1][ks]);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            bh[set][t] = *reinterpret_cast<const h8*>(st + b_row + t * 4096 + foff[0][ks]);
            bl[set][t] = *reinterpret_cast<const h8*>(st + b_row + t * 4096 + foff[1][ks]);
        }
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pt == 0 ? al[set][tm] : ah[set][tm],
                                                                         pt == 1 ? bl[set][tn] : bh[set][tn], acc[tm][tn], 0, 0, 0);
    };
    if (PIPE == 3) {
        // Explicit ping-pong: the 4 wavefronts of group g = wavefront >> 2 (one of each group per SIMD) alternate LOAD phases
        // (fragment reads of one 16-k step; the LDS-DMA of the next stage rides in phase 0) and COMPUTE phases (its 3 TM TN
        // MFMAs), group 1 one phase behind group 0, a raw s_barrier (no fence: LDS-DMAs stay in flight across it) between
        // phases: while one wavefront of a SIMD computes, the other one loads, and the matrix pipe never waits for LDS.
        //   phase 4s+q    q = 0                       1          2          3
        //   group 0       L(s,0) + DMA(s+1)           C(s,0)     L(s,1)     C(s,1) + vmcnt(0)
        //   group 1       C(s-1,1) + DMA(s+1)         L(s,0)     C(s,0)     L(s,1) + vmcnt(0)      (+ a last C(nk-1,1))
        const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);
        auto bar = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
        auto wait_lds = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
        auto wait_dma = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        issue(0, 0);
        wait_dma();
        bar();
        if (grp == 0) {
            for (int st = 0; st < nk; ++st) {
                if (st + 1 < nk) issue(st + 1, (st + 1) & 1);
                read_frags(st & 1, 0, 0); wait_lds(); bar();
                mfmas(0); bar();
                read_frags(st & 1, 1, 0); wait_lds(); bar();
                mfmas(0); wait_dma(); bar();
            }
            bar();
        } else {
            for (int st = 0; st < nk; ++st) {
                if (st + 1 < nk) issue(st + 1, (st + 1) & 1);
                if (st > 0) mfmas(0);
                bar();
                read_frags(st & 1, 0, 0); wait_lds(); bar();
                mfmas(0); bar();
                read_frags(st & 1, 1, 0); wait_lds(); wait_dma(); bar();
            }
            mfmas(0);
            bar();
        }
    } else if (PIPE == 1) {
        // PIPE 1: as PIPE 0, but the fragments are double-buffered in registers: the LDS reads of k-step s + 1 are issued
        // before the MFMAs of k-step s (the first k-step of the NEXT stage right behind the barrier that publishes it), so
        // that a wavefront never waits for LDS in front of its matrix burst.
        issue(0, 0);
        __syncthreads();
        read_frags(0, 0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
            read_frags(buf, 1, 1);
            mfmas(0);
            __syncthreads();          // (drains the LDS-DMA of stage kt + 1: it had the MFMAs of k-step 0 to land)
            if (kt + 1 < nk) read_frags(buf ^ 1, 0, 0);
            mfmas(1);
        }
    } else {
        issue(0, 0);
        __syncthreads();   // (the compiler drains vmcnt before the barrier while an LDS-DMA is in flight)
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                read_frags(buf, ks, 0);
                mfmas(0);
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm_ * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int col = n0 + wn_ * 32 * TN + tn * 32 + fi;
                if (row < M) C[(long)row * N + col] = acc[tm][tn][r] * inv_scale;
            }
}

template <int TM, int TN, int WR, int WC, int PIPE>
double run_gemm(const _Float16* A, const _Float16* B, float* C, int M, int N, int K, float inv, int reps) {
    constexpr int BM = 32 * TM * WR, BN = 32 * TN * WC;
    const size_t lds = 2 * (size_t)(BM + BN) * 128;
    CHECK(hipFuncSetAttribute((const void*)gemm_hl_kernel<TM, TN, WR, WC, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid(((M + BM - 1) / BM) * (N / BN)), block(512);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm_hl_kernel<TM, TN, WR, WC, PIPE>), grid, block, lds, 0, A, B, C, M, N, K, inv);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_hl_kernel<TM, TN, WR, WC, PIPE>), grid, block, lds, 0, A, B, C, M, N, K, inv);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

double check(const std::vector<float>& a, const std::vector<float>& b, const float* dC, int M, int N, int K) {
    std::vector<float> c((size_t)M * N);
    CHECK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    unsigned s = 12345;
    for (int t = 0; t < 4000; ++t) {
        s = s * 1664525u + 1013904223u;
        const int r = (t < 8 ? (M - 1 - t) : (int)((s >> 8) % (unsigned)M));
        s = s * 1664525u + 1013904223u;
        const int col = (int)((s >> 8) % (unsigned)N);
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)a[(size_t)r * K + k] * (double)b[(size_t)col * K + k];
        worst = std::fmax(worst, std::fabs(ref - (double)c[(size_t)r * N + col]));
        scale = std::fmax(scale, std::fabs(ref));
    }
    return worst / scale;
}

int main(int argc, char** argv) {
    // ---- semantics
    {
        const int n = 4096;
        std::vector<unsigned> h(n / 4);
        for (int i = 0; i < n / 4; ++i) h[i] = 0x1000u + i;
        unsigned *d, *o;
        CHECK(hipMalloc(&d, n));
        CHECK(hipMalloc(&o, 768 * 4));
        CHECK(hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(semantics_kernel, dim3(1), dim3(64), 0, 0, d, n, o);
        std::vector<unsigned> r(768);
        CHECK(hipMemcpy(r.data(), o, 768 * 4, hipMemcpyDeviceToHost));
        int ok0 = 1, ok1 = 1, zeros2 = 0, wrapped2 = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                ok0 &= r[l * 4 + j] == 0x1000u + (63 - l) * 4 + j;                                  // lane-linear LDS, per-lane source
                ok1 &= r[256 + l * 4 + j] == ((l & 1) ? 0u : 0x1000u + 256 + l * 4 + j);            // OOB lanes -> zeros; soffset added
                zeros2 += r[512 + l * 4 + j] == 0u;
                wrapped2 += r[512 + l * 4 + j] == 0xdeadbeefu;
            }
        printf("LDS-DMA semantics: lane-linear placement %s; out-of-range voffset -> zeros in LDS, soffset added %s; "
               "voffset in range + soffset past the end: %d of 256 dwords zero, %d untouched (first: %08x)\n",
               ok0 ? "OK" : "FAIL", ok1 ? "OK" : "FAIL", zeros2, wrapped2, r[512 + 4 * 40]);
    }
    // ---- GEMM core
    const int M = argc > 1 ? atoi(argv[1]) : 76800, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 4608;
    std::vector<float> a((size_t)M * K), b((size_t)N * K);
    unsigned s = 1;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : a) v = rnd();
    for (auto& v : b) v = rnd() * 0.05f;
    float *da, *db, *dc;
    _Float16 *ha, *hb;
    CHECK(hipMalloc(&da, a.size() * 4));
    CHECK(hipMalloc(&db, b.size() * 4));
    CHECK(hipMalloc(&dc, (size_t)M * N * 4));
    CHECK(hipMalloc(&ha, a.size() * 4));
    CHECK(hipMalloc(&hb, b.size() * 4));
    CHECK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(split_hl32_kernel, dim3((unsigned)((a.size() + 255) / 256)), dim3(256), 0, 0, da, ha, (long)M, K, 1.0f);
    hipLaunchKernelGGL(split_hl32_kernel, dim3((unsigned)((b.size() + 255) / 256)), dim3(256), 0, 0, db, hb, (long)N, K, 64.0f);
    CHECK(hipDeviceSynchronize());
    const double flop = 2.0 * M * N * (double)K;
    struct R { const char* name; double ms; double err; };
    std::vector<R> res;
    auto run = [&](const char* name, auto fn) {
        CHECK(hipMemset(dc, 0, (size_t)M * N * 4));
        R r;
        r.name = name;
        r.ms = fn();
        r.err = check(a, b, dc, M, N, K);
        res.push_back(r);
    };
    const float inv = 1.0f / 64.0f;
    run("256x256 (2x4 waves of 128x64), plain loop", [&] { return run_gemm<4, 2, 2, 4, 0>(ha, hb, dc, M, N, K, inv, 10); });
    run("256x256 (2x4 waves of 128x64), explicit ping-pong of two wavefront groups", [&] { return run_gemm<4, 2, 2, 4, 3>(ha, hb, dc, M, N, K, inv, 10); });
    run("256x256 (2x4 waves of 128x64), double-buffered fragments", [&] { return run_gemm<4, 2, 2, 4, 1>(ha, hb, dc, M, N, K, inv, 10); });
    run("256x256 (4x2 waves of 64x128), double-buffered fragments", [&] { return run_gemm<2, 4, 4, 2, 1>(ha, hb, dc, M, N, K, inv, 10); });
    run("256x128 (4x2 waves of 64x64), double-buffered fragments", [&] { return run_gemm<2, 2, 4, 2, 1>(ha, hb, dc, M, N, K, inv, 10); });
    run("256x256 (4x2 waves of 64x128), plain loop", [&] { return run_gemm<2, 4, 4, 2, 0>(ha, hb, dc, M, N, K, inv, 10); });
    run("256x128 (4x2 waves of 64x64), plain loop", [&] { return run_gemm<2, 2, 4, 2, 0>(ha, hb, dc, M, N, K, inv, 10); });
    for (auto& r : res)
        printf("hl32 LDS-DMA GEMM %d x %d x %d, tile %s: %.1f us, %.1f TFLOP/s algorithmic (x3 MFMA products: %.0f TF fp16), "
               "max rel err vs float64 %.2e\n", M, N, K, r.name, r.ms * 1e3, flop / (r.ms * 1e-3) / 1e12,
               3 * flop / (r.ms * 1e-3) / 1e12, r.err);
    return 0;
}
