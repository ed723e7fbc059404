// Round-5 copy of the round-3 probe (tools/hl_gemm_probe2.hip) with ONE more variant: the same two-group schedule on
// v_mfma_f32_16x16x32_f16 tiles (VAR & 8) -- tools/mfma_shape_peak.hip measured that shape at 1.9-2.0 PF sustained on random operands
// against 1.5-1.66 PF for 32x32x16 (half the accumulator-register traffic per FLOP under the power limit).
// Round-3 probe of the split-fp16 ("f16x3") GEMM core on gfx950: the two-wavefront-group, phase-offset schedule.
//   C[M][N] = A[M][K] B[N][K]^T, both operands pre-split "hl32" ([row][K/32][hi x32 | lo x32] fp16, one 128-byte line per
//   32-K chunk), 256 x 256 tiles on 8 wavefronts (2 x 4, wavefront tile 128 x 64 = 8 accumulators of 32 x 32), operands
//   filled by LDS-DMA (`buffer_load_dwordx4 ... lds`) into an XOR-swizzled lane-linear image, 3 MFMAs per product.
// Round 2's plain loop (kept here as the same-box baseline, PLAIN) issues the 8 LDS-DMA pieces of a stage at the top of
// the stage on all 8 wavefronts at once -- nothing feeds the matrix pipe while they issue (60-185 cycles per piece).
// PP: the wavefronts of group 1 (waves 4-7, one per SIMD) run ONE barrier behind group 0 (waves 0-3), a 32-K stage is four
// phases (one 64 x 32 quadrant of the wavefront tile x both 16-k steps x 3 products = 12 MFMAs = 384 matrix-pipe cycles),
// each phase = LOAD slot (fragment reads of the quadrant + 2 LDS-DMA pieces of the NEXT stage, counted vmcnt) | barrier |
// COMPUTE slot (the 12 MFMAs, s_setprio 1) | barrier: while one wavefront of a SIMD computes, the other one loads.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/hl_probe2 tools/hl_gemm_probe2.hip && /tmp/hl_probe2
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kOob = (int)0x80000000;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, void* lds_dst, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_dst, 16, voffset, soffset, 0, 0);
}

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

// ------------------------------------------------------------------------------------------ round-2 plain loop (baseline)
template <int TM, int TN, int WR, int WC>
__global__ void __launch_bounds__(512, 1)
gemm_hl_plain_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                     float inv_scale) {
    constexpr int BM = 32 * TM * WR, BN = 32 * TN * WC;
    constexpr int kStageBytes = (BM + BN) * 128;
    constexpr int IA = BM * 128 / 1024 / 8, IB = BN * 128 / 1024 / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm_ = wv / WC, wn_ = wv % WC;
    const int ntn = N / BN;
    const int mt = blockIdx.x / ntn, nt = blockIdx.x - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = K / 32;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(A), 0, (int)((long)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(B), 0, (int)((long)N * K * 4), 0x00020000);
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
        for (int i = 0; i < IA; ++i) glds16(rs_a, base + (wv * IA + i) * 1024, voa[i], soff);
#pragma unroll
        for (int i = 0; i < IB; ++i) glds16(rs_b, base + BM * 128 + (wv * IB + i) * 1024, vob[i], soff);
    };
    const int fi = lane & 31, fh = lane >> 5, swz = (fi >> 1) & 7;
    int foff[2][2];
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
    h8 ah[TM], al[TM], bh[TN], bl[TN];
    issue(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* st = lds + buf * kStageBytes;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                ah[t] = *reinterpret_cast<const h8*>(st + a_row + t * 4096 + foff[0][ks]);
                al[t] = *reinterpret_cast<const h8*>(st + a_row + t * 4096 + foff[1][ks]);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                bh[t] = *reinterpret_cast<const h8*>(st + b_row + t * 4096 + foff[0][ks]);
                bl[t] = *reinterpret_cast<const h8*>(st + b_row + t * 4096 + foff[1][ks]);
            }
#pragma unroll
            for (int pt = 0; pt < 3; ++pt)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pt == 0 ? al[tm] : ah[tm], pt == 1 ? bl[tn] : bh[tn],
                                                                             acc[tm][tn], 0, 0, 0);
        }
        __syncthreads();
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

// ------------------------------------------------------------------------------------------ phase-offset ping-pong
// Tile 256 x 256, stage = 32 K = 64 KB of LDS: [A rows 0..255][B rows 0..255] x 128 B, two stage buffers (128 KB).
// Half-tiles (16 KB = 16 LDS-DMA pieces of 8 rows, two per wavefront): H0 = A rows 0..127, H1 = B rows 0..127,
// H2 = B rows 128..255, H3 = A rows 128..255 -- the order in which a stage's phases first need them.
// Wavefront (g = wv >> 2, wn = wv & 3) owns tile rows {i*128 + g*64 + [0,64)} (i = 0, 1) and columns {j*128 + wn*32 + [0,32)}
// (j = 0, 1): quadrant (i, j) needs half-tiles A_i and B_j only.  Phase order (0,0) (0,1) (1,1) (1,0): one operand's
// fragments stay in registers from phase to phase.
// LDS-DMA discipline (every wavefront, in its LOAD slot of phase p of stage s): issue H_p of stage s+1 into the other
// buffer, then wait until at most two half-tiles (4 pieces) are in flight -> at the end of LOAD(s, p) everything up to
// H_{p-2}(s+1) has landed for this wavefront; the barrier that closes the slot publishes it.  Needs: LOAD(s,1) reads H2(s)
// [landed at the end of LOAD(s,0)], LOAD(s,2) reads H3(s) [end of LOAD(s,1)], LOAD(s+1,0) reads H0, H1 of s+1 [end of
// LOAD(s,3)]; group 1's slots are one barrier later than group 0's, so the latest publication still precedes the earliest
// read.  WAR: H_p(s+1) overwrites H_p(s-1), last read in LOAD(s-1, 3) of group 1, two barriers before LOAD(s, 0) of group 0.
// VAR bit 0: s_setprio 1 around the MFMA cluster.  bit 1: LDS-DMA issued before the fragment reads of the slot (default after).
// bit 2: no explicit lgkmcnt(0) before the slot's barrier (the compiler's own waits in front of the MFMAs remain).
template <int VAR>
__global__ void __launch_bounds__(512, 1)
gemm_hl_pp_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                  float inv_scale) {
    constexpr int BM = 256, BN = 256, kStageBytes = 65536;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wv >> 2, wn = wv & 3;
    const int ntn = N / BN;
    const int mt = blockIdx.x / ntn, nt = blockIdx.x - mt * ntn;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = K / 32;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(A), 0, (int)((long)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(B), 0, (int)((long)N * K * 4), 0x00020000);
    // half-tile h: operand (0: A, 1: B), first tile row, LDS base inside the stage buffer
    //   H0 = A 0..127 @0, H1 = B 0..127 @32768, H2 = B 128..255 @49152, H3 = A 128..255 @16384
    int vo[4][2];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool isb = h == 1 || h == 2;
            const int row = (h >= 2 ? 128 : 0) + (2 * wv + i) * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((row >> 1) & 7);
            const int g = (isb ? n0 : m0) + row;
            vo[h][i] = g < (isb ? N : M) ? g * (K * 4) + slot * 16 : kOob;
        }
    auto issue_half = [&](int kt, int buf, int h) {
        const int soff = kt * 128;
        unsigned char* base = lds + buf * kStageBytes + (h == 0 ? 0 : h == 1 ? 32768 : h == 2 ? 49152 : 16384) + wv * 2048;
        glds16((h == 1 || h == 2) ? rs_b : rs_a, base, vo[h][0], soff);
        glds16((h == 1 || h == 2) ? rs_b : rs_a, base + 1024, vo[h][1], soff);
    };
    const int fi = lane & 31, fh = lane >> 5, swz = (fi >> 1) & 7;
    int foff[2][2];   // [plane][ks]
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) foff[pl][ks] = ((pl * 4 + ks * 2 + fh) ^ swz) * 16;
    const int a_row = (grp * 64 + fi) * 128, b_row = 32768 + (wn * 32 + fi) * 128;

    f32x16 acc[4][2];   // [2 i + t][j]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    h8 fa[2][2][2], fb[2][2];   // A: [t][ks][plane], B: [ks][plane]   (plane 0 = hi, 1 = lo)
    // 16 x 16 x 32 variant: quadrant (i, j) = 4 row blocks x 2 column blocks of 16 x 16; lane (fr = lane & 15, fq = lane >> 4)
    // holds k-octet fq of row fr: ONE fragment read per block and plane covers the whole 32-K stage
    constexpr bool M16 = (VAR & 8) != 0;
    f32x4 acc16[2][2][4][2];   // [i][j][rb][cb]
    if (M16) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) acc16[i][j][rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    h8 ga[4][2], gb[2][2];      // A: [rb][plane], B: [cb][plane]
    h8 gb2[2][2];               // (VAR & 16) the second column half's B fragments, kept for the whole stage
    const int fr = lane & 15, fq = lane >> 4, swz16 = (fr >> 1) & 7;
    int goff[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) goff[pl] = fr * 128 + ((pl * 4 + fq) ^ swz16) * 16;
    const int a_row16 = (grp * 64) * 128, b_row16 = 32768 + (wn * 32) * 128;

    auto read_a = [&](int buf, int i) {
        if (M16) {
            const unsigned char* st = lds + buf * kStageBytes + a_row16 + i * 16384;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) ga[rb][pl] = *reinterpret_cast<const h8*>(st + rb * 2048 + goff[pl]);
            return;
        }
        const unsigned char* st = lds + buf * kStageBytes + a_row + i * 16384;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) fa[t][ks][pl] = *reinterpret_cast<const h8*>(st + t * 4096 + foff[pl][ks]);
    };
    auto read_b = [&](int buf, int j) {
        if (M16) {
            const unsigned char* st = lds + buf * kStageBytes + b_row16 + j * 16384;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) gb[cb][pl] = *reinterpret_cast<const h8*>(st + cb * 2048 + goff[pl]);
            return;
        }
        const unsigned char* st = lds + buf * kStageBytes + b_row + j * 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fb[ks][pl] = *reinterpret_cast<const h8*>(st + foff[pl][ks]);
    };
    auto mfma_quadrant = [&](int i, int j) {
        if (VAR & 1) __builtin_amdgcn_s_setprio(1);
        if (M16) {
#pragma unroll
            for (int pt = 0; pt < 3; ++pt)
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc16[i][j][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pt == 0 ? ga[rb][1] : ga[rb][0],
                                                                                     pt == 1 ? gb[cb][1] : gb[cb][0], acc16[i][j][rb][cb], 0, 0, 0);
            if (VAR & 1) __builtin_amdgcn_s_setprio(0);
            return;
        }
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[2 * i + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pt == 0 ? fa[t][ks][1] : fa[t][ks][0],
                                                                              pt == 1 ? fb[ks][1] : fb[ks][0], acc[2 * i + t][j], 0, 0, 0);
        if (VAR & 1) __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
#define PP_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
    // one phase; MORE: stage s+1 exists (issue its half-tile h = p, leave 2 half-tiles in flight), else drain what phase p+1 reads
    auto phase = [&](auto more_tag, int s, int p) {
        constexpr bool MORE = decltype(more_tag)::value;
        const int buf = s & 1;
        const int qi = p >> 1, qj = (p == 1 || p == 2) ? 1 : 0;
        if ((VAR & 2) && MORE) issue_half(s + 1, buf ^ 1, p);
        if (p == 0) { read_b(buf, 0); __builtin_amdgcn_sched_barrier(0); read_a(buf, 0); }
        else if (p == 1) read_b(buf, 1);
        else if (p == 2) read_a(buf, 1);
        else read_b(buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(VAR & 2) && MORE) issue_half(s + 1, buf ^ 1, p);
        __builtin_amdgcn_sched_barrier(0);
        if (MORE) PP_WAIT_VM(4);
        else if (p == 0) PP_WAIT_VM(2);
        else if (p == 1) PP_WAIT_VM(0);
        if (!(VAR & 4)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        mfma_quadrant(qi, qj);
        bar();
    };
    using T = std::true_type;
    using F = std::false_type;

    if constexpr ((VAR & 16) != 0) {
        // TWO quadrants per slot (48 MFMAs = 768 matrix-pipe cycles): slot 0 = quadrants (0,0) (0,1), slot 1 = (1,1) (1,0); the B
        // fragments of both column halves are read in slot 0 and kept.  LOAD(s,0) issues H0 H1 H2 of stage s + 1, LOAD(s,1) its H3;
        // end of LOAD(s,0): H3(s) landed = all but the 6 youngest pieces; end of LOAD(s,1): H0 H1 H2 (s + 1) = all but 2.
        auto read_b2 = [&](int buf) {
            const unsigned char* st = lds + buf * kStageBytes + b_row16 + 16384;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) gb2[cb][pl] = *reinterpret_cast<const h8*>(st + cb * 2048 + goff[pl]);
        };
        auto mfma_pair = [&](int i) {
            if (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int pt = 0; pt < 3; ++pt)
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        acc16[i][0][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pt == 0 ? ga[rb][1] : ga[rb][0], pt == 1 ? gb[cb][1] : gb[cb][0], acc16[i][0][rb][cb], 0, 0, 0);
                        acc16[i][1][rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pt == 0 ? ga[rb][1] : ga[rb][0], pt == 1 ? gb2[cb][1] : gb2[cb][0], acc16[i][1][rb][cb], 0, 0, 0);
                    }
            if (VAR & 1) __builtin_amdgcn_s_setprio(0);
        };
        auto slot = [&](auto more_tag, int s, int p) {
            constexpr bool MORE = decltype(more_tag)::value;
            const int buf = s & 1;
            if (p == 0) { read_b(buf, 0); read_b2(buf); __builtin_amdgcn_sched_barrier(0); read_a(buf, 0); }
            else read_a(buf, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (MORE) {
                if (p == 0) { issue_half(s + 1, buf ^ 1, 0); issue_half(s + 1, buf ^ 1, 1); issue_half(s + 1, buf ^ 1, 2); }
                else issue_half(s + 1, buf ^ 1, 3);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MORE) { if (p == 0) PP_WAIT_VM(6); else PP_WAIT_VM(2); }
            else if (p == 0) PP_WAIT_VM(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bar();
            mfma_pair(p);
            bar();
        };
        issue_half(0, 0, 0);
        issue_half(0, 0, 1);
        issue_half(0, 0, 2);
        issue_half(0, 0, 3);
        __builtin_amdgcn_sched_barrier(0);
        PP_WAIT_VM(2);
        bar();
        if (grp == 1) bar();
        for (int s = 0; s + 1 < nk; ++s) {
            slot(T{}, s, 0);
            slot(T{}, s, 1);
        }
        slot(F{}, nk - 1, 0);
        slot(F{}, nk - 1, 1);
        if (grp == 0) bar();
    } else {
    issue_half(0, 0, 0);
    issue_half(0, 0, 1);
    issue_half(0, 0, 2);
    issue_half(0, 0, 3);
    __builtin_amdgcn_sched_barrier(0);
    PP_WAIT_VM(4);
    bar();
    if (grp == 1) bar();
    for (int s = 0; s + 1 < nk; ++s) {
        phase(T{}, s, 0);
        phase(T{}, s, 1);
        phase(T{}, s, 2);
        phase(T{}, s, 3);
    }
    phase(F{}, nk - 1, 0);
    phase(F{}, nk - 1, 1);
    phase(F{}, nk - 1, 2);
    phase(F{}, nk - 1, 3);
    if (grp == 0) bar();
    }
#undef PP_WAIT_VM
    if (M16) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = m0 + i * 128 + grp * 64 + rb * 16 + 4 * fq + r;
                            const int col = n0 + j * 128 + wn * 32 + cb * 16 + fr;
                            if (row < M) C[(long)row * N + col] = acc16[i][j][rb][cb][r] * inv_scale;
                        }
        return;
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (tm >> 1) * 128 + grp * 64 + (tm & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int col = n0 + tn * 128 + wn * 32 + fi;
                if (row < M) C[(long)row * N + col] = acc[tm][tn][r] * inv_scale;
            }
}

template <class Kern>
double run_kernel(Kern kern, int bm, int bn, size_t lds, const _Float16* A, const _Float16* B, float* C, int M, int N, int K,
                  float inv, int reps) {
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid(((M + bm - 1) / bm) * (N / bn)), block(512);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, block, lds, 0, A, B, C, M, N, K, inv);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, block, lds, 0, A, B, C, M, N, K, inv);
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
    for (int t = 0; t < 6000; ++t) {
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
    const int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 4608;
    const unsigned mask = argc > 4 ? (unsigned)strtoul(argv[4], nullptr, 0) : 0xffffffffu;
    const int reps = argc > 5 ? atoi(argv[5]) : 10;
    const int rounds = argc > 6 ? atoi(argv[6]) : 2;
    const int relu = argc > 7 ? atoi(argv[7]) : 0;   // 1: A = max(x, 0) (post-ReLU activations: half zeros, one sign)
    std::vector<float> a((size_t)M * K), b((size_t)N * K);
    unsigned s = 1;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : a) { v = rnd(); if (relu && v < 0.f) v = 0.f; }
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
    const float inv = 1.0f / 64.0f;
    const size_t lds = 131072;
    struct V { const char* name; unsigned bit; double (*fn)(const _Float16*, const _Float16*, float*, int, int, int, float, int); };
#define PPV(v) [](const _Float16* A, const _Float16* B, float* C, int M, int N, int K, float inv, int reps) { return run_kernel(gemm_hl_pp_kernel<v>, 256, 256, 131072, A, B, C, M, N, K, inv, reps); }
    const V variants[] = {
        {"plain loop (round 2)", 1u, [](const _Float16* A, const _Float16* B, float* C, int M, int N, int K, float inv, int reps) { return run_kernel(gemm_hl_plain_kernel<4, 2, 2, 4>, 256, 256, 131072, A, B, C, M, N, K, inv, reps); }},
        {"ping-pong, 4 phases/stage", 2u, PPV(0)},
        {"ping-pong + setprio", 4u, PPV(1)},
        {"ping-pong + setprio, DMA before reads", 8u, PPV(3)},
        {"ping-pong + setprio, no lgkmcnt(0) before the barrier", 16u, PPV(5)},
        {"ping-pong, DMA before reads, no setprio", 32u, PPV(2)},
        {"ping-pong + setprio, 16x16x32 MFMA tiles", 64u, PPV(9)},
        {"ping-pong, 16x16x32 MFMA tiles, no setprio", 128u, PPV(8)},
        {"16x16x32, setprio, no lgkmcnt(0) before the barrier", 256u, PPV(13)},
        {"16x16x32, no setprio, no lgkmcnt(0) before the barrier", 512u, PPV(12)},
        {"16x16x32, setprio, DMA before reads", 1024u, PPV(11)},
        {"16x16x32, no setprio, DMA before reads, no lgkmcnt(0)", 2048u, PPV(14)},
        {"16x16x32, setprio, TWO quadrants per slot (48 MFMAs)", 4096u, PPV(25)},
        {"16x16x32, no setprio, TWO quadrants per slot (48 MFMAs)", 8192u, PPV(24)},
    };
    (void)lds;
    for (int round = 0; round < rounds; ++round)
        for (const V& v : variants) {
            if (!(mask & v.bit)) continue;
            CHECK(hipMemset(dc, 0, (size_t)M * N * 4));
            const double ms = v.fn(ha, hb, dc, M, N, K, inv, reps);
            const double err = round == 0 ? check(a, b, dc, M, N, K) : -1.0;
            printf("hl32 GEMM %d x %d x %d%s, 256x256 tile, %-55s round %d: %8.1f us, %6.1f TFLOP/s algorithmic (x3: %5.0f TF fp16)",
                   M, N, K, relu ? " relu-A" : "", v.name, round, ms * 1e3, flop / (ms * 1e-3) / 1e12, 3 * flop / (ms * 1e-3) / 1e12);
            if (err >= 0) printf(", max rel err vs float64 %.2e", err);
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
