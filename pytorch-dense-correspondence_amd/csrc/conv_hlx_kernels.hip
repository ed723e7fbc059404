// Small-tile variants of the pre-split (hl32) LDS-DMA gather-GEMM (conv_hl_kernels.hip has the operand format, the LDS
// image and the big 256 / 192 / 320-row tiles): forward convolution and dgrad for the launches that do NOT fill the 256 CUs
// with 256 x 256 output tiles -- the reference's own operating point (training.yaml:14 `batch_size: 1`, and the two separate
// forward calls of training.py:329-333: M = 4 800 .. 19 200 rows per launch in layers 3-4), layer 3 at 8 images (38 400 x 256
// outputs = 150 tiles of 256 rows), and the 128-channel layers.
//
// Tile 160 rows x BN columns on 8 wavefronts, every wavefront an 80 x 64 block of v_mfma_f32_16x16x32_f16 tiles (5 x 4
// accumulators of 4 registers; 16-row granularity is what makes 160 = 9 600 / 60 possible) in the split-fp16 arithmetic
// (lo x hi + hi x lo + hi x hi).  Two shapes:
//   KG = 1: BN = 256, wavefronts 2 (M) x 4 (N); a stage = one 32-channel chunk: 160 + 256 lines of 128 B = 52 KB, two buffers.
//   KG = 2: BN = 128, TWO K GROUPS of 2 x 2 wavefronts: a stage = two consecutive chunks (2 x (160 + 128) lines = 72 KB, two
//           buffers), group g multiplies chunk g -- each wavefront keeps the 80 x 64 block (0.6 fragment reads per MFMA) on a
//           tile of half the width; the two groups' accumulators are added through LDS once, after the loop (each keeps one
//           half of the columns: a + b, one order).
// 38 400 x 256 outputs are 240 tiles of 160 x 256 (layer 3 at 8 images: ONE round on 256 CUs instead of 200 tiles of 192);
// 9 600 x 512 and 19 200 x 256 are 240 tiles of 160 x 128.  What is smaller still is split along K over `ksplit`
// workgroups per tile (contiguous ranges of stages; partial tiles parked device-coherently, summed in FIXED order by whichever
// workgroup arrives last, inside the launch -- the stream-K completion protocol of conv_shared.h): bit-reproducible.
//
// The loop is the plain double-buffered one (no wavefront-group phase offset): per stage
//     s_waitcnt vmcnt(0) | s_barrier | LDS-DMA of stage s + 1 into the other buffer | fragment reads + 60 MFMAs of stage s
// -- the barrier publishes stage s (every wavefront has waited for its own pieces) and retires the reads of stage s - 1 (the
// buffer stage s + 1 overwrites); the DMA has the whole compute slot of a stage to land.  Two wavefronts per SIMD drift
// apart and fill each other's waits.  Fragment reads: lane (row l & 15, k-octet l >> 4) reads 16 B of slot
// (plane * 4 + octet) ^ ((row >> 1) & 7) -- the source-side XOR swizzle of conv_hl_kernels.hip; conflict-free for the
// ds_read_b128 lane groups of gfx950 as well (rows 0-3 / 12-15 of one octet with rows 4-11 of the next: 16 distinct
// (row parity, slot) pairs).
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "conv_hlx.h"
#include "dcn_tuning.h"
#include "f16_split.h"

namespace {

using namespace dcnconv;
using namespace dcnsplit;

typedef __attribute__((address_space(3))) void* hlx_lds_ptr;
typedef f32x4_t f32x4;

constexpr int kOob = (int)0x80000000;   // voffset that fails the bounds check of any buffer <= 2 GiB: LDS receives zeros
constexpr int XR = 160;                 // rows per tile
constexpr int XTM = 5, XTN = 4;         // 16 x 16 accumulator tiles per wavefront: 80 rows x 64 columns

template <int KG> struct Hlx {
    static constexpr int BN = KG == 1 ? 256 : 128;
    static constexpr int WN = KG == 1 ? 4 : 2;            // wavefronts along N inside a K group (2 along M)
    static constexpr int kChunk = (XR + BN) * 128;        // bytes of one 32-channel chunk: A lines, then B lines
    static constexpr int kStage = KG * kChunk;
    static constexpr int kLds = 2 * kStage;               // KG = 1: 106 496 B, KG = 2: 147 456 B
    static constexpr int NB = BN / 64;                    // B row groups (8 lines) per wavefront and chunk
    static constexpr int TNE = XTN / KG;                  // accumulator columns a wavefront owns in the epilogue
};

// (a NON-template function: inside a template this builtin breaks the host-side kernel stub with this compiler)
__device__ __forceinline__ void glds16x(__amdgpu_buffer_rsrc_t rs, void* lds_dst, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (hlx_lds_ptr)lds_dst, 16, voffset, soffset, 0, 0);
}

// TR: dgrad (the gather runs over the output gradient with mirrored taps).
template <bool TR, int KG>
__global__ void __launch_bounds__(512, 1)
conv_gemm_hlx_kernel(GemmConv p) {
    using G = Hlx<KG>;
    constexpr int BN = G::BN, WN = G::WN, NB = G::NB, TNE = G::TNE, kChunk = G::kChunk, kStage = G::kStage;
    __shared__ __attribute__((aligned(16))) unsigned char lds[G::kLds];   // (ONE array: see conv_hl_kernels.hip)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = KG == 1 ? 0 : (wv >> 2);                 // K group
    const int wq = KG == 1 ? wv : (wv & 3);                 // wavefront inside the group
    const int wm = wq / WN, wn = wq % WN;
    const int tiles = p.mtiles * p.ntiles;
    const int S = p.ksplit > 1 ? p.ksplit : 1;
    const int wg = xcd_remap(blockIdx.x, tiles * S);
    const int split = fdiv(wg, p.div_tiles), tile = wg - split * tiles;
    const int mt = fdiv(tile, p.div_nt), nt = tile - mt * p.ntiles;
    const int m0 = mt * XR, n0 = nt * BN;
    const int nk = p.K / 32, nss = nk / KG;
    const int ss0 = (int)(((int64_t)nss * split) / S), ss1 = (int)(((int64_t)nss * (split + 1)) / S);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, (int)p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.wh), 0, (int)p.w_bytes, 0x00020000);

    // ---- LDS-DMA pieces of this lane.  A piece = 8 tile rows x 128 B: lane l fetches row (l >> 3), PHYSICAL 16-byte slot
    // l & 7 = logical slot (l & 7) ^ ((row >> 1) & 7) of the row's hl32 line.  Row groups (of 8) per wavefront and chunk:
    //   A (20 groups): wv, wv + 8 and -- KG = 1: 16 + wv for wv < 4;  KG = 2: 16 + (wv & 3) of chunk (wv >> 2) only
    //   B (BN / 8 groups): wv + 8 b, b < NB
    const int l8 = lane >> 3, ls = lane & 7;
    const int cs4 = p.cs * 4;              // bytes per pixel of the activation image
    const int taps = p.kh * p.kw;
    int rowoff[3];
    unsigned vmask[3];                     // one validity bit per filter tap (taps <= 32)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int rg = a < 2 ? wv + 8 * a : 16 + (wv & 3);
        const int row = rg * 8 + l8;
        const int sl = ls ^ ((row >> 1) & 7);
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int img = fdiv(mm, p.div_hw), rem = mm - img * p.div_hw.d;
        const int y = fdiv(rem, p.div_w), x = rem - y * p.div_w.d;
        const int by = TR ? y + p.pad : y - p.pad, bx = TR ? x + p.pad : x - p.pad;   // (stride 1)
        rowoff[a] = (img * p.hs * p.ws + by * p.ws + bx) * cs4 + sl * 16;
        unsigned vm = 0u;
        for (int tap = 0; tap < taps; ++tap) {
            const int r = fdiv(tap, p.div_kw), s = tap - r * p.kw;
            const int dy = r * p.dil, dx = s * p.dil;
            bool v;
            if (TR) {
                const int ny = by - dy, nx = bx - dx;
                v = ((ny | nx) >= 0) & (ny < p.hs) & (nx < p.ws);
            } else {
                v = ((unsigned)(by + dy) < (unsigned)p.hs) & ((unsigned)(bx + dx) < (unsigned)p.ws);
            }
            vm |= (v ? 1u : 0u) << tap;
        }
        vmask[a] = ok ? vm : 0u;
    }
    int vob[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int row = (wv + 8 * b) * 8 + l8;
        const int sl = ls ^ ((row >> 1) & 7);
        const int n = n0 + row;
        vob[b] = n < p.cd ? n * (nk * 128) + sl * 16 : kOob;
    }
    // K traversal (as conv_hl_kernels.hip): channel-chunk groups outermost (kcg chunks), then the filter taps, then the chunks
    // of the group; a KG = 2 stage is two consecutive chunks of one group and tap (kcg is even: hlx_shape)
    const int cpt = p.cs / 32;
    const int kcg = (cpt & 3) == 0 ? 4 : ((cpt & 1) == 0 ? 2 : 1);
    int u_grp, u_tap, u_c, cur_delta = 0, cur_bit = 0;
    auto set_tap = [&](int tap) {
        const int r = fdiv(tap, p.div_kw), s = tap - r * p.kw;
        const int delta = (r * p.dil * p.ws + s * p.dil) * cs4;
        cur_delta = TR ? -delta : delta;
        cur_bit = tap;
    };
    {
        const int k0 = ss0 * KG, per_grp = taps * kcg;
        u_grp = k0 / per_grp;
        const int rem = k0 - u_grp * per_grp;
        u_tap = rem / kcg;
        u_c = rem - u_tap * kcg;
        set_tap(u_tap);
    }
    auto advance = [&]() {                 // to the next stage (KG chunks on)
        u_c += KG;
        if (u_c >= kcg) {
            u_c = 0;
            if (++u_tap == taps) { u_tap = 0; ++u_grp; }
            set_tap(u_tap);
        }
    };
    auto voa = [&](int a) { return ((vmask[a] >> cur_bit) & 1u) ? rowoff[a] + cur_delta : kOob; };
    auto issue = [&](int buf) {            // every piece of the stage the traversal state points at
        unsigned char* st = lds + buf * kStage;
        const int soa = (u_grp * kcg + u_c) * 128, sob = (u_tap * cpt + u_grp * kcg + u_c) * 128;
#pragma unroll
        for (int ch = 0; ch < KG; ++ch) {
            unsigned char* cbase = st + ch * kChunk;
            glds16x(rs_a, cbase + wv * 1024, voa(0), soa + ch * 128);
            glds16x(rs_a, cbase + (wv + 8) * 1024, voa(1), soa + ch * 128);
#pragma unroll
            for (int b = 0; b < NB; ++b) glds16x(rs_b, cbase + XR * 128 + (wv + 8 * b) * 1024, vob[b], sob + ch * 128);
        }
        if (KG == 1) {
            if (wv < 4) glds16x(rs_a, st + (16 + wv) * 1024, voa(2), soa);
        } else {
            const int ch = wv >> 2;
            glds16x(rs_a, st + ch * kChunk + (16 + (wv & 3)) * 1024, voa(2), soa + ch * 128);
        }
    };

    // ---- fragments: lane (fr = lane & 15, fq = lane >> 4) holds 8 consecutive k (octet fq of the chunk's 32) of row fr of a
    // 16-row tile; plane 0 = hi, 1 = lo
    const int fr = lane & 15, fq = lane >> 4, swz = (fr >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) foff[pl] = fr * 128 + ((pl * 4 + fq) ^ swz) * 16;
    const int a_row = (wm * 80) * 128, b_row = XR * 128 + (wn * 64) * 128;

    f32x4 acc[XTM][XTN];
#pragma unroll
    for (int i = 0; i < XTM; ++i)
#pragma unroll
        for (int j = 0; j < XTN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto bar = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
    // compute slot of a stage in two parts (rows 0-47 of the wavefront's block with the B fragments, rows 48-79), so that the
    // LDS-DMA issue of the next stage can sit in front of the first part (wavefronts 0-3) or between the parts (wavefronts
    // 4-7): wavefronts w and w + 4 share a SIMD, and with every wavefront released by the same barrier an unstaggered loop
    // has BOTH of them issuing their 7-9 DMA pieces (~100 issue cycles each) while the matrix pipe idles.
    h8 fa[XTM][2], fb[XTN][2];
    auto load_b = [&](const unsigned char* cbase) {
#pragma unroll
        for (int tn = 0; tn < XTN; ++tn)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fb[tn][pl] = *reinterpret_cast<const h8*>(cbase + b_row + tn * 2048 + foff[pl]);
    };
    auto load_a = [&](const unsigned char* cbase, auto t0, auto t1) {
#pragma unroll
        for (int tm = decltype(t0)::value; tm < decltype(t1)::value; ++tm)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fa[tm][pl] = *reinterpret_cast<const h8*>(cbase + a_row + tm * 2048 + foff[pl]);
    };
    // product type outermost (the small cross terms before hi x hi), the part's independent accumulators in between
    auto mm = [&](auto t0, auto t1) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int tm = decltype(t0)::value; tm < decltype(t1)::value; ++tm)
#pragma unroll
                for (int tn = 0; tn < XTN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pt == 0 ? fa[tm][1] : fa[tm][0],
                                                                         pt == 1 ? fb[tn][1] : fb[tn][0], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I3 = std::integral_constant<int, 3>;
    using I5 = std::integral_constant<int, XTM>;
    const bool late_issue = p.hlx_stagger == 1 && wv >= 4;
    const int grp = wv >> 2;   // wavefronts w and w + 4 share a SIMD: group 0 = wavefronts 0-3, group 1 = 4-7

    if (ss0 < ss1 && p.hlx_stagger == 2) {
        // Two wavefront groups ONE SLOT APART (the idea of the big tiles' schedule with one phase per stage): a stage is a LOAD
        // slot (all 18 fragment reads of the wavefront's 80 x 64 block) and a COMPUTE slot (its 60 MFMAs), a barrier behind each;
        // group 1 runs one slot behind group 0, so on every SIMD one wavefront computes while the other one loads.  Both groups
        // issue the LDS-DMA pieces of stage s + 1 in the SAME slot -- group 0 in LOAD(s), group 1 at the start of COMPUTE(s - 1)
        // -- and wait for them at the end of the next one: two slots to land.  WAR: the buffer of stage s + 1 held stage s - 1,
        // last read in group 1's LOAD(s - 1), the slot before.  RAW: stage s + 1 is first read in group 0's LOAD(s + 1), behind
        // the barrier that closes the slot both groups' waits sit in.
        issue(0);
        advance();
        DCN_WAIT_VMCNT(0);
        bar();
        if (grp == 1) {   // (group 1's extra slot, next to group 0's LOAD(ss0))
            if (ss0 + 1 < ss1) { issue(1); advance(); }
            bar();
        }
        int buf = 0;
        for (int ss = ss0; ss < ss1; ++ss) {
            const unsigned char* cbase = lds + buf * kStage + kg * kChunk;
            load_b(cbase);
            load_a(cbase, I0{}, I5{});
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 0) {
                if (ss + 1 < ss1) { issue(buf ^ 1); advance(); }
            } else {
                DCN_WAIT_VMCNT(0);        // stage ss + 1 (issued one slot ago)
            }
            __builtin_amdgcn_sched_barrier(0);
            DCN_WAIT_LGKMCNT0();
            bar();
            if (grp == 1 && ss + 2 < ss1) { issue(buf); advance(); }   // stage ss + 2 into the buffer just read
            __builtin_amdgcn_sched_barrier(0);
            mm(I0{}, I5{});
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 0) DCN_WAIT_VMCNT(0);   // stage ss + 1 (issued in this stage's LOAD slot)
            bar();
            buf ^= 1;
        }
        if (grp == 0) bar();
    } else if (ss0 < ss1) {   // (wave-uniform; a split without stages contributes zeros)
        issue(0);
        advance();
        int buf = 0;
        for (int ss = ss0; ss < ss1; ++ss) {
            DCN_WAIT_VMCNT(0);
            bar();
            const bool more = ss + 1 < ss1;
            const unsigned char* cbase = lds + buf * kStage + kg * kChunk;
            if (more && !late_issue) {
                issue(buf ^ 1);
                advance();
            }
            __builtin_amdgcn_sched_barrier(0);
            load_b(cbase);
            load_a(cbase, I0{}, I3{});
            mm(I0{}, I3{});
            __builtin_amdgcn_sched_barrier(0);
            if (more && late_issue) {
                issue(buf ^ 1);
                advance();
            }
            __builtin_amdgcn_sched_barrier(0);
            load_a(cbase, I3{}, I5{});
            mm(I3{}, I5{});
            buf ^= 1;
        }
    }
    __syncthreads();   // every fragment read is done: the array is scratch from here on

    const float sa = p.a_absmax ? pow2_scale(*p.a_absmax) : 1.f;   // (the scale the producer of the hl32 image applied)
    const float inv = p.b_inv_scale / sa;
    float* scratch = reinterpret_cast<float*>(lds);
    // ---- the wavefront's share of the tile: KG = 1 all 64 columns; KG = 2 the two K groups are added through LDS, group g
    // keeps columns 32 g .. 32 g + 31 of the 64 (the partner's half: [wavefront][tm][tn][lane] float4)
    f32x4 out[XTM][TNE];
    int cb;
    if constexpr (KG == 1) {
        cb = wn * 64;
#pragma unroll
        for (int tm = 0; tm < XTM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TNE; ++tn) out[tm][tn] = acc[tm][tn];
    } else {
        cb = wn * 64 + kg * 32;
        f32x4* mine = reinterpret_cast<f32x4*>(scratch) + (size_t)wv * (XTM * TNE * 64) + lane;
        const f32x4* theirs = reinterpret_cast<const f32x4*>(scratch) + (size_t)(wv ^ 4) * (XTM * TNE * 64) + lane;
        if (kg == 0) {
#pragma unroll
            for (int tm = 0; tm < XTM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TNE; ++tn) mine[(tm * TNE + tn) * 64] = acc[tm][TNE + tn];
        } else {
#pragma unroll
            for (int tm = 0; tm < XTM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TNE; ++tn) mine[(tm * TNE + tn) * 64] = acc[tm][tn];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int tm = 0; tm < XTM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TNE; ++tn) out[tm][tn] = acc[tm][tn] + theirs[(tm * TNE + tn) * 64];
        } else {
#pragma unroll
            for (int tm = 0; tm < XTM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TNE; ++tn) out[tm][tn] = theirs[(tm * TNE + tn) * 64] + acc[tm][TNE + tn];
        }
        __syncthreads();   // (the scratch is reused below)
    }
#pragma unroll
    for (int tm = 0; tm < XTM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TNE; ++tn) out[tm][tn] *= inv;

    if (S > 1) {
        // ---- K split over workgroups: park the partial tile device-coherently ([wavefront][tm][tn][lane] float4 -- 160 x BN
        // floats per (tile, split)), count in; the last arriver sums the S partials in fixed order (its own included)
        constexpr int kSc1 = 16;
        constexpr int kSlotBytes = XR * BN * 4;
        const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(p.sk_partial, 0, (int)p.sk_bytes, 0x00020000);
        const int lane_off = (wv * (XTM * TNE * 64) + lane) * 16;
        {
            const int so = (tile * S + split) * kSlotBytes;
#pragma unroll
            for (int tm = 0; tm < XTM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TNE; ++tn)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, out[tm][tn]), rs_p,
                                                           lane_off + (tm * TNE + tn) * 1024, so, kSc1);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this work-item's partial has been written through
        __syncthreads();
        int* s_last = reinterpret_cast<int*>(lds + G::kLds - 16);
        if (tid == 0) *s_last = sk_arrive_is_last(p.sk_count + tile, p.sk_id, S) ? 1 : 0;
        __syncthreads();
        const int last = *s_last;
        __syncthreads();                               // (s_last has been read by everyone: the epilogue reuses the array)
        if (!last) return;
        // sum over the splits in the order 0 .. S - 1, this workgroup's own share from its registers (the same bits it parked)
        f32x4 own[XTM][TNE];
#pragma unroll
        for (int tm = 0; tm < XTM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TNE; ++tn) { own[tm][tn] = out[tm][tn]; out[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        constexpr int kBatch = 10;   // 16-byte loads issued back to back (device-coherent round trips)
        static_assert((XTM * TNE) % kBatch == 0, "whole batches");
        for (int g = 0; g < S; ++g) {
            if (g == split) {
#pragma unroll
                for (int tm = 0; tm < XTM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TNE; ++tn) out[tm][tn] += own[tm][tn];
                continue;
            }
            const int so = (tile * S + g) * kSlotBytes;
#pragma unroll
            for (int b0 = 0; b0 < XTM * TNE; b0 += kBatch) {
                u32x4 t[kBatch];
#pragma unroll
                for (int j = 0; j < kBatch; ++j) t[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, lane_off + (b0 + j) * 1024, so, kSc1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < kBatch; ++j) out[(b0 + j) / TNE][(b0 + j) % TNE] += __builtin_bit_cast(f32x4, t[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (tid == 0) atomicExch(p.sk_count + tile, 0ull);
    }
    gemm_epilogue16<XTM, TNE, XR, BN>(p, out, mt, nt, wm, wm * 80, cb, scratch);
}

}  // namespace

namespace dcnconv {

// Which small-tile shape (if any) a launch of M x cd outputs with K = 32 nk should take, and at what cost in the unit of
// hl_shape's model (conv_hl_kernels.hip): time of one 32-K stage of a 256-column tile ROW, i.e. rows x (columns / 256).
// Forced by DCN_GEMM_HLX = "kg" or "kg,splits" (0: never).
HlxShape hlx_shape(int M, int cd, int K, int group_rows, int taps, int cs) {
    HlxShape best;
    best.ok = false;
    const dcn::Tuning& tune = dcn::tuning();
    if (tune.gemm_hlx == 0 || taps > 32 || (cs % 32) != 0 || (K % 32) != 0) return best;
    if (group_rows > 0 && (group_rows % XR) != 0) return best;
    const int nk = K / 32, cpt = cs / 32;
    const int mtiles = dcn::ceil_div(M, XR);
    double best_cost = 1e300;
    for (int kg = 1; kg <= 2; ++kg) {
        if (tune.gemm_hlx_kg > 0 && kg != tune.gemm_hlx_kg) continue;
        if (kg == 2 && ((cpt & 1) != 0 || (nk & 1) != 0)) continue;
        const int bn = kg == 1 ? 256 : 128;
        const int ntiles = dcn::ceil_div(cd, bn);
        const int tiles = mtiles * ntiles;
        const int nss = nk / kg;
        for (int s = 1; s <= 8; ++s) {
            if (tune.gemm_hlx_splits > 0 && s != tune.gemm_hlx_splits) continue;
            if (s > nss || (s > 1 && (int64_t)tiles * s * XR * bn * 4 >= ((int64_t)1 << 31))) continue;
            if (s > 1 && nss / s < 6 && tune.gemm_hlx_splits == 0) continue;   // (short ranges: prologue and fix-up dominate)
            const int wgs = tiles * s;
            const int rounds = dcn::ceil_div(wgs, 256);
            // per stage of one workgroup: rows x (columns / 256) x an efficiency factor for the smaller tile (more LDS-DMA bytes
            // per MFMA); a K split adds the parked partial: written once, read s times by the last arriver
            const double eff = kg == 1 ? tune.hlx_cost1 : tune.hlx_cost2;
            const double per_stage = XR * (bn / 256.0) * eff;
            const double stages = (double)nk / s;   // (chunks per workgroup; a KG = 2 stage is two of them on half the columns)
            // (calibrated on profiles/r5a_hlx_sweep.txt / r5b_hlx_sweep.txt: 10-14 us per split launch of the 160 x 128 tile, twice
            // that with the 160 KB partials of the 160 x 256 one, growing slowly with s)
            const double fix = s > 1 ? tune.hlx_split_cost * XR * 0.75 * (bn / 128.0) * (1.0 + 0.1 * s) : 0.0;
            const double cost = rounds * (stages * per_stage + fix);
            if (cost < best_cost) {
                best_cost = cost;
                best.ok = true; best.kg = kg; best.bn = bn; best.mtiles = mtiles; best.ntiles = ntiles; best.nk = nk; best.splits = s;
                best.cost = cost;
                const size_t part = s > 1 ? (size_t)tiles * s * XR * bn * sizeof(float) : 0;
                best.cnt_off = part;
                best.ws_bytes = s > 1 ? part + (size_t)tiles * sizeof(unsigned long long) : 0;
            }
        }
    }
    return best;
}

// Arrival words of the K splits.  They must be clean (no word that carries the id of the launch about to start) -- the
// stream-K scratch of the caller is shared with other launches' partial tiles and would have to be cleared by a fill launch
// in front of every split launch (~4 us of dependent-launch latency on a 40-us kernel).  Instead: one small buffer per
// (device, stream), owned by the library, zeroed once; launches of a stream are ordered, the last arriver of a tile leaves
// its word zero, and the id tag makes whatever an aborted launch left behind count as zero.  Not available while the stream
// is being captured into a graph (no allocation then), for more tiles than the buffer holds, or with DCN_HLX_COUNTERS=0:
// the launch then uses the words behind the partials in the caller's scratch, cleared by a fill launch.
constexpr int kPoolWords = 8192;
struct CounterPool {
    std::mutex mu;
    std::unordered_map<unsigned long long, unsigned long long*> map;
};
CounterPool& counter_pool() {
    static CounterPool* pool = new CounterPool;   // (never destroyed: no hipFree from a static destructor after the runtime is gone;
    return *pool;                                  //  dcn_release_pooled_buffers frees the device buffers on request)
}
unsigned long long* pooled_counters(hipStream_t st, int tiles) {
    if (tiles > kPoolWords || dcn::tuning().hlx_counters == 0) return nullptr;
    // A launch that is being CAPTURED never gets the pooled words, not even those a previous eager launch on the same stream
    // created: the graph would carry the pointer, and a replay on another stream concurrent with eager split launches on this
    // one would share the per-tile words.  Captured launches keep their words in the caller's scratch behind a fill node.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const unsigned long long key = ((unsigned long long)(uintptr_t)st << 8) ^ (unsigned long long)(dev & 0xff);
    CounterPool& pool = counter_pool();
    std::lock_guard<std::mutex> lock(pool.mu);
    auto it = pool.map.find(key);
    if (it != pool.map.end()) return it->second;
    void* ptr = nullptr;
    if (hipMalloc(&ptr, kPoolWords * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    // zeroed ON THE LAUNCHING STREAM: a null-stream memset has no ordering with a non-blocking stream's launches
    if (hipMemsetAsync(ptr, 0, kPoolWords * sizeof(unsigned long long), st) != hipSuccess) { (void)hipFree(ptr); (void)hipGetLastError(); return nullptr; }
    pool.map[key] = (unsigned long long*)ptr;
    return (unsigned long long*)ptr;
}

}  // namespace dcnconv

// Frees the library-owned arrival-word buffers (one 64 KB buffer per (device, stream) that ever ran a K-split launch).  The
// caller has synchronised the streams that used them; the next split launch allocates afresh.
extern "C" void dcn_release_pooled_buffers(void) {
    dcnconv::CounterPool& pool = dcnconv::counter_pool();
    std::lock_guard<std::mutex> lock(pool.mu);
    for (auto& kv : pool.map)
        if (hipFree(kv.second) != hipSuccess) (void)hipGetLastError();
    pool.map.clear();
}

namespace dcnconv {

int launch_gemm_hlx(GemmConv& p, const HlxShape& g, void* workspace, hipStream_t st) {
    if (!g.ok || (g.splits > 1 && !workspace)) return DCN_E_INVALID;
    if (p.relu || p.out_absmax || p.bnb_partial) return DCN_E_UNSUPPORTED;   // (not in gemm_epilogue16, see launch_gemm_hl)
    p.sshift = 0;
    p.div_hw = make_fastdiv(p.hd * p.wd);
    p.div_w = make_fastdiv(p.wd);
    p.div_cs = make_fastdiv(p.cs);
    p.div_kw = make_fastdiv(p.kw);
    p.mtiles = g.mtiles;
    p.ntiles = g.ntiles;
    p.div_nt = make_fastdiv(g.ntiles);
    p.div_nk = make_fastdiv(g.nk);
    p.div_tiles = make_fastdiv(g.mtiles * g.ntiles);
    p.ksplit = g.splits;
    p.hlx_stagger = dcn::tuning().hlx_stagger;
    p.sk_units = 0; p.sk_dp = 0;
    p.sk_partial = nullptr; p.sk_count = nullptr; p.sk_bytes = 0u;
    if (g.splits > 1) {
        p.sk_partial = (float*)workspace;
        p.sk_bytes = (unsigned)g.cnt_off;
        p.sk_id = next_sk_launch_id();
        p.sk_count = pooled_counters(st, g.mtiles * g.ntiles);
        if (!p.sk_count) {
            p.sk_count = (unsigned long long*)((char*)workspace + g.cnt_off);
            if (dcn::fill_bytes_async(p.sk_count, 0, (size_t)g.mtiles * g.ntiles * sizeof(unsigned long long), st) != DCN_OK)
                return DCN_E_LAUNCH;
        }
    }
    const dim3 grid(g.mtiles * g.ntiles * g.splits), block(512);
    if (g.kg == 1) {
        if (p.transposed) hipLaunchKernelGGL((conv_gemm_hlx_kernel<true, 1>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_gemm_hlx_kernel<false, 1>), grid, block, 0, st, p);
    } else {
        if (p.transposed) hipLaunchKernelGGL((conv_gemm_hlx_kernel<true, 2>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_gemm_hlx_kernel<false, 2>), grid, block, 0, st, p);
    }
    return dcn::check_launch();
}

}  // namespace dcnconv
