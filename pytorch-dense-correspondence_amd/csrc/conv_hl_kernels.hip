// Split-fp16 ("f16x3") gather-GEMM on PRE-SPLIT operands for the wide layers (destination >= 256 channels, source channels
// % 32 == 0, stride 1): forward convolution and dgrad of layers 3-4 of the dilated ResNets -- 70 % of the network's FLOPs.
//
// Both operands arrive as "hl32" tensors: per 32-channel chunk one 128-byte line [hi x32 | lo x32] fp16 with
// hi = fp16(s x), lo = fp16(s x - hi) -- the byte size of the fp32 tensor.  The activation / gradient image is written by
// the kernel that PRODUCES the tensor (the batch-norm apply passes, elementwise_kernels.hip; dcn_split_act_hl32 as a
// stand-alone pass), the weight images by one batched launch per call.  The GEMM loop then contains no conversion, no
// VGPR staging and no LDS store: tiles are filled by LDS-DMA (`buffer_load_dwordx4 ... lds`, 64 lanes x 16 B land at
// M0 + 16 lane; rows outside the image / past M carry an out-of-range offset and arrive as zeros) into a lane-linear image
// whose 16-byte slots are XOR-swizzled on the SOURCE side (slot ^ ((row >> 1) & 7): conflict-free ds_read_b128 fragments).
//
// Tile 256 x 256 on 8 wavefronts (2 x 4; wavefront tile 128 x 64 = 8 accumulators of 32 x 32), 32-K stages of 64 KB, two
// stage buffers.  Schedule (tools/hl_gemm_probe2.hip, profiles/r3a_hl_gemm_probe2.txt): the wavefronts of group 1 (waves
// 4-7, one per SIMD) run ONE barrier behind group 0 (waves 0-3); a stage is four phases -- one 64 x 32 quadrant of the
// wavefront tile x both 16-k steps x 3 products = 12 MFMAs = 384 matrix-pipe cycles -- and a phase is
//     LOAD slot (fragment reads of the quadrant, 2 LDS-DMA pieces of the NEXT stage, counted vmcnt) | s_barrier |
//     COMPUTE slot (12 MFMAs under s_setprio 1) | s_barrier
// so that on every SIMD one wavefront computes while the other one loads.
// Half-tiles (16 KB = 16 pieces of 8 rows, two per wavefront) in the order a stage's phases first need them:
//     H0 = A_0, H1 = B_0, H2 = B_1, H3 = A_1;   A_i = tile rows {128 g + 64 i + [0, 64)}, B_j = columns {64 wn + 32 j + [0, 32)}
// (quadrant (i, j) of wavefront (g, wn) reads A_i and B_j only; phase order (0,0) (0,1) (1,1) (1,0)).
// LDS-DMA discipline (every wavefront, LOAD slot of phase p of stage s): issue H_p(s + 1) into the other buffer, then wait
// until at most two half-tiles (4 pieces) are in flight; the barrier that closes the slot publishes what has landed:
//     LOAD(s, 1) reads H2(s)  [landed by the end of LOAD(s, 0)],   LOAD(s, 2) reads H3(s)  [end of LOAD(s, 1)],
//     LOAD(s + 1, 0) reads H0, H1 of s + 1  [end of LOAD(s, 3)];  group 1's slots are one barrier later than group 0's, so
// the latest publication still precedes the earliest read.  WAR: H_p(s + 1) overwrites H_p(s - 1), last read in LOAD(s - 1, 3)
// of group 1, two barriers before LOAD(s, 0) of group 0.  ALL shared memory of the kernel is ONE array (a second
// __shared__ object makes hipcc drain vmcnt in front of every fragment read, cdna_hip_programming.md section 5).
//
// 192-row variant (MT = 3: tile 192 x 256, wavefront tile 96 x 64 = 6 accumulators; hl_shape picks it when it quantises
// better -- 38 400 rows are 150 tiles of 256 rows but 200 of 192 on the 256 CUs).  A stage is THREE phases, phase p = the
// p-th 32-row block of the wavefront tile x both column blocks (12 MFMAs); the B fragments of a stage are read once, in
// LOAD(s, 0).  Pieces per wavefront and stage: A_0, A_1, A_2 (one each, rows 96 g + 32 i + 8 wn + [0, 8), private to the
// group) and Bf, Bs (two each, as B_0 / B_1 above).  Issue order, two stages ahead for what phase 0 needs:
//     LOAD(s, 0): A_1, A_2 of s + 1     LOAD(s, 1): Bf of s + 2     LOAD(s, 2): Bs, A_0 of s + 2
// RAW (a read in LOAD(x) needs every wavefront's wait by the end of the slot before it -- group 1 runs one barrier late):
//     end of LOAD(s, 0): A_1(s) landed = all but the 8 youngest pieces;  LOAD(s, 1): A_2(s), 9;  LOAD(s, 2): B, A_0 of s + 1, 7
// (the last two stages issue less and wait for 8 / 7 / 2 and 1 / 0).  WAR: B(s) and A_0(s) are read in LOAD(s, 0) only, of
// group 1 at the latest -- one barrier before LOAD(s, 1) of group 0, the first slot that overwrites them with stage s + 2;
// A_1 / A_2 of s + 1 replace those of s - 1 (same group, last read in LOAD(s - 1, 2)).
//
// 320-row variant (MT = 5, round 4: tile 320 x 256, wavefront tile 160 x 64 = 10 accumulators = 160 registers; stage =
// 40 KB of A + 32 KB of B, two buffers = 144 KB of the 160 KB LDS).  Why: 38 400 rows x 512 channels (layer 4 at 8 images)
// are 240 such tiles -- ONE data-parallel round on 256 CUs, no stream-K partials at all -- where 192-row tiles need 1.56
// rounds (256 + 144 stream-K'd tiles whose K ranges share no operand in L2: 0.6 GB of fabric reads per launch,
// profiles/r3f_hbm_counters.txt), and the tile moves 0.93x the LDS bytes per FLOP of the 256-row one.  The schedule is the
// 192-row one with FIVE phases per stage (phase p = the p-th 32-row block x both column blocks, 12 MFMAs; B fragments read once
// per stage).  Pieces per wavefront and stage: A_0 .. A_4 (one each) and Bf, Bs (two each); issue order
//     LOAD(s, 0): A_1, A_2 of s + 1     LOAD(s, 1): A_3, A_4 of s + 1     LOAD(s, 2): Bf of s + 2     LOAD(s, 3): Bs of s + 2
//     LOAD(s, 4): A_0 of s + 2
// so every piece has >= 6 phases to land.  RAW (youngest pieces that may still be in flight at the end of the slot):
//     LOAD(s, 0): A_1(s) landed = all but 10;  (s, 1): A_2(s), 11;  (s, 2): A_3(s), 12;  (s, 3): A_4(s), 13;  (s, 4): B, A_0 of
//     s + 1, 9 (the last two stages of a segment issue less: 10 / 11 / 10 / 9 / 4 and 3 / 2 / 1 / 0).  WAR as above: B(s) and
// A_0(s) are read in LOAD(s, 0) only and overwritten from LOAD(s, 2) on; A_i(s + 1) replaces A_i(s - 1), read five slots earlier.
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "conv_hlx.h"
#include "conv_shared.h"
#include "dcn_tuning.h"
#include "f16_split.h"

namespace {

using namespace dcnconv;
using namespace dcnsplit;

typedef __attribute__((address_space(3))) void* hl_lds_ptr;

constexpr int kOob = (int)0x80000000;   // voffset that fails the bounds check of any buffer <= 2 GiB: LDS receives zeros
constexpr int kHlStage = 65536;         // bytes per 32-K stage: [A rows 0..255][B rows 0..255] x 128 B
// per tile height: bytes of the A part of a stage (B follows it: 256 rows x 128 B) and of the whole stage
template <int MT> struct HlStage {
    static constexpr int kA = MT == 5 ? 320 * 128 : 32768, kStage = kA + 32768;
};
constexpr int HLK = 32;

// (a NON-template function: inside a template this builtin breaks the host-side kernel stub with this compiler)
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, void* lds_dst, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (hl_lds_ptr)lds_dst, 16, voffset, soffset, 0, 0);
}

// fp32 [rows][C] (C % 32 == 0) -> hl32 [rows][C / 32][hi x32 | lo x32], scaled by pow2_scale(*absmax).
// One work-item per 8 channels: 32 B in, 16 B of the hi half-line + 16 B of the lo half-line out.
__global__ void __launch_bounds__(256)
split_act_hl32_kernel(const float* __restrict__ src, const float* __restrict__ absmax, u32x4* __restrict__ dst, int64_t n8) {
    const float s = absmax ? pow2_scale(*absmax) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
        h4 ah, al, bh, bl;
        split4(a, s, ah, al);
        split4(b, s, bh, bl);
        const u32x2 h0 = __builtin_bit_cast(u32x2, ah), h1 = __builtin_bit_cast(u32x2, bh);
        const u32x2 l0 = __builtin_bit_cast(u32x2, al), l1 = __builtin_bit_cast(u32x2, bl);
        u32x4* line = dst + (i >> 2) * 8 + (i & 3);
        line[0] = u32x4{h0[0], h0[1], h1[0], h1[1]};
        line[4] = u32x4{l0[0], l0[1], l1[0], l1[1]};
    }
}

// One (tile, K-stage range) of the gather-GEMM.  TR: dgrad (the gather runs over the output gradient with mirrored taps).
// MT: 32-row blocks per wavefront (4: tile 256 x 256, 3: tile 192 x 256).
template <bool TR, int MT>
__device__ __forceinline__ void gemm_segment_hl(const GemmConv& p, unsigned char* lds, int tile, int k0, int k1, int nk,
                                                float* slot) {
    static_assert(MT == 3 || MT == 4 || MT == 5, "wavefront tiles of 96, 128 or 160 rows");
    constexpr int BM = 64 * MT, GR = 32 * MT;   // rows per tile / per wavefront group
    constexpr int NA = MT == 4 ? 4 : MT;        // A pieces per wavefront and stage
    constexpr int kHlStage = HlStage<MT>::kStage, kAB = HlStage<MT>::kA;   // (shadow the 256-row constants)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wv >> 2, wn = wv & 3;
    const int mt = fdiv(tile, p.div_nt), nt = tile - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * 256;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, (int)p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.wh), 0, (int)p.w_bytes, 0x00020000);

    // ---- LDS-DMA pieces of this lane: piece e (0, 1) of half-tile A_i covers tile rows 128 g + 64 i + 8 (2 wn + e) + [0, 8),
    // of B_j rows 64 (wv >> 1) + 32 j + 8 (2 (wv & 1) + e) + [0, 8); lane l fetches row (l >> 3), PHYSICAL slot l & 7 =
    // logical 16-byte slot (l & 7) ^ ((row >> 1) & 7) of the row's 128-byte line
    const int l8 = lane >> 3, ls = lane & 7;
    const int cs4 = p.cs * 4;              // bytes per pixel of the activation image
    // A piece a of this wavefront: MT = 4: (i, e) = (a >> 1, a & 1) as above; MT = 3: block i = a, rows 96 g + 32 a + 8 wn + [0, 8)
    auto a_piece_row = [&](int a) { return MT == 4 ? grp * 128 + (a >> 1) * 64 + (2 * wn + (a & 1)) * 8 : grp * GR + a * 32 + wn * 8; };
    int by[NA], bx[NA], rowoff[NA], voa[NA], vob[2][2];
    // MT = 5 (160 accumulator registers): instead of the position (by, bx) and the current offset (voa) of every piece, ONE
    // word per piece with a validity bit per filter tap (taps <= 32: hl_shape) -- the offset of a piece is formed when it is
    // issued: rowoff +- the tap's (wave-uniform) displacement, or the out-of-range offset.  10 registers instead of 20.
    unsigned vmask[NA];
    int cur_delta = 0, cur_bit = 0;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int row = a_piece_row(a) + l8;
        const int sl = ls ^ ((row >> 1) & 7);
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int img = fdiv(mm, p.div_hw), rem = mm - img * p.div_hw.d;
        const int y = fdiv(rem, p.div_w), x = rem - y * p.div_w.d;
        by[a] = ok ? (TR ? y + p.pad : y - p.pad) : -(1 << 28);   // (stride 1; rows past M: never inside the image)
        bx[a] = TR ? x + p.pad : x - p.pad;
        rowoff[a] = ok ? (img * p.hs * p.ws + by[a] * p.ws + bx[a]) * cs4 + sl * 16 : 0;
        vmask[a] = 0u;
        if constexpr (MT == 5) {
            const int ntap = p.kh * p.kw;
            for (int tap = 0; tap < ntap; ++tap) {
                const int r = fdiv(tap, p.div_kw), s = tap - r * p.kw;
                const int dy = r * p.dil, dx = s * p.dil;
                bool v;
                if (TR) {
                    const int ny = by[a] - dy, nx = bx[a] - dx;
                    v = ((ny | nx) >= 0) & (ny < p.hs) & (nx < p.ws);
                } else {
                    v = ((unsigned)(by[a] + dy) < (unsigned)p.hs) & ((unsigned)(bx[a] + dx) < (unsigned)p.ws);
                }
                vmask[a] |= (v ? 1u : 0u) << tap;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = (wv >> 1) * 64 + j * 32 + (2 * (wv & 1) + e) * 8 + l8;
            const int sl = ls ^ ((row >> 1) & 7);
            const int n = n0 + row;
            vob[j][e] = n < p.cd ? n * (nk * 128) + sl * 16 : kOob;
        }
    // K traversal: channel-chunk groups outermost (kcg chunks = up to 128 channels), then the filter taps, then the chunks
    // of the group: the same input pixels come back for the next tap after kcg stages, while they are still in this XCD's L2
    const int cpt = p.cs / HLK;
    const int kcg = (cpt & 3) == 0 ? 4 : ((cpt & 1) == 0 ? 2 : 1);
    const int taps = p.kh * p.kw;
    int u_grp, u_tap, u_c, u_kt = k0;      // the stage whose LDS-DMA is issued next
    auto set_tap = [&](int tap) {
        const int r = fdiv(tap, p.div_kw), s = tap - r * p.kw;
        const int dy = r * p.dil, dx = s * p.dil;
        const int delta = (dy * p.ws + dx) * cs4;
        if constexpr (MT == 5) {
            cur_delta = TR ? -delta : delta;
            cur_bit = tap;
            return;
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            bool ok;
            if (TR) {
                const int ny = by[a] - dy, nx = bx[a] - dx;
                ok = ((ny | nx) >= 0) & (ny < p.hs) & (nx < p.ws);
            } else {
                ok = ((unsigned)(by[a] + dy) < (unsigned)p.hs) & ((unsigned)(bx[a] + dx) < (unsigned)p.ws);
            }
            voa[a] = ok ? (TR ? rowoff[a] - delta : rowoff[a] + delta) : kOob;
        }
    };
    {
        const int per_grp = taps * kcg;
        u_grp = k0 / per_grp;
        const int rem = k0 - u_grp * per_grp;
        u_tap = rem / kcg;
        u_c = rem - u_tap * kcg;
        set_tap(u_tap);
    }
    auto advance = [&]() {                 // steps to the next stage of the segment (stays on the last one)
        if (u_kt + 1 < k1) {
            ++u_kt;
            if (++u_c == kcg) {
                u_c = 0;
                if (++u_tap == taps) { u_tap = 0; ++u_grp; }
                set_tap(u_tap);
            }
        }
    };
    // pieces of the stage the traversal state points at, into stage buffer `buf`
    auto issue_a = [&](int buf, int a) {
        const int vo = MT == 5 ? (((vmask[a] >> cur_bit) & 1u) ? rowoff[a] + cur_delta : kOob) : voa[a];
        glds16(rs_a, lds + buf * kHlStage + a_piece_row(a) * 128, vo, (u_grp * kcg + u_c) * 128);
    };
    auto issue_b = [&](int buf, int j) {
        unsigned char* d = lds + buf * kHlStage + kAB + ((wv >> 1) * 64 + j * 32 + 2 * (wv & 1) * 8) * 128;
        const int soff = (u_tap * cpt + u_grp * kcg + u_c) * 128;
        glds16(rs_b, d, vob[j][0], soff);
        glds16(rs_b, d + 1024, vob[j][1], soff);
    };
    // MT = 4: half-tile h (0: A_0, 1: B_0, 2: B_1, 3: A_1)
    auto issue_half = [&](int buf, int h) {
        if (h == 0 || h == 3) {
            const int i = h == 0 ? 0 : 1;
            issue_a(buf, 2 * i);
            issue_a(buf, 2 * i + 1);
        } else {
            issue_b(buf, h == 1 ? 0 : 1);
        }
    };

    // ---- fragments (round 5: v_mfma_f32_16x16x32_f16 tiles -- the same two-group schedule sustains 10-15 % more on that shape,
    // tools/hl_gemm_probe3.hip / profiles/r5g_hl_probe3.txt: half the accumulator-register traffic per FLOP under the power
    // limit; the LDS image, the DMA schedule and the NUMBER of fragment reads are unchanged): lane (fr = lane & 15,
    // fq = lane >> 4) holds k-octet fq of the stage's 32 k of row fr of a 16-row block; plane 0 = hi, 1 = lo.  The source-side
    // XOR swizzle (slot ^ ((row >> 1) & 7)) is conflict-free for the ds_read_b128 lane groups of this layout too (rows 0-3 /
    // 12-15 of one octet with rows 4-11 of the next: 16 distinct (row parity, slot) pairs).
    const int fr = lane & 15, fq = lane >> 4, swz = (fr >> 1) & 7;
    int foff[2];   // [plane]: byte offset of the lane's slot inside its 16-row block
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) foff[pl] = fr * 128 + ((pl * 4 + fq) ^ swz) * 16;
    const int a_row = (grp * GR) * 128, b_row = kAB + (wn * 64) * 128;

    f32x4_t acc[2 * MT][4];   // [row block of 16][column block of 16]: rows GR g + 16 tb, columns 64 wn + 16 cb
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    using T = std::true_type;
    using F = std::false_type;
    // raw s_barrier: no fence, LDS-DMA stays in flight across it
    auto bar = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
  if constexpr (MT == 4) {
    h8 fa[4][2], fb[2][2];   // A: [row block of the quadrant's 64 rows][plane], B: [column block of its 32 columns][plane]
    auto read_a = [&](int buf, int i) {
        const unsigned char* st = lds + buf * kHlStage + a_row + i * 8192;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fa[rb][pl] = *reinterpret_cast<const h8*>(st + rb * 2048 + foff[pl]);
    };
    auto read_b = [&](int buf, int j) {
        const unsigned char* st = lds + buf * kHlStage + b_row + j * 4096;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fb[cb][pl] = *reinterpret_cast<const h8*>(st + cb * 2048 + foff[pl]);
    };
    // product-type outermost, the eight accumulators of the quadrant in between; the small cross terms before hi * hi
    auto mfma_quadrant = [&](int i, int j) {
        if (p.hl_setprio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    acc[4 * i + rb][2 * j + cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pt == 0 ? fa[rb][1] : fa[rb][0],
                                                                                         pt == 1 ? fb[cb][1] : fb[cb][0],
                                                                                         acc[4 * i + rb][2 * j + cb], 0, 0, 0);
        if (p.hl_setprio) __builtin_amdgcn_s_setprio(0);
    };
    // one phase of stage s (LDS buffer s & 1).  MORE: stage s + 1 exists -- issue its half-tile p and leave two half-tiles
    // in flight; otherwise drain what the next phases read.
    auto phase = [&](auto more_tag, int buf, int ph) {
        constexpr bool MORE = decltype(more_tag)::value;
        const int qi = ph >> 1, qj = (ph == 1 || ph == 2) ? 1 : 0;
        if (ph == 0) { read_b(buf, 0); __builtin_amdgcn_sched_barrier(0); read_a(buf, 0); }
        else if (ph == 1) read_b(buf, 1);
        else if (ph == 2) read_a(buf, 1);
        else read_b(buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MORE) {
            issue_half(buf ^ 1, ph);
            if (ph == 3) advance();   // (the traversal state now points at stage s + 2)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MORE) DCN_WAIT_VMCNT(4);
        else if (ph == 0) DCN_WAIT_VMCNT(2);
        else if (ph == 1) DCN_WAIT_VMCNT(0);
        DCN_WAIT_LGKMCNT0();
        bar();
        mfma_quadrant(qi, qj);
        bar();
    };
    issue_half(0, 0);
    issue_half(0, 1);
    issue_half(0, 2);
    issue_half(0, 3);
    advance();
    __builtin_amdgcn_sched_barrier(0);
    DCN_WAIT_VMCNT(4);
    bar();
    if (grp == 1) bar();
    int buf = 0;
    for (int s = k0; s + 1 < k1; ++s) {
        phase(T{}, buf, 0);
        phase(T{}, buf, 1);
        phase(T{}, buf, 2);
        phase(T{}, buf, 3);
        buf ^= 1;
    }
    phase(F{}, buf, 0);
    phase(F{}, buf, 1);
    phase(F{}, buf, 2);
    phase(F{}, buf, 3);
  } else {
    h8 fa[2][2], fb[4][2];   // A: [16-row block][plane] of the phase's 32-row block, B: [column block of 16][plane] of the stage
    auto read_a = [&](int buf, int tm) {
        const unsigned char* st = lds + buf * kHlStage + a_row + tm * 4096;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fa[rb][pl] = *reinterpret_cast<const h8*>(st + rb * 2048 + foff[pl]);
    };
    auto read_b = [&](int buf) {
        const unsigned char* st = lds + buf * kHlStage + b_row;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fb[cb][pl] = *reinterpret_cast<const h8*>(st + cb * 2048 + foff[pl]);
    };
    auto mfma_block = [&](int tm) {
        if (p.hl_setprio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[2 * tm + rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pt == 0 ? fa[rb][1] : fa[rb][0],
                                                                                  pt == 1 ? fb[cb][1] : fb[cb][0], acc[2 * tm + rb][cb], 0, 0, 0);
        if (p.hl_setprio) __builtin_amdgcn_s_setprio(0);
    };
    // one phase of stage s (LDS buffer s & 1).  AHEAD: how many later stages of the segment exist (2 = at least two).
    auto phase = [&](auto ahead_tag, int buf, int ph) {
        constexpr int AHEAD = decltype(ahead_tag)::value;
        if (ph == 0) { read_b(buf); __builtin_amdgcn_sched_barrier(0); }
        read_a(buf, ph);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MT == 3) {
            if (ph == 0 && AHEAD >= 1) {
                issue_a(buf ^ 1, 1);
                issue_a(buf ^ 1, 2);
                advance();               // (the traversal state now points at stage s + 2)
            }
            if (ph == 1 && AHEAD == 2) issue_b(buf, 0);
            if (ph == 2 && AHEAD == 2) { issue_b(buf, 1); issue_a(buf, 0); }
        } else {
            if (ph == 0 && AHEAD >= 1) { issue_a(buf ^ 1, 1); issue_a(buf ^ 1, 2); }
            if (ph == 1 && AHEAD >= 1) {
                issue_a(buf ^ 1, 3);
                issue_a(buf ^ 1, 4);
                advance();               // (the traversal state now points at stage s + 2)
            }
            if (ph == 2 && AHEAD == 2) issue_b(buf, 0);
            if (ph == 3 && AHEAD == 2) issue_b(buf, 1);
            if (ph == 4 && AHEAD == 2) issue_a(buf, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MT == 3) {
            if (AHEAD == 2) {
                if (ph == 0) DCN_WAIT_VMCNT(8);
                else if (ph == 1) DCN_WAIT_VMCNT(9);
                else DCN_WAIT_VMCNT(7);
            } else if (AHEAD == 1) {
                if (ph == 0) DCN_WAIT_VMCNT(8);
                else if (ph == 1) DCN_WAIT_VMCNT(7);
                else DCN_WAIT_VMCNT(2);
            } else {
                if (ph == 0) DCN_WAIT_VMCNT(1);
                else if (ph == 1) DCN_WAIT_VMCNT(0);
            }
        } else {   // (the table of the file header)
            if (AHEAD == 2) {
                if (ph == 0) DCN_WAIT_VMCNT(10);
                else if (ph == 1) DCN_WAIT_VMCNT(11);
                else if (ph == 2) DCN_WAIT_VMCNT(12);
                else if (ph == 3) DCN_WAIT_VMCNT(13);
                else DCN_WAIT_VMCNT(9);
            } else if (AHEAD == 1) {
                if (ph == 0) DCN_WAIT_VMCNT(10);
                else if (ph == 1) DCN_WAIT_VMCNT(11);
                else if (ph == 2) DCN_WAIT_VMCNT(10);
                else if (ph == 3) DCN_WAIT_VMCNT(9);
                else DCN_WAIT_VMCNT(4);
            } else {
                if (ph == 0) DCN_WAIT_VMCNT(3);
                else if (ph == 1) DCN_WAIT_VMCNT(2);
                else if (ph == 2) DCN_WAIT_VMCNT(1);
                else if (ph == 3) DCN_WAIT_VMCNT(0);
            }
        }
        DCN_WAIT_LGKMCNT0();
        bar();
        mfma_block(ph);
        bar();
    };
    using A0 = std::integral_constant<int, 0>;
    using A1 = std::integral_constant<int, 1>;
    using A2 = std::integral_constant<int, 2>;
    auto stage = [&](auto tag, int buf) {
#pragma unroll
        for (int ph = 0; ph < MT; ++ph) phase(tag, buf, ph);
    };

    // prologue: stage k0 whole -- what phase 0 reads first --, then B and A_0 of stage k0 + 1: the order of the steady state
    issue_b(0, 0);
    issue_b(0, 1);
    issue_a(0, 0);
#pragma unroll
    for (int a = 1; a < MT; ++a) issue_a(0, a);
    if (k0 + 1 < k1) {
        advance();
        issue_b(1, 0);
        issue_b(1, 1);
        issue_a(1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MT == 3) DCN_WAIT_VMCNT(7); else DCN_WAIT_VMCNT(9);
    } else {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MT == 3) DCN_WAIT_VMCNT(2); else DCN_WAIT_VMCNT(4);
    }
    bar();
    if (grp == 1) bar();
    int buf = 0;
    for (int s = k0; s + 2 < k1; ++s) {
        stage(A2{}, buf);
        buf ^= 1;
    }
    if (k0 + 1 < k1) {
        stage(A1{}, buf);
        buf ^= 1;
    }
    stage(A0{}, buf);
  }
    if (grp == 0) bar();
    __syncthreads();

    const float sa = p.a_absmax ? pow2_scale(*p.a_absmax) : 1.f;   // (the scale the producer of the hl32 image applied)
    const float inv = p.b_inv_scale / sa;
#pragma unroll
    for (int tb = 0; tb < 2 * MT; ++tb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[tb][cb] *= inv;
    // ONE instance of the epilogue for whole tiles and completed stream-K tiles alike (two inlined copies cost the 256-row
    // stream-K instantiation 416 bytes of scratch per lane)
    bool complete = k0 == 0 && k1 == nk;
    int rel = 0;
    if (!complete) {
        // ---- stream-K partial: parked device-coherently, completed by the last contributor inside the launch
        // (conv_f16_kernels.hip, gemm_segment_f16: same protocol; slot layout [wavefront][row block][column block][lane] float4)
        constexpr int TM = MT;
        const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(p.sk_partial, 0, (int)p.sk_bytes, 0x00020000);
        constexpr int kSc1 = 16;
        constexpr int kPieces = 2 * MT * 4;   // float4 pieces per lane
        const int lane_off = (wv * (kPieces * 64) + lane) * 16;
        {
            const int so = (int)((slot - p.sk_partial) * 4);
#pragma unroll
            for (int pc = 0; pc < kPieces; ++pc)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[pc >> 2][pc & 3]), rs_p, lane_off + pc * 1024, so, kSc1);
        }
        rel = tile - p.sk_dp;
        const int ua = rel * nk, ub = ua + nk - 1;
        const int ga = ua / p.sk_units, gb = ub / p.sk_units;
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this work-item's partial has been written through
        __syncthreads();
        int* s_last = reinterpret_cast<int*>(lds + 2 * kHlStage - 16);   // (inside the one LDS array: see the header)
        if (tid == 0) *s_last = sk_arrive_is_last(p.sk_count + rel, p.sk_id, gb - ga + 1) ? 1 : 0;
        __syncthreads();
        complete = *s_last != 0;
        if (complete) {
#pragma unroll
            for (int tb = 0; tb < 2 * MT; ++tb)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[tb][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            // (each contributor's partial in batches of loads of 16 B issued back to back, then added: left to itself the
            // compiler waits for every 4 loads -- device-coherent round trips on the critical path of the launch)
            constexpr int kBatch = TM == 4 ? 16 : (TM == 3 ? 12 : 10);
            static_assert(kPieces % kBatch == 0, "whole batches");
            for (int g = ga; g <= gb; ++g) {               // fixed order, own partial included: deterministic
                const int first_tile = (g * p.sk_units) / nk;
                const int so = (2 * g + (first_tile == rel ? 0 : 1)) * (BM * 256 * 4);
#pragma unroll
                for (int b0 = 0; b0 < kPieces; b0 += kBatch) {
                    u32x4 t[kBatch];
#pragma unroll
                    for (int j = 0; j < kBatch; ++j) t[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, lane_off + (b0 + j) * 1024, so, kSc1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < kBatch; ++j) acc[(b0 + j) >> 2][(b0 + j) & 3] += __builtin_bit_cast(f32x4_t, t[j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();                               // (s_last has been read by everyone: the epilogue reuses the array)
        }
    }
    if (complete) {
        gemm_epilogue16<2 * MT, 4, BM, 256>(p, acc, mt, nt, grp, grp * GR, wn * 64, reinterpret_cast<float*>(lds));
        if (!(k0 == 0 && k1 == nk) && tid == 0) atomicExch(p.sk_count + rel, 0ull);
    }
}

template <bool TR, bool SK, int MT>
__global__ void __launch_bounds__(512, 1)
conv_gemm_hl_kernel(GemmConv p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * HlStage<MT>::kStage];
    const int nk = p.K / HLK;
    if (!SK) {
        gemm_segment_hl<TR, MT>(p, lds, xcd_remap(blockIdx.x, p.mtiles * p.ntiles), 0, nk, nk, nullptr);
    } else {
        // hybrid schedule (as conv_gemm_f16_kernel): whole rounds of tiles data-parallel, then ONE stream-K pass that splits
        // the K stages of the leftover tiles evenly.  ONE call site for both kinds of work item: two inlined copies of the
        // segment (round 5) cost the 256-row instantiation 416 bytes of scratch per lane in its epilogues.
        const int g = xcd_remap(blockIdx.x, gridDim.x);
        int tile = g;
        int u = p.sk_dp * nk + g * p.sk_units;
        const int total = p.mtiles * p.ntiles * nk;
        const int u_end = min(total, u + p.sk_units);
        int parked = 0;                                    // partials this workgroup has parked (at most two: slots 2 g, 2 g + 1)
        for (;;) {
            int t, k0, k1;
            float* slot = nullptr;
            if (tile < p.sk_dp) {
                t = tile; k0 = 0; k1 = nk;
                tile += gridDim.x;
            } else if (u < u_end) {
                t = fdiv(u, p.div_nk);
                k0 = u - t * nk;
                k1 = min(nk, k0 + (u_end - u));
                slot = p.sk_partial + (int64_t)(2 * g + parked) * (64 * MT * 256);
                u += k1 - k0;
                parked = 1;
            } else {
                break;
            }
            gemm_segment_hl<TR, MT>(p, lds, t, k0, k1, nk, slot);
            __syncthreads();
        }
    }
}

struct HlShape {
    int rows, mtiles, ntiles, nk, sk_wgs, sk_units, sk_dp;
    bool sk;
    size_t ws_bytes, sk_count_off;
    bool ok = true;   // false: no tile height divides the rows of a statistics group (the caller takes another kernel)
    HlxShape x;       // rows == 160: the launch goes to the small-tile kernel (conv_hlx_kernels.hip) with this shape
};
HlShape hl_shape_rows(int M, int cd, int K, int rows) {
    HlShape g;
    g.rows = rows;
    g.mtiles = dcn::ceil_div(M, rows);
    g.ntiles = dcn::ceil_div(cd, 256);
    g.nk = K / HLK;
    const int tiles = g.mtiles * g.ntiles;
    const double rounds = tiles / 256.0;
    const double waste = 1.0 - rounds / (double)(int)(rounds + 0.999999);
    const double gain = ((double)(int)(rounds + 0.999999) - rounds) * g.nk;   // stage times stream-K can save
    const dcn::Tuning& tune = dcn::tuning();
    // (a 256 x 256 partial is 256 KB -- parking it, reading the other contributors' and the second prologue cost a workgroup
    // about 50 stage times: stream-K pays from there.  Measured on ResNet50-8s at 1280 x 960, profiles/r3h_convbench_r50.txt)
    g.sk = tiles < 8 * 256 && waste > 0.08 && g.nk >= 8 && gain >= std::max(tune.gemm_sk_min_gain, 50.0);
    int wgs = 256;                                                              // one 512-work-item workgroup per CU
    if (tune.gemm_sk >= 0) {
        if (tune.gemm_sk == 0) g.sk = false;
        if (tune.gemm_sk > 1) { g.sk = g.nk >= 2; wgs = tune.gemm_sk; }
    }
    g.sk_wgs = 0; g.sk_units = 0; g.sk_dp = 0; g.ws_bytes = 0; g.sk_count_off = 0;
    if (g.sk) {
        if ((int64_t)tiles * g.nk >= ((int64_t)1 << 30)) { g.sk = false; return g; }
        g.sk_dp = tiles / wgs * wgs;
        const int64_t total = (int64_t)(tiles - g.sk_dp) * g.nk;
        if (total == 0) { g.sk = false; g.sk_dp = 0; return g; }
        // Only the leftover tiles behind at least one whole data-parallel round are split: a 256 x 256 partial is 256 KB, and
        // splitting EVERY tile of a sub-round launch parks more bytes than the launch reads (measured at N = 8 on the
        // 256-channel layers, 150 tiles: 212 us with stream-K -- 172 MB of partials written and read back, matrix pipe 25 %
        // busy -- against 154 us data-parallel on 150 CUs, profiles/r3d_hl_layer3_sk*.txt)
        if (g.sk_dp == 0 && tune.gemm_sk <= 1) { g.sk = false; return g; }
        if (g.sk_dp == 0 && wgs > total / 2) wgs = (int)(total / 2) > 0 ? (int)(total / 2) : 1;
        g.sk_units = (int)((total + wgs - 1) / wgs);
        g.sk_wgs = g.sk_dp > 0 ? wgs : (int)((total + g.sk_units - 1) / g.sk_units);
        g.ws_bytes = (size_t)2 * g.sk_wgs * rows * 256 * sizeof(float);
        g.sk_count_off = g.ws_bytes;
        g.ws_bytes += (size_t)(tiles - g.sk_dp) * sizeof(unsigned long long);
        if (g.sk_count_off >= ((size_t)1 << 31)) { g.sk = false; g.sk_wgs = 0; g.sk_units = 0; g.sk_dp = 0; g.ws_bytes = 0; g.sk_count_off = 0; }
    }
    return g;
}

// Tile height: 256 rows unless 192 quantises better on the 256 CUs.  Cost = rounds of tiles x rows, a stream-K'd launch
// counting its fractional rounds plus what the fix-up costs a workgroup (~50 stage times with 256 KB partials, ~30 with
// 192 KB); a 192-row tile moves 1.17x the LDS bytes per FLOP and measures ~8 % slower per row.  Calibrated on the N = 8 layer
// shapes (profiles/r3q_hl_rows_per_layer.txt): 256 -> 256 3x3, 150 / 200 tiles: 160 -> 139 us; 512 -> 512 3x3, 300 / 400
// tiles, both stream-K'd: 491 -> 450 us.  group_rows: rows per statistics group of the forward epilogue (0: none) -- a
// group is made of whole tiles.
// taps: filter taps of the convolution (the 320-row kernel keeps one validity bit per tap in a 32-bit word); cs: source channels.
// Round 5: the small tiles of conv_hlx_kernels.hip (160 x 256, 160 x 128 with two K groups, K split over workgroups) compete
// in the same model -- cost x stages of a workgroup, so that a K split counts -- and win where the big tiles leave CUs idle.
HlShape hl_shape(int M, int cd, int K, int group_rows, int taps, int cs) {
    HlShape g4 = hl_shape_rows(M, cd, K, 256);
    const dcn::Tuning& tune = dcn::tuning();
    const int force = tune.gemm_hl_rows;
    const bool ok3 = group_rows <= 0 || (group_rows % 192) == 0, ok4 = group_rows <= 0 || (group_rows % 256) == 0;
    const bool ok5 = (group_rows <= 0 || (group_rows % 320) == 0) && taps <= 32 && force != -320;
    g4.ok = ok4;
    // 320-row tiles (round 4) run data-parallel only: they are for the launches they fit in whole rounds (38 400 x 512 outputs
    // = 240 tiles = ONE round without any stream-K partial)
    auto shape5 = [&]() {
        HlShape g5 = hl_shape_rows(M, cd, K, 320);
        g5.sk = false; g5.sk_wgs = 0; g5.sk_units = 0; g5.sk_dp = 0; g5.ws_bytes = 0; g5.sk_count_off = 0;
        return g5;
    };
    auto cost = [](const HlShape& g) {
        const double rounds = g.mtiles * g.ntiles / 256.0;
        const double per_row = g.rows == 192 ? 1.08 : 1.0, fixup = g.rows == 192 ? 30.0 : 50.0;
        return (g.sk ? rounds + fixup / g.nk : (double)(int)(rounds + 0.999999)) * g.rows * per_row;
    };
    HlShape best = g4;
    double best_cost = ok4 ? cost(g4) : 1e300;
    if (force == 320 || force == 192 || force == 256) {   // a forced height: that one or nothing (never a tile that straddles groups)
        if (force == 320) { best = shape5(); best.ok = ok5; }
        else if (force == 192) { best = hl_shape_rows(M, cd, K, 192); best.ok = ok3; }
        return best;
    }
    if (ok3) {
        const HlShape g3 = hl_shape_rows(M, cd, K, 192);
        if (cost(g3) < 0.97 * best_cost) { best = g3; best_cost = cost(g3); }
    }
    if (ok5) {   // (DCN_GEMM_HL_ROWS=-320: never 320 -- the round-3 choice, for A/B runs)
        const HlShape g5 = shape5();
        if (cost(g5) < 0.97 * best_cost) { best = g5; best_cost = cost(g5); }
    }
    best.ok = best_cost < 1e299;
    const HlxShape x = hlx_shape(M, cd, K, group_rows, taps, cs);
    const bool x_forced = force == 160 || tune.gemm_hlx_kg > 0 || tune.gemm_hlx_splits > 0;
    if (x.ok && (x_forced || !best.ok || x.cost < 0.97 * best_cost * best.nk)) {
        HlShape gx = best;
        gx.rows = kHlxRows; gx.mtiles = x.mtiles; gx.ntiles = x.ntiles; gx.nk = x.nk;
        gx.sk = false; gx.sk_wgs = 0; gx.sk_units = 0; gx.sk_dp = 0;
        gx.ws_bytes = x.ws_bytes; gx.sk_count_off = x.cnt_off;
        gx.ok = true;
        gx.x = x;
        return gx;
    }
    if (force == 160) best.ok = false;
    return best;
}

bool valid_desc_hl(const dcn_conv_desc* c) {
    return c && c->n > 0 && c->hin > 0 && c->win > 0 && c->cin > 0 && c->hout > 0 && c->wout > 0 && c->cout > 0 && c->kh > 0 &&
           c->kw > 0 && c->stride > 0 && c->dil > 0 && c->pad >= 0 && c->ldc >= c->cout;
}

template <int MT>
void launch_gemm_hl_rows(const GemmConv& p, bool sk, dim3 grid, hipStream_t st) {
    const dim3 block(512);
    if constexpr (MT != 5) {   // (320-row tiles: data-parallel launches only, hl_shape)
        if (sk) {
            if (p.transposed) hipLaunchKernelGGL((conv_gemm_hl_kernel<true, true, MT>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((conv_gemm_hl_kernel<false, true, MT>), grid, block, 0, st, p);
            return;
        }
    }
    {
        if (p.transposed) hipLaunchKernelGGL((conv_gemm_hl_kernel<true, false, MT>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv_gemm_hl_kernel<false, false, MT>), grid, block, 0, st, p);
    }
}

int launch_gemm_hl(GemmConv& p, int group_rows, void* workspace, hipStream_t st) {
    // (gemm_epilogue16 -- this kernel's and the small-tile kernel's epilogue -- has no ReLU, no abs-max output and no
    // backward-statistics partials: a caller that asks for one of them must not get a silently different result)
    if (p.relu || p.out_absmax || p.bnb_partial) return DCN_E_UNSUPPORTED;
    p.sshift = 0;
    p.div_hw = make_fastdiv(p.hd * p.wd);
    p.div_w = make_fastdiv(p.wd);
    p.div_cs = make_fastdiv(p.cs);
    p.div_kw = make_fastdiv(p.kw);
    const HlShape g = hl_shape(p.M, p.cd, p.K, group_rows, p.kh * p.kw, p.cs);
    if (!g.ok) return DCN_E_UNSUPPORTED;
    p.hl_setprio = dcn::tuning().hl_setprio;
    if (g.rows == kHlxRows) {
        if (g.x.splits > 1 && !workspace) {   // no scratch (the tuning changed after the plan was sized): the same tiles, unsplit
            HlxShape x1 = g.x;
            x1.splits = 1; x1.ws_bytes = 0; x1.cnt_off = 0;
            return launch_gemm_hlx(p, x1, nullptr, st);
        }
        return launch_gemm_hlx(p, g.x, workspace, st);
    }
    const bool sk = g.sk && workspace != nullptr;
    p.mtiles = g.mtiles;
    p.ntiles = g.ntiles;
    p.div_nt = make_fastdiv(g.ntiles);
    p.div_nk = make_fastdiv(g.nk);
    p.sk_units = sk ? g.sk_units : 0;
    p.sk_dp = sk ? g.sk_dp : 0;
    p.sk_partial = sk ? (float*)workspace : nullptr;
    p.sk_count = sk ? (unsigned long long*)((char*)workspace + g.sk_count_off) : nullptr;
    p.sk_bytes = sk ? (unsigned)g.sk_count_off : 0u;
    if (sk) {
        p.sk_id = next_sk_launch_id();
        if (dcn::fill_bytes_async(p.sk_count, 0, (size_t)(g.mtiles * g.ntiles - g.sk_dp) * sizeof(unsigned long long), st) != DCN_OK)
            return DCN_E_LAUNCH;   // (arrival words cleared in front of every launch: see launch_gemm_f16)
    }
    const dim3 grid(sk ? g.sk_wgs : g.mtiles * g.ntiles);
    if (g.rows == 192) launch_gemm_hl_rows<3>(p, sk, grid, st);
    else if (g.rows == 320) launch_gemm_hl_rows<5>(p, false, grid, st);
    else launch_gemm_hl_rows<4>(p, sk, grid, st);
    return dcn::check_launch();
}

}  // namespace

static HlShape hl_shape_of(const dcn_conv_desc* c, int dgrad) {
    if (dgrad) return hl_shape(c->n * c->hin * c->win, c->cin, c->kh * c->kw * c->ldc, 0, c->kh * c->kw, c->ldc);
    return hl_shape(c->n * c->hout * c->wout, c->cout, c->kh * c->kw * c->cin, c->group_rows, c->kh * c->kw, c->cin);
}

// What the kernel can compute at all: whole 32-channel source chunks, stride 1, tensors addressable through 2 GiB buffer
// resources, statistics groups made of whole tiles of the height the tuning of the moment selects.
static bool hl_supported(const dcn_conv_desc* c, int dgrad) {
    if (!valid_desc_hl(c)) return false;
    if (c->stride != 1 || (c->ldc % 4) != 0) return false;
    const int cs = dgrad ? c->ldc : c->cin, cd = dgrad ? c->cin : c->cout;
    const int64_t M = (int64_t)c->n * (dgrad ? c->hin * c->win : c->hout * c->wout);
    const int64_t K = (int64_t)c->kh * c->kw * cs;
    if ((cs % HLK) != 0 || (dgrad && c->ldc != c->cout) || (cd % 4) != 0 || M >= ((int64_t)1 << 30)) return false;
    if (dgrad && (c->hin != c->hout || c->win != c->wout)) return false;       // (stride-1 "same" convolutions)
    const int64_t src_bytes = (int64_t)c->n * (dgrad ? c->hout * c->wout : c->hin * c->win) * cs * 4;
    const int64_t w_bytes = (int64_t)cd * K * 4;
    if (!(src_bytes <= ((int64_t)1 << 31) - 1 && w_bytes <= ((int64_t)1 << 31) - 1)) return false;
    return hl_shape_of(c, dgrad).ok;   // (false: every tile height the tuning allows would straddle two statistics groups)
}

// Which convolutions the engine sends down the hl32 path: supported, and wide / deep / tall enough for the 256 x 256 tile
// and its software pipeline to pay (destination >= 256 channels, K >= 512, M >= 4096).
extern "C" int dcn_conv_hl_eligible(const dcn_conv_desc* c, int dgrad) {
    if (!hl_supported(c, dgrad) || dcn::tuning().gemm_hl == 0) return 0;
    if (dcn::tuning().gemm_hl == 2) return 1;   // (tests: every supported convolution)
    const int cs = dgrad ? c->ldc : c->cin, cd = dgrad ? c->cin : c->cout;
    const int64_t M = (int64_t)c->n * (dgrad ? c->hin * c->win : c->hout * c->wout);
    const int64_t K = (int64_t)c->kh * c->kw * cs;
    // (round 6: K >= 128 instead of 512 -- the 1 x 1 convolutions fed by 128 - 511 channels: at eight images layer3 / layer4
    // downsample forward 34.5 -> 26.9 / 67.5 -> 50.3 us, ResNet50-8s conv3 1 x 1 256 -> 1024 255.6 -> 195.1 us, profiles/r6e_*, r6f_*)
    if (!(cd >= 128 && K >= std::min(512, dcn::tuning().hl_min_k) && M >= 4096)) return 0;
    const HlShape chosen = hl_shape_of(c, dgrad);
    // the small tiles (round 5): chosen by the cost model where the big ones leave CUs idle -- sub-round launches (B = 1, the
    // two-call pattern), layer 3 at 8 images; 128-channel destinations only when asked for (DCN_GEMM_HLX_NARROW)
    // (128-channel destinations fed by >= 256 source channels -- the dgrad of layer3.0.conv1: 35-66 us against 53-97 on the
    // fp32-operand kernel at 1-8 images, profiles/r5a_hlx_sweep.txt -- take it too: the gradient image exists anyway)
    if (chosen.rows == kHlxRows) return (cd >= 256 || cs >= 256 || dcn::tuning().gemm_hlx_narrow != 0) ? 1 : 0;
    if (cd < 256) return 0;
    // between one and one and a half rounds of tiles the second round leaves most of the chip idle, and splitting its tiles
    // costs about what it saves unless the K loop is long: the fp32-operand kernel's 256 x 128 tiles quantise better there
    // (measured: ResNet50-8s layer-3 3x3 at 1280 x 960 -5 %, its 1x1 1024 -> 256 -20 %; layer4.0.conv1 of ResNet34-8s at
    // N = 8 -3 %; the 512 -> 512 layer-4 convolutions, K = 4608, +7 %)
    // (a launch that 320-row tiles cover in at most one round is none of these cases)
    if (chosen.rows == 320 && chosen.mtiles * chosen.ntiles <= 256 && chosen.mtiles * chosen.ntiles >= 120) return 1;
    const double rounds = (double)dcn::ceil_div64(M, 256) * dcn::ceil_div(cd, 256) / 256.0;
    if (rounds > 1.0 && rounds < 1.5 && K < 4096) return 0;
    // fewer than ~120 tiles leave more than half of the 256 CUs idle (no stream-K below one round): config 1 (two images,
    // 38 x 2 tiles on layer 4) measured -3.5 % on the step with this kernel (profiles/r3l_config1_ab.txt)
    if (rounds < 120.0 / 256.0) return 0;
    return 1;
}

extern "C" int dcn_conv_num_mtiles_hl(const dcn_conv_desc* c) {
    if (!valid_desc_hl(c)) return DCN_E_INVALID;
    return hl_shape_of(c, 0).mtiles;
}

// Rows per tile the launch of this convolution will use (256, 192, 320 or 160): the granularity of its batch-norm partial statistics.
extern "C" int dcn_conv_tile_rows_hl(const dcn_conv_desc* c, int dgrad) {
    if (!valid_desc_hl(c)) return DCN_E_INVALID;
    return hl_shape_of(c, dgrad).rows;
}

extern "C" size_t dcn_conv_gemm_workspace_hl(const dcn_conv_desc* c, int dgrad) {
    if (!valid_desc_hl(c)) return 0;
    return hl_shape_of(c, dgrad).ws_bytes;
}

extern "C" int dcn_conv_hl_shape_info(const dcn_conv_desc* c, int dgrad, int* info6) {
    if (!valid_desc_hl(c) || !info6) return DCN_E_INVALID;
    if (!hl_supported(c, dgrad)) return DCN_E_UNSUPPORTED;
    const HlShape g = hl_shape_of(c, dgrad);
    const bool x = g.rows == kHlxRows;
    info6[0] = g.rows; info6[1] = x ? g.x.bn : 256; info6[2] = x ? g.x.kg : 1; info6[3] = x ? g.x.splits : 1;
    info6[4] = g.mtiles; info6[5] = g.ntiles;
    return DCN_OK;
}

extern "C" int dcn_split_act_hl32(const float* src, const float* absmax, void* dst, int64_t rows, int channels, void* stream) {
    if (!src || !dst || rows < 1 || channels < HLK || (channels % HLK) != 0) return DCN_E_INVALID;
    const int64_t n8 = rows * (channels / 8);
    const unsigned blocks = (unsigned)(dcn::ceil_div64(n8, 256) > 8192 ? 8192 : dcn::ceil_div64(n8, 256));
    hipLaunchKernelGGL(split_act_hl32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, absmax, (u32x4*)dst, n8);
    return dcn::check_launch();
}

// in_hl: hl32 image of the input [n, hin, win, cin], scaled by pow2_scale(*in_absmax) (dcn_split_act_hl32 or a producer
// kernel); w_hl: hl32 image [cout][taps * cin / 32][hi | lo] from dcn_split_weights_hl32 (scale w_scale).
extern "C" int dcn_conv_forward_hl(const dcn_conv_desc* c, const void* in_hl, const float* in_absmax, const void* w_hl,
                                   float w_scale, const float* bias, float* out, float* bn_partial, void* workspace,
                                   void* stream) {
    if (!in_hl || !w_hl || !out || !(w_scale > 0.f)) return DCN_E_INVALID;
    if (!hl_supported(c, 0)) return DCN_E_UNSUPPORTED;
    GemmConv p;
    p.src = (const float*)in_hl; p.wm = nullptr; p.bias = bias; p.add = nullptr; p.dst = out; p.bn_partial = bn_partial;
    p.out_absmax = nullptr; p.wh = (const _Float16*)w_hl; p.wl = nullptr; p.a_absmax = in_absmax; p.b_inv_scale = 1.f / w_scale;
    p.hs = c->hin; p.ws = c->win; p.cs = c->cin; p.hd = c->hout; p.wd = c->wout; p.cd = c->cout;
    p.kh = c->kh; p.kw = c->kw; p.stride = 1; p.pad = c->pad; p.dil = c->dil; p.ldc = c->ldc;
    p.M = c->n * c->hout * c->wout; p.K = c->kh * c->kw * c->cin; p.kp = p.K; p.transposed = 0; p.relu = 0;
    p.src_bytes = (unsigned)((int64_t)c->n * c->hin * c->win * c->cin * 4);
    p.w_bytes = (unsigned)((int64_t)c->cout * p.K * 4);
    return launch_gemm_hl(p, c->group_rows, workspace, (hipStream_t)stream);
}

// dout_hl: hl32 image of the output gradient [n, hout, wout, ldc], scaled by pow2_scale(*dout_absmax); wt_hl: hl32 image of
// the channel-transposed weights [cin][taps * ldc / 32][hi | lo] (dcn_split_weights_hl32, transposed).
extern "C" int dcn_conv_dgrad_hl(const dcn_conv_desc* c, const void* dout_hl, const void* wt_hl, float w_scale,
                                 const float* dout_absmax, const float* add, float* din, void* workspace, void* stream) {
    if (!dout_hl || !wt_hl || !din || !(w_scale > 0.f)) return DCN_E_INVALID;
    if (!hl_supported(c, 1)) return DCN_E_UNSUPPORTED;
    GemmConv p;
    p.src = (const float*)dout_hl; p.wm = nullptr; p.bias = nullptr; p.add = add; p.dst = din; p.bn_partial = nullptr;
    p.out_absmax = nullptr; p.wh = (const _Float16*)wt_hl; p.wl = nullptr; p.a_absmax = dout_absmax; p.b_inv_scale = 1.f / w_scale;
    p.hs = c->hout; p.ws = c->wout; p.cs = c->ldc; p.hd = c->hin; p.wd = c->win; p.cd = c->cin;
    p.kh = c->kh; p.kw = c->kw; p.stride = 1; p.pad = c->pad; p.dil = c->dil; p.ldc = c->cin;
    p.M = c->n * c->hin * c->win; p.K = c->kh * c->kw * c->ldc; p.kp = p.K; p.transposed = 1; p.relu = 0;
    p.src_bytes = (unsigned)((int64_t)c->n * c->hout * c->wout * c->ldc * 4);
    p.w_bytes = (unsigned)((int64_t)c->cin * p.K * 4);
    return launch_gemm_hl(p, 0, workspace, (hipStream_t)stream);
}
