// Weight gradient of the wide layers on PRE-SPLIT (hl32) operands: dW[n][kc] = sum over pixels m of dy[m][n] x[pix(m, tap)][c]
// (training.py:345 through the backbone's convolutions) as a GEMM whose reduction index is the pixel -- split-fp16 ("f16x3")
// arithmetic on v_mfma_f32_32x32x16_f16, both operands by LDS-DMA, the same two-wavefront-group schedule as the forward /
// dgrad kernel of these layers (conv_hl_kernels.hip): 256 (n) x 256 (kc) tiles on 8 wavefronts (2 x 4, wavefront tile
// 128 x 64), stages of 32 pixels, four phases of 12 MFMAs per stage, counted vmcnt, raw s_barrier.
//
// Both operands live in HBM pixel-major (hl32: per pixel and 32-channel chunk one 128-byte line [hi x32 | lo x32] fp16) but
// the matrix instruction wants them k-major -- 8 consecutive PIXELS of one channel per lane.  The transpose costs nothing:
//   * LDS holds pixel-major half-tiles [32 pixels][4 chunk lines = 512 B], filled by LDS-DMA -- one piece = two pixels x 512 B
//     (lanes 0-31 / 32-63), the source offset per lane, so the input pixel of every output pixel and filter tap (and its
//     validity: padding, image and split borders -> out-of-range offset -> zeros) is just a per-lane address;
//   * fragments come out with `ds_read_b64_tr_b16` (measured semantics, tools/tr_probe.bin: the 16 lanes of a group supply
//     8-byte runs S[s][0..3]; lane i receives S[4 j + (i >> 2)][i & 3], j = 0..3): a group that supplies pixel rows
//     k0 + (s >> 2) and channels 4 (s & 3) .. + 3 hands lane i the four pixels k0 .. k0 + 3 of channel i -- two reads make
//     the 8-k operand of one lane.
// Round 5: v_mfma_f32_16x16x32_f16 tiles (the two-group schedule sustains 10-15 % more on that shape: tools/hl_gemm_probe3.hip) --
// lane (channel fr = lane & 15, k-octet fq = lane >> 4) holds the pixels 8 fq .. 8 fq + 7 of the 32-pixel stage: lane group fq reads
// the pixel rows 8 fq + (s >> 2) (+ 4 for the second read), ONE 16-channel block per read.  A 32-lane access then touches EIGHT
// pixel rows (8 fq' + 0..3, fq' = two consecutive groups) x 32 bytes: the 32-byte units of a pixel's 512 bytes are XOR-swizzled
// with key(pixel) = (pixel & 3) | ((pixel >> 3) & 1) << 2 on the source side of the DMA, so that those eight rows fall into the
// eight 32-byte groups of the 64 banks (row pitch 512 B = 0 mod 256 otherwise).  (Rounds 3-4: 32 x 32 x 16 tiles, 64-byte slots
// swizzled with pixel & 3 -- four rows x two channel halves per access.)
// Half-tiles in the order the phases first need them, as in the forward kernel: H0 = A_0, H1 = B_0, H2 = B_1, H3 = A_1 with
//     A_i = the dy channels {128 g + 64 i + [0, 64)} (g = 0, 1: local chunk a = 2 g + t),  B_j = the x columns
//     {64 wn + 32 j + [0, 32)} (wn = 0..3: local chunk b = wn)
// and the LDS-DMA discipline / WAR / RAW argument of conv_hl_kernels.hip verbatim.  Pixel ranges are split over workgroups
// (whole 32-pixel stages), partial slabs are summed by the fixed-order reduce kernel of the other wgrad kernels:
// bit-reproducible.
#include <algorithm>
#include <type_traits>

#include "conv_shared.h"
#include "dcn_tuning.h"
#include "f16_split.h"

namespace {

using namespace dcnconv;
using namespace dcnsplit;

typedef __attribute__((address_space(3))) void* whl_lds_ptr;
typedef short s4v __attribute__((ext_vector_type(4)));

constexpr int kOob = (int)0x80000000;
constexpr int kStage = 65536;   // bytes per 32-pixel stage: A_0 | A_1 | B_0 | B_1, each [32 pixels][512 B]
constexpr int WPX = 32;         // pixels per stage

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rs, void* lds_dst, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (whl_lds_ptr)lds_dst, 16, voffset, soffset, 0, 0);
}
// four pixels of one channel (see the header): the transposing 8-byte LDS read
__device__ __forceinline__ s4v lds_tr4(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p);
}

struct WgradHl {
    const void* xh;            // hl32 activations [n, hin, win, cin], scaled by pow2_scale(*x_absmax)
    const void* dh;            // hl32 output gradient [M][ldo], scaled by pow2_scale(*d_absmax)
    float* slab;               // [splits][cout][K]
    const float* x_absmax;
    const float* d_absmax;
    unsigned x_bytes, d_bytes;
    int hin, win, cin, cout, kh, kw, pad, dil, ldo, M, K, splits, stages_per_split, ntiles_n, ntiles_k, cpt, setprio;
    FastDiv div_hw, div_w, div_cpt, div_kw;
};

__global__ void __launch_bounds__(512, 1)
conv_wgrad_hl_kernel(WgradHl p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kStage];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wv >> 2, wn = wv & 3;
    const int tiles = p.ntiles_n * p.ntiles_k;
    const int bid = xcd_remap(blockIdx.x, tiles * p.splits);
    const int split = bid / tiles, tile = bid - split * tiles;
    const int tn_ = tile / p.ntiles_k, tk_ = tile - tn_ * p.ntiles_k;
    const int n0 = tn_ * 256, kc0 = tk_ * 256;
    const int nstages = (p.M + WPX - 1) / WPX;
    const int s_begin = split * p.stages_per_split;
    const int s_end = min(nstages, s_begin + p.stages_per_split);
    const int m_end = min(p.M, s_end * WPX);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.xh), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dh), 0, (int)p.d_bytes, 0x00020000);

    // ---- LDS-DMA pieces of this lane.  Piece e (0, 1) of every half-tile covers stage pixels 4 wv + 2 e + {0, 1}: lanes
    // 0-31 the first, 32-63 the second; lane l fills PHYSICAL 16-byte slot l & 31 of its pixel's 512 bytes = 16-byte half
    // (l & 1) of physical 32-byte unit (l & 31) >> 1 = logical unit ((l & 31) >> 1) ^ key(pixel) -> (64-byte slot = unit >> 1 ->
    // local chunk = slot >> 1, plane = slot & 1; 16-channel block = unit & 1).
    const int lp = lane >> 5, p16 = lane & 31;
    const int ldo4 = p.ldo * 4, cin4 = p.cin * 4;
    int kp[2];                    // pixel index inside the stage
    int cA[2][2], cB[2][2];       // [i | j][e]: byte offset inside the source pixel (chunk line + plane + 16-byte piece), or -1: no such chunk
    int tdy[2][2], tdx[2][2];     // [j][e]: filter-tap offset (input pixel = output pixel + this) of the lane's x chunk
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        kp[e] = 4 * wv + 2 * e + lp;
        const int key = (kp[e] & 3) | (((kp[e] >> 3) & 1) << 2);
        const int unit = (p16 >> 1) ^ key;
        const int slot = unit >> 1;
        const int lc = slot >> 1, pl = slot & 1, q16 = 2 * (unit & 1) + (p16 & 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // A_i: local chunk lc = 2 g' + t  <->  tile chunk 4 g' + 2 i + t
            const int chunk = n0 / 32 + 4 * (lc >> 1) + 2 * i + (lc & 1);
            cA[i][e] = chunk * 32 < p.cout ? chunk * 128 + pl * 64 + q16 * 16 : -1;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {   // B_j: local chunk lc = wn'  <->  tile chunk 2 wn' + j  <->  (filter tap, channel chunk)
            const int G = kc0 / 32 + 2 * lc + j;
            const bool ok = G * 32 < p.K;
            const int tap = fdiv(ok ? G : 0, p.div_cpt), cc = (ok ? G : 0) - tap * p.cpt;
            const int r = fdiv(tap, p.div_kw), s = tap - r * p.kw;
            tdy[j][e] = r * p.dil - p.pad;
            tdx[j][e] = s * p.dil - p.pad;
            cB[j][e] = ok ? cc * 128 + pl * 64 + q16 * 16 : -1;
        }
    }
    // position of this lane's two pixels in the stage the LDS-DMA is issued for (carried from stage to stage: win >= 32)
    int pm[2], py[2], px[2], pimg[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        pm[e] = s_begin * WPX + kp[e];
        const int mm = pm[e] < p.M ? pm[e] : 0;
        pimg[e] = fdiv(mm, p.div_hw);
        const int rem = mm - pimg[e] * p.div_hw.d;
        py[e] = fdiv(rem, p.div_w);
        px[e] = rem - py[e] * p.div_w.d;
    }
    int voA[2][2], voB[2][2];
    auto set_offsets = [&]() {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool mok = pm[e] < m_end;
#pragma unroll
            for (int i = 0; i < 2; ++i) voA[i][e] = (mok && cA[i][e] >= 0) ? pm[e] * ldo4 + cA[i][e] : kOob;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int sy = py[e] + tdy[j][e], sx = px[e] + tdx[j][e];
                const bool ok = mok & (cB[j][e] >= 0) & ((unsigned)sy < (unsigned)p.hin) & ((unsigned)sx < (unsigned)p.win);
                voB[j][e] = ok ? ((pimg[e] * p.hin + sy) * p.win + sx) * cin4 + cB[j][e] : kOob;
            }
        }
    };
    auto advance = [&]() {   // the next stage: 32 pixels on (output width >= 32: at most one row wrap)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            pm[e] += WPX;
            px[e] += WPX;
            const bool wx = px[e] >= p.win;
            px[e] -= wx ? p.win : 0;
            py[e] += wx ? 1 : 0;
            const bool wy = py[e] >= p.hin;
            py[e] = wy ? 0 : py[e];
            pimg[e] += wy ? 1 : 0;
        }
        set_offsets();
    };
    set_offsets();
    // half-tile h (0: A_0, 1: B_0, 2: B_1, 3: A_1) of the stage the positions point at
    auto issue_half = [&](int buf, int h) {
        unsigned char* base = lds + buf * kStage + (h == 0 ? 0 : h == 3 ? 16384 : h == 1 ? 32768 : 49152) + wv * 2048;
        if (h == 0 || h == 3) {
            const int i = h == 0 ? 0 : 1;
            glds16(rs_d, base, voA[i][0], 0);
            glds16(rs_d, base + 1024, voA[i][1], 0);
        } else {
            const int j = h == 1 ? 0 : 1;
            glds16(rs_x, base, voB[j][0], 0);
            glds16(rs_x, base + 1024, voB[j][1], 0);
        }
    };

    // ---- fragments: lane (group gq = lane >> 4, s = lane & 15) supplies pixel row 8 gq + 4 h + (s >> 2), channels
    // 4 (s & 3) .. + 3 of a 16-channel block and receives the pixels 8 gq + 4 h .. + 3 of channel lane & 15: element 4 h + .. of
    // the lane's 8-k MFMA operand (k-octet gq of the stage's 32 pixels)
    const int gq = lane >> 4, s16 = lane & 15, rq = s16 >> 2;
    const int fkey = rq | ((gq & 1) << 2);                 // key(pixel) of the lane's pixel rows (both reads: + 4 leaves it)
    const int prow = (8 * gq + rq) * 512 + (s16 & 3) * 8;
    int offA[4][2], offB[2][2];   // [16-channel block of the quadrant's 64 / 32][plane]: lane constants
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) offA[rb][pl] = prow + ((2 * (2 * (2 * grp + (rb >> 1)) + pl) + (rb & 1)) ^ fkey) * 32;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) offB[cb][pl] = prow + ((2 * (2 * wn + pl) + cb) ^ fkey) * 32;
    }
    f32x4_t acc[8][4];   // [16-row block of the wavefront's 128 dy channels][16-column block of its 64 K columns]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    h8 fa[4][2], fb[2][2];   // A: [block][plane], B: [block][plane]
    auto frag = [&](const unsigned char* q) {
        const s4v lo = lds_tr4(q), hi = lds_tr4(q + 4 * 512);
        typedef short s8v __attribute__((ext_vector_type(8)));
        const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(h8, v);
    };
    auto read_a = [&](int buf, int i) {
        const unsigned char* st = lds + buf * kStage + (i ? 16384 : 0);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fa[rb][pl] = frag(st + offA[rb][pl]);
    };
    auto read_b = [&](int buf, int j) {
        const unsigned char* st = lds + buf * kStage + (j ? 49152 : 32768);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fb[cb][pl] = frag(st + offB[cb][pl]);
    };
    auto mfma_quadrant = [&](int i, int j) {
        if (p.setprio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    acc[4 * i + rb][2 * j + cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pt == 0 ? fa[rb][1] : fa[rb][0],
                                                                                         pt == 1 ? fb[cb][1] : fb[cb][0],
                                                                                         acc[4 * i + rb][2 * j + cb], 0, 0, 0);
        if (p.setprio) __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); };
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
            if (ph == 3) advance();   // (the positions now point at stage s + 2)
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
    using T = std::true_type;
    using F = std::false_type;
    if (s_begin < s_end) {   // (wave-uniform: a split past the last stage has nothing to add; its slab tile stays zero)
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
        for (int s = s_begin; s + 1 < s_end; ++s) {
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
        if (grp == 0) bar();
    }
    // C fragment of a 16 x 16 tile: row 4 (lane >> 4) + r <-> output channel n, column lane & 15 <-> K column: 64-byte row pieces
    const float inv = 1.f / ((p.d_absmax ? pow2_scale(*p.d_absmax) : 1.f) * (p.x_absmax ? pow2_scale(*p.x_absmax) : 1.f));
    float* out = p.slab + (int64_t)split * p.cout * p.K;
    const int fc = lane & 15, fq4 = 4 * (lane >> 4);
#pragma unroll
    for (int tb = 0; tb < 8; ++tb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int kc = kc0 + wn * 64 + cb * 16 + fc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + grp * 128 + tb * 16 + fq4 + r;
                if (n < p.cout && kc < p.K) out[(int64_t)n * p.K + kc] = acc[tb][cb][r] * inv;
            }
        }
}

bool wgrad_hl_supported(const dcn_conv_desc* c) {
    if (!c || c->n < 1 || c->hin < 1 || c->win < 32 || c->cin < 32 || c->cout < 32 || c->kh < 1 || c->kw < 1) return false;
    if (c->stride != 1 || c->hin != c->hout || c->win != c->wout || c->ldc != c->cout) return false;
    if ((c->cin % 32) != 0 || (c->cout % 32) != 0) return false;
    const int64_t xb = (int64_t)c->n * c->hin * c->win * c->cin * 4, db = (int64_t)c->n * c->hout * c->wout * c->ldc * 4;
    return xb <= ((int64_t)1 << 31) - 1 && db <= ((int64_t)1 << 31) - 1;
}

// pixel-range splits in whole 32-pixel stages: about one round of 256 workgroups (one per CU), the last round nearly full
int wgrad_hl_splits(const dcn_conv_desc* c, int* stages_per_split) {
    const int M = c->n * c->hout * c->wout, K = c->kh * c->kw * c->cin;
    const int tiles = dcn::ceil_div(c->cout, 256) * dcn::ceil_div(K, 256);
    const int nstages = dcn::ceil_div(M, WPX);
    int best = 1;
    double best_score = -1.0;
    const int hi = std::min(nstages, std::max(1, 2 * 256 / tiles + 1));
    for (int s = 1; s <= hi; ++s) {
        const int sps = dcn::ceil_div(nstages, s), real = dcn::ceil_div(nstages, sps);
        const int wgs = real * tiles;
        const double eff = (double)wgs / (double)(dcn::ceil_div(wgs, 256) * 256);
        // prefer full rounds; every split pays its prologue, its 256 KB slab tile and its share of the reduce pass
        const double score = eff - 0.002 * real - (wgs < 256 ? 0.5 * (1.0 - wgs / 256.0) : 0.0) - (wgs > 256 ? 0.15 : 0.0);
        if (score > best_score) { best_score = score; best = s; }
    }
    if (const int v = dcn::tuning().wgrad_splits) { if (v >= 1 && v <= nstages) best = v; }
    const int sps = dcn::ceil_div(nstages, best);
    *stages_per_split = sps;
    return dcn::ceil_div(nstages, sps);
}


// ------------------------------------------------------------------------------------------------------------------------
// Row-window kernel ("hlr") of the NARROW 3 x 3 layers (64 / 128 channels: ResNet layers 1 and 2), whose 64 x 576 / 128 x 1152
// weight gradients are a fraction of one 256 x 256 tile above.  One workgroup owns 64 output channels x ALL nine taps x 64 input
// channels -- 18 16 x 16 accumulator tiles per wavefront, the whole reduction in registers -- and walks a range of STAGES = 32
// consecutive output pixels of ONE image row.  Per stage LDS holds the 32 dy pixels (64 channels) and three row WINDOWS of x
// (rows y - 1, y, y + 1; x0 - 1 .. x0 + 32; the tile's 64 channels): filter tap (r, s) reads window r at pixel offset s, so an
// input pixel enters LDS 3.2 times per launch instead of 9 (what bounds the gather form on these layers is the L2 -> LDS traffic
// and the per-tap operand split, not the matrix pipe: 108 - 160 TF), and every border case is a per-lane out-of-range LDS-DMA
// offset (zeros) because stages never straddle an image row.  Wavefront (nh = wv & 1, cq = wv >> 1): dy blocks 2 nh, 2 nh + 1 x
// taps 0..8 x input block cq: 2 dy fragments and 9 shifted x fragments per stage -- 44 transposing reads for 54 MFMAs.  Same hl32
// lines, same 32-byte-unit swizzle key(pixel) = (pixel & 3) | ((pixel >> 3) & 1) << 2 as above (a 32-lane read touches two runs of
// four consecutive pixel rows eight apart WHATEVER the tap shift: eight distinct keys); two stage buffers, all wavefronts in
// step (DMA of stage s + 1 | wait stage s | barrier | 9 taps, fragments two taps ahead | barrier).
// (64 x 64)-channel tiles x pixel ranges fill the chip; slabs summed by the fixed-order reduce kernels: bit-reproducible.
// Measured (profiles/r5r_wgrad_hlr_64ch_tiles.txt, slab reduce included): layer 1 at eight images 103.2 -> 52.9 us, layer 2 72.0 -> 58.0 us;
// the step: config 2 +1.9 %, its two-call form +1.6 %, config 5 (ResNet50, 1280 x 960) +3.6 % (profiles/r5s_ab_wgrad_hlr.txt, r5t_*).
constexpr int RW_WP = 34;                 // window pixels: 32 + 2 (dilation 1, padding 1)
constexpr int RW_DY_BYTES = 32 * 256;     // [32 pixels][2 chunk lines]

template <int CB> struct RwGeom {
    static constexpr int kPitch = 256 * CB;                                  // bytes per window pixel (cin = 64 CB)
    static constexpr int kXInstr = (3 * RW_WP * kPitch + 8191) / 8192;        // LDS-DMA instructions per wavefront (1 KB each)
    static constexpr int kXBytes = kXInstr * 8192;
    static constexpr int kStageBytes = RW_DY_BYTES + kXBytes;
};

struct WgradHlr {
    const void* xh;
    const void* dh;
    float* slab;               // [splits][cout][K]
    const float* x_absmax;
    const float* d_absmax;
    unsigned x_bytes, d_bytes;
    int hin, win, cin, cout, ldo, K, nstages, segs, splits, stages_per_split, ntiles_n, ntiles_c;
    FastDiv div_segs, div_h;
};

template <int CB>
__global__ void __launch_bounds__(512, 1)
conv_wgrad_hlr_kernel(WgradHlr p) {
    using G = RwGeom<CB>;
    constexpr int E = G::kXInstr, PITCH = G::kPitch;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * G::kStageBytes];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nh = wv & 1, cq = wv >> 1;
    const int tiles = p.ntiles_n * p.ntiles_c;
    const int bid = xcd_remap(blockIdx.x, tiles * p.splits);
    const int split = bid / tiles, tile = bid - split * tiles;
    const int tn_ = tile / p.ntiles_c, tc_ = tile - tn_ * p.ntiles_c;
    const int n0 = tn_ * 64, c0 = tc_ * 64 * CB;
    const int s_begin = split * p.stages_per_split;
    const int s_end = min(p.nstages, s_begin + p.stages_per_split);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.xh), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dh), 0, (int)p.d_bytes, 0x00020000);
    auto key_of = [](int pixel) { return (pixel & 3) | (((pixel >> 3) & 1) << 2); };

    // ---- LDS-DMA pieces of this lane (lane constants).  One instruction fills 1 KB of LDS, lane l its 16-byte slot l.
    // dy: wavefront wv -> stage pixels 4 wv .. 4 wv + 3 (256 B each); x: instruction k = wv + 8 e -> bytes [1024 k, 1024 k + 1024)
    // of the window image [3][RW_WP][PITCH].  PHYSICAL 16-byte slot ps of a pixel holds half (ps & 1) of physical 32-byte unit
    // ps >> 1 = logical unit (ps >> 1) ^ key(pixel): unit u -> chunk line u >> 2, plane (u >> 1) & 1, 16-channel half u & 1.
    auto unit_src = [](int u, int ps) { return (u >> 2) * 128 + ((u >> 1) & 1) * 64 + (u & 1) * 32 + (ps & 1) * 16; };
    const int dpix = 4 * wv + (lane >> 4);
    const int dsrc = (n0 / 32) * 128 + unit_src(((lane & 15) >> 1) ^ key_of(dpix), lane & 15);
    int xinfo[E];   // source bytes inside the pixel | window pixel q << 16 | (window r + 1) << 24  (r + 1 = 0: past the image: zeros)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int byte = (wv + 8 * e) * 1024 + lane * 16;
        const int row = byte / PITCH, ps = (byte - row * PITCH) >> 4;
        const int r = row / RW_WP, q = row - r * RW_WP;
        xinfo[e] = unit_src((ps >> 1) ^ key_of(q), ps) | (q << 16) | ((row < 3 * RW_WP ? r + 1 : 0) << 24);
    }
    const int ldo4 = p.ldo * 4, cin4 = p.cin * 4;
    auto issue_stage = [&](int t, int buf) {   // stage t -> (image row id = img * H + y, segment)
        const int rowid = fdiv(t, p.div_segs), x0 = (t - rowid * p.segs) * 32;
        const int img = fdiv(rowid, p.div_h), y = rowid - img * p.hin;
        unsigned char* base = lds + buf * G::kStageBytes;
        const int vd = (x0 + dpix < p.win) ? (rowid * p.win + x0 + dpix) * ldo4 + dsrc : kOob;
        glds16(rs_d, base + wv * 1024, vd, 0);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int r1 = xinfo[e] >> 24, q = (xinfo[e] >> 16) & 255, src = xinfo[e] & 0xffff;
            const int yy = y + r1 - 2, xx = x0 - 1 + q;
            const bool ok = (r1 > 0) & ((unsigned)yy < (unsigned)p.hin) & ((unsigned)xx < (unsigned)p.win);
            const int vx = ok ? ((rowid + r1 - 2) * p.win + xx) * cin4 + c0 * 4 + src : kOob;
            glds16(rs_x, base + RW_DY_BYTES + (wv + 8 * e) * 1024, vx, 0);
        }
    };

    // ---- fragments.  Lane (gq = lane >> 4, s16 = lane & 15) supplies pixel row 8 gq + 4 h + (s16 >> 2) (+ the tap's column shift for
    // x), channels 4 (s16 & 3) .. + 3 of a 16-channel block, and receives pixels 8 gq + 4 h .. + 3 of channel s16 (see the header)
    const int gq = lane >> 4, s16 = lane & 15, rq = s16 >> 2;
    const int sub = (s16 & 3) * 8;
    // logical unit of (16-channel block b of the pixel, plane): chunk line b >> 1, half b & 1 -- as a byte offset (x 32)
    auto unit32 = [](int b, int pl) { return ((((b >> 1) * 2 + pl) * 2) + (b & 1)) * 32; };
    // Byte offsets of the lane's reads.  The fields of an offset do not overlap -- pixel row x pitch (bits >= 8), swizzled unit
    // (bits 5-7, + bit 8 at cin = 128), 8-byte run (bits 3-4) -- and the blocks / planes one wavefront reads differ in unit bits 0-1
    // only (dy: block i, plane pl of blocks 2 nh, 2 nh + 1 = unit 4 nh + 2 pl + i; x: 4 cq + 2 pl + c, or 4 (cq >> 1) + 2 pl + (cq & 1)
    // at cin = 64): ONE base per (column shift, read) and an XOR with (block * 32 | plane * 64) per read; window row, second read
    // and stage buffer are additive constants (the instruction's immediate offset).  The column shift can carry into bit 3 of
    // the pixel: the key is per (shift, read).
    const int abase = (8 * gq + rq) * 256 + sub + (unit32(2 * nh, 0) ^ ((rq | ((gq & 1) << 2)) * 32));   // second read: + 4 * 256
    int bbase[3][2];
#pragma unroll
    for (int sft = 0; sft < 3; ++sft)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = 8 * gq + rq + sft + 4 * h;
            bbase[sft][h] = q * PITCH + sub + (unit32(CB * cq, 0) ^ (key_of(q) * 32));
        }
    typedef short s8v __attribute__((ext_vector_type(8)));
    auto frag2 = [&](const unsigned char* q0, const unsigned char* q1) {
        const s4v lo = lds_tr4(q0), hi = lds_tr4(q1);
        const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(h8, v);
    };
    f32x4_t acc[2][9][CB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < CB; ++c) acc[i][t][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    h8 fa[2][2];        // dy: [block][plane]
    constexpr int AH = CB == 1 ? 2 : 1;   // taps read ahead of the MFMAs (6 MFMAs per tap at cin = 64 do not cover an LDS read)
    h8 fb[AH + 1][CB][2];   // x of one tap, a ring of AH + 1: [slot][block][plane]
    // (the bases of the stage at hand, re-materialised per stage behind an opaque barrier: the optimiser would otherwise keep all
    // 80 / 152 loop-invariant addresses of a stage in registers -- spills at cin = 128 -- or recompute them in full per read)
    int va = 0, vb[3][2] = {};
    auto read_a = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const unsigned char* q = lds + (va ^ (i * 32 | pl * 64));
                fa[i][pl] = frag2(q, q + 4 * 256);
            }
    };
    auto read_b = [&](int tap, int slot) {
        const int r = tap / 3, sft = tap - 3 * r;
        const int woff = RW_DY_BYTES + r * RW_WP * PITCH;
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fb[slot][c][pl] = frag2(lds + (vb[sft][0] ^ (c * 32 | pl * 64)) + woff, lds + (vb[sft][1] ^ (c * 32 | pl * 64)) + woff);
    };
    auto mfma_tap = [&](int tap, int slot) {
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < CB; ++c)
                    acc[i][tap][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pt == 0 ? fa[i][1] : fa[i][0],
                                                                            pt == 1 ? fb[slot][c][1] : fb[slot][c][0],
                                                                            acc[i][tap][c], 0, 0, 0);
    };
    if (s_begin < s_end) issue_stage(s_begin, 0);
    for (int t = s_begin; t < s_end; ++t) {
        const int buf = (t - s_begin) & 1;
        // (the other buffer was last read in stage t - 1, behind that iteration's closing barrier)
        if (t + 1 < s_end) {
            issue_stage(t + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            static_assert((CB == 1 && E + 1 == 5) || (CB == 2 && E + 1 == 8), "counted vmcnt below");   // (CB = 2: 128 input channels per workgroup, 144 accumulator registers -- measured slower than two 64-channel tiles (twice the slab bytes per launch); not instantiated)
            if constexpr (CB == 1) DCN_WAIT_VMCNT(5); else DCN_WAIT_VMCNT(8);   // all but the E + 1 instructions just issued
        } else {
            __builtin_amdgcn_sched_barrier(0);
            DCN_WAIT_VMCNT(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        va = abase + buf * G::kStageBytes;
        DCN_OPAQUE_INT(va);
#pragma unroll
        for (int sft = 0; sft < 3; ++sft)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                vb[sft][h] = bbase[sft][h] + buf * G::kStageBytes;
                DCN_OPAQUE_INT(vb[sft][h]);
            }
        read_a();
#pragma unroll
        for (int tap = 0; tap < AH; ++tap) read_b(tap, tap);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {   // (fenced: AH taps' fragments ahead, no more -- 144 accumulator registers at cin = 128)
            if (tap + AH < 9) read_b(tap + AH, (tap + AH) % (AH + 1));
            __builtin_amdgcn_sched_barrier(0);
            mfma_tap(tap, tap % (AH + 1));
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        DCN_WAIT_LGKMCNT0();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    // C fragment of a 16 x 16 tile: row 4 (lane >> 4) + r <-> output channel, column lane & 15 <-> input channel of the tap
    const float inv = 1.f / ((p.d_absmax ? pow2_scale(*p.d_absmax) : 1.f) * (p.x_absmax ? pow2_scale(*p.x_absmax) : 1.f));
    float* out = p.slab + (int64_t)split * p.cout * p.K;
    const int fc = lane & 15, fq4 = 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const int kc = tap * p.cin + c0 + (CB * cq + c) * 16 + fc;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + (2 * nh + i) * 16 + fq4 + r;
                    out[(int64_t)n * p.K + kc] = acc[i][tap][c][r] * inv;
                }
            }
}


// ---- ROW-PAIR form of the same kernel (default).  The single-row form above is bound by its transposing LDS reads (44 reads per 54
// MFMAs and wavefront: 180 KB per stage against 1728 matrix-pipe cycles; measured 4000 cycles per stage): here a stage is TWO image
// rows (y, y + 1) x 32 pixels, the two wavefront groups (wv >> 2) take one row each -- FOUR x windows (rows y - 1 .. y + 2) serve
// both, group g reads windows g .. g + 2 -- and a wavefront (cq = wv & 3) owns ALL four dy blocks x nine taps x input block cq:
// 36 accumulator tiles (144 registers), 16 + 36 reads for 108 MFMAs.  An input pixel now enters LDS 2.1 times per launch.  The
// groups' accumulators (sums over different pixels of the same 64 x 576 tile) are added through LDS at the end, group 1 into
// group 0, in two passes of 18 tiles; group 0 writes the slab.  Measured (profiles/r5u_wgrad_hlr_row_pairs.txt): layer 1 at eight
// images 52.1 -> 48.6 us, layer 2 57.8 -> 51.7 us -- of which about 18 us are the launch pair's fixed part (256 slabs of 147 KB
// written and summed: 26 us at two images, 33 us at four).
constexpr int RP_DY_BYTES = 2 * 32 * 256;                  // [2 rows][32 pixels][2 chunk lines]
constexpr int RP_XINSTR = 5;                               // LDS-DMA instructions per wavefront for [4][RW_WP][256 B] = 34 KB (40 KB held)
constexpr int RP_STAGE = RP_DY_BYTES + RP_XINSTR * 8192;   // 56 KB

__global__ void __launch_bounds__(512, 1)
conv_wgrad_hlrp_kernel(WgradHlr p) {
    constexpr int E = RP_XINSTR, PITCH = 256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * RP_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wv >> 2, cq = wv & 3;
    const int tiles = p.ntiles_n * p.ntiles_c;
    const int bid = xcd_remap(blockIdx.x, tiles * p.splits);
    const int split = bid / tiles, tile = bid - split * tiles;
    const int tn_ = tile / p.ntiles_c, tc_ = tile - tn_ * p.ntiles_c;
    const int n0 = tn_ * 64, c0 = tc_ * 64;
    const int s_begin = split * p.stages_per_split;
    const int s_end = min(p.nstages, s_begin + p.stages_per_split);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.xh), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dh), 0, (int)p.d_bytes, 0x00020000);
    auto key_of = [](int pixel) { return (pixel & 3) | (((pixel >> 3) & 1) << 2); };
    auto unit_src = [](int u, int ps) { return (u >> 2) * 128 + ((u >> 1) & 1) * 64 + (u & 1) * 32 + (ps & 1) * 16; };
    // ---- LDS-DMA pieces (see the single-row kernel).  dy: instruction wv + 8 e = pixels 4 wv .. + 3 of row y + e; x: instruction
    // wv + 8 e = bytes [1024 (wv + 8 e), + 1024) of the window image [4][RW_WP][256 B]
    const int dpix = 4 * wv + (lane >> 4);
    const int dsrc = (n0 / 32) * 128 + unit_src(((lane & 15) >> 1) ^ key_of(dpix), lane & 15);
    int xinfo[E];   // source bytes inside the pixel | window pixel q << 16 | (window r + 1) << 24  (0: past the image: zeros)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int byte = (wv + 8 * e) * 1024 + lane * 16;
        const int row = byte / PITCH, ps = (byte - row * PITCH) >> 4;
        const int r = row / RW_WP, q = row - r * RW_WP;
        xinfo[e] = unit_src((ps >> 1) ^ key_of(q), ps) | (q << 16) | ((row < 4 * RW_WP ? r + 1 : 0) << 24);
    }
    const int ldo4 = p.ldo * 4, cin4 = p.cin * 4;
    auto issue_stage = [&](int t, int buf) {   // stage t -> (row pair id = img * ceil(H / 2) + y / 2, segment)
        const int pairid = fdiv(t, p.div_segs), x0 = (t - pairid * p.segs) * 32;
        const int img = fdiv(pairid, p.div_h), y = 2 * (pairid - img * p.div_h.d);
        const int rowid = img * p.hin + y;
        unsigned char* base = lds + buf * RP_STAGE;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int vd = ((x0 + dpix < p.win) & (y + e < p.hin)) ? ((rowid + e) * p.win + x0 + dpix) * ldo4 + dsrc : kOob;
            glds16(rs_d, base + (wv + 8 * e) * 1024, vd, 0);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int r1 = xinfo[e] >> 24, q = (xinfo[e] >> 16) & 255, src = xinfo[e] & 0xffff;
            const int yy = y + r1 - 2, xx = x0 - 1 + q;
            const bool ok = (r1 > 0) & ((unsigned)yy < (unsigned)p.hin) & ((unsigned)xx < (unsigned)p.win);
            const int vx = ok ? ((rowid + r1 - 2) * p.win + xx) * cin4 + c0 * 4 + src : kOob;
            glds16(rs_x, base + RP_DY_BYTES + (wv + 8 * e) * 1024, vx, 0);
        }
    };
    // ---- fragments (byte offsets as in the single-row kernel: one base per (column shift, read), XOR (block, plane) bits per read)
    const int gq = lane >> 4, s16 = lane & 15, rq = s16 >> 2;
    const int sub = (s16 & 3) * 8;
    const int abase = grp * 8192 + (8 * gq + rq) * 256 + sub + ((rq | ((gq & 1) << 2)) * 32);   // second read: + 4 * 256
    auto unit32 = [](int b, int pl) { return ((((b >> 1) * 2 + pl) * 2) + (b & 1)) * 32; };
    int bbase[3][2];
#pragma unroll
    for (int sft = 0; sft < 3; ++sft)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = 8 * gq + rq + sft + 4 * h;
            bbase[sft][h] = RP_DY_BYTES + grp * RW_WP * PITCH + q * PITCH + sub + (unit32(cq, 0) ^ (key_of(q) * 32));
        }
    typedef short s8v __attribute__((ext_vector_type(8)));
    auto frag2 = [&](const unsigned char* q0, const unsigned char* q1) {
        const s4v lo = lds_tr4(q0), hi = lds_tr4(q1);
        const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(h8, v);
    };
    f32x4_t acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    h8 fa[4][2];       // dy: [block][plane]
    h8 fb[2][2];       // x of one tap, double-buffered: [slot][plane]
    int va = 0, vb[3][2] = {};
    auto read_a = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const unsigned char* q = lds + (va ^ unit32(i, pl));
                fa[i][pl] = frag2(q, q + 4 * 256);
            }
    };
    auto read_b = [&](int tap, int slot) {
        const int r = tap / 3, sft = tap - 3 * r;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            fb[slot][pl] = frag2(lds + (vb[sft][0] ^ (pl * 64)) + r * RW_WP * PITCH, lds + (vb[sft][1] ^ (pl * 64)) + r * RW_WP * PITCH);
    };
    auto mfma_tap = [&](int tap, int slot) {
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][tap] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pt == 0 ? fa[i][1] : fa[i][0], pt == 1 ? fb[slot][1] : fb[slot][0],
                                                                     acc[i][tap], 0, 0, 0);
    };
    if (s_begin < s_end) issue_stage(s_begin, 0);
    for (int t = s_begin; t < s_end; ++t) {
        const int buf = (t - s_begin) & 1;
        if (t + 1 < s_end) {
            issue_stage(t + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            static_assert(E + 2 == 7, "counted vmcnt below");
            DCN_WAIT_VMCNT(7);   // all but the E + 2 instructions just issued
        } else {
            __builtin_amdgcn_sched_barrier(0);
            DCN_WAIT_VMCNT(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        va = abase + buf * RP_STAGE;
        DCN_OPAQUE_INT(va);
#pragma unroll
        for (int sft = 0; sft < 3; ++sft)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                vb[sft][h] = bbase[sft][h] + buf * RP_STAGE;
                DCN_OPAQUE_INT(vb[sft][h]);
            }
        read_a();
        read_b(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {   // (fenced: one tap's fragments ahead -- 144 accumulator registers)
            if (tap + 1 < 9) read_b(tap + 1, (tap + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_tap(tap, tap & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        DCN_WAIT_LGKMCNT0();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- group 1's sums into group 0's, two passes of 18 tiles through LDS ([wavefront cq][tile][lane] float4: 72 KB)
    float4* xch = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const f32x4_t v = acc[2 * half + i][tap];
                    xch[(cq * 18 + i * 9 + tap) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
                }
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float4 v = xch[(cq * 18 + i * 9 + tap) * 64 + lane];
                    acc[2 * half + i][tap][0] += v.x; acc[2 * half + i][tap][1] += v.y;
                    acc[2 * half + i][tap][2] += v.z; acc[2 * half + i][tap][3] += v.w;
                }
        }
        __syncthreads();
    }
    if (grp != 0) return;
    const float inv = 1.f / ((p.d_absmax ? pow2_scale(*p.d_absmax) : 1.f) * (p.x_absmax ? pow2_scale(*p.x_absmax) : 1.f));
    float* out = p.slab + (int64_t)split * p.cout * p.K;
    const int fc = lane & 15, fq4 = 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kc = tap * p.cin + c0 + cq * 16 + fc;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(int64_t)(n0 + i * 16 + fq4 + r) * p.K + kc] = acc[i][tap][r] * inv;
        }
}

// which convolutions: the 3 x 3, stride-1, dilation-1 ones with whole 64-channel input and output tiles
bool wgrad_hlr_supported(const dcn_conv_desc* c) {
    if (!c || c->n < 1 || c->hin < 1 || c->win < 1 || c->hout < 1 || c->wout < 1 || c->kh != 3 || c->kw != 3 || c->dil != 1 ||
        c->pad != 1) return false;
    if (c->stride != 1 || c->hin != c->hout || c->win != c->wout || c->ldc != c->cout) return false;
    // (whole 64-channel tiles, at least one each way: a zero channel count passes `% 64` and divides by zero tiles below)
    if (c->cin < 64 || c->cout < 64 || (c->cin % 64) != 0 || (c->cout % 64) != 0) return false;
    const int64_t xb = (int64_t)c->n * c->hin * c->win * c->cin * 4, db = (int64_t)c->n * c->hout * c->wout * c->ldc * 4;
    return xb <= ((int64_t)1 << 31) - 1 && db <= ((int64_t)1 << 31) - 1;
}
bool hlr_pairs() { return dcn::tuning().wgrad_hlr_pairs != 0; }
// stages: (image row | row pair) x 32-pixel segment
int wgrad_hlr_stages(const dcn_conv_desc* c, bool pairs) {
    return c->n * (pairs ? dcn::ceil_div(c->hout, 2) : c->hout) * dcn::ceil_div(c->wout, 32);
}
// stage ranges: one round of workgroups (64-channel tiles x splits <= 256), every split at least two stages
int wgrad_hlr_splits(const dcn_conv_desc* c, int* stages_per_split, bool pairs) {
    const int nstages = wgrad_hlr_stages(c, pairs), tiles = (c->cout / 64) * (c->cin / 64);
    int s = std::max(1, std::min(std::max(256 / tiles, 1), nstages / 2));
    if (const int v = dcn::tuning().wgrad_splits) { if (v >= 1 && v <= nstages) s = v; }
    const int sps = dcn::ceil_div(nstages, s);
    *stages_per_split = sps;
    return dcn::ceil_div(nstages, sps);
}
// the row-window kernel takes the launch (instead of the 256 x 256 tile kernel above / the fp32-operand kernel)
bool use_hlr(const dcn_conv_desc* c) {
    const int v = dcn::tuning().wgrad_hlr;
    if (v == 0 || !wgrad_hlr_supported(c)) return false;
    if (v == 2) return true;
    if ((c->cout % 256) == 0) return false;   // (the 256 x 256 tile kernel's layers)
    if (dcn::tuning().wgrad_hlr_min_m > 0) return (int64_t)c->n * c->hout * c->wout >= dcn::tuning().wgrad_hlr_min_m;
    // from about eight stages per workgroup: at two images (layer 1: 4.7 stages on each of 256 workgroups) every launch is still
    // faster than the fp32-operand kernel's (26.7 vs 37.2 us, 30.6 vs 39.2 us) but the STEP is not (-0.7 %: the old kernel's few
    // workgroups ran beside the main stream's GEMMs, these take the whole chip) -- profiles/r5s_ab_wgrad_hlr.txt
    return (int64_t)c->n * c->hout * dcn::ceil_div(c->wout, 32) * (c->cout / 64) * (c->cin / 64) >= 2048;
}
int launch_wgrad_hlr(const dcn_conv_desc* c, const void* x_hl, const float* x_absmax, const void* dout_hl,
                     const float* dout_absmax, float* dw, void* slabs, hipStream_t st) {
    WgradHlr p;
    p.xh = x_hl; p.dh = dout_hl; p.x_absmax = x_absmax; p.d_absmax = dout_absmax;
    p.x_bytes = (unsigned)((int64_t)c->n * c->hin * c->win * c->cin * 4);
    p.d_bytes = (unsigned)((int64_t)c->n * c->hout * c->wout * c->ldc * 4);
    p.hin = c->hin; p.win = c->win; p.cin = c->cin; p.cout = c->cout; p.ldo = c->ldc; p.K = 9 * c->cin;
    p.segs = dcn::ceil_div(c->wout, 32); p.nstages = wgrad_hlr_stages(c, hlr_pairs());
    p.splits = wgrad_hlr_splits(c, &p.stages_per_split, hlr_pairs());
    p.ntiles_n = c->cout / 64; p.ntiles_c = c->cin / 64;
    p.div_segs = make_fastdiv(p.segs); p.div_h = make_fastdiv(hlr_pairs() ? dcn::ceil_div(c->hin, 2) : c->hin);
    p.slab = p.splits == 1 ? dw : (float*)slabs;
    if (hlr_pairs()) hipLaunchKernelGGL(conv_wgrad_hlrp_kernel, dim3(p.ntiles_n * p.ntiles_c * p.splits), dim3(512), 0, st, p);
    else hipLaunchKernelGGL(conv_wgrad_hlr_kernel<1>, dim3(p.ntiles_n * p.ntiles_c * p.splits), dim3(512), 0, st, p);
    if (p.splits > 1) launch_wgrad_reduce((const float*)slabs, dw, (int64_t)c->cout * p.K / 4, p.splits, st);
    return dcn::check_launch();
}
}  // namespace

// Which convolutions' weight gradients take the hl32 kernel: whole 256-channel output tiles, enough K columns and pixels.
extern "C" int dcn_conv_wgrad_hl_eligible(const dcn_conv_desc* c) {
    if (dcn::tuning().wgrad_hl == 0) return 0;
    if (use_hlr(c)) return 1;                    // the narrow 3 x 3 layers: row-window kernel
    if (!wgrad_hl_supported(c)) return 0;
    if (dcn::tuning().wgrad_hl == 2) return 1;   // (tests: every supported convolution)
    // (round 3 asked for M >= 16384: at two images the step lost 0.5 % -- with the main stream's GEMMs on 38-76 tiles the
    // fp32-operand kernel's launches ran for free on the idle CUs.  Round 5: the small-tile forward / dgrad kernel fills the
    // chip at that size too, the weight gradients are then paid in full, and per launch this kernel is the faster one at two
    // images already -- layer 4 137 vs 187 us, layer 3 52 vs 61 us, profiles/r5a_wgrad_n2.txt -- and at one: 82 vs 113 us,
    // 39 vs 41 us, profiles/r5b_wgrad_n1.txt)
    const int64_t min_m = dcn::tuning().wgrad_hl_min_m > 0 ? dcn::tuning().wgrad_hl_min_m : 4096;
    // (round 6: reduction length >= 128 instead of 1024 -- the 1 x 1 convolutions into >= 256 channels: at eight images the
    // layer3 / layer4 downsample 46.6 -> 29.3 / 93.8 -> 54.1 us, ResNet50-8s conv3 256 -> 1024 285 -> 143 us, 512 -> 2048 783 -> 435 us)
    return ((c->cout % 256) == 0 && c->kh * c->kw * c->cin >= std::min(1024, dcn::tuning().hl_min_k) &&
            (int64_t)c->n * c->hout * c->wout >= min_m) ? 1 : 0;
}

extern "C" int dcn_conv_wgrad_hl_kind(const dcn_conv_desc* c) { return use_hlr(c) ? 2 : (wgrad_hl_supported(c) ? 1 : 0); }

extern "C" size_t dcn_conv_wgrad_workspace_hl(const dcn_conv_desc* c) {
    // (the larger of the two kernels' slabs wherever both support the convolution: a plan sized under one setting of
    // DCN_WGRAD_HLR stays valid under another)
    size_t bytes = 0;
    int sps;
    if (wgrad_hlr_supported(c))   // (either form of the row-window kernel: DCN_WGRAD_HLR_PAIRS may change after a plan was sized)
        bytes = (size_t)std::max(wgrad_hlr_splits(c, &sps, true), wgrad_hlr_splits(c, &sps, false)) * c->cout * 9 * c->cin * sizeof(float);
    if (wgrad_hl_supported(c))
        bytes = std::max(bytes, (size_t)wgrad_hl_splits(c, &sps) * c->cout * c->kh * c->kw * c->cin * sizeof(float));
    return bytes;
}

// x_hl: hl32 image of the convolution's input [n, hin, win, cin], scaled by pow2_scale(*x_absmax); dout_hl: hl32 image of
// the output gradient [n, hout, wout, cout], scaled by pow2_scale(*dout_absmax) (dcn_split_act_hl32 or the producing
// batch-norm passes).  dw: [cout][kh][kw][cin]; slabs: dcn_conv_wgrad_workspace_hl bytes.
extern "C" int dcn_conv_wgrad_hl(const dcn_conv_desc* c, const void* x_hl, const float* x_absmax, const void* dout_hl,
                                 const float* dout_absmax, float* dw, void* slabs, void* stream) {
    if (!x_hl || !dout_hl || !dw || !slabs) return DCN_E_INVALID;
    if (use_hlr(c)) return launch_wgrad_hlr(c, x_hl, x_absmax, dout_hl, dout_absmax, dw, slabs, (hipStream_t)stream);
    if (!wgrad_hl_supported(c)) return DCN_E_UNSUPPORTED;
    WgradHl p;
    p.xh = x_hl; p.dh = dout_hl; p.x_absmax = x_absmax; p.d_absmax = dout_absmax;
    p.x_bytes = (unsigned)((int64_t)c->n * c->hin * c->win * c->cin * 4);
    p.d_bytes = (unsigned)((int64_t)c->n * c->hout * c->wout * c->ldc * 4);
    p.hin = c->hin; p.win = c->win; p.cin = c->cin; p.cout = c->cout; p.kh = c->kh; p.kw = c->kw; p.pad = c->pad; p.dil = c->dil;
    p.ldo = c->ldc; p.M = c->n * c->hout * c->wout; p.K = c->kh * c->kw * c->cin; p.cpt = c->cin / 32;
    p.splits = wgrad_hl_splits(c, &p.stages_per_split);
    p.setprio = dcn::tuning().hl_setprio;
    p.ntiles_n = dcn::ceil_div(c->cout, 256); p.ntiles_k = dcn::ceil_div(p.K, 256);
    p.div_hw = make_fastdiv(c->hout * c->wout); p.div_w = make_fastdiv(c->wout);
    p.div_cpt = make_fastdiv(p.cpt); p.div_kw = make_fastdiv(c->kw);
    p.slab = p.splits == 1 ? dw : (float*)slabs;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv_wgrad_hl_kernel, dim3(p.ntiles_n * p.ntiles_k * p.splits), dim3(512), 0, st, p);
    if (p.splits > 1) launch_wgrad_reduce((const float*)slabs, dw, (int64_t)c->cout * p.K / 4, p.splits, st);
    return dcn::check_launch();
}
