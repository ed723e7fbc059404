// K1/K3/K4/K5/K6: NHWC fp32 convolutions as implicit GEMMs on the gfx950 matrix cores
// (v_mfma_f32_32x32x2_f32: exact-fp32 MFMA, 64 FLOP/clk/SIMD -- the only matrix path that can hold the
// 1e-4 parity bound against the fp32 CPU reference; gfx950 has no xf32).
//
//   forward : out[m][n]  = sum_{tap,c} in[pix(m,tap)][c] * W[n][tap][c]            (+ bias[n])
//   dgrad   : din[m][c]  = sum_{tap,n} dout[pixT(m,tap)][n] * Wt[c][tap][n]        (+ add[m][c])
//   wgrad   : dW[n][tap][c] = sum_m dout[m][n] * in[pix(m,tap)][c]
//
// forward and dgrad are the same "gather-GEMM" kernel (M = N*H*W destination pixels, N = destination
// channels, K = taps * source channels); dgrad gathers with the transposed index map and a channel-
// transposed weight copy, so stride-2 convolutions need no separate kernel.  wgrad reduces over pixels
// (split over the grid, partial slabs + a deterministic second pass).
//
// Tiling (wave64): 256 work-items = 4 wavefronts in a 2x2 arrangement, workgroup tile 128 x 128, each
// wavefront 64 x 64 = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).  K advances 16 at a time through a
// double-buffered LDS tile (one barrier per K step); global->LDS goes through registers because the
// gather needs per-row zero fill (padding / dilation halo / ragged M) which LDS-DMA cannot express.
//   gather-GEMM LDS image: [row][16 k + 4 pad] so that one ds_read_b128 per lane fetches 4 consecutive k
//   for row (lane & 31) at k-offset 4*(lane >> 5): MFMA step j then contracts k = {j, 4 + j} -- any
//   k permutation is legal as long as A and B use the same one.  The 80-byte row pitch makes the
//   16-lane ds_read_b128 groups hit 16 distinct 16-byte slots (conflict-free).
//   wgrad LDS image: [pixel][128 channels], fragments by ds_read_b64 (lane i holds channels 2i, 2i+1 of
//   pixel 2q + (lane >> 5)): 64 consecutive dwords per half-wave, conflict-free without padding.
#include "dcn_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16, LDK = BK + 4, NT = 256;

struct GemmConv {
    const float* src;   // [n, hs, ws, cs] NHWC
    const float* wm;    // [cd][taps][cs]
    const float* bias;  // [cd] or null
    const float* add;   // [M][ldc] or null
    float* dst;         // [M][ldc]
    float* bn_partial;  // [mtiles][2][cd] or null
    int hs, ws, cs, hd, wd, cd, kh, kw, stride, pad, dil, ldc, M, K, transposed, mtiles, ntiles;
};

// bijective XCD-aware remap: consecutive logical tiles (sharing an M tile) land on the same XCD / L2
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

__global__ void __launch_bounds__(NT)
conv_gemm_kernel(GemmConv p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDK];
    constexpr int kStage = (BM + BN) * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm_ = wv >> 1, wn_ = wv & 1;
    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-thread gather coordinates: this thread stages rows r0, r0+64 (A and B) at k-quad kq
    const int kq = tid & 3, r0 = tid >> 2;
    int by[2], bx[2], pixbase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + r0 + 64 * j;
        if (m < p.M) {
            const int hw = p.hd * p.wd;
            const int img = m / hw, rem = m - img * hw;
            const int y = rem / p.wd, x = rem - y * p.wd;
            by[j] = p.transposed ? y + p.pad : y * p.stride - p.pad;
            bx[j] = p.transposed ? x + p.pad : x * p.stride - p.pad;
            pixbase[j] = img * p.hs * p.ws;
        } else {
            by[j] = -(1 << 28); bx[j] = -(1 << 28); pixbase[j] = 0;
        }
    }
    const int nk = (p.K + BK - 1) / BK;
    float4 ra[2], rb[2];

    auto load_tile = [&](int kt) {
        const int k = kt * BK + kq * 4;
        const bool kval = k < p.K;
        const int tap = k / p.cs, c = k - tap * p.cs;
        const int r = tap / p.kw, s = tap - r * p.kw;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int sy, sx;
            bool ok = kval;
            if (p.transposed) {
                const int ny = by[j] - r * p.dil, nx = bx[j] - s * p.dil;
                sy = ny / p.stride; sx = nx / p.stride;
                ok = ok && ny >= 0 && nx >= 0 && (sy * p.stride == ny) && (sx * p.stride == nx);
            } else {
                sy = by[j] + r * p.dil; sx = bx[j] + s * p.dil;
                ok = ok && sy >= 0 && sx >= 0;
            }
            ok = ok && sy < p.hs && sx < p.ws;
            if (ok) {
                const int64_t off = (int64_t)(pixbase[j] + sy * p.ws + sx) * p.cs + c;
                ra[j] = *reinterpret_cast<const float4*>(p.src + off);
            } else {
                ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const int n = n0 + r0 + 64 * j;
            if (kval && n < p.cd) rb[j] = *reinterpret_cast<const float4*>(p.wm + (int64_t)n * p.K + k);
            else rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int stage) {
        float* a = lds + stage * kStage;
        float* b = a + BM * LDK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<float4*>(a + (r0 + 64 * j) * LDK + kq * 4) = ra[j];
            *reinterpret_cast<float4*>(b + (r0 + 64 * j) * LDK + kq * 4) = rb[j];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int fi = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const float* a = lds + cur * kStage + (wm_ * 64 + fi) * LDK + 4 * fh;
        const float* b = lds + cur * kStage + BM * LDK + (wn_ * 64 + fi) * LDK + 4 * fh;
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            const float4 a0 = *reinterpret_cast<const float4*>(a + ks * 8);
            const float4 a1 = *reinterpret_cast<const float4*>(a + 32 * LDK + ks * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(b + ks * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(b + 32 * LDK + ks * 8);
            const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
            const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][j], bv[0][j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][j], bv[1][j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][j], bv[0][j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][j], bv[1][j], acc[1][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D fragment -> NHWC rows (32 consecutive channels per half-wave = 128 B segments)
    float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int col = n0 + wn_ * 64 + tn * 32 + fi;
        const bool cok = col < p.cd;
        const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm_ * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                float v = acc[tm][tn][r] + bv;
                if (cok && row < p.M) {
                    const int64_t o = (int64_t)row * p.ldc + col;
                    if (p.add) v += p.add[o];
                    p.dst[o] = v;
                }
                csum[tn] += acc[tm][tn][r];
                csq[tn] = fmaf(acc[tm][tn][r], acc[tm][tn][r], csq[tn]);
            }
        }
    }
    if (p.bn_partial) {
        // rows >= M and columns >= cd are exactly zero in acc (zero-filled fragments), so no masking is needed
        float* red = lds;  // [2 (wm)][2 (sum, sq)][128]
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            csum[tn] += __shfl_xor(csum[tn], 32, 64);
            csq[tn] += __shfl_xor(csq[tn], 32, 64);
            if (fh == 0) {
                red[(wm_ * 2 + 0) * BN + wn_ * 64 + tn * 32 + fi] = csum[tn];
                red[(wm_ * 2 + 1) * BN + wn_ * 64 + tn * 32 + fi] = csq[tn];
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.cd) {
            const float s = red[(0 * 2 + 0) * BN + tid] + red[(1 * 2 + 0) * BN + tid];
            const float q = red[(0 * 2 + 1) * BN + tid] + red[(1 * 2 + 1) * BN + tid];
            p.bn_partial[((int64_t)mt * 2 + 0) * p.cd + n0 + tid] = s;
            p.bn_partial[((int64_t)mt * 2 + 1) * p.cd + n0 + tid] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------- wgrad
constexpr int WK = 16;  // pixels per reduction step

struct WgradConv {
    const float* in;    // [n, hin, win, cin]
    const float* dout;  // [M][ldo]
    float* slab;        // [splits][cout][K]
    int hin, win, cin, hout, wout, cout, kh, kw, stride, pad, dil, ldo, M, K, splits, rows_per_split, ntiles_n,
        ntiles_k;
};

__global__ void __launch_bounds__(NT)
conv_wgrad_kernel(WgradConv p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * WK * 128];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm_ = wv >> 1, wn_ = wv & 1;
    const int tiles = p.ntiles_n * p.ntiles_k;
    const int bid = xcd_remap(blockIdx.x, tiles * p.splits);
    const int split = bid / tiles, tile = bid - split * tiles;
    const int tn_ = tile / p.ntiles_k, tk_ = tile - tn_ * p.ntiles_k;
    const int n0 = tn_ * 128, j0 = tk_ * 128;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    // this thread stages channel quad c4 of pixel rows p0, p0 + 8
    const int c4 = tid & 31, p0 = tid >> 5;
    const int ncol = n0 + c4 * 4;
    const bool nval = ncol < p.ldo;          // dout rows are padded to ldo (multiple of 4) with zeros
    const int kcol = j0 + c4 * 4;
    const bool kval = kcol < p.K;
    const int tap = kcol / p.cin, cc = kcol - tap * p.cin;
    const int tr = tap / p.kw, ts = tap - tr * p.kw;
    const int hw = p.hout * p.wout;
    float4 rd[2], rx[2];

    auto load_tile = [&](int m_base) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m_base + p0 + 8 * j;
            rd[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            rx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end) {
                if (nval) rd[j] = *reinterpret_cast<const float4*>(p.dout + (int64_t)m * p.ldo + ncol);
                if (kval) {
                    const int img = m / hw, rem = m - img * hw;
                    const int y = rem / p.wout, x = rem - y * p.wout;
                    const int sy = y * p.stride - p.pad + tr * p.dil, sx = x * p.stride - p.pad + ts * p.dil;
                    if (sy >= 0 && sy < p.hin && sx >= 0 && sx < p.win)
                        rx[j] = *reinterpret_cast<const float4*>(
                            p.in + ((int64_t)(img * p.hin + sy) * p.win + sx) * p.cin + cc);
                }
            }
        }
    };
    auto store_tile = [&](int stage) {
        float* d = lds + stage * 2 * WK * 128;
        float* x = d + WK * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<float4*>(d + (p0 + 8 * j) * 128 + c4 * 4) = rd[j];
            *reinterpret_cast<float4*>(x + (p0 + 8 * j) * 128 + c4 * 4) = rx[j];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fi = lane & 31, fh = lane >> 5;
    const int nsteps = (m_end - m_begin + WK - 1) / WK;
    if (nsteps > 0) {
        load_tile(m_begin);
        store_tile(0);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        if (st + 1 < nsteps) load_tile(m_begin + (st + 1) * WK);
        const float* d = lds + cur * 2 * WK * 128 + fh * 128 + wm_ * 64 + 2 * fi;
        const float* x = lds + cur * 2 * WK * 128 + WK * 128 + fh * 128 + wn_ * 64 + 2 * fi;
#pragma unroll
        for (int q = 0; q < WK / 2; ++q) {
            const float2 a = *reinterpret_cast<const float2*>(d + 2 * q * 128);
            const float2 b = *reinterpret_cast<const float2*>(x + 2 * q * 128);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
        }
        if (st + 1 < nsteps) store_tile(cur ^ 1);
        __syncthreads();
    }
    // fragment (tm, r) <-> output channel n0 + wm*64 + 2*row_r + tm ; (tn, lane) <-> k column j0 + wn*64 + 2*fi + tn
    float* out = p.slab + (int64_t)split * p.cout * p.K;
    const int kc = j0 + wn_ * 64 + 2 * fi;
    if (kc < p.K) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm_ * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * fh) + tm;
                if (n < p.cout)
                    *reinterpret_cast<float2*>(out + (int64_t)n * p.K + kc) = make_float2(acc[tm][0][r], acc[tm][1][r]);
            }
        }
    }
}

// dw[i] = sum_s slab[s][i]   (fixed order: deterministic)
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int64_t n4, int splits) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 a = reinterpret_cast<const float4*>(slab)[i];
    for (int s = 1; s < splits; ++s) {
        const float4 b = reinterpret_cast<const float4*>(slab)[(int64_t)s * n4 + i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(dw)[i] = a;
}

// wt[c][tap][n] = w[n][tap][c]
__global__ void __launch_bounds__(256)
transpose_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int taps, int cin, int ldn) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, c = c0 + tx;
        tile[i][tx] = (n < cout && c < cin) ? w[((int64_t)n * taps + tap) * cin + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, n = n0 + tx;
        if (c < cin && n < ldn) wt[((int64_t)c * taps + tap) * ldn + n] = tile[tx][i];  // n >= cout: zero pad
    }
}

int wgrad_splits(const dcn_conv_desc* c, int* rows_per_split) {
    const int M = c->n * c->hout * c->wout, K = c->kh * c->kw * c->cin;
    const int tiles = dcn::ceil_div(c->cout, 128) * dcn::ceil_div(K, 128);
    // aim for ~4 workgroup waves over 256 CUs x 2 resident workgroups, at least 8 reduction steps per workgroup
    int splits = dcn::ceil_div(2048, tiles);
    const int max_by_rows = (M / (8 * WK)) > 1 ? (M / (8 * WK)) : 1;
    if (splits > max_by_rows) splits = max_by_rows;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    int rps = dcn::ceil_div(dcn::ceil_div(M, splits), WK) * WK;
    splits = dcn::ceil_div(M, rps);
    *rows_per_split = rps;
    return splits;
}

bool valid_desc(const dcn_conv_desc* c) {
    return c && c->n > 0 && c->hin > 0 && c->win > 0 && c->cin > 0 && (c->cin % 4) == 0 && c->hout > 0 && c->wout > 0 &&
           c->cout > 0 && c->kh > 0 && c->kw > 0 && c->stride > 0 && c->dil > 0 && c->pad >= 0 && c->ldc >= c->cout;
}

}  // namespace

extern "C" int dcn_conv_num_mtiles(const dcn_conv_desc* c) {
    if (!valid_desc(c)) return DCN_E_INVALID;
    return dcn::ceil_div(c->n * c->hout * c->wout, BM);
}

extern "C" int dcn_conv_forward(const dcn_conv_desc* c, const float* in, const float* w, const float* bias, float* out,
                                float* bn_partial, void* stream) {
    if (!valid_desc(c) || !in || !w || !out) return DCN_E_INVALID;
    GemmConv p;
    p.src = in; p.wm = w; p.bias = bias; p.add = nullptr; p.dst = out; p.bn_partial = bn_partial;
    p.hs = c->hin; p.ws = c->win; p.cs = c->cin; p.hd = c->hout; p.wd = c->wout; p.cd = c->cout;
    p.kh = c->kh; p.kw = c->kw; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.ldc = c->ldc;
    p.M = c->n * c->hout * c->wout; p.K = c->kh * c->kw * c->cin; p.transposed = 0;
    p.mtiles = dcn::ceil_div(p.M, BM); p.ntiles = dcn::ceil_div(p.cd, BN);
    hipLaunchKernelGGL(conv_gemm_kernel, dim3(p.mtiles * p.ntiles), dim3(NT), 0, (hipStream_t)stream, p);
    return dcn::check_launch();
}

// The description is the FORWARD convolution's; dout is [n,hout,wout,ld = c->ldc], din is [n,hin,win,cin].
extern "C" int dcn_conv_dgrad(const dcn_conv_desc* c, const float* dout, const float* wt, const float* add, float* din,
                              void* stream) {
    if (!valid_desc(c) || !dout || !wt || !din || (c->ldc % 4) != 0) return DCN_E_INVALID;
    GemmConv p;
    p.src = dout; p.wm = wt; p.bias = nullptr; p.add = add; p.dst = din; p.bn_partial = nullptr;
    p.hs = c->hout; p.ws = c->wout; p.cs = c->ldc;   // source channels = (padded) forward output channels
    p.hd = c->hin; p.wd = c->win; p.cd = c->cin;
    p.kh = c->kh; p.kw = c->kw; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.ldc = c->cin;
    p.M = c->n * c->hin * c->win; p.K = c->kh * c->kw * c->ldc; p.transposed = 1;
    p.mtiles = dcn::ceil_div(p.M, BM); p.ntiles = dcn::ceil_div(p.cd, BN);
    hipLaunchKernelGGL(conv_gemm_kernel, dim3(p.mtiles * p.ntiles), dim3(NT), 0, (hipStream_t)stream, p);
    return dcn::check_launch();
}

extern "C" size_t dcn_conv_wgrad_workspace(const dcn_conv_desc* c) {
    if (!valid_desc(c)) return 0;
    int rps;
    const int splits = wgrad_splits(c, &rps);
    return (size_t)splits * c->cout * c->kh * c->kw * c->cin * sizeof(float);
}

extern "C" int dcn_conv_wgrad(const dcn_conv_desc* c, const float* in, const float* dout, float* dw, void* slabs,
                              void* stream) {
    if (!valid_desc(c) || !in || !dout || !dw || !slabs || (c->ldc % 4) != 0) return DCN_E_INVALID;
    WgradConv p;
    p.in = in; p.dout = dout;
    p.hin = c->hin; p.win = c->win; p.cin = c->cin; p.hout = c->hout; p.wout = c->wout; p.cout = c->cout;
    p.kh = c->kh; p.kw = c->kw; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.ldo = c->ldc;
    p.M = c->n * c->hout * c->wout; p.K = c->kh * c->kw * c->cin;
    p.splits = wgrad_splits(c, &p.rows_per_split);
    p.ntiles_n = dcn::ceil_div(c->cout, 128); p.ntiles_k = dcn::ceil_div(p.K, 128);
    p.slab = p.splits == 1 ? dw : (float*)slabs;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(p.ntiles_n * p.ntiles_k * p.splits), dim3(NT), 0, st, p);
    if (p.splits > 1) {
        const int64_t n4 = (int64_t)c->cout * p.K / 4;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)dcn::ceil_div64(n4, 256)), dim3(256), 0, st,
                           (const float*)slabs, dw, n4, p.splits);
    }
    return dcn::check_launch();
}

extern "C" int dcn_transpose_weight(const float* w, float* wt, int cout, int taps, int cin, int ldn, void* stream) {
    if (!w || !wt || cout < 1 || taps < 1 || cin < 1 || ldn < cout) return DCN_E_INVALID;
    hipLaunchKernelGGL(transpose_weight_kernel, dim3(dcn::ceil_div(cin, 32), dcn::ceil_div(ldn, 32), taps), dim3(256),
                       0, (hipStream_t)stream, w, wt, cout, taps, cin, ldn);
    return dcn::check_launch();
}
