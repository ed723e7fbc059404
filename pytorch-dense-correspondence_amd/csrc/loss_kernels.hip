// K9: fused pixelwise-contrastive loss (gather -> L2 distance -> hinge -> wave/block reduction) and its
// backward (scatter-add), for gfx950.
//
// Reference arithmetic being replaced (dense_correspondence/loss_functions/):
//   pixelwise_contrastive_loss.py:132-167  match_loss                  1/P_m * sum ||A[a_i]-B[b_i]||^2
//   pixelwise_contrastive_loss.py:171-213  non_match_descriptor_loss   l_j = max(0, M - ||A[a_j]-B[b_j]||_2)^2, #{l_j != 0}
//   pixelwise_contrastive_loss.py:307-352  l2_pixel_loss               w_j = min(||uv(gt_j)-uv(b_j)||, M_pixel)/M_pixel
//   loss_composer.py:70-212                composition + hard-negative scaling
//
// HBM-bound gather: a group of LP = 4 / 8 / 16 / 32 lanes per pixel pair reads the pair's two int64 indices, then the two
// D-float descriptors straight from the [pairs, HW, D] descriptor maps as ONE contiguous 4*D-byte run each (the
// channels_last descriptor layout makes a descriptor contiguous; lane k of the group takes component k), reduces the
// squared distance inside the group with xor-shuffles, the loss terms across the 64-lane wavefront with shuffles and
// across the workgroup through LDS, and writes ONE partial per workgroup.  A single-workgroup finalize kernel adds the
// partials in a fixed order in fp64 (run-to-run deterministic forward) and composes the 5-tuple on the device, so the
// hard-negative count never travels to the host (the reference syncs twice per step at pcl.py:210-211).  The backward
// scatter-adds a pair's gradient as D consecutive fp32 atomics per descriptor (one cache line per group).
// Algorithmic traffic per pair, fwd+bwd: 2*8 B indices + 2*4D B reads (x2, fwd and bwd) + 2*4D B atomics.
#include "dcn_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kItems = 8;  // pairs per lane group

// Lane mapping: LP = 4 / 8 / 16 / 32 consecutive lanes share one pixel pair and each owns the descriptor components
// sub, sub + LP, ... -- so the two gathers of a pair are ONE contiguous 4*D-byte run across the group (not D strided
// loads of one lane), the squared distance is a log2(LP)-step xor-shuffle reduction inside the group, and the backward
// scatter-add of a pair's gradient is one run of D consecutive fp32 atomics per descriptor: a wave-instruction touches
// 64 / LP cache lines with LP dwords each instead of 64 lines with one dword each (16x fewer L2 atomic line operations
// at D = 16).  SINGLE: D <= LP, every lane holds at most one component (kept in a register between the two phases).
template <int LP> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
constexpr int pairs_per_block(int lp) { return kThreads / lp * kItems; }
inline int lanes_per_pair(int d) { return d <= 4 ? 4 : (d <= 8 ? 8 : (d <= 16 ? 16 : 32)); }

__device__ __forceinline__ float pixel_weight(int64_t gt, int64_t nb, int width, float m_pixel) {
    const float du = (float)((gt % width) - (nb % width));
    const float dv = (float)((gt / width) - (nb / width));
    return fminf(sqrtf(du * du + dv * dv), m_pixel) / m_pixel;
}

// grid = (chunks, 4*num_pairs).  Every workgroup writes its partial, chunks beyond the list's end write zeros.
template <int LP, bool SINGLE>
__global__ void __launch_bounds__(kThreads)
loss_fwd_kernel(const float* __restrict__ A, const float* __restrict__ B, int64_t hw, int D,
                const int64_t* __restrict__ idx_a, const int64_t* __restrict__ idx_b,
                const int64_t* __restrict__ offsets, dcn_loss_config cfg, double* __restrict__ part_sum,
                int* __restrict__ part_cnt, float* __restrict__ per_term, int* __restrict__ part_oob,
                float* __restrict__ rec_d, float* __restrict__ rec_s) {
    // rec_d [total][D], rec_s [total] (optional, together): per pixel pair the difference vector a - b and the factor s with
    //     d loss / d a = coef(list, image pair) * s * (a - b)      (= - d loss / d b)
    // -- everything of the pair's gradient that does not depend on the hard-negative counts of the whole image pair.  The
    // backward pass (loss_bwd_saved_kernel) then reads these coalesced records instead of gathering both descriptors again.
    // part_oob[workgroup]: 1 if the workgroup met an index outside [0, hw) -- OR-ed into the status word by the finalize
    // kernel (no cleared status word needed in front of this launch)
    __shared__ double s_sum[kThreads / dcn::kWave];
    __shared__ int s_cnt[kThreads / dcn::kWave];
    __shared__ int s_oob;
    if (threadIdx.x == 0) s_oob = 0;
    __syncthreads();
    constexpr int GROUPS = kThreads / LP, PPB = GROUPS * kItems;
    const int seg = blockIdx.y, p = seg >> 2, t = seg & 3;
    const int64_t beg = offsets[seg], len = offsets[seg + 1] - beg;
    const int64_t chunk0 = (int64_t)blockIdx.x * PPB;
    const int grp = threadIdx.x / LP, sub = threadIdx.x % LP;
    double sum = 0.0;
    int cnt = 0;
    if (chunk0 < len) {
        const float* Ap = A + (int64_t)p * hw * D;
        const float* Bp = B + (int64_t)p * hw * D;
        const int64_t mbeg = offsets[4 * p], mlen = offsets[4 * p + 1] - mbeg;
        const int64_t per = (cfg.pixel_weight[t] && mlen > 0) ? len / mlen : 0;
        const float M = cfg.margin[t];
        float acc = 0.f;
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int64_t j = chunk0 + (int64_t)it * GROUPS + grp;
            // (every lane of a group takes the same branches: j, ia, ib are group-uniform; the shuffles below only mix
            //  lanes of one group, and all 64 lanes of the wavefront reach them)
            const bool live = j < len;
            const int64_t ia = live ? idx_a[beg + j] : -1, ib = live ? idx_b[beg + j] : -1;
            const bool skip = ia < 0 || ib < 0;        // the reference's `[-1]` "empty list" sentinel
            const bool oob = !skip && (ia >= hw || ib >= hw);
            if (oob && sub == 0) s_oob = 1;
            const bool ok = !skip && !oob;
            float s = 0.f;
            if (ok) {
                const float* a = Ap + ia * D;
                const float* b = Bp + ib * D;
                if (SINGLE) {
                    if (sub < D) {
                        const float df = a[sub] - b[sub];
                        s = df * df;
                        if (rec_d) rec_d[(beg + j) * D + sub] = df;
                    }
                } else {
                    for (int c = sub; c < D; c += LP) {
                        const float df = a[c] - b[c];
                        s = fmaf(df, df, s);
                        if (rec_d) rec_d[(beg + j) * D + c] = df;
                    }
                }
            }
            const float d2 = group_sum<LP>(s);
            float term = 0.f, sfac = 0.f;
            if (ok) {
                if (t == DCN_LIST_MATCH) {
                    term = d2;
                    acc += sub == 0 ? term : 0.f;
                    sfac = 2.f;
                } else {
                    float w = 1.f;
                    if (per > 0) {
                        const int64_t mi = j / per;
                        w = mi < mlen ? pixel_weight(idx_b[mbeg + mi], ib, cfg.image_width, cfg.m_pixel) : 0.f;
                    }
                    if (cfg.invert[t] == 2) {   // legacy hinge on the SQUARED distance, not squared again (pcl.py:399-404)
                        term = fmaxf(M - d2, 0.f);
                        sfac = (M - d2 > 0.f) ? -2.f : 0.f;
                    } else {
                        const float dist = sqrtf(d2);
                        const float hinge = cfg.invert[t] ? dist - M : M - dist;
                        const float h = fmaxf(hinge, 0.f);
                        term = h * h;
                        // (the expressions of loss_bwd_kernel: same bits; d||x||/dx := 0 at x = 0 like torch)
                        if (hinge > 0.f && dist > 0.f) sfac = (cfg.invert[t] ? 2.f : -2.f) * hinge / dist * w;
                    }
                    cnt += (sub == 0 && term != 0.f) ? 1 : 0;
                    acc += sub == 0 ? term * w : 0.f;
                }
            }
            if (per_term && live && sub == 0) per_term[beg + j] = term;
            if (rec_s && live && sub == 0) rec_s[beg + j] = sfac;
        }
        sum = (double)acc;
    }
    const double bs = dcn::block_sum<kThreads>(sum, s_sum);
    const int bc = dcn::block_sum<kThreads>(cnt, s_cnt);
    if (threadIdx.x == 0) {   // (block_sum's barriers order the s_oob writes before this read)
        part_sum[(int64_t)seg * gridDim.x + blockIdx.x] = bs;
        part_cnt[(int64_t)seg * gridDim.x + blockIdx.x] = bc;
        part_oob[(int64_t)seg * gridDim.x + blockIdx.x] = s_oob;
    }
}

// Scale factors shared by the finalize and the backward kernel (loss_composer.py:107-134, :179-187, :205-211).
struct PairScales {
    float match_coef;      // d loss_p / d (sum ||a-b||^2)
    float nonmatch_coef;   // d loss_p / d (S_masked + S_background)
    float blind_coef;      // d loss_p / d S_blind
};

__device__ __forceinline__ PairScales pair_scales(const dcn_loss_config& cfg, const int* h, const int64_t* len) {
    PairScales s;
    s.match_coef = 0.f; s.nonmatch_coef = 0.f; s.blind_coef = 0.f;
    if (cfg.compose == DCN_COMPOSE_WITHIN_SCENE) {
        s.match_coef = len[0] > 0 ? cfg.match_loss_weight * (float)(1.0 / (double)len[0]) : 0.f;
        int64_t scale;
        if (cfg.scale_by_hard_negatives) {
            scale = (int64_t)h[1] + h[2];
            if (scale < 1) scale = 1;
        } else {
            scale = (len[1] > 1 ? len[1] : 1) + (len[2] > 1 ? len[2] : 1);
        }
        s.nonmatch_coef = cfg.non_match_loss_weight * (float)(1.0 / (double)scale);
    } else if (cfg.compose == DCN_COMPOSE_RAW_SUMS) {
        s.match_coef = cfg.match_loss_weight;
        s.nonmatch_coef = cfg.non_match_loss_weight;
        s.blind_coef = cfg.non_match_loss_weight;
    } else {
        int64_t scale = cfg.scale_by_hard_negatives ? (int64_t)h[3] : len[3];
        if (scale < 1) scale = 1;
        s.blind_coef = len[3] > 0 ? (float)(1.0 / (double)scale) : 0.f;
    }
    return s;
}

// composes image pair p's 5-tuple from its four fp64 sums S[] and hard-negative counts hcnt[] (ONE work-item)
__device__ __forceinline__ void compose_pair(int p, const double* s_S, const int* s_h, const int64_t* __restrict__ offsets,
                                             const dcn_loss_config& cfg, float* __restrict__ terms, float* __restrict__ sums,
                                             int* __restrict__ hard_neg, double* __restrict__ pair_loss) {
    int64_t len[4];
    int h[4];
    for (int t = 0; t < 4; ++t) { len[t] = offsets[4 * p + t + 1] - offsets[4 * p + t]; h[t] = s_h[t]; }
    h[0] = (int)len[0];
    const PairScales sc = pair_scales(cfg, h, len);
    float out[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (cfg.compose == DCN_COMPOSE_WITHIN_SCENE) {
        const float match_loss = len[0] > 0 ? (float)(s_S[0] / (double)len[0]) : 0.f;
        const float Sk = (float)s_S[1], Sg = (float)s_S[2], Sb = (float)s_S[3];
        int64_t dk, dg, db;
        if (cfg.scale_by_hard_negatives) {
            dk = h[1] > 1 ? h[1] : 1; dg = h[2] > 1 ? h[2] : 1;
            db = len[3] > 0 ? (h[3] > 1 ? h[3] : 1) : 1;
        } else {
            dk = len[1] > 1 ? len[1] : 1; dg = len[2] > 1 ? len[2] : 1; db = len[3] > 1 ? len[3] : 1;
        }
        out[1] = match_loss;
        out[2] = Sk / (float)dk;
        out[3] = Sg / (float)dg;
        out[4] = Sb / (float)db;
        out[0] = cfg.match_loss_weight * match_loss + sc.nonmatch_coef * (Sk + Sg);
    } else if (cfg.compose == DCN_COMPOSE_RAW_SUMS) {
        out[1] = (float)s_S[0]; out[2] = (float)s_S[1]; out[3] = (float)s_S[2]; out[4] = (float)s_S[3];
        out[0] = sc.match_coef * out[1] + sc.nonmatch_coef * (out[2] + out[3] + out[4]);
    } else {
        const float Sb = (float)s_S[3];
        out[0] = sc.blind_coef * Sb;
        // different_object returns the scaled blind loss, across_scene the raw sum, as 5th element
        out[4] = cfg.compose == DCN_COMPOSE_DIFFERENT_OBJECT ? out[0] : Sb;
    }
    for (int k = 0; k < 5; ++k) terms[5 * p + k] = out[k];
    for (int t = 0; t < 4; ++t) { sums[4 * p + t] = (float)s_S[t]; hard_neg[4 * p + t] = h[t]; }
    pair_loss[p] = (double)out[0];
}

// One workgroup PER IMAGE PAIR: adds that pair's partials in a fixed order in fp64 and composes its 5-tuple; the per-pair
// losses go to pair_loss[] and a second, tiny kernel averages them in pair order (deterministic; a single workgroup walking
// all pairs took 200 us at 32 pairs x 110 000 pixel pairs).
// finalizes image pair p (all work-items of the workgroup take part); returns, in work-item 0, whether the pair met an
// out-of-range index
__device__ __forceinline__ int finalize_pair(int p, const double* __restrict__ part_sum, const int* __restrict__ part_cnt,
                                             const int* __restrict__ part_oob, int chunks, const int64_t* __restrict__ offsets,
                                             const dcn_loss_config& cfg, float* __restrict__ terms, float* __restrict__ sums,
                                             int* __restrict__ hard_neg, double* __restrict__ pair_loss, double* s_sum, int* s_cnt,
                                             double* s_S, int* s_h) {
    int oob = 0;
    for (int t = 0; t < 4; ++t) {
        double a = 0.0;
        int c = 0, o = 0;
        const int64_t base = (int64_t)(4 * p + t) * chunks;
        for (int i = threadIdx.x; i < chunks; i += kThreads) { a += part_sum[base + i]; c += part_cnt[base + i]; o |= part_oob[base + i]; }
        a = dcn::block_sum<kThreads>(a, s_sum);
        c = dcn::block_sum<kThreads>(c, s_cnt);
        o = dcn::block_sum<kThreads>(o, s_cnt);
        if (threadIdx.x == 0) { s_S[t] = a; s_h[t] = c; oob |= o; }
    }
    __syncthreads();
    if (threadIdx.x != 0) return 0;
    compose_pair(p, s_S, s_h, offsets, cfg, terms, sums, hard_neg, pair_loss);
    return oob ? 1 : 0;
}

// many pairs: one workgroup per pair (pair_oob[p] = out-of-range flag), then loss_mean_kernel
__global__ void __launch_bounds__(kThreads)
loss_finalize_kernel(const double* __restrict__ part_sum, const int* __restrict__ part_cnt, const int* __restrict__ part_oob,
                     int chunks, const int64_t* __restrict__ offsets, dcn_loss_config cfg, float* __restrict__ terms,
                     float* __restrict__ sums, int* __restrict__ hard_neg, double* __restrict__ pair_loss,
                     int* __restrict__ pair_oob) {
    __shared__ double s_sum[kThreads / dcn::kWave];
    __shared__ int s_cnt[kThreads / dcn::kWave];
    __shared__ double s_S[4];
    __shared__ int s_h[4];
    const int o = finalize_pair(blockIdx.x, part_sum, part_cnt, part_oob, chunks, offsets, cfg, terms, sums, hard_neg, pair_loss,
                                s_sum, s_cnt, s_S, s_h);
    if (threadIdx.x == 0) pair_oob[blockIdx.x] = o;
}

__global__ void __launch_bounds__(64)
loss_mean_kernel(const double* __restrict__ pair_loss, const int* __restrict__ pair_oob, int num_pairs, float* __restrict__ loss,
                 int* __restrict__ status) {
    if (threadIdx.x != 0) return;
    double total = 0.0;
    int o = 0;
    for (int p = 0; p < num_pairs; ++p) { total += pair_loss[p]; o |= pair_oob[p]; }   // pair order: deterministic
    loss[0] = (float)(total / (double)num_pairs);
    status[0] = o;
}

// few pairs (the training configurations: 1-8 per step): ONE workgroup -- each wavefront reduces (pair, term) sums with a
// fixed shuffle tree, then one work-item per pair composes its 5-tuple and work-item 0 averages in pair order and writes the
// status word: one launch instead of three (status clear, finalize, mean), no serial walk over the pairs
__global__ void __launch_bounds__(kThreads)
loss_finalize_all_kernel(const double* __restrict__ part_sum, const int* __restrict__ part_cnt, const int* __restrict__ part_oob,
                         int chunks, int num_pairs, const int64_t* __restrict__ offsets, dcn_loss_config cfg,
                         float* __restrict__ terms, float* __restrict__ sums, int* __restrict__ hard_neg,
                         double* __restrict__ pair_loss, float* __restrict__ loss, int* __restrict__ status) {
    __shared__ double s_S[8][4];
    __shared__ int s_h[8][4];
    __shared__ int s_o[8][4];
    const int lane = threadIdx.x & (dcn::kWave - 1), wv = threadIdx.x / dcn::kWave;
    for (int q = wv; q < 4 * num_pairs; q += kThreads / dcn::kWave) {
        double a = 0.0;
        int c = 0, o = 0;
        const int64_t base = (int64_t)q * chunks;
        for (int i = lane; i < chunks; i += dcn::kWave) { a += part_sum[base + i]; c += part_cnt[base + i]; o |= part_oob[base + i]; }
#pragma unroll
        for (int m = dcn::kWave / 2; m >= 1; m >>= 1) {
            a += __shfl_xor(a, m, dcn::kWave);
            c += __shfl_xor(c, m, dcn::kWave);
            o |= __shfl_xor(o, m, dcn::kWave);
        }
        if (lane == 0) { s_S[q >> 2][q & 3] = a; s_h[q >> 2][q & 3] = c; s_o[q >> 2][q & 3] = o; }
    }
    __syncthreads();
    if ((int)threadIdx.x < num_pairs) compose_pair(threadIdx.x, s_S[threadIdx.x], s_h[threadIdx.x], offsets, cfg, terms, sums, hard_neg, pair_loss);
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        int o = 0;
        for (int p = 0; p < num_pairs; ++p) {
            total += pair_loss[p];
            o |= s_o[p][0] | s_o[p][1] | s_o[p][2] | s_o[p][3];
        }
        loss[0] = (float)(total / (double)num_pairs);
        status[0] = o ? 1 : 0;
    }
}

// grid = (chunks, 4*num_pairs).  Scatter-adds d loss / d descriptor with hardware fp32 atomics (lane mapping above).
template <int LP, bool SINGLE>
__global__ void __launch_bounds__(kThreads)
loss_bwd_kernel(const float* __restrict__ A, const float* __restrict__ B, int64_t hw, int D, int num_pairs,
                const int64_t* __restrict__ idx_a, const int64_t* __restrict__ idx_b,
                const int64_t* __restrict__ offsets, dcn_loss_config cfg, const int* __restrict__ hard_neg,
                const float* __restrict__ grad_loss, const float* __restrict__ pair_grad,
                float* __restrict__ gA, float* __restrict__ gB) {
    constexpr int GROUPS = kThreads / LP, PPB = GROUPS * kItems;
    const int seg = blockIdx.y, p = seg >> 2, t = seg & 3;
    const int64_t beg = offsets[seg], len = offsets[seg + 1] - beg;
    const int64_t chunk0 = (int64_t)blockIdx.x * PPB;
    if (chunk0 >= len) return;
    const int grp = threadIdx.x / LP, sub = threadIdx.x % LP;
    int64_t lens[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lens[k] = offsets[4 * p + k + 1] - offsets[4 * p + k];
    float coef = 1.f;
    if (!pair_grad) {
        int h[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = hard_neg[4 * p + k];
        const PairScales sc = pair_scales(cfg, h, lens);
        coef = t == DCN_LIST_MATCH ? sc.match_coef : (t == DCN_LIST_BLIND ? sc.blind_coef : sc.nonmatch_coef);
        if (coef == 0.f) return;
        coef *= grad_loss[0] / (float)num_pairs;
    }
    const float* Ap = A + (int64_t)p * hw * D;
    const float* Bp = B + (int64_t)p * hw * D;
    float* gAp = gA + (int64_t)p * hw * D;
    float* gBp = gB + (int64_t)p * hw * D;
    const int64_t mbeg = offsets[4 * p], mlen = lens[0];
    const int64_t per = (cfg.pixel_weight[t] && mlen > 0) ? len / mlen : 0;
    const float M = cfg.margin[t];
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
        const int64_t j = chunk0 + (int64_t)it * GROUPS + grp;
        const bool live = j < len;
        const int64_t ia = live ? idx_a[beg + j] : -1, ib = live ? idx_b[beg + j] : -1;
        const bool ok = (uint64_t)ia < (uint64_t)hw && (uint64_t)ib < (uint64_t)hw;
        const float* a = Ap + (ok ? ia : 0) * D;
        const float* b = Bp + (ok ? ib : 0) * D;
        float df = 0.f, s = 0.f;
        if (ok) {
            if (SINGLE) {
                if (sub < D) { df = a[sub] - b[sub]; s = df * df; }
            } else {
                for (int c = sub; c < D; c += LP) { const float e = a[c] - b[c]; s = fmaf(e, e, s); }
            }
        }
        const float d2 = group_sum<LP>(s);    // (all lanes of the wavefront get here)
        if (!ok) continue;
        float g;  // d term / d diff = g * diff
        const float cj = pair_grad ? pair_grad[beg + j] : coef;
        if (t == DCN_LIST_MATCH) {
            g = 2.f * cj;
        } else if (cfg.invert[t] == 2) {        // legacy: d max(0, M - d2) / d diff = -2 diff where the hinge is active
            if (!(M - d2 > 0.f)) continue;
            g = -2.f * cj;
        } else {
            const float dist = sqrtf(d2);
            const float hinge = cfg.invert[t] ? dist - M : M - dist;
            if (!(hinge > 0.f) || !(dist > 0.f)) continue;  // clamp is flat; d||x||/dx := 0 at x = 0 (torch)
            float w = 1.f;
            if (per > 0) {
                const int64_t mi = j / per;
                w = mi < mlen ? pixel_weight(idx_b[mbeg + mi], ib, cfg.image_width, cfg.m_pixel) : 0.f;
            }
            if (pair_grad) w = 1.f;  // per_term is the unweighted l_j
            g = (cfg.invert[t] ? 2.f : -2.f) * hinge / dist * w * cj;
        }
        if (SINGLE) {
            if (sub < D) {
                const float v = g * df;
                unsafeAtomicAdd(gAp + ia * D + sub, v);
                unsafeAtomicAdd(gBp + ib * D + sub, -v);
            }
        } else {
            for (int c = sub; c < D; c += LP) {
                const float v = g * (a[c] - b[c]);
                unsafeAtomicAdd(gAp + ia * D + c, v);
                unsafeAtomicAdd(gBp + ib * D + c, -v);
            }
        }
    }
}

// Backward from the records the forward pass saved (loss_fwd_kernel: rec_d, rec_s): per pixel pair two int64 indices, the
// factor and the D-float difference -- coalesced streams -- then the same run of D fp32 atomics per descriptor as above.  No
// descriptor is gathered again.  grid = (chunks, 4*num_pairs).
template <int LP, bool SINGLE>
__global__ void __launch_bounds__(kThreads)
loss_bwd_saved_kernel(int64_t hw, int D, int num_pairs, const int64_t* __restrict__ idx_a, const int64_t* __restrict__ idx_b,
                      const int64_t* __restrict__ offsets, dcn_loss_config cfg, const int* __restrict__ hard_neg,
                      const float* __restrict__ grad_loss, const float* __restrict__ rec_d, const float* __restrict__ rec_s,
                      float* __restrict__ gA, float* __restrict__ gB) {
    constexpr int GROUPS = kThreads / LP, PPB = GROUPS * kItems;
    const int seg = blockIdx.y, p = seg >> 2, t = seg & 3;
    const int64_t beg = offsets[seg], len = offsets[seg + 1] - beg;
    const int64_t chunk0 = (int64_t)blockIdx.x * PPB;
    if (chunk0 >= len) return;
    const int grp = threadIdx.x / LP, sub = threadIdx.x % LP;
    int64_t lens[4];
    int h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { lens[k] = offsets[4 * p + k + 1] - offsets[4 * p + k]; h[k] = hard_neg[4 * p + k]; }
    const PairScales sc = pair_scales(cfg, h, lens);
    float coef = t == DCN_LIST_MATCH ? sc.match_coef : (t == DCN_LIST_BLIND ? sc.blind_coef : sc.nonmatch_coef);
    if (coef == 0.f) return;
    coef *= grad_loss[0] / (float)num_pairs;
    float* gAp = gA + (int64_t)p * hw * D;
    float* gBp = gB + (int64_t)p * hw * D;
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
        const int64_t j = chunk0 + (int64_t)it * GROUPS + grp;
        if (j >= len) continue;
        const float sf = rec_s[beg + j];
        if (sf == 0.f) continue;            // skipped / out-of-range pair, flat hinge, zero distance
        const int64_t ia = idx_a[beg + j], ib = idx_b[beg + j];
        const float g = sf * coef;          // (loss_bwd_kernel's g: factor x coefficient)
        if (SINGLE) {
            if (sub < D) {
                const float v = g * rec_d[(beg + j) * D + sub];
                unsafeAtomicAdd(gAp + ia * D + sub, v);
                unsafeAtomicAdd(gBp + ib * D + sub, -v);
            }
        } else {
            for (int c = sub; c < D; c += LP) {
                const float v = g * rec_d[(beg + j) * D + c];
                unsafeAtomicAdd(gAp + ia * D + c, v);
                unsafeAtomicAdd(gBp + ib * D + c, -v);
            }
        }
    }
}

// ---- the same backward pass, ORDER-INDEPENDENT (round 6): every contribution v is converted to 64-bit fixed point under a
// per-image-pair power-of-two scale and accumulated with INTEGER atomics -- integer addition is associative, so the dense
// gradient maps carry the same bits whatever order the hardware retires the atomics in (the fp32 atomics above make the loss
// backward the one kernel of a training step that is not run-to-run reproducible).  Three launches behind the zero-fill of
// the int64 maps:
//   1. loss_bwd_vmax_kernel:        vmax[p] = max |v| over all contributions of pair p (same expression as the scatter);
//   2. loss_bwd_saved_exact_kernel: q = rint(v * 2^e_p), e_p = 40 - exponent(vmax[p]):  |q| <= 2^40, and a pair has fewer than
//                                   2^22 contributions (checked by the launcher), so no sum can leave 63 bits; the rounding of a
//                                   contribution is 2^-40 of the largest one -- 16 bits below fp32's own resolution;
//   3. loss_exact_convert_kernel:   grad = (float)(acc * 2^-e_p)  (an exact double product, ONE rounding).
// A non-finite contribution makes vmax non-finite: the pair's maps are then filled with NaN, as the float path would make them
// wherever the value lands.
__device__ __forceinline__ double exact_scale(float vmax) {      // 2^e with vmax * 2^e in [2^39, 2^40); 0: nothing to add
    if (!(vmax > 0.f) || !(vmax < __builtin_huge_valf())) return 0.0;
    int x;
    (void)frexpf(vmax, &x);                                      // vmax = m 2^x, m in [0.5, 1)
    return ldexp(1.0, 40 - x);
}

template <int LP, bool SINGLE>
__global__ void __launch_bounds__(kThreads)
loss_bwd_vmax_kernel(int D, int num_pairs, const int64_t* __restrict__ offsets, dcn_loss_config cfg,
                     const int* __restrict__ hard_neg, const float* __restrict__ grad_loss, const float* __restrict__ rec_d,
                     const float* __restrict__ rec_s, unsigned* __restrict__ vmax_bits) {
    constexpr int GROUPS = kThreads / LP, PPB = GROUPS * kItems;
    const int seg = blockIdx.y, p = seg >> 2, t = seg & 3;
    const int64_t beg = offsets[seg], len = offsets[seg + 1] - beg;
    const int64_t chunk0 = (int64_t)blockIdx.x * PPB;
    if (chunk0 >= len) return;
    const int grp = threadIdx.x / LP, sub = threadIdx.x % LP;
    int64_t lens[4];
    int h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { lens[k] = offsets[4 * p + k + 1] - offsets[4 * p + k]; h[k] = hard_neg[4 * p + k]; }
    const PairScales sc = pair_scales(cfg, h, lens);
    float coef = t == DCN_LIST_MATCH ? sc.match_coef : (t == DCN_LIST_BLIND ? sc.blind_coef : sc.nonmatch_coef);
    if (coef == 0.f) return;
    coef *= grad_loss[0] / (float)num_pairs;
    float m = 0.f;
    bool bad = false;
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
        const int64_t j = chunk0 + (int64_t)it * GROUPS + grp;
        if (j >= len) continue;
        const float sf = rec_s[beg + j];
        if (sf == 0.f) continue;
        const float g = sf * coef;
        for (int c = sub; c < (SINGLE ? (sub < D ? sub + 1 : 0) : D); c += LP) {
            const float v = fabsf(g * rec_d[(beg + j) * D + c]);
            bad |= !(v == v);                                    // (fmaxf drops a NaN)
            m = fmaxf(m, v);
        }
    }
    if (bad) m = __builtin_huge_valf();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) {
        const unsigned bits = __float_as_uint(m);
        if (bits > __atomic_load_n(vmax_bits + p, __ATOMIC_RELAXED)) atomicMax(vmax_bits + p, bits);
    }
}

template <int LP, bool SINGLE>
__global__ void __launch_bounds__(kThreads)
loss_bwd_saved_exact_kernel(int64_t hw, int D, int num_pairs, const int64_t* __restrict__ idx_a,
                            const int64_t* __restrict__ idx_b, const int64_t* __restrict__ offsets, dcn_loss_config cfg,
                            const int* __restrict__ hard_neg, const float* __restrict__ grad_loss,
                            const float* __restrict__ rec_d, const float* __restrict__ rec_s,
                            const float* __restrict__ vmax, unsigned long long* __restrict__ accA,
                            unsigned long long* __restrict__ accB) {
    constexpr int GROUPS = kThreads / LP, PPB = GROUPS * kItems;
    const int seg = blockIdx.y, p = seg >> 2, t = seg & 3;
    const int64_t beg = offsets[seg], len = offsets[seg + 1] - beg;
    const int64_t chunk0 = (int64_t)blockIdx.x * PPB;
    if (chunk0 >= len) return;
    const double scale = exact_scale(vmax[p]);
    if (scale == 0.0) return;                                    // nothing to add, or a non-finite pair (the convert pass writes NaN)
    const int grp = threadIdx.x / LP, sub = threadIdx.x % LP;
    int64_t lens[4];
    int h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { lens[k] = offsets[4 * p + k + 1] - offsets[4 * p + k]; h[k] = hard_neg[4 * p + k]; }
    const PairScales sc = pair_scales(cfg, h, lens);
    float coef = t == DCN_LIST_MATCH ? sc.match_coef : (t == DCN_LIST_BLIND ? sc.blind_coef : sc.nonmatch_coef);
    if (coef == 0.f) return;
    coef *= grad_loss[0] / (float)num_pairs;
    unsigned long long* gAp = accA + (int64_t)p * hw * D;
    unsigned long long* gBp = accB + (int64_t)p * hw * D;
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
        const int64_t j = chunk0 + (int64_t)it * GROUPS + grp;
        if (j >= len) continue;
        const float sf = rec_s[beg + j];
        if (sf == 0.f) continue;
        const int64_t ia = idx_a[beg + j], ib = idx_b[beg + j];
        const float g = sf * coef;
        for (int c = sub; c < (SINGLE ? (sub < D ? sub + 1 : 0) : D); c += LP) {
            const float v = g * rec_d[(beg + j) * D + c];
            const long long q = __double2ll_rn((double)v * scale);       // |q| <= 2^40; exact product, one rounding
            if (q != 0) {
                atomicAdd(gAp + ia * D + c, (unsigned long long)q);       // (two's complement: adding -q is adding 2^64 - q)
                atomicAdd(gBp + ib * D + c, (unsigned long long)(-q));
            }
        }
    }
}

// grad = acc * 2^-e of the map's image pair; maps: [2][num_pairs][per_pair] (A maps, then B maps).  grid = (x, 2 * num_pairs):
// one map per blockIdx.y, so the scale is formed once per work-item; two elements (16 B in, 8 B out) per work-item and trip.
__global__ void __launch_bounds__(256)
loss_exact_convert_kernel(const long long* __restrict__ acc, const float* __restrict__ vmax, float* __restrict__ grad_a,
                          float* __restrict__ grad_b, int64_t per_pair, int num_pairs) {
    const int m = blockIdx.y;                                     // map index: [0, P) of A, [P, 2P) of B
    const int p = m >= num_pairs ? m - num_pairs : m;
    const float vm = vmax[p];
    const bool bad = !(vm < __builtin_huge_valf());               // a non-finite contribution somewhere in this pair
    const double scale = exact_scale(vm);
    const double inv = scale == 0.0 ? 0.0 : 1.0 / scale;          // (a power of two: exact)
    const long long* src = acc + (int64_t)m * per_pair;
    float* dst = (m >= num_pairs ? grad_b : grad_a) + (int64_t)p * per_pair;
    const float nanv = __builtin_nanf("");
    if ((per_pair & 1) == 0) {
        typedef long long ll2 __attribute__((ext_vector_type(2)));
        const int64_t n2 = per_pair >> 1;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
            const ll2 q = reinterpret_cast<const ll2*>(src)[i];
            float2 o;
            o.x = bad ? nanv : (float)((double)q[0] * inv);
            o.y = bad ? nanv : (float)((double)q[1] * inv);
            reinterpret_cast<float2*>(dst)[i] = o;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_pair; i += (int64_t)gridDim.x * 256)
            dst[i] = bad ? nanv : (float)((double)src[i] * inv);
    }
}

// workgroups per list for descriptor dimension d (d <= 0: the worst case over all d, for workspace sizing)
int chunks_for(int64_t max_list_len, int d) {
    const int ppb = pairs_per_block(d > 0 ? lanes_per_pair(d) : 32);
    return (int)dcn::ceil_div64(max_list_len > 0 ? max_list_len : 1, ppb);
}

int64_t max_len(const int64_t* offsets_host, int num_pairs) {
    int64_t m = 0;
    for (int s = 0; s < 4 * num_pairs; ++s) {
        const int64_t l = offsets_host[s + 1] - offsets_host[s];
        if (l > m) m = l;
    }
    return m;
}

// ------------------------------------------------------------------------------------------------ triplet loss
// pixelwise_contrastive_loss.py:104-129 (get_triplet_loss):  with a = A[non_a[i]], m = B[match_b[i / multiplier]],
// n = B[non_b[i]]:   loss = 1/P * sum_i sum_d max(0, (a_d - m_d)^2 - (a_d - n_d)^2 + alpha)
// -- the hinge is applied PER DESCRIPTOR COMPONENT (the reference never sums over d before the clamp), and the match list
// is expanded by `multiplier` = P / P_match non-matches per match ("matches_b_long").  One work-item per triplet.
constexpr int kTripThreads = 256;

__global__ void __launch_bounds__(kTripThreads)
triplet_fwd_kernel(const float* __restrict__ A, const float* __restrict__ B, int64_t hw, int d,
                   const int64_t* __restrict__ non_a, const int64_t* __restrict__ match_b,
                   const int64_t* __restrict__ non_b, int64_t n, int64_t multiplier, float alpha,
                   double* __restrict__ part, int* __restrict__ status) {
    __shared__ double s[kTripThreads / dcn::kWave];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kTripThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTripThreads) {
        const int64_t ia = non_a[i], im = match_b[i / multiplier], in = non_b[i];
        if (ia < 0 || ia >= hw || im < 0 || im >= hw || in < 0 || in >= hw) { *status = 1; continue; }
        const float* a = A + ia * d;
        const float* m = B + im * d;
        const float* q = B + in * d;
        float t = 0.f;
        for (int k = 0; k < d; ++k) {
            const float dm = a[k] - m[k], dn = a[k] - q[k];
            t += fmaxf(dm * dm - dn * dn + alpha, 0.f);
        }
        acc += (double)t;
    }
    acc = dcn::block_sum<kTripThreads>(acc, s);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(kTripThreads)
triplet_finalize_kernel(const double* __restrict__ part, int blocks, int64_t n, float* __restrict__ loss) {
    __shared__ double s[kTripThreads / dcn::kWave];
    double acc = 0.0;
    for (int i = threadIdx.x; i < blocks; i += kTripThreads) acc += part[i];   // fixed order
    acc = dcn::block_sum<kTripThreads>(acc, s);
    if (threadIdx.x == 0) *loss = (float)(acc / (double)n);
}

// gA / gB += d loss / d descriptors (hardware fp32 atomics; the caller zero-fills or accumulates on purpose)
__global__ void __launch_bounds__(kTripThreads)
triplet_bwd_kernel(const float* __restrict__ A, const float* __restrict__ B, int64_t hw, int d,
                   const int64_t* __restrict__ non_a, const int64_t* __restrict__ match_b,
                   const int64_t* __restrict__ non_b, int64_t n, int64_t multiplier, float alpha,
                   const float* __restrict__ grad_loss, float* __restrict__ gA, float* __restrict__ gB) {
    const float g2 = 2.f * (*grad_loss) / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * kTripThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTripThreads) {
        const int64_t ia = non_a[i], im = match_b[i / multiplier], in = non_b[i];
        if (ia < 0 || ia >= hw || im < 0 || im >= hw || in < 0 || in >= hw) continue;
        for (int k = 0; k < d; ++k) {
            const float a = A[ia * d + k], m = B[im * d + k], q = B[in * d + k];
            const float dm = a - m, dn = a - q;
            if (dm * dm - dn * dn + alpha > 0.f) {
                unsafeAtomicAdd(gA + ia * d + k, g2 * (dm - dn));
                unsafeAtomicAdd(gB + im * d + k, -g2 * dm);
                unsafeAtomicAdd(gB + in * d + k, g2 * dn);
            }
        }
    }
}

int triplet_blocks(int64_t n) {
    int64_t b = dcn::ceil_div64(n, kTripThreads);
    if (b > 2048) b = 2048;
    return b < 1 ? 1 : (int)b;
}

}  // namespace

extern "C" size_t dcn_triplet_loss_workspace_bytes(int64_t n) { return (size_t)triplet_blocks(n) * sizeof(double) + 64; }

extern "C" int dcn_triplet_loss_forward(const float* desc_a, const float* desc_b, int64_t hw, int d, const int64_t* non_a,
                                        const int64_t* match_b, const int64_t* non_b, int64_t n, int64_t n_match, float alpha,
                                        float* loss, int32_t* status, void* workspace, void* stream) {
    if (!desc_a || !desc_b || !non_a || !match_b || !non_b || !loss || !status || !workspace || hw < 1 || d < 1 || n < 1 ||
        n_match < 1 || (n % n_match) != 0)   // the reference's index_select shapes only agree for whole multiples
        return DCN_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (dcn::fill_bytes_async(status, 0, sizeof(int32_t), st) != DCN_OK) return DCN_E_LAUNCH;
    const int blocks = triplet_blocks(n);
    hipLaunchKernelGGL(triplet_fwd_kernel, dim3(blocks), dim3(kTripThreads), 0, st, desc_a, desc_b, hw, d, non_a, match_b,
                       non_b, n, n / n_match, alpha, (double*)workspace, (int*)status);
    hipLaunchKernelGGL(triplet_finalize_kernel, dim3(1), dim3(kTripThreads), 0, st, (const double*)workspace, blocks, n, loss);
    return dcn::check_launch();
}

extern "C" int dcn_triplet_loss_backward(const float* desc_a, const float* desc_b, int64_t hw, int d, const int64_t* non_a,
                                         const int64_t* match_b, const int64_t* non_b, int64_t n, int64_t n_match,
                                         float alpha, const float* grad_loss, float* grad_a, float* grad_b, void* stream) {
    if (!desc_a || !desc_b || !non_a || !match_b || !non_b || !grad_loss || !grad_a || !grad_b || hw < 1 || d < 1 || n < 1 ||
        n_match < 1 || (n % n_match) != 0)
        return DCN_E_INVALID;
    hipLaunchKernelGGL(triplet_bwd_kernel, dim3(triplet_blocks(n)), dim3(kTripThreads), 0, (hipStream_t)stream, desc_a,
                       desc_b, hw, d, non_a, match_b, non_b, n, n / n_match, alpha, grad_loss, grad_a, grad_b);
    return dcn::check_launch();
}

namespace {
}  // namespace

extern "C" size_t dcn_loss_workspace_bytes(int num_pairs, int64_t max_list_len) {
    const size_t n = (size_t)4 * (size_t)num_pairs * (size_t)chunks_for(max_list_len, 0);
    return n * sizeof(double) + (size_t)num_pairs * sizeof(double) + 2 * n * sizeof(int) + (size_t)num_pairs * sizeof(int) + 64;
}

namespace {
int loss_forward_impl(const float* desc_a, const float* desc_b, int num_pairs, int64_t hw, int d, const int64_t* idx_a,
                      const int64_t* idx_b, const int64_t* offsets_host, const int64_t* offsets_dev, const dcn_loss_config* cfg,
                      float* terms, float* sums, int32_t* hard_neg, float* loss, float* per_term, int32_t* status,
                      void* workspace, float* rec_d, float* rec_s, void* stream);
}  // namespace

extern "C" int dcn_contrastive_loss_forward(const float* desc_a, const float* desc_b, int num_pairs, int64_t hw, int d,
                                            const int64_t* idx_a, const int64_t* idx_b, const int64_t* offsets_host,
                                            const int64_t* offsets_dev, const dcn_loss_config* cfg, float* terms,
                                            float* sums, int32_t* hard_neg, float* loss, float* per_term,
                                            int32_t* status, void* workspace, void* stream) {
    return loss_forward_impl(desc_a, desc_b, num_pairs, hw, d, idx_a, idx_b, offsets_host, offsets_dev, cfg, terms, sums, hard_neg,
                             loss, per_term, status, workspace, nullptr, nullptr, stream);
}

extern "C" size_t dcn_loss_saved_floats(int64_t total_pairs, int d) {
    return total_pairs < 1 || d < 1 ? 0 : (size_t)total_pairs * ((size_t)d + 1);
}

extern "C" int dcn_contrastive_loss_forward_save(const float* desc_a, const float* desc_b, int num_pairs, int64_t hw, int d,
                                                 const int64_t* idx_a, const int64_t* idx_b, const int64_t* offsets_host,
                                                 const int64_t* offsets_dev, const dcn_loss_config* cfg, float* terms,
                                                 float* sums, int32_t* hard_neg, float* loss, float* per_term,
                                                 int32_t* status, void* workspace, float* pair_records, void* stream) {
    if (!pair_records || !offsets_host || num_pairs < 1 || d < 1) return DCN_E_INVALID;
    const int64_t total = offsets_host[4 * num_pairs];
    return loss_forward_impl(desc_a, desc_b, num_pairs, hw, d, idx_a, idx_b, offsets_host, offsets_dev, cfg, terms, sums, hard_neg,
                             loss, per_term, status, workspace, pair_records, pair_records + (size_t)total * d, stream);
}

namespace {
int loss_forward_impl(const float* desc_a, const float* desc_b, int num_pairs, int64_t hw, int d, const int64_t* idx_a,
                      const int64_t* idx_b, const int64_t* offsets_host, const int64_t* offsets_dev, const dcn_loss_config* cfg,
                      float* terms, float* sums, int32_t* hard_neg, float* loss, float* per_term, int32_t* status,
                      void* workspace, float* rec_d, float* rec_s, void* stream) {
    if (!desc_a || !desc_b || !offsets_host || !offsets_dev || !cfg || !terms || !sums || !hard_neg || !loss ||
        !status || !workspace || num_pairs < 1 || hw < 1 || d < 1)
        return DCN_E_INVALID;
    for (int s = 0; s < 4 * num_pairs; ++s)
        if (offsets_host[s + 1] < offsets_host[s]) return DCN_E_INVALID;
    if (offsets_host[4 * num_pairs] > 0 && (!idx_a || !idx_b)) return DCN_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int chunks = chunks_for(max_len(offsets_host, num_pairs), d);
    const size_t n = (size_t)4 * num_pairs * chunks;
    double* part_sum = (double*)workspace;
    double* pair_loss = part_sum + n;
    int* part_cnt = (int*)(pair_loss + num_pairs);
    int* part_oob = part_cnt + n;
    int* pair_oob = part_oob + n;
    const dim3 grid(chunks, 4 * num_pairs), block(kThreads);
#define DCN_LAUNCH_FWD(LP, SINGLE)                                                                                       \
    hipLaunchKernelGGL((loss_fwd_kernel<LP, SINGLE>), grid, block, 0, st, desc_a, desc_b, hw, d, idx_a, idx_b, offsets_dev, \
                       *cfg, part_sum, part_cnt, per_term, part_oob, rec_d, rec_s)
    switch (lanes_per_pair(d)) {
        case 4: DCN_LAUNCH_FWD(4, true); break;
        case 8: DCN_LAUNCH_FWD(8, true); break;
        case 16: DCN_LAUNCH_FWD(16, true); break;
        default:
            if (d <= 32) DCN_LAUNCH_FWD(32, true);
            else DCN_LAUNCH_FWD(32, false);
            break;
    }
#undef DCN_LAUNCH_FWD
    if (num_pairs <= 8) {
        hipLaunchKernelGGL(loss_finalize_all_kernel, dim3(1), block, 0, st, part_sum, part_cnt, part_oob, chunks, num_pairs,
                           offsets_dev, *cfg, terms, sums, (int*)hard_neg, pair_loss, loss, (int*)status);
    } else {
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(num_pairs), block, 0, st, part_sum, part_cnt, part_oob, chunks, offsets_dev,
                           *cfg, terms, sums, (int*)hard_neg, pair_loss, pair_oob);
        hipLaunchKernelGGL(loss_mean_kernel, dim3(1), dim3(64), 0, st, (const double*)pair_loss, (const int*)pair_oob, num_pairs,
                           loss, (int*)status);
    }
    return dcn::check_launch();
}
}  // namespace

extern "C" int dcn_contrastive_loss_backward(const float* desc_a, const float* desc_b, int num_pairs, int64_t hw, int d,
                                             const int64_t* idx_a, const int64_t* idx_b, const int64_t* offsets_host,
                                             const int64_t* offsets_dev, const dcn_loss_config* cfg, const float* sums,
                                             const int32_t* hard_neg, const float* grad_loss,
                                             const float* pair_grad, float* grad_a, float* grad_b, void* stream) {
    (void)sums;
    if (!desc_a || !desc_b || !offsets_host || !offsets_dev || !cfg || !grad_a || !grad_b || num_pairs < 1 || hw < 1 ||
        d < 1)
        return DCN_E_INVALID;
    if (!pair_grad && (!hard_neg || !grad_loss)) return DCN_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = (size_t)num_pairs * (size_t)hw * (size_t)d * sizeof(float);
    if ((char*)grad_b == (char*)grad_a + bytes) {   // the two maps are one allocation (dcn_hip/loss.py): one fill launch
        if (dcn::fill_bytes_async(grad_a, 0, 2 * bytes, st) != DCN_OK) return DCN_E_LAUNCH;
    } else {
        if (dcn::fill_bytes_async(grad_a, 0, bytes, st) != DCN_OK) return DCN_E_LAUNCH;
        if (dcn::fill_bytes_async(grad_b, 0, bytes, st) != DCN_OK) return DCN_E_LAUNCH;
    }
    const int64_t ml = max_len(offsets_host, num_pairs);
    if (ml == 0) return DCN_OK;
    const dim3 grid(chunks_for(ml, d), 4 * num_pairs), block(kThreads);
#define DCN_LAUNCH_BWD(LP, SINGLE)                                                                                        \
    hipLaunchKernelGGL((loss_bwd_kernel<LP, SINGLE>), grid, block, 0, st, desc_a, desc_b, hw, d, num_pairs, idx_a, idx_b,   \
                       offsets_dev, *cfg, (const int*)hard_neg, grad_loss, pair_grad, grad_a, grad_b)
    switch (lanes_per_pair(d)) {
        case 4: DCN_LAUNCH_BWD(4, true); break;
        case 8: DCN_LAUNCH_BWD(8, true); break;
        case 16: DCN_LAUNCH_BWD(16, true); break;
        default:
            if (d <= 32) DCN_LAUNCH_BWD(32, true);
            else DCN_LAUNCH_BWD(32, false);
            break;
    }
#undef DCN_LAUNCH_BWD
    return dcn::check_launch();
}

// Backward from the forward pass's pair records (dcn_contrastive_loss_forward_save).  prefilled != 0: grad_a / grad_b have
// already been zero-filled by the caller (e.g. on another stream while the forward kernels ran: dcn_fill_bytes) -- the call
// then only accumulates; otherwise it zero-fills them first, like dcn_contrastive_loss_backward.
extern "C" int dcn_contrastive_loss_backward_saved(int num_pairs, int64_t hw, int d, const int64_t* idx_a, const int64_t* idx_b,
                                                   const int64_t* offsets_host, const int64_t* offsets_dev,
                                                   const dcn_loss_config* cfg, const int32_t* hard_neg, const float* grad_loss,
                                                   const float* pair_records, int prefilled, float* grad_a, float* grad_b,
                                                   void* stream) {
    if (!offsets_host || !offsets_dev || !cfg || !hard_neg || !grad_loss || !pair_records || !grad_a || !grad_b || num_pairs < 1 ||
        hw < 1 || d < 1)
        return DCN_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = (size_t)num_pairs * (size_t)hw * (size_t)d * sizeof(float);
    if (!prefilled) {
        if ((char*)grad_b == (char*)grad_a + bytes) {
            if (dcn::fill_bytes_async(grad_a, 0, 2 * bytes, st) != DCN_OK) return DCN_E_LAUNCH;
        } else {
            if (dcn::fill_bytes_async(grad_a, 0, bytes, st) != DCN_OK) return DCN_E_LAUNCH;
            if (dcn::fill_bytes_async(grad_b, 0, bytes, st) != DCN_OK) return DCN_E_LAUNCH;
        }
    }
    const int64_t ml = max_len(offsets_host, num_pairs);
    if (ml == 0) return DCN_OK;
    if (!idx_a || !idx_b) return DCN_E_INVALID;
    const int64_t total = offsets_host[4 * num_pairs];
    const float* rec_d = pair_records;
    const float* rec_s = pair_records + (size_t)total * d;
    const dim3 grid(chunks_for(ml, d), 4 * num_pairs), block(kThreads);
#define DCN_LAUNCH_BWDS(LP, SINGLE)                                                                                          \
    hipLaunchKernelGGL((loss_bwd_saved_kernel<LP, SINGLE>), grid, block, 0, st, hw, d, num_pairs, idx_a, idx_b, offsets_dev, *cfg, \
                       (const int*)hard_neg, grad_loss, rec_d, rec_s, grad_a, grad_b)
    switch (lanes_per_pair(d)) {
        case 4: DCN_LAUNCH_BWDS(4, true); break;
        case 8: DCN_LAUNCH_BWDS(8, true); break;
        case 16: DCN_LAUNCH_BWDS(16, true); break;
        default:
            if (d <= 32) DCN_LAUNCH_BWDS(32, true);
            else DCN_LAUNCH_BWDS(32, false);
            break;
    }
#undef DCN_LAUNCH_BWDS
    return dcn::check_launch();
}

// Workspace of the order-independent backward: the two int64 accumulation maps + one abs-max word per image pair.
extern "C" size_t dcn_loss_exact_workspace_bytes(int num_pairs, int64_t hw, int d) {
    if (num_pairs < 1 || hw < 1 || d < 1) return 0;
    return (size_t)2 * num_pairs * (size_t)hw * d * sizeof(long long) + (((size_t)num_pairs * 4 + 255) / 256) * 256;
}

// dcn_contrastive_loss_backward_saved with bit-reproducible gradient maps (64-bit fixed-point accumulation under a per-pair
// power-of-two scale; see loss_bwd_saved_exact_kernel).  grad_a / grad_b are written in full (no zero-fill needed).
// DCN_E_UNSUPPORTED when an image pair has 2^22 pixel pairs or more (the headroom of the 63-bit sums): use the float path.
extern "C" int dcn_contrastive_loss_backward_saved_exact(int num_pairs, int64_t hw, int d, const int64_t* idx_a,
                                                         const int64_t* idx_b, const int64_t* offsets_host,
                                                         const int64_t* offsets_dev, const dcn_loss_config* cfg,
                                                         const int32_t* hard_neg, const float* grad_loss,
                                                         const float* pair_records, void* workspace, float* grad_a,
                                                         float* grad_b, void* stream) {
    if (!offsets_host || !offsets_dev || !cfg || !hard_neg || !grad_loss || !pair_records || !workspace || !grad_a || !grad_b ||
        num_pairs < 1 || hw < 1 || d < 1)
        return DCN_E_INVALID;
    if (num_pairs > 16383) return DCN_E_UNSUPPORTED;                  // (grid.y of the scatter / conversion launches: 4 / 2 per pair)
    for (int p = 0; p < num_pairs; ++p)
        if (offsets_host[4 * p + 4] - offsets_host[4 * p] >= ((int64_t)1 << 22)) return DCN_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const size_t per_pair = (size_t)hw * d, map_bytes = (size_t)num_pairs * per_pair * sizeof(long long);
    unsigned long long* accA = (unsigned long long*)workspace;
    unsigned long long* accB = accA + (size_t)num_pairs * per_pair;
    float* vmax = (float*)((char*)workspace + 2 * map_bytes);
    if (dcn::fill_bytes_async(workspace, 0, dcn_loss_exact_workspace_bytes(num_pairs, hw, d), st) != DCN_OK) return DCN_E_LAUNCH;
    const int64_t ml = max_len(offsets_host, num_pairs);
    if (ml > 0) {
        if (!idx_a || !idx_b) return DCN_E_INVALID;
        const int64_t total = offsets_host[4 * num_pairs];
        const float* rec_d = pair_records;
        const float* rec_s = pair_records + (size_t)total * d;
        const dim3 grid(chunks_for(ml, d), 4 * num_pairs), block(kThreads);
#define DCN_LAUNCH_BWDX(LP, SINGLE)                                                                                          \
        do {                                                                                                                  \
            hipLaunchKernelGGL((loss_bwd_vmax_kernel<LP, SINGLE>), grid, block, 0, st, d, num_pairs, offsets_dev, *cfg,       \
                               (const int*)hard_neg, grad_loss, rec_d, rec_s, (unsigned*)vmax);                              \
            hipLaunchKernelGGL((loss_bwd_saved_exact_kernel<LP, SINGLE>), grid, block, 0, st, hw, d, num_pairs, idx_a, idx_b, \
                               offsets_dev, *cfg, (const int*)hard_neg, grad_loss, rec_d, rec_s, (const float*)vmax, accA, accB); \
        } while (0)
        switch (lanes_per_pair(d)) {
            case 4: DCN_LAUNCH_BWDX(4, true); break;
            case 8: DCN_LAUNCH_BWDX(8, true); break;
            case 16: DCN_LAUNCH_BWDX(16, true); break;
            default:
                if (d <= 32) DCN_LAUNCH_BWDX(32, true);
                else DCN_LAUNCH_BWDX(32, false);
                break;
        }
#undef DCN_LAUNCH_BWDX
    }
    const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(dcn::ceil_div64((int64_t)per_pair / 2 + 1, 256 * 4),
                                                                         (256 * 16) / (2 * num_pairs) + 1));
    hipLaunchKernelGGL(loss_exact_convert_kernel, dim3(bx, 2 * num_pairs), dim3(256), 0, st, (const long long*)workspace,
                       (const float*)vmax, grad_a, grad_b, (int64_t)per_pair, num_pairs);
    return dcn::check_launch();
}

// Stream-ordered fill of n bytes (n % 4 == 0) with a byte value, as a kernel (an ordinary node under hipGraph capture).
extern "C" int dcn_fill_bytes(void* p, int byte_value, size_t n, void* stream) {
    if (!p && n) return DCN_E_INVALID;
    return dcn::fill_bytes_async(p, byte_value, n, (hipStream_t)stream);
}
