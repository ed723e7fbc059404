// Host launchers of the HBM-bound kernels (elementwise_kernels.hip), used by the backbone engine.
#pragma once
#include "dcn_common.h"

namespace dcn {

// Launch observer (bench.py's roofline_elementwise / kernel_ms_sum): while a plan is being profiled
// (dcn_plan_profile_begin) the engine installs one for the calling thread, and every launcher below reports each kernel
// it launches as begin(category, algorithmic HBM bytes) ... end() around the launch -- the engine brackets it with HIP
// events on the launch stream.  Null (the default): no cost.  Categories: DCN_PROF_* of include/dcn_hip.h.
struct LaunchObserver {
    void* ctx;
    void (*begin)(void* ctx, int cat, double bytes, hipStream_t st);
    void (*end)(void* ctx, hipStream_t st);
};
extern thread_local const LaunchObserver* launch_observer;
struct ObservedLaunch {   // RAII bracket of ONE kernel launch
    hipStream_t st;
    const LaunchObserver* o;
    ObservedLaunch(int cat, double bytes, hipStream_t s) : st(s), o(launch_observer) { if (o) o->begin(o->ctx, cat, bytes, st); }
    ~ObservedLaunch() { if (o) o->end(o->ctx, st); }
    ObservedLaunch(const ObservedLaunch&) = delete;
    ObservedLaunch& operator=(const ObservedLaunch&) = delete;
};

// absmax (optional): raised to max |img|
void launch_nchw3_to_nhwc4(const float* img, float* out, int n, int hw, float* absmax, hipStream_t st);
void launch_pad_c3_to_c4(const float* w, float* wp, int64_t rows, hipStream_t st);
void launch_unpad_c4_to_c3(const float* wp, float* w, int64_t rows, hipStream_t st);
void launch_pad_rows(const float* src, float* dst, int64_t rows, int d, int ld, hipStream_t st);

// Batch norm.  `groups` (1 or 2) = independent batches stacked along the rows of one launch (forward(img_a) and
// forward(img_b) of a training step as one launch sequence): statistics per group; rows % groups == 0 and conv M tiles /
// backward chunks never straddle a group boundary.  stats = [groups][4][C]: scale, shift, mean, invstd.
// partial[groups * tiles_per_group][3][C] (sum, sum of squares, max |x|) -> stats (+ running update, once per group, in
// order).  out_bound (optional, training only): device scalar raised to a bound of max |x scale + shift| (+ *res_bound when
// given: the bound of the residual that the apply pass adds) -- the pre-scale of the split-fp16 convolution that reads y.
void launch_bn_finalize(const float* partial, int tiles_per_group, int groups, int C, double count_per_group,
                        const float* gamma, const float* beta, float* rmean, float* rvar, float momentum, float eps,
                        int training, float* stats, float* out_bound, const float* res_bound, hipStream_t st);
// y = [relu](x*scale1 + shift1 + (res ? (stats2 ? res*scale2 + shift2 : res) : 0))
// relu_mask (optional, with relu): one byte per float4 of y, bit j = y[4i + j] > 0 (read back by launch_bn_bwd)
// hl_out (optional, with hl_absmax; C % 32 == 0): y also as the hl32 image of conv_hl_kernels.hip, scaled by the power of two
// chosen from *hl_absmax (the bound launch_bn_finalize stored for y)
void launch_bn_apply(const float* x, const float* stats1, const float* res, const float* stats2, int relu, float* y,
                     unsigned char* relu_mask, int C, int64_t rows, int groups, hipStream_t st, void* hl_out = nullptr,
                     const float* hl_absmax = nullptr);
int bn_bwd_chunks(int64_t rows_per_group);
// partial: groups*bn_bwd_chunks(rows/groups)*4*C floats, k123: groups*3*C floats.  g_out (nullable) receives the
// relu-masked dy.  absmax (optional): raised to an upper bound of max |dx|
// the ReLU mask of dy comes from relu_mask (bytes written by launch_bn_apply) if given, else from relu_out, else none
void launch_bn_bwd(const float* dy, const float* relu_out, const unsigned char* relu_mask, const float* x, const float* stats,
                   const float* gamma, int C,
                   int64_t rows, int groups, float* partial, float* dgamma, float* dbeta, float* k123, float* dx,
                   float* g_out, float* absmax, void* dq, hipStream_t st,    // dq (optional, with absmax): dx also as the
                   int reduced_tiles_per_group = 0,                          // pixel-blocked split-fp16 tensor (f16_split.h)
                   const float* dy2 = nullptr,    // dy2 (optional): the upstream gradient is dy + dy2 (residual branch's share)
                   void* hl_dx = nullptr,         // hl_dx (optional, with dq; C % 32 == 0): dx also as the hl32 image the pre-split
                   int keep_dx = 0);              // dgrad reads; the fp32 dx is then NOT written unless keep_dx
// reduced_tiles_per_group > 0: `partial` already holds that many rows per group of per-tile sums, written by the epilogue
// of the dgrad that produced dy (GemmConv::bnb_partial) -- the reduce pass is skipped; dy is then already ReLU-masked
// (pass relu_out = relu_mask = nullptr)
void launch_add(const float* a, const float* b, float* out, int64_t n, hipStream_t st);

// bn_stats (optional): [groups][scale C | shift C | mean C | invstd C] of the batch norm whose output is pooled -- `in` is then
// the convolution output, y = relu(in * scale + shift) is formed on the fly and relu_mask receives y's sign mask
void launch_maxpool_fwd(const float* in, float* out, unsigned char* argmax, int n, int hin, int win, int hout,
                        int wout, int C, hipStream_t st, const float* bn_stats = nullptr, int groups = 1,
                        unsigned char* relu_mask = nullptr);
void launch_maxpool_bwd(const float* gout, const unsigned char* argmax, float* gin, int n, int hin, int win, int hout,
                        int wout, int C, hipStream_t st, const float* gout2 = nullptr);   // gout2 (optional): gradient = gout + gout2

void launch_upsample_fwd(const float* low, int n, int hl, int wl, int ldl, int d, int h, int w, int normalize,
                         float* out, hipStream_t st);
size_t upsample_bwd_tmp_bytes(int n, int hl, int w, int d);
// gv = gradient w.r.t. the un-normalised upsampled map, given gout = gradient w.r.t. the L2-normalised one
void launch_normalize_bwd(const float* low, int n, int hl, int wl, int ldl, int d, int h, int w, const float* gout,
                          float* gv, hipStream_t st);
// gout_b (optional, n even): the gradient of images [n / 2, n) lives in a tensor of its own (forward_pair's second output)
void launch_upsample_bwd(const float* gout, int n, int hl, int wl, int ldl, int d, int h, int w, float* tmp,
                         float* glow, float* absmax, hipStream_t st, const float* gout_b = nullptr);

}  // namespace dcn
