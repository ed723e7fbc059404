/*
 * dcn_hip.h -- C ABI of libdcn_hip.so: the MI355X (gfx950) implementation of the dense-correspondence
 * training hot path of RobotLocomotion/pytorch-dense-correspondence.
 *
 * The reference has no native code and no FFI of its own: its hot path is Python that calls into
 * torch.  The entry points below are what a binding for that path replaces;
 * each cites the reference interface it stands in for (paths relative to the reference root).  The
 * host-side mirror of the reference's Python API that calls them through ctypes lives in
 * pytorch-dense-correspondence_amd/ (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.  All tensor pointers are DEVICE pointers owned
 *     by the caller (the PyTorch caching allocator in practice); the library allocates no device
 *     memory.  Pointer *arrays* (params / grads) are HOST arrays of device pointers.
 *   - every call is asynchronous on the caller's stream (`hipStream_t` passed as void*) and performs no host
 *     synchronisation.  The only process-wide state is the table of DCN_* environment overrides (read once, see
 *     dcn_reload_env).  A dcn_plan carries per-plan state (a side stream, events, profiling slots, the conv mode):
 *     use ONE plan per stream -- two streams driving the same plan concurrently would race; one process per GPU.
 *   - return value: 0 on success, negative DCN_E_* otherwise (never throws across the ABI).  The
 *     Python wrapper raises RuntimeError / ValueError like the reference does
 *     (dense_correspondence/network/dense_correspondence_network.py:381).
 *   - layouts: activations are NHWC fp32 (logical [N,C,H,W] in torch.channels_last memory), so the
 *     reference's own `view(N, D, W*H).permute(0, 2, 1)` (network.py:317-318) of the descriptor map is a
 *     contiguous [N, H*W, D] tensor.  Convolution weights are [Cout][kh][kw][Cin] (logical OIHW in
 *     channels_last memory).  Pixel indices are int64 `u + W*v`
 *     (dense_correspondence/dataset/spartan_dataset_masked.py:1256-1264).
 */
#ifndef DCN_HIP_H
#define DCN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCN_OK 0
#define DCN_E_INVALID (-1)     /* bad argument (shape, null pointer, unsupported architecture name) */
#define DCN_E_LAUNCH (-2)      /* a kernel launch failed (hipGetLastError != hipSuccess) */
#define DCN_E_UNSUPPORTED (-3) /* valid request the library does not implement */

/* Build identification: "dcn_hip <version> gfx950" for the shipped library, "... hostemu" for the
 * test-only host emulation build (tests/hostemu). */
const char* dcn_version(void);

/* The DCN_* environment overrides -- the full list with their meaning is the header comment of csrc/dcn_tuning.h:
 * DCN_CONV_MODE, DCN_BACKWARD_OVERLAP, DCN_GEMM_TILE_M, DCN_STEM8, DCN_GEMM_SK, DCN_GEMM_SK_MIN_GAIN,
 * DCN_GEMM_UNI, DCN_GEMM_SK_FIXUP, DCN_BN_BWD_FUSED, DCN_DEFER_RESIDUAL_ADD, DCN_WGRAD_TILE, DCN_WGRAD_DEEP, DCN_WGRAD_ROLES,
 * DCN_WGRAD_SPLITS, DCN_GEMM_HL, DCN_HL_MIN_K, DCN_GEMM_HL_ROWS, DCN_WGRAD_HL, DCN_HL_PRODUCERS, DCN_HL_ONLY_MID, DCN_STEM_POOL_FUSED, DCN_BN_REVERSE, DCN_BN_NT, DCN_BN_REDUCE_WIDE, DCN_WSPLIT_OVERLAP -- are read ONCE, at the first call that needs them -- never on the
 * launch path.  dcn_reload_env re-reads them (tests / tuning scripts that change a variable in-process); not to be called
 * while another thread is inside the library. */
void dcn_reload_env(void);

/* The ONE exception to "the library allocates no device memory": a 64 KB buffer of arrival words per (device, stream) that
 * ever ran a K-split launch of the small-tile gather-GEMM (csrc/conv_hlx_kernels.hip; never while the stream is being
 * captured).  This call frees them all; the caller has synchronised those streams.  The Python binding registers it atexit. */
void dcn_release_pooled_buffers(void);

/* =====================================================================================================
 * 1. Pixelwise contrastive loss  (kernel K9)
 *
 * Replaces, fused into one pass over all pair lists of all image pairs of a step:
 *   PixelwiseContrastiveLoss.match_loss                  dense_correspondence/loss_functions/pixelwise_contrastive_loss.py:132-167
 *   PixelwiseContrastiveLoss.non_match_descriptor_loss   ...pixelwise_contrastive_loss.py:171-213
 *   PixelwiseContrastiveLoss.non_match_loss_descriptor_only / _with_l2_pixel_norm / l2_pixel_loss
 *                                                        ...pixelwise_contrastive_loss.py:215-352
 *   loss_composer.get_within_scene_loss / get_different_object_loss / get_same_object_across_scene_loss
 *                                                        dense_correspondence/loss_functions/loss_composer.py:70-212
 *   and their autograd (index_select backward == index_add_ into a zeroed [1,HW,D] buffer).
 *
 * Pair lists are ragged: for image pair p (0 <= p < num_pairs) and list type t the pairs are
 * idx_a[offsets[4p+t] .. offsets[4p+t+1]) / idx_b[...] with t = DCN_LIST_*.  An empty list (the
 * reference's `[-1]` sentinel, dense_correspondence_dataset_masked.py:209-223) has equal offsets.
 * num_pairs == 1 is exactly one reference iteration; for num_pairs > 1 every pair keeps its own
 * hard-negative normaliser and `loss = mean_p loss_p` (SURVEY.md section 8a note B).
 * ===================================================================================================== */

enum { DCN_LIST_MATCH = 0, DCN_LIST_MASKED = 1, DCN_LIST_BACKGROUND = 2, DCN_LIST_BLIND = 3, DCN_NUM_LISTS = 4 };

/* How the per-list sums are composed into `loss` (loss_composer.py). */
enum {
    DCN_COMPOSE_WITHIN_SCENE = 0,     /* loss_composer.py:70-143  (also MULTI_OBJECT / SYNTHETIC_MULTI_OBJECT) */
    DCN_COMPOSE_DIFFERENT_OBJECT = 1, /* loss_composer.py:168-191 blind list only, margin M_background */
    DCN_COMPOSE_ACROSS_SCENE = 2,     /* loss_composer.py:193-212 blind list only, inverted hinge, margin M_masked */
    DCN_COMPOSE_RAW_SUMS = 3          /* loss = match_weight * S_match + non_match_weight * (S_masked + S_background + S_blind):
                                         the un-normalised sums the PixelwiseContrastiveLoss building blocks return
                                         (pixelwise_contrastive_loss.py:271-304) */
};

typedef struct dcn_loss_config {
    float margin[DCN_NUM_LISTS];       /* hinge margin per list type (match entry unused) */
    int32_t invert[DCN_NUM_LISTS];     /* 0: max(0, M - d)^2;  1: max(0, d - M)^2 (pcl.py:205-208);  2: max(0, M - d^2), the legacy
                                          hinge on the squared distance of get_loss_original (pcl.py:399-404) */
    int32_t pixel_weight[DCN_NUM_LISTS]; /* 1: weight each term by min(|uv(gt) - uv(b)|, M_pixel)/M_pixel (pcl.py:307-334);
                                            the list must hold (len/len_match) consecutive entries per match */
    float m_pixel;
    int32_t image_width;
    float match_loss_weight;
    float non_match_loss_weight;
    int32_t scale_by_hard_negatives;   /* training.yaml:59 (or scale_by_hard_negatives_DIFFERENT_OBJECT for mode 1) */
    int32_t compose;                   /* DCN_COMPOSE_* */
} dcn_loss_config;

/* Bytes of scratch `dcn_contrastive_loss_forward` needs for `num_pairs` image pairs whose longest list
 * has `max_list_len` entries. */
size_t dcn_loss_workspace_bytes(int num_pairs, int64_t max_list_len);

/*
 * Forward.  desc_a / desc_b: [num_pairs, HW, D] contiguous fp32.
 *   terms      [num_pairs][5] : (loss_p, match_loss, masked_scaled, background_scaled, blind_scaled)  -- the 5-tuple
 *                               loss_composer.get_loss returns (loss_composer.py:143)
 *   sums       [num_pairs][4] : raw per-list sums  (match: sum ||a-b||^2; others: sum l_j [*w_j])
 *   hard_neg   [num_pairs][4] : int32 #{j : l_j != 0} per list (pcl.py:210-211); match entry = list length
 *   loss       [1]            : mean_p loss_p
 *   per_term   nullable [offsets[4*num_pairs]] : every pair's own term (match: ||a-b||^2, others l_j) --
 *                               the vector PixelwiseContrastiveLoss.non_match_descriptor_loss returns
 *   status     [1] int32      : set to 1 if any index was outside [0, HW) (such pairs are skipped)
 */
int dcn_contrastive_loss_forward(const float* desc_a, const float* desc_b, int num_pairs, int64_t hw, int d,
                                 const int64_t* idx_a, const int64_t* idx_b, const int64_t* offsets_host,
                                 const int64_t* offsets_dev, const dcn_loss_config* cfg, float* terms, float* sums,
                                 int32_t* hard_neg, float* loss, float* per_term, int32_t* status, void* workspace,
                                 void* stream);

/*
 * Backward of `loss` w.r.t. desc_a / desc_b.  grad_loss: device scalar (the upstream gradient).
 * grad_a / grad_b [num_pairs, HW, D] are zero-filled by this call and then accumulated into (duplicate
 * indices are legal and accumulate, correspondence_finder.py:326-328).  `sums` / `hard_neg` are the
 * forward outputs (read on device; no host round trip for the hard-negative count, unlike
 * pcl.py:210-211's nonzero()/len()).
 * pair_grad (nullable, [offsets[4*num_pairs]]): when given, the call instead back-propagates the
 * per-pair vector `per_term` of the forward: d/d desc of sum_j pair_grad[j] * per_term[j]  (the autograd of the
 * vector PixelwiseContrastiveLoss.non_match_descriptor_loss returns); grad_loss / sums / hard_neg are ignored.
 */
int dcn_contrastive_loss_backward(const float* desc_a, const float* desc_b, int num_pairs, int64_t hw, int d,
                                  const int64_t* idx_a, const int64_t* idx_b, const int64_t* offsets_host,
                                  const int64_t* offsets_dev, const dcn_loss_config* cfg, const float* sums,
                                  const int32_t* hard_neg, const float* grad_loss, const float* pair_grad,
                                  float* grad_a, float* grad_b, void* stream);

/* The same forward / backward pair with the backward's gathers traded for coalesced streams (round 4): the forward also
 * writes, per pixel pair, the difference vector a - b and the factor s with  d loss / d a = coef(list, image pair) * s * (a - b)
 * into pair_records (dcn_loss_saved_floats(total pairs, d) floats: [total][d] differences, then [total] factors); the backward
 * reads those records -- no descriptor is gathered a second time -- and scatter-adds the same values, bit for bit, as
 * dcn_contrastive_loss_backward.  prefilled != 0: the caller has already zero-filled grad_a / grad_b (dcn_fill_bytes on another
 * stream, overlapped with the forward kernels); the call then only accumulates. */
size_t dcn_loss_saved_floats(int64_t total_pairs, int d);
int dcn_contrastive_loss_forward_save(const float* desc_a, const float* desc_b, int num_pairs, int64_t hw, int d,
                                      const int64_t* idx_a, const int64_t* idx_b, const int64_t* offsets_host,
                                      const int64_t* offsets_dev, const dcn_loss_config* cfg, float* terms, float* sums,
                                      int32_t* hard_neg, float* loss, float* per_term, int32_t* status, void* workspace,
                                      float* pair_records, void* stream);
int dcn_contrastive_loss_backward_saved(int num_pairs, int64_t hw, int d, const int64_t* idx_a, const int64_t* idx_b,
                                        const int64_t* offsets_host, const int64_t* offsets_dev, const dcn_loss_config* cfg,
                                        const int32_t* hard_neg, const float* grad_loss, const float* pair_records,
                                        int prefilled, float* grad_a, float* grad_b, void* stream);

/* The same backward pass with BIT-REPRODUCIBLE gradient maps: every contribution is accumulated as 64-bit fixed point under a
 * per-image-pair power-of-two scale with integer atomics (order-independent), then converted once.  `workspace`:
 * dcn_loss_exact_workspace_bytes(num_pairs, hw, d) bytes (two int64 maps the size of the gradient maps x 2 + a word per pair).
 * grad_a / grad_b are written in full.  DCN_E_UNSUPPORTED when an image pair has >= 2^22 pixel pairs (use the float path).
 * Replaces the index_add_ backward of pixelwise_contrastive_loss.py:154-165,192-210 like the call above. */
size_t dcn_loss_exact_workspace_bytes(int num_pairs, int64_t hw, int d);
int dcn_contrastive_loss_backward_saved_exact(int num_pairs, int64_t hw, int d, const int64_t* idx_a, const int64_t* idx_b,
                                              const int64_t* offsets_host, const int64_t* offsets_dev,
                                              const dcn_loss_config* cfg, const int32_t* hard_neg, const float* grad_loss,
                                              const float* pair_records, void* workspace, float* grad_a, float* grad_b,
                                              void* stream);
/* Stream-ordered fill of n bytes (n % 4 == 0) with a byte value, as a kernel (an ordinary node under hipGraph capture). */
int dcn_fill_bytes(void* p, int byte_value, size_t n, void* stream);

/* Triplet variant (pixelwise_contrastive_loss.py:104-129, loss_composer.py:145-166):
 *   loss = 1/n * sum_i sum_k max(0, (a_k - m_k)^2 - (a_k - q_k)^2 + alpha),   a = A[non_a[i]], m = B[match_b[i / (n / n_match)]],
 *   q = B[non_b[i]]  -- hinge per descriptor component, exactly as the reference computes it.  n % n_match == 0.
 * desc_a / desc_b: [hw, d] of ONE image pair.  backward ACCUMULATES into grad_a / grad_b (zero-fill them first). */
size_t dcn_triplet_loss_workspace_bytes(int64_t n);
int dcn_triplet_loss_forward(const float* desc_a, const float* desc_b, int64_t hw, int d, const int64_t* non_a,
                             const int64_t* match_b, const int64_t* non_b, int64_t n, int64_t n_match, float alpha,
                             float* loss, int32_t* status, void* workspace, void* stream);
int dcn_triplet_loss_backward(const float* desc_a, const float* desc_b, int64_t hw, int d, const int64_t* non_a,
                              const int64_t* match_b, const int64_t* non_b, int64_t n, int64_t n_match, float alpha,
                              const float* grad_loss, float* grad_a, float* grad_b, void* stream);

/* =====================================================================================================
 * 2. Dilated-ResNet FCN backbone  (kernels K1-K8, K10)
 *
 * Replaces `self.fcn(img_tensor)` in DenseCorrespondenceNetwork.forward
 * (dense_correspondence/network/dense_correspondence_network.py:239-263) and its autograd, for
 * fcn = resnet_dilated.Resnet{18,34,50,101}_8s(num_classes=D) (network.py:373-375; the module itself is
 * third-party, see oracle/resnet_dilated_oracle.py).
 *
 * A plan fixes (architecture, N, H, W, D).  Parameters are passed as a host array of device pointers in
 * the order dcn_plan_param_name enumerates them, which is the reference checkpoint's state_dict order
 * (`conv1.weight, bn1.weight, bn1.bias, layer1.0.conv1.weight, ... fc.weight, fc.bias`); BN running
 * statistics as a second array (`bn1.running_mean, bn1.running_var, layer1.0.bn1.running_mean, ...`).
 * ===================================================================================================== */

typedef struct dcn_plan dcn_plan;

/* arch: "Resnet18_8s" | "Resnet34_8s" | "Resnet50_8s" | "Resnet101_8s".  base_width = 64 for the real
 * networks (smaller widths, multiples of 4, exist for tests). */
int dcn_plan_create(const char* arch, int base_width, int n, int h, int w, int d, dcn_plan** out);
/* `groups` (1 or 2) independent batches of n / groups images stacked along N: ONE launch sequence computes what
 * `groups` consecutive forward calls of the reference compute (training.py:329-333 forwards img_a and img_b through the
 * same weights) -- batch-norm statistics, their backward and the running-statistics updates are per group, in order.
 * Larger launches fill the 256 CUs better at small batch.  DCN_E_UNSUPPORTED if a group's rows are not tile-aligned. */
int dcn_plan_create_grouped(const char* arch, int base_width, int n, int groups, int h, int w, int d, dcn_plan** out);
void dcn_plan_destroy(dcn_plan* plan);
/* A training-mode forward call leaves a small host-side record keyed by its `saved` pointer (what the matching backward call
 * must agree with).  dcn_plan_forget_saved drops it when the caller releases that arena without (or after) differentiating
 * it -- returns how many records were dropped; dcn_plan_num_forward_records counts the records a plan holds. */
int dcn_plan_forget_saved(dcn_plan* plan, const void* saved);
int dcn_plan_num_forward_records(const dcn_plan* plan);

/* How the plan's convolutions multiply.  Both modes take and return fp32 tensors and accumulate in fp32.
 *   DCN_CONV_FP32  : fp32 MFMA (v_mfma_f32_32x32x2_f32), 157 TFLOP/s peak.
 *   DCN_CONV_F16X3 : split-fp16 -- every operand element x is split on the fly into fp16 hi + lo with
 *                    s*x = hi + lo (s a power of two: 64 for weights, chosen from the tensor's abs-max for
 *                    gradients AND activations -- every tensor that feeds a convolution carries a device scalar
 *                    with (a bound of) its abs-max, written by the kernel that produced it) and
 *                    hi*hi + hi*lo + lo*hi runs on the fp16 MFMA pipe.  ~22 mantissa bits per operand:
 *                    indistinguishable from fp32 on this network, ~2x faster.  Operand range: any finite
 *                    fp32 tensor (the pre-scale brings its abs-max to <= 4096); |weight| < 1023.
 * Default: DCN_CONV_F16X3, or the environment variable DCN_CONV_MODE = "fp32" | "f16x3" at plan creation.
 * The saved / workspace arenas are sized for either mode, so the mode may be switched between steps (not between a
 * forward and its backward). */
#define DCN_CONV_FP32 0
#define DCN_CONV_F16X3 1
int dcn_plan_set_conv_mode(dcn_plan* plan, int mode);
int dcn_plan_conv_mode(const dcn_plan* plan);

int dcn_plan_num_params(const dcn_plan* plan);
int dcn_plan_num_bn(const dcn_plan* plan);
/* name (without the "resnet34_8s." prefix) and logical OIHW / [C] shape of parameter i; ndim is 4 or 1. */
int dcn_plan_param_info(const dcn_plan* plan, int i, char* name, int name_cap, int64_t shape[4], int* ndim);
/* name prefix ("layer1.0.bn1") and channel count of batch-norm j. */
int dcn_plan_bn_info(const dcn_plan* plan, int j, char* name, int name_cap, int64_t* channels);

/* number of batch norms of the last dcn_backbone_backward call whose backward reduction ran in the epilogue of the dgrad
 * that produced their upstream gradient (split-fp16 mode; DCN_BN_BWD_FUSED=0 disables) */
int dcn_plan_fused_bn_backward(const dcn_plan* plan);

/* Inside the `saved` buffer of a forward call, at byte offset dcn_plan_activation_absmax_offset: one float per
 * activation tensor that feeds a convolution (dcn_plan_num_activation_slots of them; slot 0 = the input image) with its
 * abs-max as the split-fp16 kernels used it (all zero in fp32 mode), followed by one int32 STATUS word: bit 0 = some
 * convolution input was not finite (inf / NaN) in that call.  Read it after the call (device memory; no sync is forced). */
int dcn_plan_num_activation_slots(const dcn_plan* plan);
size_t dcn_plan_activation_absmax_offset(const dcn_plan* plan);

/* Gradient buckets for data-parallel training: the parameters split into
 * dcn_plan_num_grad_buckets contiguous ranges of the state-dict order -- bucket k = parameters
 * [first_param(k), first_param(k - 1)), bucket 0 ending at the last parameter -- numbered in the order the backward pass
 * completes them (k = 0: fc + layer4, 62 % of Resnet34_8s; 1: layer3; 2: the rest).  dcn_backbone_backward records an
 * event per bucket on its stream once every launch that writes one of the bucket's gradients has been enqueued;
 * dcn_plan_stream_wait_grad_bucket makes `stream` wait for that event, so that a communication stream can all-reduce
 * bucket k (RCCL) while the caller's stream is still computing buckets k + 1, ...  Call it after
 * dcn_backbone_backward has returned; DCN_E_UNSUPPORTED before the plan's first backward pass. */
int dcn_plan_num_grad_buckets(const dcn_plan* plan);
int dcn_plan_grad_bucket_first_param(const dcn_plan* plan, int k);
int dcn_plan_stream_wait_grad_bucket(dcn_plan* plan, int k, void* stream);

size_t dcn_plan_saved_bytes(const dcn_plan* plan);     /* activations kept from forward to backward */
size_t dcn_plan_workspace_bytes(const dcn_plan* plan); /* scratch shared by forward and backward */
double dcn_plan_forward_flops(const dcn_plan* plan);   /* algorithmic conv FLOPs of one forward (2*MAC) */

/*
 * Forward.  image: [N,3,H,W] fp32 NCHW (what the reference's DataLoader yields, training.py:311-312).
 * descriptors: [N,H,W,D] fp32 (== logical [N,D,H,W] channels_last).
 * training != 0: batch statistics over the N images of this call + running-stat update with `momentum`
 * (nn.BatchNorm2d semantics); training == 0: running statistics.  normalize != 0 applies
 * network.py:256-259 (per-pixel L2 normalisation over D).
 * saved (dcn_plan_saved_bytes) and workspace (dcn_plan_workspace_bytes) are always required.
 */
int dcn_backbone_forward(dcn_plan* plan, const float* image, const float* const* params,
                         float* const* bn_running, float momentum, float eps, int training, int normalize,
                         float* descriptors, void* saved, void* workspace, void* stream);

/* The same call for a grouped plan (dcn_plan_create_grouped, groups == 2) whose two image batches are separate tensors:
 * images [0, N/2) are read from image_a, [N/2, N) from image_b ([N/2,3,H,W] each) -- forward(img_a), forward(img_b) of
 * training.py:329-333 as one launch sequence without a concatenated copy of the batches. */
int dcn_backbone_forward_pair(dcn_plan* plan, const float* image_a, const float* image_b, const float* const* params,
                              float* const* bn_running, float momentum, float eps, int training, int normalize,
                              float* descriptors, void* saved, void* workspace, void* stream);

/*
 * Launch-level timing of the two matrix-core kernels (for bench.py's roofline): between begin and end every
 * conv_gemm_kernel (forward + dgrad; category 0) and conv_wgrad_kernel (category 1) launch made through this
 * plan is bracketed by hipEventRecord on the caller's stream.  `end` synchronises on the recorded events and
 * returns, per category, the summed kernel milliseconds, the number of launches and the algorithmic FLOPs
 * (2 * MACs of the convolution being computed, padding not counted).
 */
int dcn_plan_profile_begin(dcn_plan* plan);
int dcn_plan_profile_end(dcn_plan* plan, double ms[2], int64_t launches[2], double flops[2]);
/* The same with a third category: [2] = the part of category 0 that ran on the pre-split (hl32) LDS-DMA kernel
 * (conv_hl_kernels.hip; an operand split pass that had to run in front of a launch is inside its bracket). */
int dcn_plan_profile_end3(dcn_plan* plan, double ms[3], int64_t launches[3], double flops[3]);
/* EVERY launch the engine makes between begin and end, by category (bench.py: `roofline_elementwise`, `kernel_ms_sum`):
 * work[c] = algorithmic FLOPs for the three matrix-core categories, algorithmic HBM bytes (4 B x elements read and
 * written, the operands a pass needs once) for the streaming ones, 0 where neither is meaningful.  Category
 * DCN_PROF_GEMM_HL is a subset of DCN_PROF_GEMM (as in dcn_plan_profile_end3); the others are disjoint, so the launches
 * of a step are  sum over c != DCN_PROF_GEMM_HL.  Arrays of DCN_PROF_NCAT entries. */
enum {
    DCN_PROF_GEMM = 0,           /* gather-GEMM convolutions, forward + dgrad (all kernels) */
    DCN_PROF_WGRAD = 1,          /* weight-gradient GEMMs (their slab-reduce launch included) */
    DCN_PROF_GEMM_HL = 2,        /* the part of DCN_PROF_GEMM on the pre-split (hl32) kernel */
    DCN_PROF_BN_APPLY = 3,       /* batch-norm apply (+ residual, ReLU, mask, hl32 image): one streaming pass */
    DCN_PROF_BN_BWD_REDUCE = 4,  /* batch-norm backward reduction pass */
    DCN_PROF_BN_BWD_APPLY = 5,   /* batch-norm backward apply pass (+ the gradient's operand images) */
    DCN_PROF_BN_FINALIZE = 6,    /* the two per-channel finalize kernels (latency-bound) */
    DCN_PROF_RESAMPLE = 7,       /* max pool, bilinear upsample (forward and backward), NCHW -> NHWC4 of the input */
    DCN_PROF_OTHER = 8,          /* fills, weight splits, stand-alone operand splits, status / bias-gradient kernels */
    DCN_PROF_NCAT = 9
};
int dcn_plan_profile_end_all(dcn_plan* plan, double ms[DCN_PROF_NCAT], int64_t launches[DCN_PROF_NCAT],
                             double work[DCN_PROF_NCAT]);

/* Backward.  grad_descriptors: [N,H,W,D]; grads[i] receives dL/d params[i] (overwritten, same layout as
 * params[i]).  `saved` is the buffer the matching forward filled; `normalize` must be the forward's flag. */
int dcn_backbone_backward(dcn_plan* plan, const float* grad_descriptors, const float* const* params,
                          const void* saved, void* workspace, float* const* grads, int normalize, void* stream);
/* Grouped plan, the two batches' descriptor gradients as separate [N/2,H,W,D] tensors (the two outputs of a
 * dcn_backbone_forward_pair call receive their gradients separately).  Both backward entry points return DCN_E_INVALID when
 * `saved` was not filled by a training-mode forward call of this plan in the plan's current arithmetic, or when the tuning
 * switches (dcn_reload_env) changed since in a way that would make the pass read a tensor that call did not write. */
int dcn_backbone_backward_pair(dcn_plan* plan, const float* grad_a, const float* grad_b, const float* const* params,
                               const void* saved, void* workspace, float* const* grads, int normalize, void* stream);

/* =====================================================================================================
 * 3. Individual kernels, exported for unit tests and micro-benchmarks (same conventions).
 * ===================================================================================================== */

typedef struct dcn_conv_desc {
    int32_t n, hin, win, cin;   /* input  [n, hin, win, cin] NHWC, cin % 4 == 0 */
    int32_t hout, wout, cout;   /* output [n, hout, wout, cout], leading dimension ldc >= cout */
    int32_t kh, kw, stride, pad, dil;
    int32_t ldc;
    int32_t group_rows;         /* 0, or: output rows (pixels) per batch-norm group -- the forward kernel then picks an M
                                   tile that divides it, so that no row of bn_partial mixes two groups (% 64 == 0 required) */
} dcn_conv_desc;

/* out = conv(in, w) [+ bias];  w: [cout][kh][kw][cin].  If bn_partial != NULL also writes per-M-tile
 * partial sums for batch-norm statistics: bn_partial[tile][3][cout] (sum, sum of squares, max |x|); the number of
 * tiles is returned by dcn_conv_num_mtiles. */
/* workspace (nullable): dcn_conv_gemm_workspace(c, dgrad) bytes; enables the stream-K work split used when the layer
 * has too few output tiles to load all 256 CUs evenly (small batch). */
int dcn_conv_forward(const dcn_conv_desc* c, const float* in, const float* w, const float* bias, float* out,
                     float* bn_partial, void* workspace, void* stream);
size_t dcn_conv_gemm_workspace(const dcn_conv_desc* c, int dgrad);
int dcn_conv_num_mtiles(const dcn_conv_desc* c);
/* din = conv_transpose(dout, w) [+ add];  wt: [cin][kh][kw][cout] (see dcn_transpose_weight). */
int dcn_conv_dgrad(const dcn_conv_desc* c, const float* dout, const float* wt, const float* add, float* din,
                   void* workspace, void* stream);
/* dw[cout][kh][kw][cin] = sum_m dout[m][cout] * in[pix(m,tap)][cin]; slabs: scratch of dcn_conv_wgrad_workspace bytes */
int dcn_conv_wgrad(const dcn_conv_desc* c, const float* in, const float* dout, float* dw, void* slabs, void* stream);
size_t dcn_conv_wgrad_workspace(const dcn_conv_desc* c);
/* wt[c][tap][0..ldn) = (w[0..cout)[tap][c], zeros);  ldn >= cout is the row pitch of dout in dcn_conv_dgrad */
int dcn_transpose_weight(const float* w, float* wt, int cout, int taps, int cin, int ldn, void* stream);

/* ---- split-fp16 ("f16x3") variants: same tensors and results to ~fp32 accuracy, products on the fp16 matrix pipe
 * (see csrc/conv_f16_kernels.hip).  Weights are pre-split with dcn_split_rows_f16 into two fp16 arrays of
 * rows x dcn_f16_kpad(K) (hi, lo), scaled by the power of two `w_scale`. */
int dcn_f16_kpad(int k);
int dcn_split_rows_f16(const float* w, void* hi, void* lo, int64_t rows, int k, float scale, void* stream);
int dcn_conv_num_mtiles_f16(const dcn_conv_desc* c);
size_t dcn_conv_gemm_workspace_f16(const dcn_conv_desc* c, int dgrad);
/* in_absmax: device scalar >= max|in| (picks the power-of-two pre-scale of the activation tensor) or NULL for scale 1 */
int dcn_conv_forward_f16(const dcn_conv_desc* c, const float* in, const float* in_absmax, const void* w_hi, const void* w_lo,
                         float w_scale, const float* bias, float* out, float* bn_partial, void* workspace, void* stream);
/* dout_absmax: device scalar >= max|dout| (picks the power-of-two pre-scale of the gradient tensor) or NULL */
int dcn_conv_dgrad_f16(const dcn_conv_desc* c, const float* dout, const void* wt_hi, const void* wt_lo, float w_scale,
                       const float* dout_absmax, const float* add, float* din, void* workspace, void* stream);
/* dcn_conv_dgrad_f16 whose result din is the upstream gradient of a train-mode batch norm (the one that produced this
 * convolution's input, training.py:345 through the backbone's BasicBlock): the batch norm's backward REDUCTION runs in the
 * GEMM epilogue.  din receives the gradient masked by relu_mask (the bytes dcn_bn_forward wrote for that batch norm's
 * output; NULL: no ReLU) and bn_partial[dcn_conv_dgrad_bn_num_mtiles_f16(c)][cin][4] the per-tile sums that
 * dcn_bn_backward_from_partial consumes.  bn_x: the batch norm's input [n, hin, win, cin]; bn_stats: its statistics
 * [4][cin] (per group when c->group_rows is set: M tiles then never straddle a group). */
int dcn_conv_dgrad_bn_num_mtiles_f16(const dcn_conv_desc* c);
int dcn_conv_dgrad_bn_f16(const dcn_conv_desc* c, const float* dout, const void* wt_hi, const void* wt_lo, float w_scale,
                          const float* dout_absmax, const float* add, float* din, const float* bn_x,
                          const unsigned char* relu_mask, const float* bn_stats, float* bn_partial, void* workspace,
                          void* stream);

/* dcn_split_weights_scaled_f16 (forward images; row_scale nullable) with a range check: bit 1 of the device word *status (not
 * cleared by the call) is raised when some |scale * row_scale * w| leaves fp16's range or is NaN -- with the engine's fixed
 * weight scale 64 that is |w| >= 1023.  The backbone engine passes the status word behind its activation abs-max slots
 * (dcn_plan_activation_absmax_offset). */
int dcn_split_weights_checked_f16(int n, const float* const* w, const float* const* row_scale, void* const* hi, void* const* lo,
                                  const int* cout, const int* taps, const int* cin, float scale, int* status, void* stream);

/* ---- pre-split ("hl32") operand path of the wide layers (csrc/conv_hl_kernels.hip): the forward convolution and dgrad of
 * every stride-1 convolution with >= 256 destination channels and source channels % 32 == 0 (layers 3-4 of the backbone behind
 * network.py:255 / training.py:345).  Operands are "hl32" tensors -- per 32-channel chunk one 128-byte line
 * [hi x32 | lo x32] fp16, the byte size of the fp32 tensor -- so that the GEMM loop is LDS-DMA + MFMA only (256 x 256 tiles,
 * two wavefront groups one phase apart; round 5: 160 x 256 / 160 x 128 tiles with a K split for the launches the big tiles do
 * not fill the chip with -- B = 1, training.yaml:14 -- from 128 destination channels on: csrc/conv_hlx_kernels.hip).  Results equal dcn_conv_forward_f16 / dcn_conv_dgrad_f16 (same products, other
 * summation order).  dcn_conv_hl_eligible: 1 when the descriptor qualifies (forward: dgrad = 0). */
int dcn_conv_hl_eligible(const dcn_conv_desc* c, int dgrad);
int dcn_conv_num_mtiles_hl(const dcn_conv_desc* c);
/* rows per tile of the launch (256, or 192 / 320 where that fills the 256 CUs better; DCN_GEMM_HL_ROWS forces one) */
int dcn_conv_tile_rows_hl(const dcn_conv_desc* c, int dgrad);
size_t dcn_conv_gemm_workspace_hl(const dcn_conv_desc* c, int dgrad);
/* the tile the launch of this convolution takes under the tuning of the moment: info[0..5] = rows (256 / 192 / 320: the big
 * tiles; 160: the small-tile kernel of csrc/conv_hlx_kernels.hip -- round 5: the reference's batch_size 1,
 * training.yaml:14, and its two separate forward calls, training.py:329-333), columns, K groups inside the workgroup,
 * workgroups per tile along K, M tiles, N tiles.  Returns 0, or DCN_E_UNSUPPORTED when no tile height fits. */
int dcn_conv_hl_shape_info(const dcn_conv_desc* c, int dgrad, int* info6);
/* fp32 [rows][channels] (channels % 32 == 0) -> hl32, scaled by the power of two chosen from *absmax (NULL: 1) */
int dcn_split_act_hl32(const float* src, const float* absmax, void* dst, int64_t rows, int channels, void* stream);
/* n weight tensors w[i] = [cout][taps][cin] -> out[i] = hl32 [cout][taps*cin/32][hi|lo], or (transposed) the dgrad image
 * [cin][taps*ldn/32][hi|lo]; all arrays are HOST arrays */
int dcn_split_weights_hl32(int n, const float* const* w, void* const* out, const int* cout, const int* taps, const int* cin,
                           const int* ldn, int transposed, float scale, void* stream);
int dcn_conv_forward_hl(const dcn_conv_desc* c, const void* in_hl, const float* in_absmax, const void* w_hl, float w_scale,
                        const float* bias, float* out, float* bn_partial, void* workspace, void* stream);
int dcn_conv_dgrad_hl(const dcn_conv_desc* c, const void* dout_hl, const void* wt_hl, float w_scale, const float* dout_absmax,
                      const float* add, float* din, void* workspace, void* stream);

/* Weight gradient on hl32 operands (csrc/wgrad_hl_kernels.hip; the wgrad of training.py:345 for the same wide layers): x_hl /
 * dout_hl are the hl32 images of the convolution's input and of the output gradient (scaled by the powers of two chosen from
 * *x_absmax / *dout_absmax).  256 x 256 tiles, pixel-major tiles by LDS-DMA, k-major fragments by transposing LDS reads.
 * Same result as dcn_conv_wgrad_f16 (same products, other summation order); bit-reproducible.
 * Round 5: the narrow 3 x 3 layers (64 / 128 input channels, stride 1, dilation 1: ResNet layers 1 and 2) take a second kernel behind
 * the same entry points -- 64 output channels x nine taps x all input channels per workgroup, three row windows of x per
 * 32-pixel stage (conv_wgrad_hlr_kernel).  dcn_conv_wgrad_hl_kind: which kernel dcn_conv_wgrad_hl launches for c now --
 * 0 none (unsupported), 1 the 256 x 256 tile kernel, 2 the row-window kernel. */
int dcn_conv_wgrad_hl_eligible(const dcn_conv_desc* c);
int dcn_conv_wgrad_hl_kind(const dcn_conv_desc* c);
size_t dcn_conv_wgrad_workspace_hl(const dcn_conv_desc* c);
int dcn_conv_wgrad_hl(const dcn_conv_desc* c, const void* x_hl, const float* x_absmax, const void* dout_hl,
                      const float* dout_absmax, float* dw, void* slabs, void* stream);

/* The backbone's stem (7x7 / stride 2 / pad 3 on 3 + 1 zero input channels; K1 of SURVEY.md section 8a) as a uniform-tap
 * convolution: a filter ROW is one 32-K chunk (8 pixels x 4 channels = 128 contiguous bytes of the NHWC4 image, the 8th
 * pixel with zero weights), so the gather is the wide layers' per-row buffer load instead of a per-element tap decode.
 * w4: [cout][7][7][4] fp32; hi / lo receive [cout][7][8][4] fp16.  Same result as dcn_conv_forward_f16. */
int dcn_split_stem_weights_f16(const float* w4, void* hi, void* lo, int cout, float scale, void* stream);
int dcn_conv_stem_forward_f16(const dcn_conv_desc* c, const float* in, const float* in_absmax, const void* w_hi, const void* w_lo,
                              float w_scale, float* out, float* bn_partial, void* stream);

/* All weight tensors of a network in one launch: w[i] = [cout[i]][taps[i]][cin[i]] (device), hi[i] / lo[i] (device) receive
 * the forward image [cout][kpad(taps*cin)] or, transposed != 0, the dgrad image [cin][kpad(taps*ldn[i])] of
 * dcn_transpose_weight + dcn_split_rows_f16.  The seven arrays themselves are HOST arrays of length n. */
int dcn_split_weights_f16(int n, const float* const* w, void* const* hi, void* const* lo, const int* cout, const int* taps,
                          const int* cin, const int* ldn, int transposed, float scale, void* stream);

/* as above with optional per-output-channel factors (forward images only): row_scale[i] = NULL or device vector [cout[i]] */
int dcn_split_weights_scaled_f16(int n, const float* const* w, const float* const* row_scale, void* const* hi,
                                 void* const* lo, const int* cout, const int* taps, const int* cin, const int* ldn,
                                 int transposed, float scale, void* stream);
/* inference: out = [relu](conv(in, w) + bias [+ add]) in one pass (eval-mode batch norm folded into w and bias);
 * out_absmax (nullable): device scalar raised to max|out| -- the in_absmax of the convolution that reads `out` next */
int dcn_conv_forward_fused_f16(const dcn_conv_desc* c, const float* in, const float* in_absmax, const void* w_hi,
                               const void* w_lo, float w_scale, const float* bias, const float* add, int relu, float* out,
                               float* out_absmax, void* workspace, void* stream);

/* wgrad consumes PRE-SPLIT operands (every element takes part in many tiles, so the fp32 -> fp16 hi/lo split is done
 * once per tensor).  Both split tensors have the byte size of their fp32 source:
 *   activations  xs[pixel][c/4][hi x4 | lo x4]                       dcn_split_act_f16 (n elements, n % 4 == 0)
 *   out-gradient dq[m/4][4 sub-planes][ldc/4][2 channels x 4 pixels], scaled by the power of two chosen from *absmax
 *                (dcn_split_grad_blocked_f16; dcn_grad_blocked_bytes(M, ldc) bytes).  dout_absmax = the same scalar. */
int dcn_split_act_f16(const float* src, void* xs, int64_t n, void* stream);
size_t dcn_grad_blocked_bytes(int m, int ld);
int dcn_split_grad_blocked_f16(const float* dy, int m, int ld, const float* absmax, void* dq, void* stream);
/* xs_is_fp32 != 0: `xs` is the fp32 activation tensor itself, split on the fly (cheaper than a split pass when every
 * element is only used by a few tiles, e.g. 1x1 convolutions); x_absmax (nullable, fp32 operand only): device scalar
 * >= max|xs| for its power-of-two pre-scale */
int dcn_conv_wgrad_f16(const dcn_conv_desc* c, const void* xs, int xs_is_fp32, const float* x_absmax, const void* dq,
                       const float* dout_absmax, float* dw, void* slabs, void* stream);
size_t dcn_conv_wgrad_workspace_f16(const dcn_conv_desc* c);

/* Train-mode batch norm (+ residual) (+ ReLU) of a convolution output x [rows][c] (kernel K7; nn.BatchNorm2d as the
 * backbone uses it): statistics from the per-M-tile partial sums the convolution's epilogue wrote (bn_partial
 * [mtiles][3][c], dcn_conv_forward), running statistics updated with `momentum` (unbiased variance); training == 0: the
 * running statistics are used instead.  y = [relu](x * scale + shift [+ res]); relu_mask (nullable): one byte per
 * float4 of y, bit j = y[4 i + j] > 0.  stats [4][c] receives scale, shift, mean, invstd (read by dcn_bn_backward). */
int dcn_bn_forward(const float* x, const float* bn_partial, int mtiles, int c, int64_t rows, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float momentum, float eps, int training,
                   const float* res, int relu, float* y, unsigned char* relu_mask, float* stats, void* stream);
/* dx, dgamma, dbeta of the above given dy (masked by relu_mask when given); g_out (nullable) receives the masked dy
 * (the residual branch's gradient).  workspace: dcn_bn_backward_workspace(rows, c) bytes. */
size_t dcn_bn_backward_workspace(int64_t rows, int c);
int dcn_bn_backward(const float* dy, const unsigned char* relu_mask, const float* x, const float* stats, const float* gamma,
                    int c, int64_t rows, float* dgamma, float* dbeta, float* dx, float* g_out, void* workspace,
                    void* stream);
/* dcn_bn_backward with the reduction already done by dcn_conv_dgrad_bn_f16 (dy masked, bn_partial [mtiles][c][4]);
 * workspace: 3 * c floats */
int dcn_bn_backward_from_partial(const float* dy, const float* bn_partial, int mtiles, const float* x, const float* stats,
                                 const float* gamma, int c, int64_t rows, float* dgamma, float* dbeta, float* dx,
                                 void* workspace, void* stream);
/* 3x3 / stride 2 / pad 1 max pool (kernel K2) of in [n,hin,win,c] -> out [n,(hin+1)/2,(win+1)/2,c]; argmax (nullable in
 * forward): one byte per output element (window position 0..8); backward gathers with it (deterministic). */
int dcn_maxpool_forward(const float* in, int n, int hin, int win, int c, float* out, unsigned char* argmax, void* stream);
int dcn_maxpool_backward(const float* gout, const unsigned char* argmax, int n, int hin, int win, int c, float* gin,
                         void* stream);

/* bilinear xS upsample, align_corners=True (F.upsample_bilinear): low [n,hl,wl,ldl] -> out [n,h,w,d] */
int dcn_upsample_forward(const float* low, int n, int hl, int wl, int ldl, int d, int h, int w, int normalize,
                         float* out, void* stream);
/* glow [n,hl,wl,ldl] (pad channels zeroed); tmp: scratch of dcn_upsample_backward_tmp_bytes(n, hl, w, d) bytes */
int dcn_upsample_backward(const float* gout, int n, int hl, int wl, int ldl, int d, int h, int w, float* glow,
                          float* tmp, void* stream);
size_t dcn_upsample_backward_tmp_bytes(int n, int hl, int w, int d);

/* =====================================================================================================
 * 4. Best-match search over a descriptor image (evaluation / heat-map side)
 *
 * Replaces DenseCorrespondenceNetwork.find_best_match / find_best_match_for_descriptor
 * (dense_correspondence/network/dense_correspondence_network.py:488-550: numpy
 * `sqrt(sum(square(res_b - d), axis=2))` + argmin, once per query) for Q queries in one pass.
 *   res        [HW][D] fp32 descriptor image (the [H,W,D] tensor forward_single_image_tensor returns)
 *   queries    [Q][D]
 *   mask       nullable [HW] uint8: only pixels with mask != 0 are candidates
 *   best_idx   [Q] int64 flat index u + W*v of the first minimum (np.argmin order); -1 if the mask is empty
 *   best_dist  [Q] the distance at best_idx
 *   norm_diffs nullable [Q][HW]: the full distance images
 *   workspace  dcn_find_best_match_workspace(Q) bytes
 * ===================================================================================================== */
/* Match statistics of evaluation.py:1046-1100 for q query matches in ONE pass over the descriptor image res [hw][d]
 * (row length w):  queries[i] = res_a[uv_a_i], gt_idx[i] = flat index of the ground-truth match in image b.
 *   best_idx / best_dist  [2][q]: argmin / min of d over the image, and of d + (1 - mask) * 1e6 (mask NULL: same as the image)
 *   count                 [2][q]: pixels with d < ||queries[i] - res[gt_idx[i]]|| (image / masked)
 *   dist_sum              [2][q]: sum of the pixel distances of those pixels to the ground-truth pixel
 *   gt_dist               [q]:    that ground-truth descriptor distance */
size_t dcn_match_statistics_workspace(int q);
int dcn_match_statistics(const float* res, int64_t hw, int w, int d, const float* queries, const int64_t* gt_idx, int q,
                         const unsigned char* mask, int64_t* best_idx, float* best_dist, int32_t* count, float* dist_sum,
                         float* gt_dist, void* workspace, void* stream);

int dcn_find_best_match(const float* res, int64_t hw, int d, const float* queries, int q, const unsigned char* mask,
                        int64_t* best_idx, float* best_dist, float* norm_diffs, void* workspace, void* stream);
size_t dcn_find_best_match_workspace(int q);

/* =====================================================================================================
 * 5. Pair generation on the device -- replaces, for device-resident depth images,
 *    dense_correspondence/correspondence_tools/correspondence_finder.py
 *      batch_find_pixel_correspondences (:409-619, from the point where the candidate pixels are chosen, :486)
 *      create_non_correspondences       (:276-405, sampling part; its perturbation step is inert in the reference)
 *    Depth images: [h][w] uint16 millimetres.  K, K_inv: row-major 3x3 fp32; pose_a, pose_b_inv: row-major 4x4 fp32
 *    camera-to-world of image a and world-to-camera of image b (HOST pointers: they travel as kernel arguments).
 *    Outputs keep candidate order; *out_count (device scalar) matches were written to the front of the out_* arrays.
 * ===================================================================================================== */
size_t dcn_find_correspondences_workspace(int64_t n);
int dcn_find_correspondences(const uint16_t* depth_a, const uint16_t* depth_b, int h, int w, const float* K,
                             const float* K_inv, const float* pose_a, const float* pose_b_inv, const int64_t* cand_u,
                             const int64_t* cand_v, int64_t n, int64_t* out_ua, int64_t* out_va, float* out_ub,
                             float* out_vb, int64_t* out_count, void* workspace, void* stream);
/* list[0 .. *count) = flat indices of the non-zero mask pixels, increasing (torch.nonzero order) */
size_t dcn_mask_nonzero_workspace(int64_t hw);
int dcn_mask_nonzero(const float* mask, int64_t hw, int64_t* list, int64_t* count, void* workspace, void* stream);
/* list == NULL: (u, v) = (floor(rand[i] * w), floor(rand[n + i] * h))            (pytorch_rand_select_pixel, :29-34)
 * otherwise   : p = list[floor(rand[i] * *count)], (u, v) = (p % w, p / w)       (:319-324).  u, v: float [n]. */
int dcn_sample_pixels(const float* rand, int64_t n, int w, int h, const int64_t* list, const int64_t* count, float* u,
                      float* v, void* stream);

/* =====================================================================================================
 * 6. Optimizer step -- replaces `optimizer.step()` (dense_correspondence/training/training.py:346) of the
 *    torch.optim.Adam built at training.py:133-145 (lr 1e-4, weight_decay 1e-4 from training.yaml:3,6; default betas,
 *    eps; no amsgrad): one pass over n dense fp32 tensors, ceil(n / 80) launches.
 *      g' = g + weight_decay * p;  m = m + (1 - beta1)(g' - m);  v = beta2 v + (1 - beta2) g'^2
 *      p  = p - lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 *    param / grad / exp_avg / exp_avg_sq: HOST arrays of n device pointers (element i of each addresses numel[i] floats in
 *    the same memory order); they travel as kernel arguments.  step >= 1 is the count INCLUDING this update.
 * ===================================================================================================== */
int dcn_adam_step(int n, void* const* param, const void* const* grad, void* const* exp_avg, void* const* exp_avg_sq,
                  const int64_t* numel, double lr, double beta1, double beta2, double eps, double weight_decay,
                  int64_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCN_HIP_H */
