// Process-wide tuning / test overrides of the convolution launchers.  The environment is read ONCE (first use), not on
// every launch; dcn_reload_env() (include/dcn_hip.h) re-reads it -- the test-suite calls it after changing a variable.
//   DCN_CONV_MODE           fp32 | f16x3    default arithmetic of new plans
//   DCN_BACKWARD_OVERLAP    0: weight-gradient GEMMs stay on the caller's stream
//   DCN_GEMM_TILE_M         32 | 64 | 128 | 256   workgroup-tile height of the gather-GEMM kernels
//   DCN_STEM8               0: the 7x7 stem runs the generic gather path (1: as a uniform-tap convolution over filter rows)
//   DCN_GEMM_SK             0: no stream-K, 1: as decided, N > 1: force N workgroups
//   DCN_GEMM_SK_MIN_GAIN    stage times stream-K must save to be chosen (split-fp16 kernel)
//   DCN_GEMM_UNI            0: disable the uniform-tap fast path
//   DCN_GEMM_SK_FIXUP       kernel: stream-K tiles are completed by the separate fix-up kernel instead of by the last
//                           contributing workgroup of the GEMM launch itself (split-fp16 kernel)
//   DCN_BN_BWD_FUSED        1: the batch-norm backward reduction runs in the epilogue of the dgrad that produces its upstream
//                           gradient (split-fp16 mode) instead of as a separate pass.  Default 0: measured SLOWER on the
//                           MI355X (+1.3 ms per config-2 step, profiles/r2b_ab_fence_variant.txt: the one-workgroup-per-CU GEMM exposes the
//                           epilogue's extra loads, the separate pass streams at HBM speed)
//   DCN_DEFER_RESIDUAL_ADD  0: the residual branch's gradient is added in the dgrad epilogue (1: in the BN backward passes)
//   DCN_WGRAD_TILE          128: keep the 128-channel / 4-wavefront tile of the split-fp16 wgrad kernel on wide layers
//   DCN_WGRAD_DEEP          mask of the wgrad tile variants that prefetch through two register sets (1: 64-channel tile,
//                           2: 128, 4: 256; default 4)
//   DCN_WGRAD_ROLES         0: the wide tile's gradient copy is spread over all 8 wavefronts (default 1: wavefronts 4-7 only)
//   DCN_WGRAD_SPLITS        force the pixel-range split count of the split-fp16 wgrad kernel
//   DCN_GEMM_HL             0: the wide layers stay on the fp32-operand gather-GEMM (conv_f16_kernels.hip) instead of the
//                           pre-split (hl32) LDS-DMA kernel (conv_hl_kernels.hip); 2: every convolution that kernel supports
//                           takes it, whatever its size (tests)
//   DCN_HL_MIN_K            reduction length (filter taps x source channels) from which a wide convolution's forward / dgrad and
//                           weight gradient take the hl32 kernels (default 128; DCN_HL_MIN_K=1024 restores the thresholds up to round 5 -- 512 / 1024 --, which
//                           kept the 1 x 1 convolutions fed by fewer than 512 / 1024 channels -- the downsample branches, the
//                           bottleneck blocks' conv3 -- on the fp32-operand kernels: 1.3 - 2x slower there, profiles/r6e_*, r6f_*)
//   DCN_GEMM_HL_ROWS        192 / 256 / 320: tile height of the hl32 gather-GEMM (default 0: whichever quantises better on the
//                           256 CUs, hl_shape in conv_hl_kernels.hip); -320: as decided, but never 320 (the round-3 choice)
//   DCN_GEMM_HLX            0: never the small-tile variants of the hl32 gather-GEMM (conv_hlx_kernels.hip: 160 x 256 / 160 x 128
//                           tiles, K split over workgroups); 1: where hl_shape's cost model picks them (default); "kg,splits":
//                           force the K groups per workgroup (1 | 2) and the workgroups per tile along K wherever the kernel
//                           applies (0 = as decided)
//   DCN_GEMM_HLX_NARROW     0: only destinations of >= 256 channels (or fed by >= 256) take the small tiles (default 1: the 128-channel
//                           layers too -- layer 2: +1.5 % on a config-1 step, profiles/r5d_ab_config1.txt)
//   DCN_HLX_COST            "e1,e2,split": cost-model constants of hlx_shape (per-stage cost factor of the 160 x 256 and of the
//                           160 x 128 tile relative to a 256-row tile of conv_hl_kernels.hip; stages one parked partial costs)
//   DCN_WGRAD_HL            0: the wide layers' weight gradients stay on the fp32-operand kernel (conv_f16_kernels.hip) instead of
//                           the pre-split (hl32) LDS-DMA kernel (wgrad_hl_kernels.hip); 2: every supported convolution (tests)
//   DCN_HL_SETPRIO          0: no s_setprio around the MFMA bursts of the big-tile hl32 kernels (conv_hl_kernels.hip, wgrad_hl_kernels.hip)
//   DCN_HLX_STAGGER         schedule of the small-tile kernel.  0: every wavefront issues the next stage's LDS-DMA in front of its
//                           compute slot; 1: wavefronts 4-7 between the two parts of theirs (the partner on the SIMD computes
//                           meanwhile); 2: two wavefront groups one slot apart (LOAD slot | barrier | COMPUTE slot | barrier)
//   DCN_HLX_COUNTERS        0: the arrival words of the small-tile kernel's K splits live in the caller's scratch and are cleared by
//                           a fill launch in front of every split launch (default 1: a library-owned clean buffer per stream)
//   DCN_WGRAD_HLR           0: the 64 / 128-channel 3 x 3 layers' weight gradients stay on the fp32-operand kernel; 1 (default): on the
//                           row-window hl32 kernel (conv_wgrad_hlr_kernel) from about eight 32-pixel stages per workgroup (four images at 640 x 480), or
//                           from DCN_WGRAD_HLR_MIN_M output pixels when that is set;
//                           2: every convolution that kernel supports (tests)
//   DCN_WGRAD_HLR_PAIRS     0: the row-window kernel's single-row form (stage = one image row x 32 pixels) instead of the row-pair one
//   DCN_WGRAD_HL_MIN_M      output pixels from which the weight gradients of the wide layers take the hl32 kernel (default 4096)
//   DCN_HL_PRODUCERS        0: hl32 activation / gradient images are made by stand-alone split passes instead of by the
//                           batch-norm apply kernels that produce the tensors
//   DCN_STEM_POOL_FUSED     0: the stem's batch norm + ReLU as an apply pass of its own in front of the max pool (default 1: applied
//                           inside the pooling pass, the activation itself is never stored)
//   DCN_HL_ONLY_MID         0: the batch-norm apply pass inside a block also writes the fp32 activation when both of its readers
//                           (the next convolution and that convolution's weight gradient) take the hl32 image (default 1: it does not)
//   DCN_BN_REVERSE          bit mask: the batch-norm streaming passes walk their tensors BACK TO FRONT (1: the forward apply pass,
//                           whose input the convolution has just written front to back; 2: the backward apply pass, whose two
//                           inputs the reduce pass has just read front to back) -- what was touched last is then read first
//                           and is still in the 256 MB Infinity Cache when the tensors do not fit it whole.  Same results bit for bit.
//   DCN_BN_REDUCE_WIDE      16 / 32 / 64 / 128: channel quads per workgroup of the batch-norm backward reduction at most (16 = 64
//                           channels: 256-byte pieces of a row per tensor; 128 = whole rows of 512 channels, but 8x fewer workgroups)
//   DCN_BN_NT               bit mask: non-temporal loads of tensors a batch-norm pass reads for the last time (1: the conv output in
//                           the forward apply pass, 2: dy and the conv output in the backward apply pass)
//   DCN_WSPLIT_OVERLAP      0: the per-call weight images are made on the caller's stream in front of the forward / backward pass
//                           (default 1: all of them on the side stream during the stem of the forward pass)
#pragma once

namespace dcn {

struct Tuning {
    int conv_mode = -1;          // -1: unset
    int conv_mode_invalid = 0;   // DCN_CONV_MODE holds something else than fp32 / f16x3
    int backward_overlap = 1;
    int gemm_tile_m = 0;         // 0: unset
    int gemm_sk = -1;            // -1: unset
    double gemm_sk_min_gain = 20.0;
    int gemm_uni = 1;
    int stem8 = 1;
    int gemm_sk_inline = 1;      // stream-K tiles completed inside the GEMM launch (0: separate fix-up kernel)
    int bn_bwd_fused = 0;
    int defer_residual_add = 1;
    int wgrad_splits = 0;        // 0: unset
    int wgrad_tile = 0;          // 0: unset
    int wgrad_deep = 4;
    int gemm_hl = 1;             // wide layers on the pre-split (hl32) LDS-DMA gather-GEMM
    int hl_min_k = 128;          // see DCN_HL_MIN_K
    int gemm_hl_rows = 0;        // hl32 gather-GEMM tile height: 0 = by tile quantisation, 192 / 256 = forced
    int gemm_hlx = 1;            // small-tile variants of the hl32 gather-GEMM where hl_shape's cost model picks them
    int gemm_hlx_kg = 0;         // forced K groups per workgroup (0: as decided)
    int gemm_hlx_splits = 0;     // forced workgroups per tile along K (0: as decided)
    int gemm_hlx_narrow = 1;     // 128-channel destinations on the 160 x 128 tile
    double hlx_cost1 = 1.32, hlx_cost2 = 1.42, hlx_split_cost = 14.0;   // (profiles/r5a_hlx_sweep.txt, r5b_hlx_sweep.txt)
    int hl_setprio = 1;          // see DCN_HL_SETPRIO
    int hlx_stagger = 1;         // see DCN_HLX_STAGGER
    int hlx_counters = 1;        // see DCN_HLX_COUNTERS
    int wgrad_hl_min_m = 0;      // 0: the default of dcn_conv_wgrad_hl_eligible
    int wgrad_hl = 1;            // wide layers' weight gradients on the pre-split (hl32) LDS-DMA kernel
    int wgrad_hlr = 1;           // see DCN_WGRAD_HLR above
    int wgrad_hlr_min_m = 0;     // 0: the default of use_hlr (wgrad_hl_kernels.hip)
    int wgrad_hlr_pairs = 1;     // see DCN_WGRAD_HLR_PAIRS above
    int hl_producers = 1;        // hl32 images written by the producing batch-norm passes (0: stand-alone split passes)
    int hl_only_mid = 1;         // mid-block activations whose two readers (next conv, its wgrad) take the hl32 image: no fp32 copy (0: keep it)
    int stem_pool_fused = 1;     // the stem's batch norm + ReLU applied inside the max-pool pass (0: an apply pass of its own)
    int bn_reduce_wide = 32;     // see DCN_BN_REDUCE_WIDE above (32: +0.2 % on the step, 128: -1 %, profiles/r4c_ab_bn_reduce_wide.txt)
    int wsplit_overlap = 1;      // see DCN_WSPLIT_OVERLAP above
    int bn_nt = 0;               // see DCN_BN_NT above
    int bn_reverse = 0;          // see DCN_BN_REVERSE above
    int wgrad_roles = 1;         // wide tile: wavefronts 0-3 stage the activations, 4-7 the gradient (0: copy spread over all 8)
};

const Tuning& tuning();

}  // namespace dcn
