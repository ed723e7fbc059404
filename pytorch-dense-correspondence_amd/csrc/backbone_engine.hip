// Backbone engine: plans and sequences the gfx950 kernels for the dilated-ResNet FCN
// (`fcn = resnet_dilated.Resnet34_8s(num_classes=D)`, dense_correspondence_network.py:373-375; forward
// called at network.py:255, backward through loss.backward() at training.py:345).
//
// Architecture restated from the published description of the (un-vendored) backbone -- see
// oracle/resnet_dilated_oracle.py for the statement and its provenance:
//   conv7x7/2 -> BN -> ReLU -> maxpool3x3/2 -> layer1..4 (BasicBlock or Bottleneck; output stride 8, so
//   layer3 / layer4 trade their stride for dilation 2 / 4 in every 3x3 conv) -> 1x1 conv (+bias) to D
//   channels -> bilinear upsample (align_corners=True) back to H x W.
//
// Memory plan (sized for 288 GB of HBM3E: nothing is recomputed, nothing is aliased):
//   saved arena     : NHWC4 input, every conv output x (pre-BN), every BN's (scale, shift, mean, invstd),
//                     every post-activation tensor y, max-pool argmax          -- kept forward -> backward
//   workspace arena : 6 activation-gradient buffers, transposed-weight scratch, wgrad split slabs, BN
//                     partial sums, stem padding buffers, low-resolution descriptor map and its gradient
// All launches go to the caller's stream; there is no host synchronisation anywhere.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "dcn_tuning.h"
#include "elementwise_kernels.h"

namespace {

using dcn::ceil_div;

struct ConvL {
    double flops = 0;  // algorithmic 2*MAC of the forward convolution
    dcn_conv_desc d;   // as executed (stem: cin padded 3 -> 4)
    int w = -1, b = -1;  // parameter indices
    int bn = -1;
    size_t x = 0;      // conv output offset (floats) in the saved arena
    int cin_true = 0;
    int mtiles[2] = {0, 0};   // BN partial-sum rows of the forward kernel, per conv mode
    size_t wsplit = 0;        // offset (halves) of this conv's fp16 weight image inside w_wh / w_wl (forward or dgrad image)
    size_t whl = 0;           // offset (floats) of its hl32 weight image inside w_whl (conv_hl_kernels.hip), forward or dgrad image
    bool hl_any = false;      // some launch of this convolution may take the hl32 path (an image slot is reserved)
    size_t hl_x = 0;          // saved-arena offset of the hl32 image of this convolution's INPUT (weight gradient on hl32 operands)
    bool has_hl_x = false;
    int idx = 0;
    int in_act = -1;          // abs-max slot of the activation tensor this convolution reads (split-fp16 operand pre-scale)
    std::string name;
};
struct BnL {
    int C = 0, g = -1, b = -1, idx = 0;
    size_t stats = 0;  // saved arena: scale[C], shift[C], mean[C], invstd[C]
    int64_t rows = 0;
    std::string name;
};
struct BlockL {
    int nconv = 2;
    int conv[3] = {-1, -1, -1};
    int down = -1;
    size_t in = 0, mid[2] = {0, 0}, out = 0;
    int64_t out_rows = 0;
    int out_c = 0, in_c = 0;
    int64_t in_rows = 0;
    int act_in = -1, act_mid[2] = {-1, -1}, act_out = -1;   // abs-max slots of the block's input / mid / output activations
    int act_down = -1;                                      // ... and of the downsample branch's batch-norm output
    // saved hl32 images (wgrad_hl_kernels.hip reads the activations as hl32 tensors): of the mid activations and of the block's
    // INPUT (= the previous block's output), where a convolution of this block takes the hl32 weight-gradient kernel
    size_t hl_mid[2] = {0, 0}, hl_in = 0;
    bool has_hl_mid[2] = {false, false}, has_hl_in = false;
};
struct ParamInfo {
    std::string name;
    int64_t shape[4];
    int ndim;
};

}  // namespace

struct dcn_plan {
    std::string arch, prefix;
    int N = 0, H = 0, W = 0, D = 0, Dp = 0, base = 64;
    int groups = 1;   // independent batches stacked along N (batch-norm statistics per group); N % groups == 0
    bool bottleneck = false;
    std::vector<ConvL> convs;
    std::vector<BnL> bns;
    std::vector<BlockL> blocks;
    std::vector<ParamInfo> params;
    std::vector<std::pair<size_t, size_t>> relu_masks;   // (post-ReLU activation offset, offset of its one-byte-per-float4 mask)
    int stem = -1, fc = -1;
    int hl = 0, wl = 0, feat_c = 0;
    // saved arena offsets (floats)
    size_t s_in4 = 0, s_stem_y = 0, s_pool = 0, s_argmax = 0, s_low = 0, s_actmax = 0, saved_floats = 0;
    // the backward pass's (channel-transposed) weight images, written by the FORWARD call on the side stream (round 4):
    // fp16 hi / lo planes and hl32 images, laid out like w_wh / w_wl / w_whl
    size_t s_wht = 0, s_wlt = 0, s_whlt = 0;
    int n_act = 0;    // activation tensors that feed a convolution: s_actmax[n_act] abs-max scalars + one status word behind them
    // workspace offsets (floats)
    size_t w_buf[6] = {0, 0, 0, 0, 0, 0}, w_wt = 0, w_slab = 0, w_part = 0, w_k123 = 0, w_wstem = 0, w_dwstem = 0,
           w_glow = 0, w_ups = 0, w_sk = 0, w_stem8 = 0, w_gnorm = 0, w_wh = 0, w_wl = 0, w_amax = 0, w_dq = 0, w_dq2 = 0, w_whl = 0,
           w_hl = 0, w_hl2 = 0, ws_floats = 0;
    int conv_mode = DCN_CONV_F16X3;
    size_t max_act = 0, sk_bytes = 0;   // sk_bytes: size of the stream-K scratch at w_sk
    double flops = 0;
    // optional launch-level timing (dcn_plan_profile_begin/end)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;   // start/stop pairs
    std::vector<int> prof_cat;
    std::vector<double> prof_flops;    // algorithmic FLOPs (matrix-core categories) or HBM bytes (streaming categories)
    size_t prof_used = 0;
    // What a training-mode forward call decided, keyed by its saved arena: the backward pass of that arena must agree with it
    // (the tuning table or the plan's conv mode may have changed in between -- dcn_reload_env, dcn_plan_set_conv_mode -- and a
    // backward pass that re-derived the decisions would then read tensors the forward pass never wrote)
    struct FwdRecord {
        const void* saved = nullptr;
        int conv_mode = 0;
        std::vector<unsigned char> mid_hl_only;   // per convolution: its INPUT activation exists as the saved hl32 image only
        std::vector<unsigned char> hl_x_written;  // per convolution: the saved hl32 image of its input was written
        bool wt_saved = false;                    // the backward pass's weight images are in the saved arena (s_wht / s_wlt / s_whlt)
        std::vector<unsigned char> hl_t_written;  // ... per convolution: its transposed hl32 image among them
    };
    // newest last; one per forward call awaiting its backward.  A record is a few hundred bytes; it is replaced by the next
    // forward call that fills the same arena and dropped when the caller says the arena is gone (dcn_plan_forget_saved: the
    // Python binding ties that to the arena tensor's lifetime).  The bound only keeps a C caller that never does either from
    // growing without limit (round 5: was 8 -- five image pairs forwarded separately and differentiated by one backward() exceed that)
    static constexpr size_t kMaxFwdRecords = 4096;
    std::vector<FwdRecord> fwd_records;
    // backward pass, split-fp16 mode: the weight-gradient GEMMs run on a second, low-priority stream next to the
    // dgrad -> BN-backward chain of the following layer (created on first use; DCN_BACKWARD_OVERLAP=0 disables)
    hipStream_t side = nullptr;
    hipEvent_t ev_dq[2] = {nullptr, nullptr}, ev_wg[2] = {nullptr, nullptr}, ev_join = nullptr;
    hipEvent_t ev_ws[2] = {nullptr, nullptr};   // forward: weight images on the side stream (fork after the stem's padding, join before layer 1)
    int side_state = 0;   // 0: not tried, 1: ready, -1: unavailable
    // gradient buckets (data-parallel training): bucket k = parameters [bucket_first[k], bucket_first[k - 1]) in
    // state-dict order (bucket 0 ends at the last parameter); backward finishes them in the order 0, 1, ... and records
    // ev_bucket[k] on the caller's stream as soon as every gradient of bucket k has been enqueued
    std::vector<int> bucket_first, bucket_block;
    std::vector<hipEvent_t> ev_bucket;
    hipEvent_t ev_bucket_side = nullptr;
    int bucket_state = 0;   // 0: events not created yet, 1: ready, -1: unavailable
    int fused_bn_bwd = 0;   // batch norms of the last backward call whose reduction ran in a dgrad epilogue
};

namespace {

size_t align64(size_t x) { return (x + 63) & ~size_t(63); }

struct Builder {
    dcn_plan& p;
    size_t saved = 0;
    explicit Builder(dcn_plan& pl) : p(pl) {}

    size_t alloc_saved(size_t floats) {
        const size_t o = saved;
        saved = align64(saved + floats);
        return o;
    }
    // a post-ReLU activation of `floats` elements plus its ReLU mask (one byte per float4, read by the BN backward passes)
    size_t alloc_relu_out(size_t floats) {
        const size_t o = alloc_saved(floats);
        p.relu_masks.emplace_back(o, alloc_saved((floats / 4 + 3) / 4));
        return o;
    }
    int add_param(const std::string& name, std::initializer_list<int64_t> shape) {
        ParamInfo pi;
        pi.name = name;
        pi.ndim = (int)shape.size();
        int i = 0;
        for (int k = 0; k < 4; ++k) pi.shape[k] = 1;
        for (auto s : shape) pi.shape[i++] = s;
        p.params.push_back(pi);
        return (int)p.params.size() - 1;
    }
    // returns conv index; registers "<name>.weight"
    int add_conv(const std::string& name, int n, int hin, int win, int cin, int cout, int k, int stride, int pad, int dil,
                 bool bias, int ldc = 0) {
        ConvL c;
        c.name = name;
        c.cin_true = cin;
        c.d.n = n; c.d.hin = hin; c.d.win = win; c.d.cin = (cin + 3) / 4 * 4;
        c.d.kh = k; c.d.kw = k; c.d.stride = stride; c.d.pad = pad; c.d.dil = dil;
        c.d.hout = (hin + 2 * pad - dil * (k - 1) - 1) / stride + 1;
        c.d.wout = (win + 2 * pad - dil * (k - 1) - 1) / stride + 1;
        c.d.cout = cout;
        c.d.ldc = ldc ? ldc : cout;
        // M tiles (= rows of the BN partial sums) must not straddle a group boundary
        c.d.group_rows = p.groups > 1 ? n / p.groups * c.d.hout * c.d.wout : 0;
        c.w = add_param(name + ".weight", {cout, cin, k, k});
        if (bias) c.b = add_param(name + ".bias", {cout});
        const int64_t M = (int64_t)n * c.d.hout * c.d.wout;
        c.mtiles[DCN_CONV_FP32] = dcn_conv_num_mtiles(&c.d);  // M tiles of the forward kernel == rows of its BN partial sums
        c.mtiles[DCN_CONV_F16X3] = dcn_conv_num_mtiles_f16(&c.d);
        c.idx = (int)p.convs.size();
        c.flops = 2.0 * (double)M * cout * (double)(k * k * cin);
        p.flops += c.flops;
        p.convs.push_back(c);
        return (int)p.convs.size() - 1;
    }
    int add_bn(const std::string& name, int C, int64_t rows) {
        BnL b;
        b.name = name;
        b.C = C;
        b.rows = rows;
        b.g = add_param(name + ".weight", {C});
        b.b = add_param(name + ".bias", {C});
        b.idx = (int)p.bns.size();
        b.stats = alloc_saved((size_t)4 * C * p.groups);
        p.bns.push_back(b);
        return b.idx;
    }
    int64_t rows_of(const ConvL& c) const { return (int64_t)c.d.n * c.d.hout * c.d.wout; }
    void conv_out(int ci) {
        ConvL& c = p.convs[ci];
        const size_t fl = (size_t)rows_of(c) * c.d.ldc;
        c.x = alloc_saved(fl);
        if (fl > p.max_act) p.max_act = fl;
    }
};

int build_plan(dcn_plan& p) {
    Builder B(p);
    static const struct { const char* name; bool bott; int layers[4]; const char* prefix; } archs[] = {
        {"Resnet18_8s", false, {2, 2, 2, 2}, "resnet18_8s"},
        {"Resnet34_8s", false, {3, 4, 6, 3}, "resnet34_8s"},
        {"Resnet50_8s", true, {3, 4, 6, 3}, "resnet50_8s"},
        {"Resnet101_8s", true, {3, 4, 23, 3}, "resnet101_8s"},
    };
    const int* layers = nullptr;
    for (auto& a : archs)
        if (p.arch == a.name) { layers = a.layers; p.bottleneck = a.bott; p.prefix = a.prefix; }
    if (!layers) return DCN_E_INVALID;
    const int N = p.N, w = p.base, exp = p.bottleneck ? 4 : 1;
    p.Dp = (p.D + 3) / 4 * 4;

    p.s_in4 = B.alloc_saved((size_t)N * p.H * p.W * 4);
    int n_act = 0;
    // stem
    p.stem = B.add_conv("conv1", N, p.H, p.W, 3, w, 7, 2, 3, 1, false);
    p.convs[p.stem].in_act = n_act++;              // slot 0: the input image
    const int act_pool = n_act++;                   // slot 1: max |stem activation| >= max |max-pool output|
    B.conv_out(p.stem);
    {
        ConvL& c = p.convs[p.stem];
        c.bn = B.add_bn("bn1", w, B.rows_of(c));
        p.s_stem_y = B.alloc_relu_out((size_t)B.rows_of(c) * w);
    }
    int h = p.convs[p.stem].d.hout, wd = p.convs[p.stem].d.wout;
    const int hp = (h + 2 - 3) / 2 + 1, wp = (wd + 2 - 3) / 2 + 1;
    p.s_pool = B.alloc_saved((size_t)N * hp * wp * w);
    p.s_argmax = B.alloc_saved(((size_t)N * hp * wp * w + 3) / 4);
    h = hp; wd = wp;
    size_t cur = p.s_pool;
    int cur_act = act_pool;
    int inplanes = w, cur_stride = 4, cur_dil = 1;
    for (int li = 0; li < 4; ++li) {
        const int planes = w << li;
        int stride = li == 0 ? 1 : 2;
        for (int bi = 0; bi < layers[li]; ++bi) {
            const std::string bname = "layer" + std::to_string(li + 1) + "." + std::to_string(bi);
            BlockL blk;
            blk.act_in = cur_act;
            blk.in = cur;
            blk.in_c = inplanes;
            blk.in_rows = (int64_t)N * h * wd;
            int bstride = 1;
            bool need_down = false;
            if (bi == 0 && (stride != 1 || inplanes != planes * exp)) {
                need_down = true;
                if (cur_stride == 8) { cur_dil *= stride; }
                else { cur_stride *= stride; bstride = stride; }
            }
            const int dil = cur_dil;
            // NB: like the reference's _make_layer, the first block of a layer already uses the layer's dilation
            if (!p.bottleneck) {
                blk.nconv = 2;
                blk.conv[0] = B.add_conv(bname + ".conv1", N, h, wd, inplanes, planes, 3, bstride, dil, dil, false);
                B.conv_out(blk.conv[0]);
                const ConvL c0 = p.convs[blk.conv[0]];
                p.convs[blk.conv[0]].bn = B.add_bn(bname + ".bn1", planes, B.rows_of(c0));
                blk.mid[0] = B.alloc_relu_out((size_t)B.rows_of(c0) * planes);
                blk.conv[1] = B.add_conv(bname + ".conv2", N, c0.d.hout, c0.d.wout, planes, planes, 3, 1, dil, dil, false);
                B.conv_out(blk.conv[1]);
                p.convs[blk.conv[1]].bn = B.add_bn(bname + ".bn2", planes, B.rows_of(p.convs[blk.conv[1]]));
            } else {
                blk.nconv = 3;
                blk.conv[0] = B.add_conv(bname + ".conv1", N, h, wd, inplanes, planes, 1, 1, 0, 1, false);
                B.conv_out(blk.conv[0]);
                const ConvL c0 = p.convs[blk.conv[0]];
                p.convs[blk.conv[0]].bn = B.add_bn(bname + ".bn1", planes, B.rows_of(c0));
                blk.mid[0] = B.alloc_relu_out((size_t)B.rows_of(c0) * planes);
                blk.conv[1] = B.add_conv(bname + ".conv2", N, h, wd, planes, planes, 3, bstride, dil, dil, false);
                B.conv_out(blk.conv[1]);
                const ConvL c1 = p.convs[blk.conv[1]];
                p.convs[blk.conv[1]].bn = B.add_bn(bname + ".bn2", planes, B.rows_of(c1));
                blk.mid[1] = B.alloc_relu_out((size_t)B.rows_of(c1) * planes);
                blk.conv[2] = B.add_conv(bname + ".conv3", N, c1.d.hout, c1.d.wout, planes, planes * 4, 1, 1, 0, 1, false);
                B.conv_out(blk.conv[2]);
                p.convs[blk.conv[2]].bn = B.add_bn(bname + ".bn3", planes * 4, B.rows_of(p.convs[blk.conv[2]]));
            }
            for (int i = 0; i < blk.nconv; ++i) {
                if (i + 1 < blk.nconv) blk.act_mid[i] = n_act++;
                p.convs[blk.conv[i]].in_act = i == 0 ? blk.act_in : blk.act_mid[i - 1];
            }
            blk.act_out = n_act++;
            const ConvL last = p.convs[blk.conv[blk.nconv - 1]];
            if (need_down) {
                blk.down = B.add_conv(bname + ".downsample.0", N, h, wd, inplanes, planes * exp, 1, bstride, 0, 1, false);
                B.conv_out(blk.down);
                p.convs[blk.down].bn = B.add_bn(bname + ".downsample.1", planes * exp, B.rows_of(p.convs[blk.down]));
                p.convs[blk.down].in_act = blk.act_in;
                blk.act_down = n_act++;
            }
            blk.out_rows = B.rows_of(last);
            blk.out_c = planes * exp;
            blk.out = B.alloc_relu_out((size_t)blk.out_rows * blk.out_c);
            if ((size_t)blk.out_rows * blk.out_c > p.max_act) p.max_act = (size_t)blk.out_rows * blk.out_c;
            p.blocks.push_back(blk);
            cur = blk.out;
            cur_act = blk.act_out;
            h = last.d.hout; wd = last.d.wout;
            inplanes = planes * exp;
        }
    }
    p.hl = h; p.wl = wd; p.feat_c = inplanes;
    p.fc = B.add_conv("fc", N, h, wd, inplanes, p.D, 1, 1, 0, 1, true, p.Dp);
    p.convs[p.fc].in_act = cur_act;
    p.s_low = B.alloc_saved((size_t)N * h * wd * p.Dp);  // low-resolution descriptor map (needed by the normalise backward)
    for (BlockL& blk : p.blocks) {   // hl32 copies of the activations whose consumer's weight gradient takes the hl32 kernel
        for (int i = 0; i < blk.nconv; ++i) {
            const ConvL& c = p.convs[blk.conv[i]];
            if (!dcn_conv_wgrad_hl_eligible(&c.d)) continue;
            const size_t fl = (size_t)c.d.n * c.d.hin * c.d.win * c.d.cin;
            if (i == 0) { if (!blk.has_hl_in) { blk.hl_in = B.alloc_saved(fl); blk.has_hl_in = true; } }
            else if (!blk.has_hl_mid[i - 1]) { blk.hl_mid[i - 1] = B.alloc_saved(fl); blk.has_hl_mid[i - 1] = true; }
            p.convs[blk.conv[i]].hl_x = i == 0 ? blk.hl_in : blk.hl_mid[i - 1];
            p.convs[blk.conv[i]].has_hl_x = true;
        }
        if (blk.down >= 0 && dcn_conv_wgrad_hl_eligible(&p.convs[blk.down].d)) {
            ConvL& c = p.convs[blk.down];
            if (!blk.has_hl_in) {
                blk.hl_in = B.alloc_saved((size_t)c.d.n * c.d.hin * c.d.win * c.d.cin);
                blk.has_hl_in = true;
            }
            c.hl_x = blk.hl_in;
            c.has_hl_x = true;
        }
    }
    p.n_act = n_act;
    p.s_actmax = B.alloc_saved((size_t)n_act + 1);        // abs-max of every convolution input (kept for wgrad) + status word
    p.saved_floats = B.saved;
    {   // gradient buckets, in the order backward completes them: fc + layer4 | layer3 | layer2 + layer1 + stem
        int first_block[4], b0 = 0;
        for (int li = 0; li < 4; ++li) { first_block[li] = b0; b0 += layers[li]; }
        p.bucket_block = {first_block[3], first_block[2], 0};
        p.bucket_first = {p.convs[p.blocks[first_block[3]].conv[0]].w, p.convs[p.blocks[first_block[2]].conv[0]].w, 0};
    }
    {
        const size_t in4 = (size_t)N * p.H * p.W * 4;
        if (in4 > p.max_act) p.max_act = in4;
    }

    if (p.groups > 1)
        for (const ConvL& c : p.convs)   // every tile size the kernels may pick must divide the rows of a group
            if (c.bn >= 0 && (c.d.group_rows % 64) != 0) return DCN_E_UNSUPPORTED;

    // ---- workspace
    size_t ws = 0;
    auto alloc = [&](size_t fl) { const size_t o = ws; ws = align64(ws + fl); return o; };
    for (int i = 0; i < 6; ++i) p.w_buf[i] = alloc(p.max_act);
    size_t max_w = 0, max_slab = 0, max_part = 0, max_sk = 0;   // sized for either conv mode
    int max_c = 4;
    for (const ConvL& c : p.convs) {
        const size_t welems = (size_t)c.d.ldc * c.d.kh * c.d.kw * c.d.cin;
        if (welems > max_w) max_w = welems;
        const size_t sl = std::max(std::max(dcn_conv_wgrad_workspace(&c.d), dcn_conv_wgrad_workspace_f16(&c.d)),
                                   dcn_conv_wgrad_workspace_hl(&c.d)) / sizeof(float);
        if (sl > max_slab) max_slab = sl;
        for (int dg = 0; dg < 2; ++dg) {
            const size_t sk = std::max(std::max(dcn_conv_gemm_workspace(&c.d, dg), dcn_conv_gemm_workspace_f16(&c.d, dg)),
                                       dcn_conv_gemm_workspace_hl(&c.d, dg)) / sizeof(float);
            if (sk > max_sk) max_sk = sk;
        }
        // (sized for the smallest M tile any tuning override can select -- 32 rows --, not for the tile chosen today)
        const size_t pf = ((size_t)c.d.n * c.d.hout * c.d.wout / 32 + 2) * 3 * c.d.cout;
        if (pf > max_part) max_part = pf;
        const size_t pb = ((size_t)c.d.n * c.d.hin * c.d.win / 32 + 2) * 4 * c.d.cin;   // fused BN-backward sums (dgrad M tiles)
        if (pb > max_part) max_part = pb;
        if (c.d.cout > max_c) max_c = c.d.cout;
    }
    for (const BnL& b : p.bns) {
        const size_t pb = (size_t)p.groups * dcn::bn_bwd_chunks(b.rows / p.groups) * 4 * b.C;
        if (pb > max_part) max_part = pb;
    }
    p.w_wt = alloc(max_w);
    p.w_slab = alloc(max_slab);
    p.w_sk = alloc(max_sk);
    p.sk_bytes = max_sk * sizeof(float);
    p.w_part = alloc(max_part);
    p.w_k123 = alloc((size_t)3 * max_c * p.groups);
    p.w_wstem = alloc((size_t)p.base * 49 * 4);
    p.w_dwstem = alloc((size_t)p.base * 49 * 4);
    p.w_stem8 = alloc((size_t)p.base * 224);   // stem weights as [cout][7][8][4] fp16 hi | lo (uniform-tap stem path): 2 planes x 224 halves per channel
    p.w_glow = alloc((size_t)N * p.hl * p.wl * p.Dp);
    p.w_gnorm = alloc((size_t)N * p.H * p.W * p.D);
    p.w_ups = alloc(dcn::upsample_bwd_tmp_bytes(N, p.hl, p.W, p.D) / sizeof(float));
    {   // fp16 hi / lo images of ALL weight tensors (one batched split per forward / backward call); a conv's forward image
        // [cout][kpad(taps*cin)] and its dgrad image [cin][kpad(taps*ldc)] share the slot (max of the two, 16-byte aligned)
        size_t halves = 0;
        for (ConvL& c : p.convs) {
            const int taps = c.d.kh * c.d.kw;
            const size_t a = (size_t)c.d.cout * dcn_f16_kpad(taps * c.d.cin), b = (size_t)c.d.cin * dcn_f16_kpad(taps * c.d.ldc);
            c.wsplit = halves;
            halves += (std::max(a, b) + 7) / 8 * 8;
        }
        p.w_wh = alloc((halves + 1) / 2);
        p.w_wl = alloc((halves + 1) / 2);
        p.s_wht = B.alloc_saved((halves + 1) / 2);
        p.s_wlt = B.alloc_saved((halves + 1) / 2);
    }
    {   // hl32 weight images (conv_hl_kernels.hip) of the convolutions whose forward or dgrad may take that path (the image of
        // the forward pass and the one of the backward pass share the slot), and ONE transient hl32 activation / gradient image
        size_t fl = 0, max_img = 0;
        for (ConvL& c : p.convs) {
            if (c.idx == p.stem || c.idx == p.fc || c.d.stride != 1 || ((c.d.cin % 32) != 0 && (c.d.ldc % 32) != 0)) continue;
            c.hl_any = true;
            c.whl = fl;
            fl += align64((size_t)c.d.kh * c.d.kw * std::max((size_t)c.d.cout * c.d.cin, (size_t)c.d.cin * c.d.ldc));
            // the transient images are those of such a convolution's input (forward) or output gradient (backward): sized by
            // them, not by the network's largest tensor (the stem's, which never takes this path)
            max_img = std::max(max_img, std::max((size_t)c.d.n * c.d.hin * c.d.win * c.d.cin, (size_t)c.d.n * c.d.hout * c.d.wout * c.d.ldc));
        }
        p.w_whl = alloc(fl);
        p.s_whlt = B.alloc_saved(fl);
        p.w_hl = alloc(max_img);    // image of a block's input / output (forward), of a batch-norm backward's dx (backward)
        p.w_hl2 = alloc(max_img);   // image of a block's mid activation
    }
    p.w_amax = alloc(p.convs.size());   // abs-max of the gradient w.r.t. each convolution's output
    {   // pixel-blocked split (fp16 hi | lo) copy of one gradient tensor: wgrad's dy operand, written by the BN backward pass
        size_t max_dq = 0;
        for (const ConvL& c : p.convs)
            max_dq = std::max(max_dq, dcn_grad_blocked_bytes(c.d.n * c.d.hout * c.d.wout, c.d.ldc) / sizeof(float));
        p.w_dq = alloc(max_dq);
        p.w_dq2 = alloc(max_dq);   // second image: wgrad of layer k reads one while BN backward of layer k - 1 writes the other
    }
    p.ws_floats = ws;
    p.saved_floats = B.saved;   // (the saved copies of the backward weight images were added behind the activations)
    return DCN_OK;
}

// status word behind the activation abs-max slots: bit 0 = some convolution input is not finite (inf / NaN: the abs-max
// producers report a NaN as an infinite bound); bit 1 (set by the weight split, conv_f16_kernels.hip) = a weight outside the
// range of its fp16 image.  The word is cleared at the start of the forward call.
__global__ void __launch_bounds__(64)
act_status_kernel(const float* __restrict__ actmax, int n, int* __restrict__ status) {
    int bad = 0;
    for (int i = threadIdx.x; i < n; i += 64) {
        const float v = actmax[i];
        bad |= (v != v || v > 3.0e38f) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o);
    if (threadIdx.x == 0 && bad) atomicOr(status, 1);
}

// column sums of a [rows][ld] matrix (fc bias gradient): one workgroup per column, fixed-order reduction.  1024 work-items
// with four independent partial sums each: the D (= 3) workgroups of this launch are latency-bound -- 150 dependent
// iterations per work-item measured 45 us at 38 400 rows (profiles/r3f_kernel_stats.txt), 10 iterations of 4 loads do not
__global__ void __launch_bounds__(1024)
colsum_kernel(const float* __restrict__ m, int64_t rows, int ld, float* __restrict__ out) {
    __shared__ double s[16];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t r = threadIdx.x;
    for (; r + 3 * 1024 < rows; r += 4 * 1024) {
        a0 += (double)m[r * ld + blockIdx.x];
        a1 += (double)m[(r + 1024) * ld + blockIdx.x];
        a2 += (double)m[(r + 2048) * ld + blockIdx.x];
        a3 += (double)m[(r + 3072) * ld + blockIdx.x];
    }
    for (; r < rows; r += 1024) a0 += (double)m[r * ld + blockIdx.x];
    const double a = dcn::block_sum<1024>((a0 + a1) + (a2 + a3), s);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)a;
}

// ONE low-priority side stream per device for the whole process, shared by every plan (a plan is only ever in one backward
// call at a time, and the call joins the side stream before it returns, so the stream's FIFO order is all the ordering two
// plans need).  A stream per plan -- rounds 1-3 -- maps onto the runtime's few hardware queues round-robin: the side stream of
// the fourth plan of a process landed on the hardware queue of the caller's stream, whose wgrad kernels and cross-stream waits
// then serialised with the main chain (seen as 2.4x slower steps, host-bound, for whichever workload came fourth:
// profiles/r4c_debug_c1sep.txt; the 52 ... 63 images/s "run to run" spread of config 5 as an in-process bench variant was this).
hipStream_t shared_side_stream() {
    static std::mutex mu;
    static std::vector<std::pair<int, hipStream_t>> streams;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto& e : streams)
        if (e.first == dev) return e.second;
    int lo_prio = 0, hi_prio = 0;
    if (hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio) != hipSuccess) lo_prio = 0;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo_prio) != hipSuccess) return nullptr;
    streams.emplace_back(dev, st);
    return st;
}

// the plan's handle on the side stream + the events that order it against the caller's stream (created on first use)
bool ensure_side(dcn_plan& p) {
    if (p.side_state == 0) {
        bool ok = dcn::tuning().backward_overlap != 0;
        p.side = ok ? shared_side_stream() : nullptr;
        ok = ok && p.side != nullptr;
        for (hipEvent_t* ev : {&p.ev_dq[0], &p.ev_dq[1], &p.ev_wg[0], &p.ev_wg[1], &p.ev_join, &p.ev_ws[0], &p.ev_ws[1]})
            ok = ok && hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess;
        p.side_state = ok ? 1 : -1;
    }
    return p.side_state == 1;
}

#define DCN_TRY(expr)                    \
    do {                                 \
        const int rc__ = (expr);         \
        if (rc__ != DCN_OK) return rc__; \
    } while (0)

constexpr float kWeightScale = 64.f;   // power-of-two pre-scale of the fp16 weight images

struct Run {
    dcn_plan& p;
    const float* const* params;
    float* saved;
    float* ws;
    hipStream_t st;
    bool stem8 = false;   // this call runs the stem through dcn_conv_stem_forward_f16
    const float* hl_src[2] = {nullptr, nullptr};   // the fp32 tensors whose hl32 images currently sit in w_hl / w_hl2
    std::vector<std::pair<const float*, const float*>> hl_saved;   // (fp32 tensor, its hl32 image in the SAVED arena) of this call
    const float* saved_hl_of(const float* t) const {
        for (const auto& e : hl_saved)
            if (e.first == t) return e.second;
        return nullptr;
    }

    // bracket one launch (or one launcher call) with events when profiling is on
    static bool prof_open(dcn_plan& p, int cat, double work, hipStream_t s) {
        if (p.prof_used + 2 > p.prof_ev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
            p.prof_ev.push_back(a);
            p.prof_ev.push_back(b);
        }
        p.prof_used += 2;
        p.prof_cat.push_back(cat);
        p.prof_flops.push_back(work);
        return hipEventRecord(p.prof_ev[p.prof_used - 2], s) == hipSuccess;
    }
    static bool prof_close(dcn_plan& p, hipStream_t s) { return hipEventRecord(p.prof_ev[p.prof_used - 1], s) == hipSuccess; }
    template <class F> int timed(int cat, double flops, F&& launch) {
        if (!p.prof_on) return launch();
        const dcn::LaunchObserver* outer = dcn::launch_observer;
        dcn::launch_observer = nullptr;            // (launchers called inside this bracket are part of it)
        if (!prof_open(p, cat, flops, st)) { dcn::launch_observer = outer; return DCN_E_LAUNCH; }
        const int rc = launch();
        const bool ok = prof_close(p, st);
        dcn::launch_observer = outer;
        return ok ? rc : DCN_E_LAUNCH;
    }
    // While a plan is being profiled, the launchers of elementwise_kernels.hip report every kernel they launch to this
    // observer (elementwise_kernels.h): category + algorithmic bytes, bracketed by events like the matrix-core launches
    struct ObserverGuard {
        dcn::LaunchObserver obs;
        const dcn::LaunchObserver* outer;
        bool on;
        explicit ObserverGuard(dcn_plan& pl) : outer(dcn::launch_observer), on(pl.prof_on) {
            obs.ctx = &pl;
            obs.begin = [](void* c, int cat, double bytes, hipStream_t s) { prof_open(*(dcn_plan*)c, cat, bytes, s); };
            obs.end = [](void* c, hipStream_t s) { prof_close(*(dcn_plan*)c, s); };
            if (on) dcn::launch_observer = &obs;
        }
        ~ObserverGuard() { if (on) dcn::launch_observer = outer; }
    };
    // engine-level launches outside the launchers (fills, weight / operand splits, status and bias-gradient kernels)
    template <class F> int other(F&& launch) { return timed(DCN_PROF_OTHER, 0.0, launch); }

    // stream-K scratch for one gather-GEMM launch, or null (no stream-K) when the tile shape selected by the tuning table
    // of the moment needs more than the plan reserved (the table may have changed since the plan was made)
    void* SK(const ConvL& c, int dgrad) const {
        const size_t need = p.conv_mode == DCN_CONV_FP32 ? dcn_conv_gemm_workspace(&c.d, dgrad) : dcn_conv_gemm_workspace_f16(&c.d, dgrad);
        return need <= p.sk_bytes ? (void*)(ws + p.w_sk) : nullptr;
    }
    float* S(size_t off) const { return saved + off; }
    // abs-max scalar of activation slot `a` (split-fp16 mode only: null otherwise, i.e. "no pre-scale")
    float* A(int a) const { return (p.conv_mode == DCN_CONV_F16X3 && a >= 0) ? saved + p.s_actmax + a : nullptr; }
    // ReLU mask bytes of the post-ReLU activation at saved-arena offset `off`
    unsigned char* M(size_t off) const {
        for (const auto& m : p.relu_masks)
            if (m.first == off) return (unsigned char*)(saved + m.second);
        return nullptr;
    }
    float* Wk(size_t off) const { return ws + off; }
    const float* P(int i) const { return params[i]; }

    // forward convolution in the plan's conv mode (w: [cout][taps][d.cin] fp32)
    int conv_fwd(const ConvL& c, const float* in, const float* w, const float* bias, float* out, float* part) {
        if (p.conv_mode == DCN_CONV_FP32)
            return timed(0, c.flops, [&] { return dcn_conv_forward(&c.d, in, w, bias, out, part, SK(c, 0), st); });
        (void)w;   // split-fp16 mode: the image was produced by split_all_weights at the start of the call
        if (c.idx == p.stem && stem8) {   // the 7x7 stem through the uniform-tap path (image prepared by dcn_backbone_forward)
            _Float16* hi = (_Float16*)Wk(p.w_stem8);
            return timed(0, c.flops, [&] {
                return dcn_conv_stem_forward_f16(&c.d, in, A(c.in_act), hi, hi + (size_t)p.base * 224, kWeightScale, out, part, st);
            });
        }
        if (use_hl(c, 0)) {
            // (the operand image was written by the batch-norm apply pass that produced `in` -- hl_image_for --, or is made
            // here by a stand-alone pass)
            return timed(2, c.flops, [&] {
                const float* img = saved_hl_of(in);
                if (!img) {
                    int k = hl_src[0] == in ? 0 : (hl_src[1] == in ? 1 : -1);
                    if (k < 0) {
                        k = 0;
                        DCN_TRY(dcn_split_act_hl32(in, A(c.in_act), hlbuf(0), (int64_t)c.d.n * c.d.hin * c.d.win, c.d.cin, st));
                        hl_src[0] = in;
                    }
                    img = hlbuf(k);
                }
                return dcn_conv_forward_hl(&c.d, img, A(c.in_act), whl(c), kWeightScale, bias, out, part, SKhl(c, 0), st);
            });
        }
        return timed(0, c.flops, [&] {
            return dcn_conv_forward_f16(&c.d, in, A(c.in_act), wimg(p.w_wh, c), wimg(p.w_wl, c), kWeightScale, bias, out, part,
                                        SK(c, 0), st);
        });
    }

    // weight images: in the workspace, or (backward pass, images saved by the forward call) wherever the overrides point
    float* wh_over = nullptr;
    float* wl_over = nullptr;
    float* whl_over = nullptr;
    void* wimg(size_t plane, const ConvL& c) const {
        float* base = plane == p.w_wh ? (wh_over ? wh_over : Wk(p.w_wh)) : (wl_over ? wl_over : Wk(p.w_wl));
        return (void*)((_Float16*)base + c.wsplit);
    }

    // the saved hl32 copy of activation y (channels C, absmax slot act) by a stand-alone pass, when the apply pass that produced
    // y did not write it (DCN_HL_PRODUCERS=0)
    int ensure_saved_hl(const float* y, float* slot, int64_t rows, int C, int act) {
        if (!slot || dcn::tuning().hl_producers != 0) return DCN_OK;
        return other([&] { return dcn_split_act_hl32(y, A(act), slot, rows, C, st); });
    }
    // ---- pre-split (hl32) path of the wide layers (conv_hl_kernels.hip)
    bool use_hl(const ConvL& c, int dgrad) const {
        return p.conv_mode == DCN_CONV_F16X3 && c.hl_any && dcn_conv_hl_eligible(&c.d, dgrad) != 0;
    }
    void* whl(const ConvL& c) const { return (void*)((whl_over ? whl_over : Wk(p.w_whl)) + c.whl); }
    float* hlbuf(int k) const { return Wk(k == 0 ? p.w_hl : p.w_hl2); }
    // buffer k will hold the hl32 image of the activation `y` (written by the bn_apply pass that is about to produce y), if a
    // convolution that reads y takes the hl32 path; returns the buffer or null
    // `slot` (optional): the tensor has an hl32 copy in the saved arena (its consumer's weight gradient reads it in the
    // backward pass): the image goes there, whoever reads it in this call
    void* hl_image_for(const float* y, int k, const ConvL* consumer_a, const ConvL* consumer_b = nullptr, float* slot = nullptr) {
        if (slot) {
            hl_saved.emplace_back(y, slot);
            return dcn::tuning().hl_producers != 0 ? (void*)slot : nullptr;   // (producers off: ensure_saved_hl makes it)
        }
        if (dcn::tuning().hl_producers == 0) return nullptr;
        const bool want = (consumer_a && use_hl(*consumer_a, 0)) || (consumer_b && use_hl(*consumer_b, 0));
        if (!want) return nullptr;
        hl_src[k] = y;
        return hlbuf(k);
    }
    void* SKhl(const ConvL& c, int dgrad) const {
        return dcn_conv_gemm_workspace_hl(&c.d, dgrad) <= p.sk_bytes ? (void*)(ws + p.w_sk) : nullptr;
    }
    // rows of the batch-norm partial sums the forward kernel of the moment writes
    int fwd_mtiles(const ConvL& c) const {
        if (p.conv_mode == DCN_CONV_FP32) return dcn_conv_num_mtiles(&c.d);
        return use_hl(c, 0) ? dcn_conv_num_mtiles_hl(&c.d) : dcn_conv_num_mtiles_f16(&c.d);
    }
    // weight gradient on hl32 operands (wgrad_hl_kernels.hip): the plan reserved a saved hl32 image of the convolution's input
    // (written by every training-mode forward call in the split-fp16 arithmetic)
    bool use_wgrad_hl(const ConvL& c) const {
        return p.conv_mode == DCN_CONV_F16X3 && c.has_hl_x && dcn_conv_wgrad_hl_eligible(&c.d) != 0;
    }
    int split_hl_weights(bool transposed) {
        std::vector<const float*> w;
        std::vector<void*> out;
        std::vector<int> cout, taps, cin, ldn;
        for (const ConvL& c : p.convs) {
            if (!use_hl(c, transposed ? 1 : 0)) continue;
            w.push_back(P(c.w));
            out.push_back(whl(c));
            cout.push_back(c.d.cout); taps.push_back(c.d.kh * c.d.kw); cin.push_back(c.d.cin); ldn.push_back(c.d.ldc);
        }
        if (w.empty()) return DCN_OK;
        return other([&] {
            return dcn_split_weights_hl32((int)w.size(), w.data(), out.data(), cout.data(), taps.data(), cin.data(), ldn.data(),
                                          transposed ? 1 : 0, kWeightScale, st);
        });
    }

    // fp16 hi / lo images of every convolution's weights in one launch: forward images, or the channel-transposed dgrad
    // images (the stem has no dgrad).  `stem_w`: the stem's weights padded to 4 input channels.
    int split_all_weights(bool transposed, const float* stem_w, bool fold_bn = false) {
        const int n = (int)p.convs.size();
        std::vector<const float*> w, rs;
        std::vector<void*> hi, lo;
        std::vector<int> cout, taps, cin, ldn;
        for (int i = 0; i < n; ++i) {
            const ConvL& c = p.convs[i];
            if (transposed && i == p.stem) continue;
            w.push_back(i == p.stem ? stem_w : P(c.w));
            rs.push_back(fold_bn && c.bn >= 0 ? S(p.bns[c.bn].stats) : nullptr);   // eval-mode BN scale gamma / sqrt(var + eps)
            hi.push_back(wimg(p.w_wh, c));
            lo.push_back(wimg(p.w_wl, c));
            cout.push_back(c.d.cout); taps.push_back(c.d.kh * c.d.kw); cin.push_back(c.d.cin); ldn.push_back(c.d.ldc);
        }
        if (!transposed)   // forward images: with the weight-range check (status bit 1, see act_status_kernel)
            return other([&] {
                return dcn_split_weights_checked_f16((int)w.size(), w.data(), fold_bn ? rs.data() : nullptr, hi.data(), lo.data(),
                                                     cout.data(), taps.data(), cin.data(), kWeightScale,
                                                     (int*)(S(p.s_actmax) + p.n_act), st);
            });
        return other([&] {
            return dcn_split_weights_scaled_f16((int)w.size(), w.data(), nullptr, hi.data(), lo.data(), cout.data(), taps.data(),
                                                cin.data(), ldn.data(), 1, kWeightScale, st);
        });
    }

    // inference: conv + folded batch norm (+ residual) (+ ReLU) in one pass; bias = the BN shift beta - mean * scale
    // out_act: abs-max slot of the tensor being produced (it is the next convolution's operand), or -1
    int conv_fused(const ConvL& c, const float* in, const float* add, int relu, float* out, int out_act) {
        const float* shift = S(p.bns[c.bn].stats) + p.bns[c.bn].C;
        return timed(0, c.flops, [&] {
            return dcn_conv_forward_fused_f16(&c.d, in, A(c.in_act), wimg(p.w_wh, c), wimg(p.w_wl, c), kWeightScale, shift, add,
                                              relu, out, A(out_act), SK(c, 0), st);
        });
    }

    // conv + BN statistics -> scale/shift in the saved arena.  out_act: abs-max slot of the tensor the apply pass will
    // produce from this batch norm (its bound is computed here, from the statistics); res_act: slot of the residual added there
    int conv_bn(const ConvL& c, const float* in, const float* w, float* const* bn_running, float momentum, float eps,
                int training, int out_act, int res_act = -1) {
        float* part = training ? Wk(p.w_part) : nullptr;
        DCN_TRY(conv_fwd(c, in, w, nullptr, S(c.x), part));
        const BnL& b = p.bns[c.bn];
        float* stats = S(b.stats);
        float* rm = bn_running ? bn_running[2 * b.idx] : nullptr;
        float* rv = bn_running ? bn_running[2 * b.idx + 1] : nullptr;
        if (!training && (!rm || !rv)) return DCN_E_INVALID;
        // (M tiles of the launch just made: asked again, the tile shape follows the tuning table of the moment)
        const int mtiles = fwd_mtiles(c);
        dcn::launch_bn_finalize(part, mtiles / p.groups, p.groups, b.C, (double)(b.rows / p.groups), P(b.g),
                                P(b.b), rm, rv, momentum, eps, training, stats, training ? A(out_act) : nullptr,
                                training ? A(res_act) : nullptr, st);
        return DCN_OK;
    }
};

}  // namespace

extern "C" int dcn_plan_create(const char* arch, int base_width, int n, int h, int w, int d, dcn_plan** out) {
    return dcn_plan_create_grouped(arch, base_width, n, 1, h, w, d, out);
}
extern "C" int dcn_plan_create_grouped(const char* arch, int base_width, int n, int groups, int h, int w, int d,
                                       dcn_plan** out) {
    if (!arch || !out || n < 1 || h < 8 || w < 8 || d < 1 || base_width < 4 || (base_width % 4) || groups < 1 ||
        groups > 2 || (n % groups))
        return DCN_E_INVALID;
    dcn_plan* p = new dcn_plan();
    p->arch = arch; p->N = n; p->H = h; p->W = w; p->D = d; p->base = base_width; p->groups = groups;
    const dcn::Tuning& tune = dcn::tuning();   // (environment read once: dcn_tuning.h)
    if (tune.conv_mode_invalid) { delete p; return DCN_E_INVALID; }
    if (tune.conv_mode >= 0) p->conv_mode = tune.conv_mode;
    const int rc = build_plan(*p);
    if (rc != DCN_OK) { delete p; return rc; }
    *out = p;
    return DCN_OK;
}
// The caller has released the saved arena at this address (or will overwrite it with something else): the plan drops what it
// remembered about the forward call that filled it.  Returns the number of records dropped (0: none was held).
extern "C" int dcn_plan_forget_saved(dcn_plan* plan, const void* saved) {
    if (!plan) return DCN_E_INVALID;
    int n = 0;
    for (size_t i = 0; i < plan->fwd_records.size();)
        if (plan->fwd_records[i].saved == saved) { plan->fwd_records.erase(plan->fwd_records.begin() + i); ++n; }
        else ++i;
    return n;
}

extern "C" int dcn_plan_num_forward_records(const dcn_plan* plan) {
    return plan ? (int)plan->fwd_records.size() : DCN_E_INVALID;
}

extern "C" void dcn_plan_destroy(dcn_plan* plan) {
    if (!plan) return;
    for (hipEvent_t e : plan->prof_ev) hipEventDestroy(e);
    for (hipEvent_t e : {plan->ev_dq[0], plan->ev_dq[1], plan->ev_wg[0], plan->ev_wg[1], plan->ev_join, plan->ev_ws[0], plan->ev_ws[1]})
        if (e) hipEventDestroy(e);
    for (hipEvent_t e : plan->ev_bucket)
        if (e) hipEventDestroy(e);
    if (plan->ev_bucket_side) hipEventDestroy(plan->ev_bucket_side);
    // (plan->side is the process-wide side stream of its device -- shared_side_stream -- and is not the plan's to destroy)
    delete plan;
}
extern "C" int dcn_plan_set_conv_mode(dcn_plan* plan, int mode) {
    if (!plan || (mode != DCN_CONV_FP32 && mode != DCN_CONV_F16X3)) return DCN_E_INVALID;
    plan->conv_mode = mode;   // arenas are sized for either mode
    return DCN_OK;
}
extern "C" int dcn_plan_conv_mode(const dcn_plan* plan) { return plan ? plan->conv_mode : DCN_E_INVALID; }
extern "C" int dcn_plan_profile_begin(dcn_plan* plan) {
    if (!plan) return DCN_E_INVALID;
    plan->prof_on = true;
    plan->prof_used = 0;
    plan->prof_cat.clear();
    plan->prof_flops.clear();
    return DCN_OK;
}
extern "C" int dcn_plan_profile_end_all(dcn_plan* plan, double ms[DCN_PROF_NCAT], int64_t launches[DCN_PROF_NCAT],
                                        double work[DCN_PROF_NCAT]) {
    if (!plan || !ms || !launches || !work) return DCN_E_INVALID;
    plan->prof_on = false;
    for (int c = 0; c < DCN_PROF_NCAT; ++c) { ms[c] = 0; launches[c] = 0; work[c] = 0; }
    int rc = DCN_OK;
    for (size_t i = 0; i < plan->prof_cat.size(); ++i) {
        hipEvent_t e0 = plan->prof_ev[2 * i], e1 = plan->prof_ev[2 * i + 1];
        float t = 0.f;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) { rc = DCN_E_LAUNCH; break; }
        const int c = plan->prof_cat[i];
        if (c < 0 || c >= DCN_PROF_NCAT) continue;
        // category 2 (hl32 launches) also counts as a gather-GEMM (0)
        for (int k : {c == DCN_PROF_GEMM_HL ? (int)DCN_PROF_GEMM : c, c == DCN_PROF_GEMM_HL ? (int)DCN_PROF_GEMM_HL : -1}) {
            if (k < 0) continue;
            ms[k] += (double)t;
            launches[k] += 1;
            work[k] += plan->prof_flops[i];
        }
    }
    plan->prof_used = 0;
    plan->prof_cat.clear();
    plan->prof_flops.clear();
    return rc;
}
extern "C" int dcn_plan_profile_end3(dcn_plan* plan, double ms[3], int64_t launches[3], double flops[3]) {
    if (!ms || !launches || !flops) return DCN_E_INVALID;
    double m[DCN_PROF_NCAT], f[DCN_PROF_NCAT];
    int64_t n[DCN_PROF_NCAT];
    const int rc = dcn_plan_profile_end_all(plan, m, n, f);
    for (int c = 0; c < 3; ++c) { ms[c] = m[c]; launches[c] = n[c]; flops[c] = f[c]; }
    return rc;
}
extern "C" int dcn_plan_profile_end(dcn_plan* plan, double ms[2], int64_t launches[2], double flops[2]) {
    if (!ms || !launches || !flops) return DCN_E_INVALID;
    double m3[3], f3[3];
    int64_t n3[3];
    const int rc = dcn_plan_profile_end3(plan, m3, n3, f3);
    for (int c = 0; c < 2; ++c) { ms[c] = m3[c]; launches[c] = n3[c]; flops[c] = f3[c]; }
    return rc;
}
extern "C" int dcn_plan_num_params(const dcn_plan* plan) { return plan ? (int)plan->params.size() : DCN_E_INVALID; }
extern "C" int dcn_plan_num_bn(const dcn_plan* plan) { return plan ? (int)plan->bns.size() : DCN_E_INVALID; }
extern "C" int dcn_plan_param_info(const dcn_plan* plan, int i, char* name, int name_cap, int64_t shape[4], int* ndim) {
    if (!plan || i < 0 || i >= (int)plan->params.size() || !name || name_cap < 2 || !shape || !ndim) return DCN_E_INVALID;
    const ParamInfo& pi = plan->params[i];
    snprintf(name, (size_t)name_cap, "%s", pi.name.c_str());
    for (int k = 0; k < 4; ++k) shape[k] = pi.shape[k];
    *ndim = pi.ndim;
    return DCN_OK;
}
extern "C" int dcn_plan_bn_info(const dcn_plan* plan, int j, char* name, int name_cap, int64_t* channels) {
    if (!plan || j < 0 || j >= (int)plan->bns.size() || !name || name_cap < 2 || !channels) return DCN_E_INVALID;
    snprintf(name, (size_t)name_cap, "%s", plan->bns[j].name.c_str());
    *channels = plan->bns[j].C;
    return DCN_OK;
}
extern "C" int dcn_plan_num_grad_buckets(const dcn_plan* plan) { return plan ? (int)plan->bucket_first.size() : DCN_E_INVALID; }
extern "C" int dcn_plan_grad_bucket_first_param(const dcn_plan* plan, int k) {
    if (!plan || k < 0 || k >= (int)plan->bucket_first.size()) return DCN_E_INVALID;
    return plan->bucket_first[k];
}
extern "C" int dcn_plan_stream_wait_grad_bucket(dcn_plan* plan, int k, void* stream) {
    if (!plan || k < 0 || k >= (int)plan->bucket_first.size()) return DCN_E_INVALID;
    if (plan->bucket_state != 1) return DCN_E_UNSUPPORTED;   // no backward pass has run yet (or events unavailable)
    return hipStreamWaitEvent((hipStream_t)stream, plan->ev_bucket[k], 0) == hipSuccess ? DCN_OK : DCN_E_LAUNCH;
}
extern "C" int dcn_plan_fused_bn_backward(const dcn_plan* plan) { return plan ? plan->fused_bn_bwd : DCN_E_INVALID; }
extern "C" int dcn_plan_num_activation_slots(const dcn_plan* plan) { return plan ? plan->n_act : DCN_E_INVALID; }
extern "C" size_t dcn_plan_activation_absmax_offset(const dcn_plan* plan) { return plan ? plan->s_actmax * sizeof(float) : 0; }
extern "C" size_t dcn_plan_saved_bytes(const dcn_plan* plan) { return plan ? plan->saved_floats * sizeof(float) : 0; }
extern "C" size_t dcn_plan_workspace_bytes(const dcn_plan* plan) { return plan ? plan->ws_floats * sizeof(float) : 0; }
extern "C" double dcn_plan_forward_flops(const dcn_plan* plan) { return plan ? plan->flops : 0.0; }

namespace {

// image_b (optional, grouped plans): the second batch of a forward_pair call -- images [N/2, N) come from there instead of
// from image + N/2 images (no concatenated copy of the two batches is needed)
int forward_impl(dcn_plan* plan, const float* image, const float* image_b, const float* const* params,
                 float* const* bn_running, float momentum, float eps, int training, int normalize,
                 float* descriptors, void* saved, void* workspace, void* stream) {
    if (!plan || !image || !params || !descriptors || !saved || !workspace) return DCN_E_INVALID;
    dcn_plan& p = *plan;
    if (image_b && p.groups != 2) return DCN_E_INVALID;
    Run R{p, params, (float*)saved, (float*)workspace, (hipStream_t)stream};
    Run::ObserverGuard observe(p);
    hipStream_t st = R.st;
    const int N = p.N;

    // abs-max slots of the activation tensors (+ the status word behind them)
    DCN_TRY(R.other([&] { return dcn::fill_bytes_async(R.S(p.s_actmax), 0, ((size_t)p.n_act + 1) * sizeof(float), st); }));
    // stem: NCHW(3) -> NHWC(4), weight [w][7][7][3] -> [w][7][7][4]
    const ConvL& stem = p.convs[p.stem];
    if (image_b) {
        dcn::launch_nchw3_to_nhwc4(image, R.S(p.s_in4), N / 2, p.H * p.W, R.A(stem.in_act), st);
        dcn::launch_nchw3_to_nhwc4(image_b, R.S(p.s_in4) + (size_t)(N / 2) * p.H * p.W * 4, N / 2, p.H * p.W, R.A(stem.in_act), st);
    } else {
        dcn::launch_nchw3_to_nhwc4(image, R.S(p.s_in4), N, p.H * p.W, R.A(stem.in_act), st);
    }
    dcn::launch_pad_c3_to_c4(R.P(stem.w), R.Wk(p.w_wstem), (int64_t)p.base * 49, st);
    dcn_plan::FwdRecord rec;
    rec.saved = saved;
    rec.conv_mode = p.conv_mode;
    rec.mid_hl_only.assign(p.convs.size(), 0);
    rec.hl_x_written.assign(p.convs.size(), 0);
    const bool fused_eval = !training && p.conv_mode == DCN_CONV_F16X3;
    const bool f16_mode = p.conv_mode == DCN_CONV_F16X3;
    if (fused_eval) {
        // Inference: the BN scale / shift only depend on the running statistics, so they are computed up front, the scale
        // is folded into the fp16 weight images and every conv + BN (+ residual) + ReLU is ONE kernel pass.
        if (!bn_running) return DCN_E_INVALID;
        for (const BnL& b : p.bns) {
            if (!bn_running[2 * b.idx] || !bn_running[2 * b.idx + 1]) return DCN_E_INVALID;
            dcn::launch_bn_finalize(nullptr, 0, p.groups, b.C, (double)(b.rows / p.groups), R.P(b.g), R.P(b.b),
                                    bn_running[2 * b.idx], bn_running[2 * b.idx + 1], momentum, eps, 0, R.S(b.stats), nullptr,
                                    nullptr, st);
        }
        DCN_TRY(R.split_all_weights(false, R.Wk(p.w_wstem), true));
        DCN_TRY(R.conv_fused(stem, R.S(p.s_in4), nullptr, 1, R.S(p.s_stem_y), p.blocks[0].act_in));
        {
            const BnL& b = p.bns[stem.bn];
            const int hp = (stem.d.hout + 2 - 3) / 2 + 1, wp = (stem.d.wout + 2 - 3) / 2 + 1;
            dcn::launch_maxpool_fwd(R.S(p.s_stem_y), R.S(p.s_pool), (unsigned char*)R.S(p.s_argmax), N, stem.d.hout,
                                    stem.d.wout, hp, wp, b.C, st);
        }
        for (const BlockL& blk : p.blocks) {
            const float* in = R.S(blk.in);
            const float* cur = in;
            for (int i = 0; i + 1 < blk.nconv; ++i) {
                DCN_TRY(R.conv_fused(p.convs[blk.conv[i]], cur, nullptr, 1, R.S(blk.mid[i]), blk.act_mid[i]));
                cur = R.S(blk.mid[i]);
            }
            const ConvL& last = p.convs[blk.conv[blk.nconv - 1]];
            const float* res = in;
            if (blk.down >= 0) {
                const ConvL& dc = p.convs[blk.down];
                DCN_TRY(R.conv_fused(dc, in, nullptr, 0, R.S(dc.x), -1));
                res = R.S(dc.x);
            }
            DCN_TRY(R.conv_fused(last, cur, res, 1, R.S(blk.out), blk.act_out));
        }
    } else {
    const bool stem8 = p.conv_mode == DCN_CONV_F16X3 && dcn::tuning().stem8 != 0 && dcn::tuning().gemm_uni != 0 && stem.d.win >= 8 &&
                       stem.d.kh == 7 && stem.d.cin == 4 && (int64_t)stem.d.n * stem.d.hin * stem.d.win * 16 <= ((int64_t)1 << 31);
    // Weight images on the side stream (round 4, training, split-fp16, stem on its own image): the batched splits of ALL layers'
    // weights -- the forward images AND the channel-transposed ones of this call's backward pass, the latter into the saved arena
    // -- run next to the input layout pass, the stem convolution, its batch norm and the max pool instead of in front of them
    // (and in front of the backward pass): ~0.2 ms per step off the critical path.  Fork: after the stem's weight padding
    // (the forward image of the stem is part of the batch); join: before the first block.  Not while launches are being timed
    // one by one, not inside a hipGraph capture.
    bool hoist = training && f16_mode && stem8 && !p.prof_on && dcn::tuning().wsplit_overlap != 0 && ensure_side(p);
    if (hoist) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) hoist = false;
    }
    // an error return between fork and join must not leave side-stream kernels writing into arenas the caller is about to free:
    // the guard makes the caller's stream wait for whatever the side stream has been given
    struct ForkGuard {
        dcn_plan& p;
        hipStream_t st;
        bool armed = false;
        ~ForkGuard() {
            if (armed && hipEventRecord(p.ev_ws[1], p.side) == hipSuccess) (void)hipStreamWaitEvent(st, p.ev_ws[1], 0);
        }
    } fork_guard{p, st};
    if (p.conv_mode == DCN_CONV_F16X3) {
        if (hoist) {
            Run Rs{p, params, (float*)saved, (float*)workspace, p.side};
            bool ok = hipEventRecord(p.ev_ws[0], st) == hipSuccess && hipStreamWaitEvent(p.side, p.ev_ws[0], 0) == hipSuccess;
            if (!ok) return DCN_E_LAUNCH;
            fork_guard.armed = true;
            DCN_TRY(Rs.split_all_weights(false, Rs.Wk(p.w_wstem)));
            DCN_TRY(Rs.split_hl_weights(false));
            Rs.wh_over = Rs.S(p.s_wht); Rs.wl_over = Rs.S(p.s_wlt); Rs.whl_over = Rs.S(p.s_whlt);
            DCN_TRY(Rs.split_all_weights(true, nullptr));
            DCN_TRY(Rs.split_hl_weights(true));
            rec.wt_saved = true;
            rec.hl_t_written.assign(p.convs.size(), 0);
            for (const ConvL& c : p.convs) rec.hl_t_written[c.idx] = Rs.use_hl(c, 1) ? 1 : 0;
            if (hipEventRecord(p.ev_ws[1], p.side) != hipSuccess) return DCN_E_LAUNCH;
        } else {
            DCN_TRY(R.split_all_weights(false, R.Wk(p.w_wstem)));
            DCN_TRY(R.split_hl_weights(false));
        }
    }
    if (stem8) {
        _Float16* hi = (_Float16*)R.Wk(p.w_stem8);
        DCN_TRY(R.other([&] { return dcn_split_stem_weights_f16(R.Wk(p.w_wstem), hi, hi + (size_t)p.base * 224, p.base, kWeightScale, st); }));
        R.stem8 = true;
    }
    DCN_TRY(R.conv_bn(stem, R.S(p.s_in4), R.Wk(p.w_wstem), bn_running, momentum, eps, training, p.blocks[0].act_in));
    {
        const BnL& b = p.bns[stem.bn];
        const float* s = R.S(b.stats);
        const int hp = (stem.d.hout + 2 - 3) / 2 + 1, wp = (stem.d.wout + 2 - 3) / 2 + 1;
        if (dcn::tuning().stem_pool_fused != 0) {
            // batch norm + ReLU applied inside the pooling pass: the stem's activation (the largest tensor of the network) is
            // never stored; its sign mask -- what the batch norm's backward pass reads -- is
            dcn::launch_maxpool_fwd(R.S(stem.x), R.S(p.s_pool), (unsigned char*)R.S(p.s_argmax), N, stem.d.hout, stem.d.wout, hp,
                                    wp, b.C, st, s, p.groups, R.M(p.s_stem_y));
        } else {
            dcn::launch_bn_apply(R.S(stem.x), s, nullptr, nullptr, 1, R.S(p.s_stem_y), R.M(p.s_stem_y), b.C, b.rows, p.groups, st);
            dcn::launch_maxpool_fwd(R.S(p.s_stem_y), R.S(p.s_pool), (unsigned char*)R.S(p.s_argmax), N, stem.d.hout,
                                    stem.d.wout, hp, wp, b.C, st);
        }
    }
    if (hoist && hipStreamWaitEvent(st, p.ev_ws[1], 0) != hipSuccess) return DCN_E_LAUNCH;   // join: the weight images are there
    fork_guard.armed = false;
    if (training && f16_mode && p.blocks[0].has_hl_in) {   // (the first block's input is the max-pool output: no apply pass writes it)
        const BlockL& b0 = p.blocks[0];
        R.hl_saved.emplace_back(R.S(b0.in), R.S(b0.hl_in));
        DCN_TRY(R.other([&] { return dcn_split_act_hl32(R.S(b0.in), R.A(b0.act_in), R.S(b0.hl_in), b0.in_rows, b0.in_c, st); }));
    }
    if (training && f16_mode)   // every saved hl32 image the plan reserved is written by this call (producers or stand-alone passes)
        for (const ConvL& c : p.convs) rec.hl_x_written[c.idx] = c.has_hl_x ? 1 : 0;
    for (const BlockL& blk : p.blocks) {
        const float* in = R.S(blk.in);
        const float* cur = in;
        if (blk.down >= 0) {   // (first: the last batch norm's output bound needs the downsample branch's)
            const ConvL& dc = p.convs[blk.down];
            DCN_TRY(R.conv_bn(dc, in, R.P(dc.w), bn_running, momentum, eps, training, blk.act_down));
        }
        for (int i = 0; i < blk.nconv; ++i) {
            const ConvL& c = p.convs[blk.conv[i]];
            if (i + 1 < blk.nconv) {
                DCN_TRY(R.conv_bn(c, cur, R.P(c.w), bn_running, momentum, eps, training, blk.act_mid[i]));
                const BnL& b = p.bns[c.bn];
                const float* s = R.S(b.stats);
                float* slot = (training && f16_mode && blk.has_hl_mid[i]) ? R.S(blk.hl_mid[i]) : nullptr;
                const ConvL& nxt = p.convs[blk.conv[i + 1]];
                void* hl = training ? R.hl_image_for(R.S(blk.mid[i]), 1, &nxt, nullptr, slot) : nullptr;
                // the tensor's readers: the next convolution, that convolution's weight gradient (the batch norm's own backward
                // pass takes the ReLU mask).  Both on the hl32 kernels: the image in the saved arena is the only copy written
                const bool hl_only = hl && slot && hl == (void*)slot && (b.C % 32) == 0 && dcn::tuning().hl_only_mid != 0 &&
                                     R.use_hl(nxt, 0) && R.use_wgrad_hl(nxt);
                rec.mid_hl_only[nxt.idx] = hl_only ? 1 : 0;
                dcn::launch_bn_apply(R.S(c.x), s, nullptr, nullptr, 1, hl_only ? nullptr : R.S(blk.mid[i]), R.M(blk.mid[i]), b.C,
                                     b.rows, p.groups, st, hl, R.A(blk.act_mid[i]));
                DCN_TRY(R.ensure_saved_hl(R.S(blk.mid[i]), slot, b.rows, b.C, blk.act_mid[i]));
                cur = R.S(blk.mid[i]);
            } else {
                DCN_TRY(R.conv_bn(c, cur, R.P(c.w), bn_running, momentum, eps, training, blk.act_out,
                                  blk.down >= 0 ? blk.act_down : blk.act_in));
            }
        }
        const ConvL& last = p.convs[blk.conv[blk.nconv - 1]];
        const BnL& bl = p.bns[last.bn];
        const float* sl = R.S(bl.stats);
        // the block's output feeds the next block's first convolution (and its downsample branch)
        const BlockL* nb = (&blk != &p.blocks.back()) ? &blk + 1 : nullptr;
        float* slot = (training && f16_mode && nb && nb->has_hl_in) ? R.S(nb->hl_in) : nullptr;
        void* hl = (training && nb) ? R.hl_image_for(R.S(blk.out), 0, &p.convs[nb->conv[0]], nb->down >= 0 ? &p.convs[nb->down] : nullptr,
                                                     slot)
                                    : nullptr;
        if (blk.down >= 0) {
            const ConvL& dc = p.convs[blk.down];
            const float* sd = R.S(p.bns[dc.bn].stats);
            dcn::launch_bn_apply(R.S(last.x), sl, R.S(dc.x), sd, 1, R.S(blk.out), R.M(blk.out), bl.C, bl.rows, p.groups, st, hl,
                                 R.A(blk.act_out));
        } else {
            dcn::launch_bn_apply(R.S(last.x), sl, in, nullptr, 1, R.S(blk.out), R.M(blk.out), bl.C, bl.rows, p.groups, st, hl,
                                 R.A(blk.act_out));
        }
        DCN_TRY(R.ensure_saved_hl(R.S(blk.out), slot, bl.rows, bl.C, blk.act_out));
    }
    }   // !fused_eval
    // scoring layer (1x1 conv + bias) into the padded low-resolution map, then bilinear upsample
    const ConvL& fc = p.convs[p.fc];
    const size_t low_bytes = (size_t)N * p.hl * p.wl * p.Dp * sizeof(float);
    DCN_TRY(R.other([&] { return dcn::fill_bytes_async(R.S(p.s_low), 0, low_bytes, st); }));
    DCN_TRY(R.conv_fwd(fc, R.S(p.blocks.back().out), R.P(fc.w), R.P(fc.b), R.S(p.s_low), nullptr));
    dcn::launch_upsample_fwd(R.S(p.s_low), N, p.hl, p.wl, p.Dp, p.D, p.H, p.W, normalize, descriptors, st);
    DCN_TRY(R.other([&] {
        hipLaunchKernelGGL(act_status_kernel, dim3(1), dim3(64), 0, st, (const float*)R.S(p.s_actmax), p.n_act,
                           (int*)(R.S(p.s_actmax) + p.n_act));
        return dcn::check_launch();
    }));
    if (training) {   // what the backward pass of this arena must agree with (dcn_plan::FwdRecord)
        for (size_t i = 0; i < p.fwd_records.size();)
            if (p.fwd_records[i].saved == saved) p.fwd_records.erase(p.fwd_records.begin() + i);
            else ++i;
        if (p.fwd_records.size() >= dcn_plan::kMaxFwdRecords) p.fwd_records.erase(p.fwd_records.begin());
        p.fwd_records.push_back(std::move(rec));
    }
    return dcn::check_launch();
}

}  // namespace

extern "C" int dcn_backbone_forward(dcn_plan* plan, const float* image, const float* const* params,
                                    float* const* bn_running, float momentum, float eps, int training, int normalize,
                                    float* descriptors, void* saved, void* workspace, void* stream) {
    return forward_impl(plan, image, nullptr, params, bn_running, momentum, eps, training, normalize, descriptors, saved,
                        workspace, stream);
}
extern "C" int dcn_backbone_forward_pair(dcn_plan* plan, const float* image_a, const float* image_b,
                                         const float* const* params, float* const* bn_running, float momentum, float eps,
                                         int training, int normalize, float* descriptors, void* saved, void* workspace,
                                         void* stream) {
    if (!image_b) return DCN_E_INVALID;
    return forward_impl(plan, image_a, image_b, params, bn_running, momentum, eps, training, normalize, descriptors, saved,
                        workspace, stream);
}

namespace {

// grad_b (optional, grouped plans): the gradient of the second batch's descriptors (images [N/2, N)) when the two
// gradients of a forward_pair call are separate tensors
int backward_impl(dcn_plan* plan, const float* grad_descriptors, const float* grad_b, const float* const* params,
                  const void* saved, void* workspace, float* const* grads, int normalize, void* stream) {
    if (!plan || !grad_descriptors || !params || !saved || !workspace || !grads) return DCN_E_INVALID;
    dcn_plan& p = *plan;
    if (grad_b && p.groups != 2) return DCN_E_INVALID;
    Run R{p, params, (float*)saved, (float*)workspace, (hipStream_t)stream};
    Run::ObserverGuard observe(p);
    hipStream_t st = R.st;
    const int N = p.N;
    // the forward call that filled this arena: its decisions must still hold (see dcn_plan::FwdRecord)
    dcn_plan::FwdRecord rec;
    {
        bool found = false;
        for (size_t i = p.fwd_records.size(); i-- > 0;)   // (kept: the same arena may be differentiated again)
            if (p.fwd_records[i].saved == saved) {
                rec = p.fwd_records[i];
                found = true;
                break;
            }
        if (!found || rec.conv_mode != p.conv_mode) return DCN_E_INVALID;   // no training-mode forward of this arena, or another arithmetic
        for (const ConvL& c : p.convs) {
            if (R.use_wgrad_hl(c) && !rec.hl_x_written[c.idx]) return DCN_E_INVALID;    // would read an hl32 image nobody wrote
            if (rec.mid_hl_only[c.idx] && !R.use_wgrad_hl(c)) return DCN_E_INVALID;     // would read an fp32 activation nobody wrote
        }
    }
    float* part = R.Wk(p.w_part);
    float* k123 = R.Wk(p.w_k123);
    float* wt = R.Wk(p.w_wt);
    float* slab = R.Wk(p.w_slab);
    float* amax = R.Wk(p.w_amax);
    const bool f16 = p.conv_mode == DCN_CONV_F16X3;

    // Overlap (split-fp16 mode, not while launches are being timed one by one): wgrad of layer k only feeds the optimizer,
    // so it runs on the plan's side stream while the main stream goes on with dgrad(k) -> BN backward(k - 1) -> ...; the
    // gradient's pixel-blocked image alternates between two buffers and events order writer and reader of each.
    bool overlap = f16 && !p.prof_on;
    overlap = overlap && ensure_side(p);
    if (p.bucket_state == 0) {   // events behind dcn_plan_stream_wait_grad_bucket
        bool ok = true;
        p.ev_bucket.assign(p.bucket_first.size(), nullptr);
        for (hipEvent_t& ev : p.ev_bucket) ok = ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&p.ev_bucket_side, hipEventDisableTiming) == hipSuccess;
        p.bucket_state = ok ? 1 : -1;
    }
    // event / stream-dependency calls of the overlap and bucket logic: a failure must not pass silently (a missed
    // dependency is a data race), so every one is checked
    int rt_fail = 0;
    auto RT = [&](hipError_t e) { if (e != hipSuccess) rt_fail = 1; };
    if (overlap) {   // inside a hipGraph capture the fork / join pattern replays slower than the serial chain (measured): stay serial
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) overlap = false;
    }
    int n_bn = 0, cur = 0;              // BN-backward launches so far; buffer holding the newest split gradient
    bool wg_pending[2] = {false, false};
    float* const dqbuf[2] = {R.Wk(p.w_dq), R.Wk(p.w_dq2)};
    float* const hlimg[2] = {R.hlbuf(0), R.hlbuf(1)};   // hl32 images of dx, alternating like dqbuf (same events order their users)
    const float* dq_of = nullptr;       // gradient tensor whose pixel-blocked split copy is in dqbuf[cur]
    // BN backward of conv c's batch norm: dy (+ optional relu mask from relu_out) -> dx; g_out optional
    // Fused reduction (split-fp16 mode): the dgrad that produces a batch norm's upstream gradient dy also masks it with the
    // ReLU bits and leaves the per-tile sums of the BN backward reduction in `part` (GemmConv::bnb_*): bn_bwd then only
    // runs finalize + apply.  red_dy / red_tiles: the tensor that currently has its sums in `part`, and how many rows per group.
    const bool fuse_red = f16 && dcn::tuning().bn_bwd_fused != 0;
    p.fused_bn_bwd = 0;
    const float* red_dy = nullptr;
    int red_tiles = 0;
    const float* hl_dx_of = nullptr;    // gradient tensor whose hl32 image sits in w_hl (written by the BN backward apply pass)
    // dy2 (optional): the upstream gradient is dy + dy2 -- the residual branch's share, added in the batch norm's streaming
    // passes instead of in the epilogue of the dgrad that produced dy
    auto bn_bwd = [&](const ConvL& c, const float* dy, const float* relu_out, float* dx, float* g_out, const float* dy2 = nullptr) {
        const BnL& b = p.bns[c.bn];
        const float* s = R.S(b.stats);
        // (the mask bytes the forward wrote next to that activation; relu_out itself is then not read)
        const unsigned char* mask = relu_out ? R.M((size_t)(relu_out - R.saved)) : nullptr;
        cur = overlap ? (n_bn & 1) : 0;
        if (overlap && wg_pending[cur]) RT(hipStreamWaitEvent(st, p.ev_wg[cur], 0));   // the wgrad that read this buffer two layers ago
        const int tiles = (red_dy == dy && dy) ? red_tiles : 0;
        red_dy = nullptr;
        if (tiles > 0) { relu_out = nullptr; mask = nullptr; g_out = nullptr; ++p.fused_bn_bwd; }   // dy is already the masked gradient
        // the dgrad of this convolution on the hl32 path: the apply pass writes dx as the hl32 image INSTEAD of the fp32 tensor
        // (wgrad reads the pixel-blocked image, nobody reads the fp32 one)
        const bool prod = f16 && dcn::tuning().hl_producers != 0;
        const bool hl_d = prod && !fuse_red && R.use_hl(c, 1);       // this convolution's dgrad reads the hl32 image
        const bool hl_w = prod && R.use_wgrad_hl(c);                  // ... its weight gradient too: no pixel-blocked image then
        hl_dx_of = (hl_d || hl_w) ? dx : nullptr;
        dcn::launch_bn_bwd(dy, relu_out, mask, R.S(c.x), s, R.P(b.g), b.C, b.rows, p.groups, part, grads[b.g],
                           grads[b.b], k123, dx, g_out, f16 ? amax + c.idx : nullptr,
                           (f16 && !hl_w) ? (void*)dqbuf[cur] : nullptr, st, tiles, dy2,
                           (hl_d || hl_w) ? (void*)hlimg[cur] : nullptr, hl_d ? 0 : 1);
        if (overlap) RT(hipEventRecord(p.ev_dq[cur], st));
        ++n_bn;
        dq_of = (f16 && !hl_w) ? dx : nullptr;   // the pixel-blocked split copy of this dx now sits in dqbuf[cur]
    };
    auto wgrad = [&](const ConvL& c, const float* in, const float* dx, float* dw) -> int {
        if (!f16) return R.timed(1, c.flops, [&] { return dcn_conv_wgrad(&c.d, in, dx, dw, slab, st); });
        if (R.use_wgrad_hl(c)) {   // both operands as hl32 tensors: the saved image of the input, the image of dx
            const float* ximg = R.S(c.hl_x);
            if (hl_dx_of != dx) {  // (DCN_HL_PRODUCERS=0, or a gradient that no batch-norm backward produced)
                DCN_TRY(R.other([&] { return dcn_split_act_hl32(dx, amax + c.idx, hlimg[cur], (int64_t)c.d.n * c.d.hout * c.d.wout, c.d.ldc, st); }));
                if (overlap) RT(hipEventRecord(p.ev_dq[cur], st));
                hl_dx_of = dx;
            }
            if (overlap) {
                RT(hipStreamWaitEvent(p.side, p.ev_dq[cur], 0));
                DCN_TRY(dcn_conv_wgrad_hl(&c.d, ximg, R.A(c.in_act), hlimg[cur], amax + c.idx, dw, slab, p.side));
                RT(hipEventRecord(p.ev_wg[cur], p.side));
                wg_pending[cur] = true;
                return DCN_OK;
            }
            return R.timed(1, c.flops, [&] {
                return dcn_conv_wgrad_hl(&c.d, ximg, R.A(c.in_act), hlimg[cur], amax + c.idx, dw, slab, st);
            });
        }
        // The activation operand is the fp32 tensor itself, split on the fly inside the kernel: measured faster than a
        // split pass + pre-split operand on every layer of ResNet34 / ResNet50 (the pass costs more than the conversions).
        if (dq_of != dx) {   // (BN backward emits it directly; only the scoring layer's gradient needs the separate pass)
            DCN_TRY(R.other([&] { return dcn_split_grad_blocked_f16(dx, c.d.n * c.d.hout * c.d.wout, c.d.ldc, amax + c.idx, dqbuf[0], st); }));
            return R.timed(1, c.flops, [&] {
                return dcn_conv_wgrad_f16(&c.d, in, 1, R.A(c.in_act), dqbuf[0], amax + c.idx, dw, slab, st);
            });
        }
        if (overlap) {
            RT(hipStreamWaitEvent(p.side, p.ev_dq[cur], 0));
            DCN_TRY(dcn_conv_wgrad_f16(&c.d, in, 1, R.A(c.in_act), dqbuf[cur], amax + c.idx, dw, slab, p.side));
            RT(hipEventRecord(p.ev_wg[cur], p.side));
            wg_pending[cur] = true;
            return DCN_OK;
        }
        return R.timed(1, c.flops, [&] {
            return dcn_conv_wgrad_f16(&c.d, in, 1, R.A(c.in_act), dqbuf[cur], amax + c.idx, dw, slab, st);
        });
    };
    // bn_of: the convolution whose batch norm's backward consumes din as its upstream gradient (din = gradient w.r.t. that
    // batch norm's ReLU'd output `relu_out`), or null
    auto dgrad = [&](const ConvL& c, const float* dx, const float* add, float* din, const ConvL* bn_of = nullptr,
                     const float* relu_out = nullptr) -> int {
        if (!f16) {
            DCN_TRY(R.other([&] { return dcn_transpose_weight(R.P(c.w), wt, c.d.cout, c.d.kh * c.d.kw, c.d.cin, c.d.ldc, st); }));
            return R.timed(0, c.flops, [&] { return dcn_conv_dgrad(&c.d, dx, wt, add, din, R.SK(c, 1), st); });
        }
        if (fuse_red && bn_of) {
            const BnL& b = p.bns[bn_of->bn];
            const int tiles = dcn_conv_dgrad_bn_num_mtiles_f16(&c.d);
            if (tiles > 0 && tiles % p.groups == 0 && b.C == c.d.cin && b.rows == (int64_t)c.d.n * c.d.hin * c.d.win) {
                const unsigned char* mask = relu_out ? R.M((size_t)(relu_out - R.saved)) : nullptr;
                red_dy = din;
                red_tiles = tiles / p.groups;
                return R.timed(0, c.flops, [&] {
                    return dcn_conv_dgrad_bn_f16(&c.d, dx, R.wimg(p.w_wh, c), R.wimg(p.w_wl, c), kWeightScale, amax + c.idx, add,
                                                 din, R.S(bn_of->x), mask, R.S(b.stats), part, R.SK(c, 1), st);
                });
            }
        }
        if (R.use_hl(c, 1)) {
            return R.timed(2, c.flops, [&] {
                if (hl_dx_of != dx) {
                    // (the side stream's weight-gradient kernel may still be reading hlimg[cur]: the image is only rewritten when
                    // it holds another tensor, i.e. after the wait at the top of the bn_bwd that owns the buffer)
                    DCN_TRY(dcn_split_act_hl32(dx, amax + c.idx, hlimg[cur], (int64_t)c.d.n * c.d.hout * c.d.wout, c.d.ldc, st));
                    hl_dx_of = dx;
                }
                return dcn_conv_dgrad_hl(&c.d, hlimg[cur], R.whl(c), kWeightScale, amax + c.idx, add, din, R.SKhl(c, 1), st);
            });
        }
        return R.timed(0, c.flops, [&] {   // (transposed weight images: split_all_weights(true) below)
            return dcn_conv_dgrad_f16(&c.d, dx, R.wimg(p.w_wh, c), R.wimg(p.w_wl, c), kWeightScale, amax + c.idx, add, din,
                                      R.SK(c, 1), st);
        });
    };
    if (f16) {
        // the forward call of this arena already made this pass's weight images (dcn_plan::FwdRecord::wt_saved) -- unless a
        // convolution that takes the hl32 dgrad NOW was not among them (the tuning changed in between): then they are made here
        bool have = rec.wt_saved;
        if (have)
            for (const ConvL& c : p.convs)
                if (R.use_hl(c, 1) && !rec.hl_t_written[c.idx]) have = false;
        if (have) {
            R.wh_over = R.S(p.s_wht); R.wl_over = R.S(p.s_wlt); R.whl_over = R.S(p.s_whlt);
        } else {
            DCN_TRY(R.split_all_weights(true, nullptr));
            DCN_TRY(R.split_hl_weights(true));
        }
    }
    // split-fp16 mode: every gradient tensor that feeds a convolution records its abs-max (pre-scale selection)
    if (f16) DCN_TRY(R.other([&] { return dcn::fill_bytes_async(amax, 0, p.convs.size() * sizeof(float), st); }));

    // ---- upsample + scoring layer
    float* glow = R.Wk(p.w_glow);
    if (normalize) {  // network.py:256-259 was fused into the forward upsample: undo it first
        if (grad_b) {
            const size_t lo = (size_t)(N / 2) * p.hl * p.wl * p.Dp, hi = (size_t)(N / 2) * p.H * p.W * p.D;
            dcn::launch_normalize_bwd(R.S(p.s_low), N / 2, p.hl, p.wl, p.Dp, p.D, p.H, p.W, grad_descriptors, R.Wk(p.w_gnorm), st);
            dcn::launch_normalize_bwd(R.S(p.s_low) + lo, N / 2, p.hl, p.wl, p.Dp, p.D, p.H, p.W, grad_b, R.Wk(p.w_gnorm) + hi, st);
            grad_b = nullptr;
        } else {
            dcn::launch_normalize_bwd(R.S(p.s_low), N, p.hl, p.wl, p.Dp, p.D, p.H, p.W, grad_descriptors, R.Wk(p.w_gnorm), st);
        }
        grad_descriptors = R.Wk(p.w_gnorm);
    }
    dcn::launch_upsample_bwd(grad_descriptors, N, p.hl, p.wl, p.Dp, p.D, p.H, p.W, R.Wk(p.w_ups), glow,
                             f16 ? amax + p.convs[p.fc].idx : nullptr, st, grad_b);
    const ConvL& fc = p.convs[p.fc];
    const float* feat = R.S(p.blocks.back().out);
    DCN_TRY(wgrad(fc, feat, glow, grads[fc.w]));
    DCN_TRY(R.other([&] {
        hipLaunchKernelGGL(colsum_kernel, dim3(p.D), dim3(1024), 0, st, (const float*)glow, (int64_t)N * p.hl * p.wl, p.Dp,
                           grads[fc.b]);
        return dcn::check_launch();
    }));
    float* dout = R.Wk(p.w_buf[0]);   // gradient w.r.t. the current block's output
    float* dnext = R.Wk(p.w_buf[1]);  // gradient w.r.t. its input (swapped after every block)
    float* gbuf = R.Wk(p.w_buf[2]);   // relu-masked dout (residual branch)
    float* dxa = R.Wk(p.w_buf[3]);
    float* dxb = R.Wk(p.w_buf[4]);
    float* dpart = R.Wk(p.w_buf[5]);
    {
        const BlockL& lb = p.blocks.back();
        DCN_TRY(dgrad(fc, glow, nullptr, dout, &p.convs[lb.conv[lb.nconv - 1]], R.S(lb.out)));
    }

    // The gradient w.r.t. a block's input is  dgrad(conv1) + (identity gradient | dgrad(downsample) + ...): the second
    // summand is NOT added in the GEMM epilogue (scalar loads on the critical path of a one-workgroup-per-CU kernel:
    // measured +22 % / +43 % / +58 % on such a dgrad of layer 4 / 3 / 2) but handed to the previous block's batch-norm
    // backward, whose streaming passes read dy + dy2 (the first block's: to the max-pool backward); DCN_DEFER_RESIDUAL_ADD=0
    // restores the epilogue add
    const bool defer_add = dcn::tuning().defer_residual_add != 0 && !fuse_red;
    const float* dout_add = nullptr;   // second summand of dout (the current block's output gradient), or null
    const float* next_add = nullptr;
    for (int bi = (int)p.blocks.size() - 1; bi >= 0; --bi) {
        const BlockL& blk = p.blocks[bi];
        const float* in = R.S(blk.in);
        // last conv's BN: relu mask from the block output, emits g for the identity branch
        const ConvL& last = p.convs[blk.conv[blk.nconv - 1]];
        // (with the fused reduction dout already IS the relu-masked gradient: it serves as the residual branch's gradient)
        const float* gres = (red_dy == dout) ? dout : gbuf;
        bn_bwd(last, dout, R.S(blk.out), dxa, gbuf, dout_add);
        dout_add = nullptr;
        float* dx = dxa;
        float* dy = dxb;
        // the gradient w.r.t. this block's input is the upstream gradient of the previous block's last batch norm
        const ConvL* up_bn = bi > 0 ? &p.convs[p.blocks[bi - 1].conv[p.blocks[bi - 1].nconv - 1]] : nullptr;
        const float* up_relu = bi > 0 ? R.S(p.blocks[bi - 1].out) : nullptr;
        for (int i = blk.nconv - 1; i >= 0; --i) {
            const ConvL& c = p.convs[blk.conv[i]];
            const float* cin = i == 0 ? in : R.S(blk.mid[i - 1]);
            DCN_TRY(wgrad(c, cin, dx, grads[c.w]));
            if (i > 0) {
                const ConvL& prev = p.convs[blk.conv[i - 1]];
                DCN_TRY(dgrad(c, dx, nullptr, dy, &prev, R.S(blk.mid[i - 1])));  // dy = grad w.r.t. mid[i-1]
                bn_bwd(prev, dy, R.S(blk.mid[i - 1]), dx, nullptr);  // dx reused: grad w.r.t. prev conv output
            } else if (blk.down >= 0) {
                DCN_TRY(dgrad(c, dx, nullptr, dpart));
            } else if (defer_add) {
                DCN_TRY(dgrad(c, dx, nullptr, dnext));   // the identity gradient joins in the previous block's BN backward
                next_add = gres;
            } else {
                DCN_TRY(dgrad(c, dx, gres, dnext, up_bn, up_relu));  // + identity gradient
            }
        }
        if (blk.down >= 0) {
            const ConvL& dc = p.convs[blk.down];
            bn_bwd(dc, gres, nullptr, dxa, nullptr);
            DCN_TRY(wgrad(dc, in, dxa, grads[dc.w]));
            if (defer_add) {
                DCN_TRY(dgrad(dc, dxa, nullptr, dnext));
                next_add = dpart;
            } else {
                DCN_TRY(dgrad(dc, dxa, dpart, dnext, up_bn, up_relu));
            }
        }
        dout_add = next_add;
        next_add = nullptr;
        float* t = dout; dout = dnext; dnext = t;
        // gradient bucket complete?  (every launch that writes one of its gradients has been enqueued: the side stream's
        // weight-gradient GEMMs are joined into the caller's stream first)
        for (size_t k = 0; k + 1 < p.bucket_block.size(); ++k)
            if (p.bucket_block[k] == bi && p.bucket_state == 1) {
                if (overlap) {
                    RT(hipEventRecord(p.ev_bucket_side, p.side));
                    RT(hipStreamWaitEvent(st, p.ev_bucket_side, 0));
                }
                RT(hipEventRecord(p.ev_bucket[k], st));
            }
    }
    // ---- max pool, stem
    const ConvL& stem = p.convs[p.stem];
    const BnL& sb = p.bns[stem.bn];
    const int hp = (stem.d.hout + 2 - 3) / 2 + 1, wp = (stem.d.wout + 2 - 3) / 2 + 1;
    dcn::launch_maxpool_bwd(dout, (const unsigned char*)R.S(p.s_argmax), dnext, N, stem.d.hout, stem.d.wout, hp, wp, sb.C,
                            st, dout_add);   // (the first block's deferred identity gradient joins here)
    bn_bwd(stem, dnext, R.S(p.s_stem_y), dxa, nullptr);
    DCN_TRY(wgrad(stem, R.S(p.s_in4), dxa, R.Wk(p.w_dwstem)));
    if (overlap) {   // join: everything the side stream produced is ordered before whatever follows on the caller's stream
        RT(hipEventRecord(p.ev_join, p.side));
        RT(hipStreamWaitEvent(st, p.ev_join, 0));
    }
    dcn::launch_unpad_c4_to_c3(R.Wk(p.w_dwstem), grads[stem.w], (int64_t)p.base * 49, st);
    if (p.bucket_state == 1) RT(hipEventRecord(p.ev_bucket.back(), st));   // last bucket: the whole backward pass
    if (rt_fail) return DCN_E_LAUNCH;
    return dcn::check_launch();
}

}  // namespace

extern "C" int dcn_backbone_backward(dcn_plan* plan, const float* grad_descriptors, const float* const* params,
                                     const void* saved, void* workspace, float* const* grads, int normalize,
                                     void* stream) {
    return backward_impl(plan, grad_descriptors, nullptr, params, saved, workspace, grads, normalize, stream);
}
extern "C" int dcn_backbone_backward_pair(dcn_plan* plan, const float* grad_a, const float* grad_b,
                                          const float* const* params, const void* saved, void* workspace,
                                          float* const* grads, int normalize, void* stream) {
    if (!grad_b) return DCN_E_INVALID;
    return backward_impl(plan, grad_a, grad_b, params, saved, workspace, grads, normalize, stream);
}
