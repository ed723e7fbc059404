// Environment overrides, read once (dcn_tuning.h).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "dcn_common.h"
#include "dcn_tuning.h"

namespace dcn {
namespace {

Tuning g_tuning;
std::once_flag g_once;
std::mutex g_mu;

void read_env(Tuning& t) {
    t = Tuning();
    if (const char* m = getenv("DCN_CONV_MODE")) {
        if (!strcmp(m, "fp32")) t.conv_mode = DCN_CONV_FP32;
        else if (!strcmp(m, "f16x3")) t.conv_mode = DCN_CONV_F16X3;
        else t.conv_mode_invalid = 1;
    }
    if (const char* e = getenv("DCN_BACKWARD_OVERLAP")) t.backward_overlap = atoi(e) != 0;
    if (const char* e = getenv("DCN_GEMM_TILE_M")) t.gemm_tile_m = atoi(e);
    if (const char* e = getenv("DCN_GEMM_SK")) t.gemm_sk = atoi(e);
    if (const char* e = getenv("DCN_GEMM_SK_MIN_GAIN")) t.gemm_sk_min_gain = atof(e);
    if (const char* e = getenv("DCN_STEM8")) t.stem8 = atoi(e) != 0;
    if (const char* e = getenv("DCN_GEMM_UNI")) t.gemm_uni = atoi(e) != 0;
    if (const char* e = getenv("DCN_GEMM_SK_FIXUP")) t.gemm_sk_inline = strcmp(e, "kernel") != 0;
    if (const char* e = getenv("DCN_BN_BWD_FUSED")) t.bn_bwd_fused = atoi(e) != 0;
    if (const char* e = getenv("DCN_DEFER_RESIDUAL_ADD")) t.defer_residual_add = atoi(e) != 0;
    if (const char* e = getenv("DCN_WGRAD_DEEP")) t.wgrad_deep = atoi(e);
    if (const char* e = getenv("DCN_WGRAD_ROLES")) t.wgrad_roles = atoi(e);
    if (const char* e = getenv("DCN_WGRAD_TILE")) t.wgrad_tile = atoi(e);
    if (const char* e = getenv("DCN_WGRAD_SPLITS")) t.wgrad_splits = atoi(e);
    if (const char* e = getenv("DCN_GEMM_HL")) t.gemm_hl = atoi(e);
    if (const char* e = getenv("DCN_GEMM_HL_ROWS")) t.gemm_hl_rows = atoi(e);
    if (const char* e = getenv("DCN_HL_MIN_K")) { const int v = atoi(e); if (v >= 32) t.hl_min_k = v; }
    if (const char* e = getenv("DCN_GEMM_HLX")) {   // "0" / "1": off / on (as decided); "kg,splits": forced (0 = as decided)
        int a = 0, b = 0;
        if (strchr(e, ',')) {
            if (sscanf(e, "%d,%d", &a, &b) == 2) {
                t.gemm_hlx = 1;
                t.gemm_hlx_kg = (a == 1 || a == 2) ? a : 0;
                t.gemm_hlx_splits = b > 0 ? b : 0;
            }
        } else {
            t.gemm_hlx = atoi(e) != 0;
        }
    }
    if (const char* e = getenv("DCN_GEMM_HLX_NARROW")) t.gemm_hlx_narrow = atoi(e) != 0;
    if (const char* e = getenv("DCN_HLX_COST")) {
        double a = 0, b = 0, c = 0;
        if (sscanf(e, "%lf,%lf,%lf", &a, &b, &c) == 3 && a > 0 && b > 0 && c >= 0) { t.hlx_cost1 = a; t.hlx_cost2 = b; t.hlx_split_cost = c; }
    }
    if (const char* e = getenv("DCN_HL_ONLY_MID")) t.hl_only_mid = atoi(e);
    if (const char* e = getenv("DCN_STEM_POOL_FUSED")) t.stem_pool_fused = atoi(e);
    if (const char* e = getenv("DCN_WGRAD_HL")) t.wgrad_hl = atoi(e);
    if (const char* e = getenv("DCN_HL_SETPRIO")) t.hl_setprio = atoi(e) != 0;
    if (const char* e = getenv("DCN_HLX_STAGGER")) { const int v = atoi(e); t.hlx_stagger = (v >= 0 && v <= 2) ? v : 1; }
    if (const char* e = getenv("DCN_HLX_COUNTERS")) t.hlx_counters = atoi(e) != 0;
    if (const char* e = getenv("DCN_WGRAD_HL_MIN_M")) t.wgrad_hl_min_m = atoi(e);
    if (const char* e = getenv("DCN_WGRAD_HLR")) t.wgrad_hlr = atoi(e);
    if (const char* e = getenv("DCN_WGRAD_HLR_MIN_M")) t.wgrad_hlr_min_m = atoi(e);
    if (const char* e = getenv("DCN_WGRAD_HLR_PAIRS")) t.wgrad_hlr_pairs = atoi(e);
    if (const char* e = getenv("DCN_HL_PRODUCERS")) t.hl_producers = atoi(e) != 0;
    if (const char* e = getenv("DCN_BN_REVERSE")) t.bn_reverse = atoi(e);
    if (const char* e = getenv("DCN_BN_NT")) t.bn_nt = atoi(e);
    if (const char* e = getenv("DCN_WSPLIT_OVERLAP")) t.wsplit_overlap = atoi(e) != 0;
    if (const char* e = getenv("DCN_BN_REDUCE_WIDE")) t.bn_reduce_wide = atoi(e);
}

}  // namespace

const Tuning& tuning() {
    std::call_once(g_once, [] { read_env(g_tuning); });
    return g_tuning;
}

}  // namespace dcn

extern "C" void dcn_reload_env(void) {
    (void)dcn::tuning();   // make sure the one-time read cannot run after (and undo) this one
    std::lock_guard<std::mutex> lock(dcn::g_mu);
    dcn::read_env(dcn::g_tuning);
}
