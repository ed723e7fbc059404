// TEMPORARY during bring-up: removed once every entry point is implemented.
#include "dcn_common.h"
extern "C" {
int dcn_plan_create(const char*, int, int, int, int, int, dcn_plan**) { return DCN_E_UNSUPPORTED; }
void dcn_plan_destroy(dcn_plan*) {}
int dcn_plan_num_params(const dcn_plan*) { return 0; }
int dcn_plan_num_bn(const dcn_plan*) { return 0; }
int dcn_plan_param_info(const dcn_plan*, int, char*, int, int64_t*, int*) { return DCN_E_UNSUPPORTED; }
int dcn_plan_bn_info(const dcn_plan*, int, char*, int, int64_t*) { return DCN_E_UNSUPPORTED; }
size_t dcn_plan_saved_bytes(const dcn_plan*) { return 0; }
size_t dcn_plan_workspace_bytes(const dcn_plan*) { return 0; }
double dcn_plan_forward_flops(const dcn_plan*) { return 0; }
int dcn_backbone_forward(dcn_plan*, const float*, const float* const*, float* const*, float, float, int, int, float*, void*, void*, void*) { return DCN_E_UNSUPPORTED; }
int dcn_backbone_backward(dcn_plan*, const float*, const float* const*, const void*, void*, float* const*, void*) { return DCN_E_UNSUPPORTED; }
int dcn_conv_forward(const dcn_conv_desc*, const float*, const float*, const float*, float*, float*, void*) { return DCN_E_UNSUPPORTED; }
int dcn_conv_num_mtiles(const dcn_conv_desc*) { return 0; }
int dcn_conv_dgrad(const dcn_conv_desc*, const float*, const float*, const float*, float*, void*) { return DCN_E_UNSUPPORTED; }
int dcn_conv_wgrad(const dcn_conv_desc*, const float*, const float*, float*, void*, void*) { return DCN_E_UNSUPPORTED; }
size_t dcn_conv_wgrad_workspace(const dcn_conv_desc*) { return 0; }
int dcn_transpose_weight(const float*, float*, int, int, int, void*) { return DCN_E_UNSUPPORTED; }
int dcn_upsample_forward(const float*, int, int, int, int, int, int, int, int, float*, void*) { return DCN_E_UNSUPPORTED; }
int dcn_upsample_backward(const float*, int, int, int, int, int, int, int, float*, void*) { return DCN_E_UNSUPPORTED; }
}
