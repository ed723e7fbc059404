// Adam update of the training step (reference: `optimizer.step()` at dense_correspondence/training/training.py:346 on the
// torch.optim.Adam built at training.py:133-145: lr 1e-4, weight_decay 1e-4, default betas / eps).
// HBM-bound streaming kernel: per element 4 reads (p, g, m, v) + 3 writes = 28 bytes, one pass, all tensors of the model
// in ceil(n / 80) launches (the tensor table travels in the kernel arguments; stock torch needs ~10 passes over 85 MB).
#include "dcn_common.h"

namespace {

struct AdamEntry {
    float* p;
    const float* g;
    float* m;
    float* v;
    int numel;
    int first_block;
};
constexpr int kAdamBatch = 80;     // 80 x 40 bytes + header < 4 KB of kernel arguments
constexpr int kAdamChunk = 4096;   // elements per workgroup: 256 work-items x 4 float4
struct AdamTable {
    AdamEntry e[kAdamBatch];
    int n;
    float w1, beta2, w2, eps, weight_decay;   // w1 = 1 - beta1, w2 = 1 - beta2 (rounded once from double, like torch's scalars)
    float step_size;       // lr / (1 - beta1^t)
    float bc2_sqrt;        // sqrt(1 - beta2^t)
};

// the arithmetic of torch's _single_tensor_adam / _multi_tensor_adam, operation for operation:
//   g += wd * p;  m = lerp(m, g, 1 - beta1);  v = v * beta2 + (1 - beta2) * g * g;
//   p += -step_size * (m / (sqrt(v) / bc2_sqrt + eps))
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, const AdamTable& t) {
    if (t.weight_decay != 0.f) g = fmaf(t.weight_decay, p, g);
    m = fmaf(t.w1, g - m, m);
    v = fmaf(t.w2 * g, g, v * t.beta2);
    const float denom = sqrtf(v) / t.bc2_sqrt + t.eps;
    p = fmaf(-t.step_size, m / denom, p);
}

__global__ void __launch_bounds__(256)
adam_step_kernel(AdamTable t) {
    int ei = 0;
    while (ei + 1 < t.n && (int)blockIdx.x >= t.e[ei + 1].first_block) ++ei;   // wave-uniform, scalar loads
    const AdamEntry& E = t.e[ei];
    const int base = ((int)blockIdx.x - E.first_block) * kAdamChunk;
    const int end = min(E.numel, base + kAdamChunk);
    const bool vec = ((((uintptr_t)E.p | (uintptr_t)E.g | (uintptr_t)E.m | (uintptr_t)E.v) & 15) == 0);
    if (vec) {
        const int end4 = base + ((end - base) & ~3);
#pragma unroll 4
        for (int i = base + 4 * (int)threadIdx.x; i < end4; i += 4 * 256) {
            float4 p = *reinterpret_cast<const float4*>(E.p + i);
            const float4 g = *reinterpret_cast<const float4*>(E.g + i);
            float4 m = *reinterpret_cast<const float4*>(E.m + i);
            float4 v = *reinterpret_cast<const float4*>(E.v + i);
            adam_update(p.x, g.x, m.x, v.x, t);
            adam_update(p.y, g.y, m.y, v.y, t);
            adam_update(p.z, g.z, m.z, v.z, t);
            adam_update(p.w, g.w, m.w, v.w, t);
            *reinterpret_cast<float4*>(E.p + i) = p;
            *reinterpret_cast<float4*>(E.m + i) = m;
            *reinterpret_cast<float4*>(E.v + i) = v;
        }
        for (int i = end4 + (int)threadIdx.x; i < end; i += 256) adam_update(E.p[i], E.g[i], E.m[i], E.v[i], t);
    } else {
        for (int i = base + (int)threadIdx.x; i < end; i += 256) adam_update(E.p[i], E.g[i], E.m[i], E.v[i], t);
    }
}

}  // namespace

extern "C" int dcn_adam_step(int n, void* const* param, const void* const* grad, void* const* exp_avg,
                             void* const* exp_avg_sq, const int64_t* numel, double lr, double beta1, double beta2,
                             double eps, double weight_decay, int64_t step, void* stream) {
    if (n < 0 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq || !numel))) return DCN_E_INVALID;
    if (step < 1 || !(beta1 >= 0. && beta1 < 1.) || !(beta2 >= 0. && beta2 < 1.) || !(eps >= 0.) || !(lr >= 0.) || !(weight_decay >= 0.))
        return DCN_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    // bias corrections in double like the Python scalars of torch/optim/adam.py, then rounded once
    double b1t = 1.0, b2t = 1.0;
    {
        double a = beta1, b = beta2;
        for (int64_t e = step; e > 0; e >>= 1) {
            if (e & 1) { b1t *= a; b2t *= b; }
            a *= a; b *= b;
        }
    }
    for (int base = 0; base < n; base += kAdamBatch) {
        AdamTable t;
        t.n = 0;
        t.w1 = (float)(1.0 - beta1); t.beta2 = (float)beta2; t.w2 = (float)(1.0 - beta2); t.eps = (float)eps;
        t.weight_decay = (float)weight_decay;
        t.step_size = (float)(lr / (1.0 - b1t));
        t.bc2_sqrt = (float)sqrt(1.0 - b2t);
        int64_t blocks = 0;
        for (int i = base; i < n && i < base + kAdamBatch; ++i) {
            if (numel[i] < 0 || numel[i] > 0x7fffffff - kAdamChunk) return DCN_E_UNSUPPORTED;
            if (numel[i] == 0) continue;
            if (!param[i] || !grad[i] || !exp_avg[i] || !exp_avg_sq[i]) return DCN_E_INVALID;
            AdamEntry& E = t.e[t.n++];
            E.p = (float*)param[i]; E.g = (const float*)grad[i]; E.m = (float*)exp_avg[i]; E.v = (float*)exp_avg_sq[i];
            E.numel = (int)numel[i];
            E.first_block = (int)blocks;
            blocks += dcn::ceil_div64(numel[i], kAdamChunk);
            if (blocks > 0x7fffffff) return DCN_E_UNSUPPORTED;
        }
        if (t.n == 0) continue;
        hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, st, t);
        if (int rc = dcn::check_launch()) return rc;
    }
    return DCN_OK;
}
