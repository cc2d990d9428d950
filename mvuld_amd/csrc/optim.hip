// Optimizer-side kernels over the flat fp32 parameter / gradient buffers:
// global grad norm + clip coefficient (utils_multi.py:229-232 -> torch clip_grad_norm_(5.0)) and fused AdamW
// (optimizer.py:27-31: eps 1e-8, betas (0.9, 0.999), decoupled weight decay; no-decay group passes wd = 0),
// which also refreshes the bf16 working copy the MFMA GEMMs read.  ~28 B/param/step, HBM-bound.
#include "common.h"

__global__ __launch_bounds__(256) void sumsq_k(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    const int64_t n4 = n >> 2;
    const float4* x4 = (const float4*)x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = x4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += x[i] * x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

// The buffer holds grad_scale^-1 times the true gradient (sum over ranks: grad_scale = 1/world).
// norm_out[0] = grad_scale * sqrt(sumsq); norm_out[1] = grad_scale * min(1, max_norm / (norm + 1e-6))  (max_norm <= 0: no clip)
__global__ void clip_coef_k(const float* __restrict__ sumsq, float max_norm, float grad_scale, float* __restrict__ norm_out) {
    const float n = sqrtf(*sumsq) * grad_scale;
    norm_out[0] = n;
    norm_out[1] = grad_scale * (max_norm > 0.f ? fminf(1.0f, max_norm / (n + 1e-6f)) : 1.0f);
}

__global__ __launch_bounds__(256) void adamw_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                               bf16* __restrict__ p16, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                               float bc1, float bc2_sqrt, const float* __restrict__ coef) {
    const float c = coef ? coef[1] : 1.0f;
    const float step = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * c;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        pi -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (p16) p16[i] = (bf16)pi;
    }
}

extern "C" int mvuld_sumsq(const float* x, int64_t n, float* out, hipStream_t stream) {
    MV_CHECK_ARG(x && out && n > 0 && (((uintptr_t)x & 15) == 0), "sumsq: bad args (16-byte aligned fp32 buffer)");
    const int grid = (int)min((int64_t)2048, cdiv(n, 1024));
    hipLaunchKernelGGL(sumsq_k, dim3(grid), dim3(256), 0, stream, x, n, out);
    MV_LAUNCH_CHECK("sumsq");
    return 0;
}
extern "C" int mvuld_clip_coef(const float* sumsq, float max_norm, float grad_scale, float* norm_out, hipStream_t stream) {
    MV_CHECK_ARG(sumsq && norm_out, "clip_coef: null pointer");
    hipLaunchKernelGGL(clip_coef_k, dim3(1), dim3(1), 0, stream, sumsq, max_norm, grad_scale, norm_out);
    MV_LAUNCH_CHECK("clip_coef");
    return 0;
}
extern "C" int mvuld_adamw(float* p, const float* g, float* m, float* v, void* p16, int64_t n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int step, const float* coef, hipStream_t stream) {
    MV_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adamw: bad args");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    const int grid = (int)min((int64_t)8192, cdiv(n, 256));
    hipLaunchKernelGGL(adamw_k, dim3(grid), dim3(256), 0, stream, p, g, m, v, (bf16*)p16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, coef);
    MV_LAUNCH_CHECK("adamw");
    return 0;
}
