// Optimizer-side kernels over the flat fp32 parameter / gradient buffers:
// global grad norm + clip coefficient (utils_multi.py:229-232 -> torch clip_grad_norm_(5.0)) and fused AdamW
// (optimizer.py:27-31: eps 1e-8, betas (0.9, 0.999), decoupled weight decay; no-decay group passes wd = 0),
// which also refreshes the bf16 working copy the MFMA GEMMs read.  ~28 B/param/step, HBM-bound.
#include "common.h"

// Deterministic: every rank must derive the SAME clip coefficient from the same all-reduced gradient, or the replicas drift
// apart (nothing re-synchronises parameters in data-parallel training).  So no float atomics here: fixed grid, fixed per-thread
// order, one partial per block, and a single block folds the partials in a fixed tree.
#define SUMSQ_MAX_BLOCKS 2048
__global__ __launch_bounds__(256) void sumsq_k(const float* __restrict__ x, int64_t n, float* __restrict__ partials) {
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
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_k(const float* __restrict__ partials, int nblk, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partials[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] += s;
}

// The buffer holds grad_scale^-1 times the true gradient (sum over ranks: grad_scale = 1/world).
// norm_out[0] = grad_scale * sqrt(sumsq); norm_out[1] = grad_scale * min(1, max_norm / (norm + 1e-6))  (max_norm <= 0: no clip)
__global__ void clip_coef_k(const float* __restrict__ sumsq, float max_norm, float grad_scale, float* __restrict__ norm_out) {
    const float n = sqrtf(*sumsq) * grad_scale;
    norm_out[0] = n;
    norm_out[1] = grad_scale * (max_norm > 0.f ? fminf(1.0f, max_norm / (n + 1e-6f)) : 1.0f);
}

// 16-byte accesses (the flat store pads every tensor to 64 elements); zero_grad != 0 also clears the gradient it has just consumed,
// which replaces the separate zero_grad() fill pass over the 0.9 GB buffer.
__global__ __launch_bounds__(256) void adamw_k(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                               bf16* __restrict__ p16, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                               float bc1, float bc2_sqrt, const float* __restrict__ coef, int zero_grad,
                                               const float* __restrict__ dev_hyper) {
    if (dev_hyper) { lr = dev_hyper[0]; bc1 = dev_hyper[1]; bc2_sqrt = dev_hyper[2]; }      // per-step values from device memory (captured step)
    const float c = coef ? coef[1] : 1.0f;
    const float step = lr / bc1;
    const float decay = 1.0f - lr * wd;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 g4 = ((const float4*)g)[i];
        float4 p4 = ((const float4*)p)[i], m4 = ((const float4*)m)[i], v4 = ((const float4*)v)[i];
        float gg[4] = {g4.x * c, g4.y * c, g4.z * c, g4.w * c};
        float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mm[e] = b1 * mm[e] + (1.0f - b1) * gg[e];
            vv[e] = b2 * vv[e] + (1.0f - b2) * gg[e] * gg[e];
            pp[e] = pp[e] * decay - step * mm[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
        }
        ((float4*)p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        ((float4*)m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        ((float4*)v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (zero_grad) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p16) {
            union { uint2 u; bf16 e[4]; } o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.e[e] = (bf16)pp[e];
            ((uint2*)p16)[i] = o.u;
        }
    }
    // tail (n not a multiple of 4)
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * c;
        float pi = p[i] * decay;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        pi -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (zero_grad) g[i] = 0.f;
        if (p16) p16[i] = (bf16)pi;
    }
}

extern "C" int mvuld_sumsq(const float* x, int64_t n, float* partials, float* out, hipStream_t stream) {
    MV_CHECK_ARG(x && partials && out && n > 0 && (((uintptr_t)x & 15) == 0), "sumsq: bad args (16-byte aligned fp32 buffer, 2048-float scratch)");
    const int grid = (int)min((int64_t)SUMSQ_MAX_BLOCKS, cdiv(n, 1024));
    hipLaunchKernelGGL(sumsq_k, dim3(grid), dim3(256), 0, stream, x, n, partials);
    hipLaunchKernelGGL(sumsq_final_k, dim3(1), dim3(256), 0, stream, partials, grid, out);
    MV_LAUNCH_CHECK("sumsq");
    return 0;
}
extern "C" int mvuld_clip_coef(const float* sumsq, float max_norm, float grad_scale, float* norm_out, hipStream_t stream) {
    MV_CHECK_ARG(sumsq && norm_out, "clip_coef: null pointer");
    hipLaunchKernelGGL(clip_coef_k, dim3(1), dim3(1), 0, stream, sumsq, max_norm, grad_scale, norm_out);
    MV_LAUNCH_CHECK("clip_coef");
    return 0;
}
extern "C" int mvuld_adamw(float* p, float* g, float* m, float* v, void* p16, int64_t n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int step, const float* coef, int zero_grad, const float* dev_hyper, hipStream_t stream) {
    MV_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adamw: bad args");
    MV_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && (((uintptr_t)p16) & 7) == 0, "adamw: buffers must be 16-byte aligned");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    const int grid = (int)min((int64_t)8192, cdiv(n, 1024));
    hipLaunchKernelGGL(adamw_k, dim3(grid), dim3(256), 0, stream, p, g, m, v, (bf16*)p16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, coef, zero_grad, dev_hyper);
    MV_LAUNCH_CHECK("adamw");
    return 0;
}
