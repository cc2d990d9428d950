// LayerNorm (+ fused residual / DropPath row scale) and BatchNorm1d kernels.
//
// LayerNorm covers: Swin res-post-norm  x = shortcut + DropPath(LN(branch))  (swin_transformer_v2.py:301,304),
// patch-embed / patch-merging / final norms (:492,:362,:632), RoBERTa embedding + post-LN blocks.
// BatchNorm1d covers the head's swinbn / bn_text / final_fc_bn ([B,C]), bn_gat / bn_bbox
// (BatchNorm1d(100) on [B,100,F]: channel = node slot, GraphModel.py:135-137,186-187) and Rs_GCN's
// W[1] (Rs_GCN.py:27-34), all through one strided kernel: element (o,c,i) at o*so + c*sc + i*si,
// statistics over (o,i).
#include "common.h"

#define LN_MAXPL 16   // elements per lane: C <= 1024

template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_k(const T* __restrict__ x, const T* __restrict__ pre,
                                                       T* __restrict__ xsum, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const T* __restrict__ res,
                                                       const float* __restrict__ rowscale, int rows_per_sample,
                                                       T* __restrict__ y, float* __restrict__ mean,
                                                       float* __restrict__ rstd, int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int npl = (C + 63) >> 6;
    for (int64_t r = (int64_t)blockIdx.x * 4 + w; r < rows; r += (int64_t)gridDim.x * 4) {
        const T* xr = x + r * C;
        float v[LN_MAXPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXPL; ++i) {
            const int c = lane + 64 * i;
            v[i] = (i < npl && c < C) ? ldf(xr + c) : 0.f;
            if (pre && i < npl && c < C) {                 // post-LN residual: LN(x + pre); keep the sum for backward
                v[i] += ldf(pre + r * C + c);
                if (xsum) { stf(xsum + r * C + c, v[i]); v[i] = ldf(xsum + r * C + c); }
            }
            s += v[i];
        }
        const float mu = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXPL; ++i) {
            const int c = lane + 64 * i;
            const float d = (i < npl && c < C) ? v[i] - mu : 0.f;
            q += d * d;
        }
        const float rs = rsqrtf(wave_sum(q) / C + eps);
        if (lane == 0) { if (mean) mean[r] = mu; if (rstd) rstd[r] = rs; }
        const float sc = rowscale ? rowscale[r / rows_per_sample] : 1.0f;
#pragma unroll
        for (int i = 0; i < LN_MAXPL; ++i) {
            const int c = lane + 64 * i;
            if (i < npl && c < C) {
                float o = ((v[i] - mu) * rs * gamma[c] + beta[c]) * sc;
                if (res) o += ldf(res + r * C + c);
                stf(y + r * C + c, o);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_k(const T* __restrict__ dy, const T* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ rowscale,
                                                       int rows_per_sample, T* __restrict__ dx, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, int64_t rows, int C) {
    __shared__ float red[2][4][64 * LN_MAXPL / 4];   // reduced in 4 passes of 256 columns to bound LDS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int npl = (C + 63) >> 6;
    float pg[LN_MAXPL], pb[LN_MAXPL];
#pragma unroll
    for (int i = 0; i < LN_MAXPL; ++i) { pg[i] = 0.f; pb[i] = 0.f; }
    for (int64_t r = (int64_t)blockIdx.x * 4 + w; r < rows; r += (int64_t)gridDim.x * 4) {
        const float mu = mean[r], rs = rstd[r];
        const float sc = rowscale ? rowscale[r / rows_per_sample] : 1.0f;
        float g[LN_MAXPL], xh[LN_MAXPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXPL; ++i) {
            const int c = lane + 64 * i;
            if (i < npl && c < C) {
                const float d = ldf(dy + r * C + c) * sc;
                xh[i] = (ldf(x + r * C + c) - mu) * rs;
                pg[i] += d * xh[i];
                pb[i] += d;
                g[i] = d * gamma[c];
                s1 += g[i];
                s2 += g[i] * xh[i];
            } else { g[i] = 0.f; xh[i] = 0.f; }
        }
        s1 = wave_sum(s1) / C;
        s2 = wave_sum(s2) / C;
#pragma unroll
        for (int i = 0; i < LN_MAXPL; ++i) {
            const int c = lane + 64 * i;
            if (i < npl && c < C) stf(dx + r * C + c, rs * (g[i] - s1 - xh[i] * s2));
        }
    }
    // cross-wave reduction of the column partials, 4 register slots (256 columns) at a time
    for (int base = 0; base < npl; base += 4) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LN_MAXPL; ++i)
            if (i >= base && i < base + 4) { red[0][w][(i - base) * 64 + lane] = pg[i]; red[1][w][(i - base) * 64 + lane] = pb[i]; }
        __syncthreads();
        const int t = threadIdx.x;                       // 256 threads <-> 256 columns of this pass
        const int c = (base + (t >> 6)) * 64 + (t & 63);
        if ((base + (t >> 6)) < npl && c < C) {
            const float a = red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t];
            const float b = red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t];
            if (dgamma) atomicAdd(dgamma + c, a);
            if (dbeta) atomicAdd(dbeta + c, b);
        }
    }
}

// y = residual + rowscale[row / rows_per_sample] * (LN(x + pre) * gamma + beta); pre/xsum/residual/rowscale optional
extern "C" int mvuld_layernorm_fwd(const void* x, const void* pre, void* xsum, const float* gamma, const float* beta, const void* residual,
                                   const float* rowscale, int rows_per_sample, void* y, float* mean, float* rstd,
                                   int64_t rows, int C, float eps, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(rows > 0 && C > 0 && C <= 64 * LN_MAXPL, "layernorm_fwd: rows=%lld C=%d unsupported (C<=1024)", (long long)rows, C);
    MV_CHECK_ARG(x && y && gamma && beta, "layernorm_fwd: null pointer");
    MV_CHECK_ARG(!rowscale || rows_per_sample > 0, "layernorm_fwd: rows_per_sample");
    const int grid = (int)min((int64_t)2048, cdiv(rows, 4));
    if (dtype == MVULD_F32)
        hipLaunchKernelGGL(layernorm_fwd_k<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, (const float*)pre, (float*)xsum, gamma, beta,
                           (const float*)residual, rowscale, rows_per_sample, (float*)y, mean, rstd, rows, C, eps);
    else
        hipLaunchKernelGGL(layernorm_fwd_k<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)x, (const bf16*)pre, (bf16*)xsum, gamma, beta,
                           (const bf16*)residual, rowscale, rows_per_sample, (bf16*)y, mean, rstd, rows, C, eps);
    MV_LAUNCH_CHECK("layernorm_fwd");
    return 0;
}

extern "C" int mvuld_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                   const float* rstd, const float* rowscale, int rows_per_sample, void* dx,
                                   float* dgamma, float* dbeta, int64_t rows, int C, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(rows > 0 && C > 0 && C <= 64 * LN_MAXPL, "layernorm_bwd: rows=%lld C=%d unsupported", (long long)rows, C);
    MV_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "layernorm_bwd: null pointer");
    const int grid = (int)min((int64_t)1024, cdiv(rows, 4));
    if (dtype == MVULD_F32)
        hipLaunchKernelGGL(layernorm_bwd_k<float>, dim3(grid), dim3(256), 0, stream, (const float*)dy, (const float*)x,
                           gamma, mean, rstd, rowscale, rows_per_sample, (float*)dx, dgamma, dbeta, rows, C);
    else
        hipLaunchKernelGGL(layernorm_bwd_k<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x,
                           gamma, mean, rstd, rowscale, rows_per_sample, (bf16*)dx, dgamma, dbeta, rows, C);
    MV_LAUNCH_CHECK("layernorm_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------ BatchNorm1d
// One block per channel.  training: batch mean / biased var normalise, running stats get the
// unbiased var with `momentum`; eval: running stats.  save_mean/save_rstd feed the backward.
template <typename T>
__global__ __launch_bounds__(256) void batchnorm_fwd_k(const T* __restrict__ x, T* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ run_mean, float* __restrict__ run_var,
                                                       float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                       int O, int C, int I, int64_t so, int64_t sc, int64_t si,
                                                       float eps, float momentum, int training) {
    __shared__ float red[16];
    const int c = blockIdx.x;
    const int64_t n = (int64_t)O * I;
    float mu, rs;
    if (training) {
        float s = 0.f;
        for (int64_t e = threadIdx.x; e < n; e += blockDim.x) s += ldf(x + (e / I) * so + c * sc + (e % I) * si);
        mu = block_sum(s, red) / n;
        float q = 0.f;
        for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
            const float d = ldf(x + (e / I) * so + c * sc + (e % I) * si) - mu;
            q += d * d;
        }
        const float var = block_sum(q, red) / n;
        rs = rsqrtf(var + eps);
        if (threadIdx.x == 0) {
            if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
            if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * (n > 1 ? var * n / (n - 1) : var);
        }
    } else {
        mu = run_mean[c];
        rs = rsqrtf(run_var[c] + eps);
    }
    if (threadIdx.x == 0) { if (save_mean) save_mean[c] = mu; if (save_rstd) save_rstd[c] = rs; }
    const float ga = gamma[c], be = beta[c];
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
        const int64_t off = (e / I) * so + c * sc + (e % I) * si;
        stf(y + off, (ldf(x + off) - mu) * rs * ga + be);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void batchnorm_bwd_k(const T* __restrict__ dy, const T* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                       const float* __restrict__ save_rstd, T* __restrict__ dx,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int O, int C,
                                                       int I, int64_t so, int64_t sc, int64_t si, int training) {
    __shared__ float red[16];
    const int c = blockIdx.x;
    const int64_t n = (int64_t)O * I;
    const float mu = save_mean[c], rs = save_rstd[c], ga = gamma[c];
    float s1 = 0.f, s2 = 0.f;
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
        const int64_t off = (e / I) * so + c * sc + (e % I) * si;
        const float d = ldf(dy + off);
        s1 += d;
        s2 += d * (ldf(x + off) - mu) * rs;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { if (dgamma) atomicAdd(dgamma + c, s2); if (dbeta) atomicAdd(dbeta + c, s1); }
    const float m1 = training ? s1 / n : 0.f, m2 = training ? s2 / n : 0.f;
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
        const int64_t off = (e / I) * so + c * sc + (e % I) * si;
        const float xh = (ldf(x + off) - mu) * rs;
        stf(dx + off, ga * rs * (ldf(dy + off) - m1 - xh * m2));
    }
}

extern "C" int mvuld_batchnorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* run_mean,
                                   float* run_var, float* save_mean, float* save_rstd, int O, int C, int I, int64_t so,
                                   int64_t sc, int64_t si, float eps, float momentum, int training, int dtype,
                                   hipStream_t stream) {
    MV_CHECK_ARG(O > 0 && C > 0 && I > 0, "batchnorm_fwd: empty O=%d C=%d I=%d", O, C, I);
    MV_CHECK_ARG(x && y && gamma && beta && (training || (run_mean && run_var)), "batchnorm_fwd: null pointer");
    if (dtype == MVULD_F32)
        hipLaunchKernelGGL(batchnorm_fwd_k<float>, dim3(C), dim3(256), 0, stream, (const float*)x, (float*)y, gamma, beta,
                           run_mean, run_var, save_mean, save_rstd, O, C, I, so, sc, si, eps, momentum, training);
    else
        hipLaunchKernelGGL(batchnorm_fwd_k<bf16>, dim3(C), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, gamma, beta,
                           run_mean, run_var, save_mean, save_rstd, O, C, I, so, sc, si, eps, momentum, training);
    MV_LAUNCH_CHECK("batchnorm_fwd");
    return 0;
}

extern "C" int mvuld_batchnorm_bwd(const void* dy, const void* x, const float* gamma, const float* save_mean,
                                   const float* save_rstd, void* dx, float* dgamma, float* dbeta, int O, int C, int I,
                                   int64_t so, int64_t sc, int64_t si, int training, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(O > 0 && C > 0 && I > 0, "batchnorm_bwd: empty");
    MV_CHECK_ARG(dy && x && gamma && save_mean && save_rstd && dx, "batchnorm_bwd: null pointer");
    if (dtype == MVULD_F32)
        hipLaunchKernelGGL(batchnorm_bwd_k<float>, dim3(C), dim3(256), 0, stream, (const float*)dy, (const float*)x, gamma,
                           save_mean, save_rstd, (float*)dx, dgamma, dbeta, O, C, I, so, sc, si, training);
    else
        hipLaunchKernelGGL(batchnorm_bwd_k<bf16>, dim3(C), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, gamma,
                           save_mean, save_rstd, (bf16*)dx, dgamma, dbeta, O, C, I, so, sc, si, training);
    MV_LAUNCH_CHECK("batchnorm_bwd");
    return 0;
}
