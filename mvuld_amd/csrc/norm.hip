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

// ---- bf16 fast path: 16-byte accesses.  A row of C elements is C/8 chunks; G = min(64, pow2ceil(C/8)) lanes share a row
// (CPL = ceil(C/8/G) <= 2 chunks per lane), 64/G rows per wave, statistics by xor-shuffles inside the lane group.
// Every load is unconditional on a clamped address (a predicated load makes the compiler branch and drain vmcnt per load,
// which serialises the x / pre / residual round trips); only the stores are predicated.
// Sum over the G (power of two) consecutive lanes that share a row: DPP inside a 16-lane row, v_permlane16/32_swap across rows --
// VALU only (a __shfl_xor butterfly is six dependent LDS-routed ds_bpermute per reduction).
template <int CTRL>
__device__ __forceinline__ float ln_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float group_sum(float v, int G) {
    if (G >= 2) v += ln_dpp<0xB1>(v);        // quad_perm [1,0,3,2]
    if (G >= 4) v += ln_dpp<0x4E>(v);        // quad_perm [2,3,0,1]
    if (G >= 8) v += ln_dpp<0x141>(v);       // row_half_mirror: the other quad of the 8
    if (G >= 16) v += ln_dpp<0x140>(v);      // row_mirror: the other half of the row
    if (G >= 32) {
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    }
    if (G >= 64) {
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    }
    return v;
}

// DROP (with HAS_PRE): x is first put through mvuld_dropout's mask -- bf16(keep ? x / (1 - p) : 0), element counter row * C + column,
// the same bits the separate kernel would have written -- so RoBERTa's LayerNorm(dropout(dense) + input) is one pass over the rows
template <int CPL, bool HAS_PRE, bool HAS_RES, bool DROP = false>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_k(const bf16* __restrict__ x, const bf16* __restrict__ pre, bf16* __restrict__ xsum,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const bf16* __restrict__ res, const float* __restrict__ rowscale,
                                                           int rows_per_sample, bf16* __restrict__ y, float* __restrict__ mean,
                                                           float* __restrict__ rstd, int64_t rows, int C, int G, float eps,
                                                           uint2* __restrict__ q8, float* __restrict__ qstate, float drop_p = 0.f,
                                                           uint64_t drop_seed = 0, const uint64_t* __restrict__ drop_off = nullptr) {
    if (DROP && drop_off) drop_seed += drop_off[0] * 0xD1B54A32D192ED03ULL;
    const float drop_inv = DROP ? 1.0f / (1.0f - drop_p) : 1.0f;
    const uint32_t drop_thr = DROP ? (uint32_t)(drop_p * 4294967296.0) : 0u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int sub = lane & (G - 1), rg = lane / G, rpw = 64 / G;
    const int nch = C >> 3;
    const float qinv = q8 ? 1.0f / qstate[0] : 0.f;      // e4m3 copy of y for the next fp8 product (scale of the previous step)
    float amax_l = 0.f;
    float ga[CPL][8], be[CPL][8];
    int chc[CPL];
    bool chok[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int ch = sub + G * i;
        chok[i] = ch < nch;
        chc[i] = chok[i] ? ch : nch - 1;
        const float4 g0 = *(const float4*)(gamma + chc[i] * 8), g1 = *(const float4*)(gamma + chc[i] * 8 + 4);
        const float4 b0 = *(const float4*)(beta + chc[i] * 8), b1 = *(const float4*)(beta + chc[i] * 8 + 4);
        ga[i][0] = g0.x; ga[i][1] = g0.y; ga[i][2] = g0.z; ga[i][3] = g0.w; ga[i][4] = g1.x; ga[i][5] = g1.y; ga[i][6] = g1.z; ga[i][7] = g1.w;
        be[i][0] = b0.x; be[i][1] = b0.y; be[i][2] = b0.z; be[i][3] = b0.w; be[i][4] = b1.x; be[i][5] = b1.y; be[i][6] = b1.z; be[i][7] = b1.w;
    }
    const int64_t rstep = (int64_t)gridDim.x * 4 * rpw;
    for (int64_t r0 = ((int64_t)blockIdx.x * 4 + w) * rpw; r0 < rows; r0 += rstep) {
        const int64_t r = r0 + rg;
        const bool rok = r < rows;
        const int64_t rc = rok ? r : rows - 1;
        bf16x8 a[CPL], p[CPL], rr[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            a[i] = *(const bf16x8*)(x + rc * C + chc[i] * 8);
            if (HAS_PRE) p[i] = *(const bf16x8*)(pre + rc * C + chc[i] * 8);
            if (HAS_RES) rr[i] = *(const bf16x8*)(res + rc * C + chc[i] * 8);
        }
        float sc = 1.0f;
        if (rowscale) sc = rowscale[rc / rows_per_sample];
        float v[CPL][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = (float)a[i].v[e];
                if (DROP) {
                    const bool keep = mix32(drop_seed + (uint64_t)(rc * C + chc[i] * 8 + e) * 0x9E3779B97F4A7C15ULL) >= drop_thr;
                    t = (float)(bf16)(keep ? t * drop_inv : 0.f);
                }
                if (HAS_PRE) { t += (float)p[i].v[e]; a[i].v[e] = (bf16)t; t = (float)a[i].v[e]; }
                t = chok[i] ? t : 0.f;
                v[i][e] = t;
                s += t;
            }
            if (HAS_PRE && xsum && rok && chok[i]) *(bf16x8*)(xsum + r * C + chc[i] * 8) = a[i];
        }
        s = group_sum(s, G);
        const float mu = s / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = chok[i] ? v[i][e] - mu : 0.f; q += d * d; }
        q = group_sum(q, G);
        const float rs = rsqrtf(q / C + eps);
        if (rok && sub == 0) { if (mean) mean[r] = mu; if (rstd) rstd[r] = rs; }
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = ((v[i][e] - mu) * rs * ga[i][e] + be[i][e]) * sc;
                if (HAS_RES) t += (float)rr[i].v[e];
                o.v[e] = (bf16)t;
            }
            if (rok && chok[i]) *(bf16x8*)(y + r * C + chc[i] * 8) = o;
            if (q8) {
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (float)o.v[e];
                if (rok && chok[i]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) amax_l = fmaxf(amax_l, fabsf(f[e]));
                }
                unsigned lo = 0, hi = 0;
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(q_clamp(f[0] * qinv), q_clamp(f[1] * qinv), lo, false);
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(q_clamp(f[2] * qinv), q_clamp(f[3] * qinv), lo, true);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(q_clamp(f[4] * qinv), q_clamp(f[5] * qinv), hi, false);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(q_clamp(f[6] * qinv), q_clamp(f[7] * qinv), hi, true);
                if (rok && chok[i]) q8[(r * C >> 3) + chc[i]] = make_uint2(lo, hi);
            }
        }
    }
    if (q8) {
        // one atomic per workgroup, and only when it would raise the maximum (thousands of same-address atomics serialise in L2)
        __shared__ float wmax[4];
        const float mw = wave_max(amax_l);
        if (lane == 0) wmax[w] = mw;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (m > __hip_atomic_load(qstate + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax((unsigned*)(qstate + 1), __float_as_uint(m));
        }
    }
}

// DROP: also writes dxd = dropout(dx, p, seed) -- mvuld_dropout's mask and bits on the bf16-rounded dx -- so that the text encoder's
// backward hidden dropouts (d(dense output) = mask o d(dropout(dense) + input) / (1 - p)) cost one more store here instead of a pass
template <int CPL, bool DROP = false>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_k(const bf16* __restrict__ dy, const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ rowscale, int rows_per_sample, bf16* __restrict__ dx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int C, int G, float* __restrict__ ws,
                                                           bf16* __restrict__ dxd = nullptr, float drop_p = 0.f, uint64_t drop_seed = 0,
                                                           const uint64_t* __restrict__ drop_off = nullptr) {
    __shared__ float accg[1024], accb[1024];
    if (DROP && drop_off) drop_seed += drop_off[0] * 0xD1B54A32D192ED03ULL;
    const float drop_inv = DROP ? 1.0f / (1.0f - drop_p) : 1.0f;
    const uint32_t drop_thr = DROP ? (uint32_t)(drop_p * 4294967296.0) : 0u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int sub = lane & (G - 1), rg = lane / G, rpw = 64 / G;
    const int nch = C >> 3;
    for (int i = threadIdx.x; i < C; i += 256) { accg[i] = 0.f; accb[i] = 0.f; }
    float ga[CPL][8], pg[CPL][8], pb[CPL][8];
    int chc[CPL];
    bool chok[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int ch = sub + G * i;
        chok[i] = ch < nch;
        chc[i] = chok[i] ? ch : nch - 1;
        const float4 g0 = *(const float4*)(gamma + chc[i] * 8), g1 = *(const float4*)(gamma + chc[i] * 8 + 4);
        ga[i][0] = g0.x; ga[i][1] = g0.y; ga[i][2] = g0.z; ga[i][3] = g0.w; ga[i][4] = g1.x; ga[i][5] = g1.y; ga[i][6] = g1.z; ga[i][7] = g1.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) { pg[i][e] = 0.f; pb[i][e] = 0.f; }
    }
    const int64_t rstep = (int64_t)gridDim.x * 4 * rpw;
    for (int64_t r0 = ((int64_t)blockIdx.x * 4 + w) * rpw; r0 < rows; r0 += rstep) {
        const int64_t r = r0 + rg;
        const bool rok = r < rows;
        const int64_t rc = rok ? r : rows - 1;
        bf16x8 a[CPL], b[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            a[i] = *(const bf16x8*)(dy + rc * C + chc[i] * 8);
            b[i] = *(const bf16x8*)(x + rc * C + chc[i] * 8);
        }
        const float mu = mean[rc], rs = rstd[rc];
        float sc = 1.0f;
        if (rowscale) sc = rowscale[rc / rows_per_sample];
        float g[CPL][8], xh[CPL][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const bool ok = rok && chok[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = ok ? (float)a[i].v[e] * sc : 0.f;
                xh[i][e] = ok ? ((float)b[i].v[e] - mu) * rs : 0.f;
                pg[i][e] += d * xh[i][e];
                pb[i][e] += d;
                g[i][e] = d * ga[i][e];
                s1 += g[i][e];
                s2 += g[i][e] * xh[i][e];
            }
        }
        s1 = group_sum(s1, G) / C;
        s2 = group_sum(s2, G) / C;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.v[e] = (bf16)(rs * (g[i][e] - s1 - xh[i][e] * s2));
            if (rok && chok[i]) *(bf16x8*)(dx + r * C + chc[i] * 8) = o;
            if (DROP) {
                bf16x8 od;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool keep = mix32(drop_seed + (uint64_t)(rc * C + chc[i] * 8 + e) * 0x9E3779B97F4A7C15ULL) >= drop_thr;
                    od.v[e] = (bf16)(keep ? (float)o.v[e] * drop_inv : 0.f);
                }
                if (rok && chok[i]) *(bf16x8*)(dxd + r * C + chc[i] * 8) = od;
            }
        }
    }
    // column partials: fold the row groups of a wave by shuffles, then the waves take turns adding into the LDS accumulators
    // (LDS float atomics run at about one lane per clock per CU)
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const float tg = __shfl_xor(pg[i][e], o, 64), tb = __shfl_xor(pb[i][e], o, 64);
                pg[i][e] += (o >= G) ? tg : 0.f;
                pb[i][e] += (o >= G) ? tb : 0.f;
            }
        }
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (w == wv && rg == 0) {
#pragma unroll
            for (int i = 0; i < CPL; ++i)
                if (chok[i]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { accg[chc[i] * 8 + e] += pg[i][e]; accb[chc[i] * 8 + e] += pb[i][e]; }
                }
        }
    }
    __syncthreads();
    if (ws) {                                       // per-block partials, summed by layernorm_bwd_reduce_k
        float* o = ws + (size_t)blockIdx.x * 2 * C;
        for (int i = threadIdx.x; i < C; i += 256) { o[i] = accg[i]; o[C + i] = accb[i]; }
        return;
    }
    for (int i = threadIdx.x; i < C; i += 256) {
        if (dgamma) atomicAdd(dgamma + i, accg[i]);
        if (dbeta) atomicAdd(dbeta + i, accb[i]);
    }
}

// dgamma[c] += sum_b ws[b][c], dbeta[c] += sum_b ws[b][C + c]: 64 columns x 16 row groups per block, blockIdx.y = row slab
__global__ __launch_bounds__(1024) void layernorm_bwd_reduce_k(const float* __restrict__ ws, int nblk, int C, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int cc = min(c, 2 * C - 1);
    const int per = (nblk + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = b0 + g; b < b1; b += 128) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int bb = b + 16 * u;
            const float v = ws[(size_t)min(bb, nblk - 1) * 2 * C + cc];
            a[u] += bb < b1 ? v : 0.f;
        }
    }
    red[g][threadIdx.x & 63] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (g == 0 && c < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        if (c < C) { if (dgamma) atomicAdd(dgamma + c, t); }
        else if (dbeta) atomicAdd(dbeta + c - C, t);
    }
}

static inline int ln_group(int C) {          // lanes per row
    int g = 1;
    while (g < 64 && g * 8 < C) g <<= 1;
    return g;
}

// y = residual + rowscale[row / rows_per_sample] * (LN(x + pre) * gamma + beta); pre/xsum/residual/rowscale optional
static int layernorm_fwd_impl(const void* x, const void* pre, void* xsum, const float* gamma, const float* beta, const void* residual,
                              const float* rowscale, int rows_per_sample, void* y, float* mean, float* rstd,
                              int64_t rows, int C, float eps, int dtype, void* q8, float* qstate, hipStream_t stream) {
    MV_CHECK_ARG(rows > 0 && C > 0 && C <= 64 * LN_MAXPL, "layernorm_fwd: rows=%lld C=%d unsupported (C<=1024)", (long long)rows, C);
    MV_CHECK_ARG(x && y && gamma && beta, "layernorm_fwd: null pointer");
    MV_CHECK_ARG(!rowscale || rows_per_sample > 0, "layernorm_fwd: rows_per_sample");
    if (dtype == MVULD_BF16 && C % 8 == 0 && rows > 0 &&
        (((uintptr_t)x | (uintptr_t)y | (uintptr_t)pre | (uintptr_t)xsum | (uintptr_t)residual | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0) {
        const int G = ln_group(C), rpw = 64 / G;
        const int gridv = (int)min((int64_t)4096, cdiv(rows, (int64_t)4 * rpw));
#define LN_FWD_VEC(CPL, P, R)                                                                                                          \
    hipLaunchKernelGGL((layernorm_fwd_vec_k<CPL, P, R>), dim3(gridv), dim3(256), 0, stream, (const bf16*)x, (const bf16*)pre, (bf16*)xsum, \
                       gamma, beta, (const bf16*)residual, rowscale, rows_per_sample, (bf16*)y, mean, rstd, rows, C, G, eps, (uint2*)q8, qstate)
        const int sel = (C / 8 <= G ? 0 : 4) | (pre ? 2 : 0) | (residual ? 1 : 0);
        switch (sel) {
            case 0: LN_FWD_VEC(1, false, false); break;
            case 1: LN_FWD_VEC(1, false, true); break;
            case 2: LN_FWD_VEC(1, true, false); break;
            case 3: LN_FWD_VEC(1, true, true); break;
            case 4: LN_FWD_VEC(2, false, false); break;
            case 5: LN_FWD_VEC(2, false, true); break;
            case 6: LN_FWD_VEC(2, true, false); break;
            default: LN_FWD_VEC(2, true, true); break;
        }
#undef LN_FWD_VEC
        MV_LAUNCH_CHECK("layernorm_fwd_vec");
        return 0;
    }
    MV_CHECK_ARG(!q8, "layernorm_fwd_q8: bf16, C %% 8 == 0 and 16-byte aligned tensors only");
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

extern "C" int mvuld_layernorm_fwd(const void* x, const void* pre, void* xsum, const float* gamma, const float* beta, const void* residual,
                                   const float* rowscale, int rows_per_sample, void* y, float* mean, float* rstd,
                                   int64_t rows, int C, float eps, int dtype, hipStream_t stream) {
    return layernorm_fwd_impl(x, pre, xsum, gamma, beta, residual, rowscale, rows_per_sample, y, mean, rstd, rows, C, eps, dtype, nullptr, nullptr, stream);
}
// ... and the e4m3 copy of y for the next fp8 product, under the scale in q_state[0]; max|y| folded into q_state[1]
extern "C" int mvuld_layernorm_fwd_q8(const void* x, const void* pre, void* xsum, const float* gamma, const float* beta, const void* residual,
                                      const float* rowscale, int rows_per_sample, void* y, float* mean, float* rstd,
                                      int64_t rows, int C, float eps, void* q_out, float* q_state, hipStream_t stream) {
    MV_CHECK_ARG(q_out && q_state && (((uintptr_t)q_out) & 7) == 0, "layernorm_fwd_q8: null / misaligned e4m3 output");
    return layernorm_fwd_impl(x, pre, xsum, gamma, beta, residual, rowscale, rows_per_sample, y, mean, rstd, rows, C, eps, MVULD_BF16, q_out, q_state, stream);
}

// LayerNorm(dropout(x) + pre): mvuld_dropout(x, p, seed) fused into the post-LN of the text encoder (bf16, C % 8 == 0, 16-byte aligned);
// bit-identical to the two separate launches
extern "C" int mvuld_layernorm_fwd_drop(const void* x, const void* pre, void* xsum, const float* gamma, const float* beta, void* y, float* mean,
                                        float* rstd, int64_t rows, int C, float eps, float drop_p, uint64_t drop_seed, const uint64_t* seed_offset,
                                        hipStream_t stream) {
    MV_CHECK_ARG(x && pre && gamma && beta && y && rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * LN_MAXPL, "layernorm_fwd_drop: bad args (rows=%lld C=%d)", (long long)rows, C);
    MV_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "layernorm_fwd_drop: 0 <= p < 1");
    MV_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)pre | (uintptr_t)xsum | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "layernorm_fwd_drop: 16-byte aligned tensors only");
    const int G = ln_group(C), rpw = 64 / G;
    const int gridv = (int)min((int64_t)4096, cdiv(rows, (int64_t)4 * rpw));
    if (C / 8 <= G)
        hipLaunchKernelGGL((layernorm_fwd_vec_k<1, true, false, true>), dim3(gridv), dim3(256), 0, stream, (const bf16*)x, (const bf16*)pre, (bf16*)xsum,
                           gamma, beta, (const bf16*)nullptr, (const float*)nullptr, 1, (bf16*)y, mean, rstd, rows, C, G, eps, (uint2*)nullptr, (float*)nullptr,
                           drop_p, drop_seed, seed_offset);
    else
        hipLaunchKernelGGL((layernorm_fwd_vec_k<2, true, false, true>), dim3(gridv), dim3(256), 0, stream, (const bf16*)x, (const bf16*)pre, (bf16*)xsum,
                           gamma, beta, (const bf16*)nullptr, (const float*)nullptr, 1, (bf16*)y, mean, rstd, rows, C, G, eps, (uint2*)nullptr, (float*)nullptr,
                           drop_p, drop_seed, seed_offset);
    MV_LAUNCH_CHECK("layernorm_fwd_drop");
    return 0;
}

extern "C" int mvuld_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                   const float* rstd, const float* rowscale, int rows_per_sample, void* dx,
                                   float* dgamma, float* dbeta, int64_t rows, int C, float* ws, int64_t ws_bytes, int dtype,
                                   hipStream_t stream) {
    MV_CHECK_ARG(rows > 0 && C > 0 && C <= 64 * LN_MAXPL, "layernorm_bwd: rows=%lld C=%d unsupported", (long long)rows, C);
    MV_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "layernorm_bwd: null pointer");
    if (dtype == MVULD_BF16 && C % 8 == 0 && rows > 0 && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)gamma) & 15) == 0) {
        const int G = ln_group(C), rpw = 64 / G;
        static const int capv = getenv("MVULD_LN_BWD_GRID") ? atoi(getenv("MVULD_LN_BWD_GRID")) : 1024;
        int gridv = (int)min((int64_t)capv, cdiv(rows, (int64_t)4 * rpw));
        float* part = nullptr;
        if (ws && ws_bytes >= (int64_t)8 * C * 64) {                 // two-pass column sums when the caller lends a workspace
            gridv = (int)min((int64_t)gridv, ws_bytes / ((int64_t)8 * C));
            part = ws;
        }
        if (C / 8 <= G)
            hipLaunchKernelGGL(layernorm_bwd_vec_k<1>, dim3(gridv), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, rowscale,
                               rows_per_sample, (bf16*)dx, dgamma, dbeta, rows, C, G, part);
        else
            hipLaunchKernelGGL(layernorm_bwd_vec_k<2>, dim3(gridv), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, rowscale,
                               rows_per_sample, (bf16*)dx, dgamma, dbeta, rows, C, G, part);
        if (part && (dgamma || dbeta))
            hipLaunchKernelGGL(layernorm_bwd_reduce_k, dim3(cdiv(2 * C, 64), 8), dim3(1024), 0, stream, part, gridv, C, dgamma, dbeta);
        MV_LAUNCH_CHECK("layernorm_bwd_vec");
        return 0;
    }
    // The partials-only form (dgamma == dbeta == NULL: the caller reduces `ws` itself, mvuld_layernorm_bwd_reduce_batch) exists on the vector
    // path above only: a caller that sized its deferral with mvuld_layernorm_bwd_nparts but handed misaligned operands would otherwise have
    // an uninitialised workspace reduced into its parameter gradients
    MV_CHECK_ARG(dgamma || dbeta, "layernorm_bwd: the partials-only form (dgamma = dbeta = NULL) needs bf16 operands, C %% 8 == 0 and 16-byte aligned dy / x / dx / gamma");
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

// mvuld_layernorm_bwd (no row scale) that also emits dxd = mvuld_dropout(dx, p, seed): bf16, C % 8 == 0, 16-byte aligned; bit-identical to
// the two launches.  The backward of RobertaSelfOutput / RobertaOutput: LayerNorm(dropout(dense(h)) + input).
extern "C" int mvuld_layernorm_bwd_drop(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, void* dxd,
                                        float* dgamma, float* dbeta, int64_t rows, int C, float* ws, int64_t ws_bytes, float drop_p, uint64_t drop_seed,
                                        const uint64_t* seed_offset, hipStream_t stream) {
    MV_CHECK_ARG(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * LN_MAXPL, "layernorm_bwd_drop: rows=%lld C=%d unsupported", (long long)rows, C);
    MV_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dxd && drop_p >= 0.f && drop_p < 1.f, "layernorm_bwd_drop: bad args");
    MV_CHECK_ARG((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dxd | (uintptr_t)gamma) & 15) == 0, "layernorm_bwd_drop: 16-byte aligned tensors only");
    const int G = ln_group(C), rpw = 64 / G;
    static const int capv = getenv("MVULD_LN_BWD_GRID") ? atoi(getenv("MVULD_LN_BWD_GRID")) : 1024;
    int gridv = (int)min((int64_t)capv, cdiv(rows, (int64_t)4 * rpw));
    float* part = nullptr;
    if (ws && ws_bytes >= (int64_t)8 * C * 64) {
        gridv = (int)min((int64_t)gridv, ws_bytes / ((int64_t)8 * C));
        part = ws;
    }
    if (C / 8 <= G)
        hipLaunchKernelGGL((layernorm_bwd_vec_k<1, true>), dim3(gridv), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd,
                           (const float*)nullptr, 1, (bf16*)dx, dgamma, dbeta, rows, C, G, part, (bf16*)dxd, drop_p, drop_seed, seed_offset);
    else
        hipLaunchKernelGGL((layernorm_bwd_vec_k<2, true>), dim3(gridv), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd,
                           (const float*)nullptr, 1, (bf16*)dx, dgamma, dbeta, rows, C, G, part, (bf16*)dxd, drop_p, drop_seed, seed_offset);
    if (part && (dgamma || dbeta))
        hipLaunchKernelGGL(layernorm_bwd_reduce_k, dim3(cdiv(2 * C, 64), 8), dim3(1024), 0, stream, part, gridv, C, dgamma, dbeta);
    MV_LAUNCH_CHECK("layernorm_bwd_drop");
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
                                                       int I, int64_t so, int64_t sc, int64_t si, int training,
                                                       float* __restrict__ dxsum) {
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
    float sx = 0.f;
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
        const int64_t off = (e / I) * so + c * sc + (e % I) * si;
        const float xh = (ldf(x + off) - mu) * rs;
        const float d = ga * rs * (ldf(dy + off) - m1 - xh * m2);
        sx += d;
        stf(dx + off, d);
    }
    if (dxsum) {                                   // per-channel sum of dx in fp32, before dx is rounded to its storage type: the
        sx = block_sum(sx, red);                   // bias gradient of the layer in front (~0 under batch statistics: a sum of
        if (threadIdx.x == 0) atomicAdd(dxsum + c, sx);      // cancelling terms that bf16 rounding would turn into noise)
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
                                   int64_t so, int64_t sc, int64_t si, int training, float* dxsum, int dtype,
                                   hipStream_t stream) {
    MV_CHECK_ARG(O > 0 && C > 0 && I > 0, "batchnorm_bwd: empty");
    MV_CHECK_ARG(dy && x && gamma && save_mean && save_rstd && dx, "batchnorm_bwd: null pointer");
    if (dtype == MVULD_F32)
        hipLaunchKernelGGL(batchnorm_bwd_k<float>, dim3(C), dim3(256), 0, stream, (const float*)dy, (const float*)x, gamma,
                           save_mean, save_rstd, (float*)dx, dgamma, dbeta, O, C, I, so, sc, si, training, dxsum);
    else
        hipLaunchKernelGGL(batchnorm_bwd_k<bf16>, dim3(C), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)x, gamma,
                           save_mean, save_rstd, (bf16*)dx, dgamma, dbeta, O, C, I, so, sc, si, training, dxsum);
    MV_LAUNCH_CHECK("batchnorm_bwd");
    return 0;
}

/* bytes of fp32 scratch mvuld_layernorm_bwd wants for its two-pass column sums (1024 blocks x 2C floats) */
extern "C" int64_t mvuld_layernorm_bwd_workspace_bytes(int C) { return (int64_t)1024 * 2 * (C > 0 ? C : 0) * 4; }

// Deferred parameter gradients.  mvuld_layernorm_bwd called with dgamma = dbeta = NULL and a workspace writes dx and the per-workgroup
// column partials only (ws[p][0..C) = d gamma, ws[p][C..2C) = d beta, p < nparts) and launches nothing else; the caller adds them into the
// gradients whenever it likes -- on another stream, off the backward chain: nothing in backward reads a LayerNorm's parameter gradients, and
// a 5 us reduction launched between two chip-filling kernels costs the chain far more than 5 us beside other streams' kernels.
extern "C" int mvuld_layernorm_bwd_nparts(int64_t rows, int C, int64_t ws_bytes, int dtype) {
    if (dtype != MVULD_BF16 || rows <= 0 || C <= 0 || C % 8 != 0 || C > 64 * LN_MAXPL || ws_bytes < (int64_t)8 * C * 64) return 0;
    const int G = ln_group(C), rpw = 64 / G;
    static const int capv = getenv("MVULD_LN_BWD_GRID") ? atoi(getenv("MVULD_LN_BWD_GRID")) : 1024;
    int gridv = (int)min((int64_t)capv, cdiv(rows, (int64_t)4 * rpw));
    return (int)min((int64_t)gridv, ws_bytes / ((int64_t)8 * C));
}
// ... and many of them in ONE launch (job table as a kernel argument): blockIdx.z = job, blockIdx.y = slab of partial rows, blockIdx.x = 64 columns
// of [d gamma | d beta].  Small workgroups on purpose: beside the step's persistent GEMMs (one 512-thread, 256-register workgroup per CU) a
// 1024-thread workgroup finds no CU with room until one of those kernels ends, and stalls its stream that long.
#define LN_RED_MAX_JOBS 64
struct LnRedJobs { int n; struct { const float* ws; float* dg; float* db; int nparts, C; } j[LN_RED_MAX_JOBS]; };
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_batch_k(const LnRedJobs J) {
    __shared__ float red[4][64];
    const auto& jb = J.j[blockIdx.z];
    const int C = jb.C, nblk = jb.nparts;
    if ((int)blockIdx.x * 64 >= 2 * C) return;                 // (uniform: narrower job than the widest of the launch)
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int cc = min(c, 2 * C - 1);
    const int per = (nblk + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = b0 + g; b < b1; b += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int bb = b + 4 * u;
            const float v = jb.ws[(size_t)min(bb, nblk - 1) * 2 * C + cc];
            a[u] += bb < b1 ? v : 0.f;
        }
    }
    red[g][threadIdx.x & 63] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (g == 0 && c < 2 * C) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (c < C) { if (jb.dg) atomicAdd(jb.dg + c, t); }
        else if (jb.db) atomicAdd(jb.db + c - C, t);
    }
}
// desc: njobs x 5 int64 {ws, nparts, C, dgamma, dbeta}, read on the host during the call
extern "C" int mvuld_layernorm_bwd_reduce_batch(const int64_t* desc, int njobs, hipStream_t stream) {
    MV_CHECK_ARG(desc && njobs > 0, "layernorm_bwd_reduce_batch: bad args");
    for (int i0 = 0; i0 < njobs; i0 += LN_RED_MAX_JOBS) {
        LnRedJobs J;
        J.n = min(LN_RED_MAX_JOBS, njobs - i0);
        int cmax = 0;
        for (int i = 0; i < J.n; ++i) {
            const int64_t* d = desc + 5 * (i0 + i);
            MV_CHECK_ARG(d[0] && d[1] > 0 && d[2] > 0 && (d[3] || d[4]), "layernorm_bwd_reduce_batch: job %d: bad entry", i0 + i);
            J.j[i].ws = (const float*)d[0]; J.j[i].nparts = (int)d[1]; J.j[i].C = (int)d[2]; J.j[i].dg = (float*)d[3]; J.j[i].db = (float*)d[4];
            cmax = max(cmax, (int)d[2]);
        }
        hipLaunchKernelGGL(layernorm_bwd_reduce_batch_k, dim3(cdiv(2 * cmax, 64), 16, J.n), dim3(256), 0, stream, J);
    }
    MV_LAUNCH_CHECK("layernorm_bwd_reduce_batch");
    return 0;
}

extern "C" int mvuld_layernorm_bwd_reduce(const float* ws, int nparts, int C, float* dgamma, float* dbeta, hipStream_t stream) {
    MV_CHECK_ARG(ws && nparts > 0 && C > 0 && (dgamma || dbeta), "layernorm_bwd_reduce: bad args");
    hipLaunchKernelGGL(layernorm_bwd_reduce_k, dim3(cdiv(2 * C, 64), 8), dim3(1024), 0, stream, ws, nparts, C, dgamma, dbeta);
    MV_LAUNCH_CHECK("layernorm_bwd_reduce");
    return 0;
}

