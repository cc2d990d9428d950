// Fused attention, reference-quality VALU kernels (fp32 math, f32|bf16 storage): forward with
// online softmax, backward as a dQ pass (thread per query) + a dK/dV pass (thread per key), with
// P recomputed from the saved log-sum-exp.  No [N,N] tensor ever reaches HBM.
//
// MODE 0 -- SwinV2 window attention (swin_transformer_v2.py:140-179 fused with the roll /
//   window_partition / window_reverse / roll-back of :279-299): cosine attention
//   normalize(q).normalize(k)^T * exp(min(logit_scale, ln 100)) + 16*sigmoid(cpb)[rel_index]
//   (+ 0/-100 shift mask computed from the 3x3 region ids of :248-264), softmax, .V.
//   Tokens are gathered from / scattered to image order through the window index map, so
//   qkv and the output stay [B*L, .] in image order.
// MODE 1 -- pad-masked attention of the UniXcoder encoder (unixcoder.py:35-36 with the
//   transformers 4.18 additive mask): q.k^T/sqrt(hd) + (valid_q && valid_k ? 0 : -10000).
//
// qkv rows are [3][H][hd]; out rows are [H][hd].
#include "common.h"

struct AttnGeom {
    int mode, B, H, N, nW, res, ws, shift;
    float scale;          // MODE 1: 1/sqrt(hd)
};

__device__ __forceinline__ int64_t attn_token(const AttnGeom& g, int b, int w, int n) {
    if (g.mode == 1) return (int64_t)b * g.N + n;
    const int nwx = g.res / g.ws;
    const int sy = (w / nwx) * g.ws + n / g.ws, sx = (w % nwx) * g.ws + n % g.ws;
    int oy = sy + g.shift, ox = sx + g.shift;
    if (oy >= g.res) oy -= g.res;
    if (ox >= g.res) ox -= g.res;
    return ((int64_t)b * g.res + oy) * g.res + ox;
}
// region id of a *shifted* coordinate (h_slices / w_slices of swin_transformer_v2.py:249-254)
__device__ __forceinline__ int attn_rid(const AttnGeom& g, int s) {
    return s < g.res - g.ws ? 0 : (s < g.res - g.shift ? 1 : 2);
}
// packed per-token window info: iy | ix<<8 | region<<16
__device__ __forceinline__ int attn_info(const AttnGeom& g, int w, int n) {
    const int nwx = g.res / g.ws;
    const int iy = n / g.ws, ix = n % g.ws;
    int reg = 0;
    if (g.shift > 0) reg = attn_rid(g, (w / nwx) * g.ws + iy) * 3 + attn_rid(g, (w % nwx) * g.ws + ix);
    return iy | (ix << 8) | (reg << 16);
}
__device__ __forceinline__ int attn_rel(const AttnGeom& g, int iq, int ik) {
    const int dy = (iq & 255) - (ik & 255) + g.ws - 1, dx = ((iq >> 8) & 255) - ((ik >> 8) & 255) + g.ws - 1;
    return dy * (2 * g.ws - 1) + dx;
}

#define AT_QB 256   // queries (or keys) per block, one per thread
#define AT_KT 64    // rows of the other side staged per LDS tile

// Stage AT_KT rows of K-like data ([tile][HD] floats) from qkv slot `slot`; optional L2
// normalisation (MODE 0) and optional row scale.  8-element chunks: thread c -> (row c/(HD/8), chunk c%(HD/8)).
template <typename T, int HD>
__device__ __forceinline__ void stage_rows(const AttnGeom& g, const T* __restrict__ base, int64_t rowstride, int coloff,
                                           int b, int w, int n0, float* dst, bool normalize, float mul) {
    constexpr int CPR = HD / 8;
    for (int c = threadIdx.x; c < AT_KT * CPR; c += AT_QB) {
        const int r = c / CPR, ch = c % CPR;
        const int n = n0 + r;
        float v[8];
        if (n < g.N) {
            const T* p = base + attn_token(g, b, w, n) * rowstride + coloff + ch * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ldf(p + e);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        float f = mul;
        if (normalize) {
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += v[e] * v[e];
#pragma unroll
            for (int o = 1; o < CPR; o <<= 1) ss += __shfl_xor(ss, o, 64);
            f = mul / fmaxf(sqrtf(ss), 1e-12f);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[r * HD + ch * 8 + e] = v[e] * f;
    }
}

template <int HD>
__device__ __forceinline__ float dot_lds(const float* q, const float* krow) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        const float4 k4 = *(const float4*)(krow + d);
        s = fmaf(q[d], k4.x, s); s = fmaf(q[d + 1], k4.y, s); s = fmaf(q[d + 2], k4.z, s); s = fmaf(q[d + 3], k4.w, s);
    }
    return s;
}

// ------------------------------------------------------------------------------------ forward
template <typename T, int HD, int MODE>
__global__ __launch_bounds__(AT_QB) void attn_fwd_simple(AttnGeom g, const T* __restrict__ qkv, const float* __restrict__ table16,
                                                        const float* __restrict__ logit_scale, const int* __restrict__ valid,
                                                        T* __restrict__ out, float* __restrict__ lse) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ks = sm;                          // [AT_KT][HD]
    float* Vs = Ks + AT_KT * HD;             // [AT_KT][HD]
    int* Ki = (int*)(Vs + AT_KT * HD);       // [AT_KT] info / validity
    float* tab = (float*)(Ki + AT_KT);       // MODE 0: [(2ws-1)^2] this head's 16*sigmoid(cpb)
    const int h = blockIdx.y, bw = blockIdx.z, b = bw / g.nW, w = bw % g.nW;
    const int C = g.H * HD;
    const int64_t rs = 3 * (int64_t)C;
    const int nq = blockIdx.x * AT_QB + threadIdx.x;
    const bool qok = nq < g.N;
    const int nqc = qok ? nq : g.N - 1;
    const int64_t tq = attn_token(g, b, w, nqc);
    float q[HD], o[HD];
    int qi = 0;
    {
        float ss = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) { q[d] = ldf(qkv + tq * rs + h * HD + d); ss += q[d] * q[d]; o[d] = 0.f; }
        float f;
        if (MODE == 0) {
            f = __expf(fminf(logit_scale[h], 4.605170185988092f)) / fmaxf(sqrtf(ss), 1e-12f);
            qi = attn_info(g, w, nqc);
            const int T2 = (2 * g.ws - 1) * (2 * g.ws - 1);
            for (int i = threadIdx.x; i < T2; i += AT_QB) tab[i] = table16[(int64_t)i * g.H + h];
        } else {
            f = g.scale;
            qi = valid[b * g.N + nqc];
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) q[d] *= f;
    }
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < g.N; k0 += AT_KT) {
        __syncthreads();
        stage_rows<T, HD>(g, qkv, rs, C + h * HD, b, w, k0, Ks, MODE == 0, 1.0f);
        stage_rows<T, HD>(g, qkv, rs, 2 * C + h * HD, b, w, k0, Vs, false, 1.0f);
        if (threadIdx.x < AT_KT) {
            const int n = k0 + threadIdx.x;
            Ki[threadIdx.x] = n < g.N ? (MODE == 0 ? attn_info(g, w, n) : valid[b * g.N + n]) : 0;
        }
        __syncthreads();
        const int kn = min(AT_KT, g.N - k0);
        for (int j0 = 0; j0 < kn; j0 += 8) {          // online softmax in groups of 8 keys
            float s8[8];
            float tmax = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = j0 + jj;
                float v = -INFINITY;
                if (j < kn) {
                    v = dot_lds<HD>(q, Ks + j * HD);
                    const int ki = Ki[j];
                    if (MODE == 0) {
                        v += tab[attn_rel(g, qi, ki)];
                        if ((qi >> 16) != (ki >> 16)) v -= 100.0f;
                    } else if (!(qi && ki)) v -= 10000.0f;
                }
                s8[jj] = v;
                tmax = fmaxf(tmax, v);
            }
            const float mn = fmaxf(m, tmax);
            const float corr = __expf(m - mn);
            l *= corr;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] *= corr;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = j0 + jj;
                if (j < kn) {
                    const float p = __expf(s8[jj] - mn);
                    l += p;
                    const float* vr = Vs + j * HD;
#pragma unroll
                    for (int d = 0; d < HD; d += 4) {
                        const float4 v4 = *(const float4*)(vr + d);
                        o[d] = fmaf(p, v4.x, o[d]); o[d + 1] = fmaf(p, v4.y, o[d + 1]);
                        o[d + 2] = fmaf(p, v4.z, o[d + 2]); o[d + 3] = fmaf(p, v4.w, o[d + 3]);
                    }
                }
            }
            m = mn;
        }
    }
    if (qok) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < HD; ++d) stf(out + tq * C + h * HD + d, o[d] * inv);
        lse[((int64_t)bw * g.H + h) * g.N + nq] = m + __logf(l);
    }
}

// ------------------------------------------------------------------------------------ backward: dQ (+ d table, d logit_scale)
template <typename T, int HD, int MODE>
__global__ __launch_bounds__(AT_QB) void attn_bwd_dq_simple(AttnGeom g, const T* __restrict__ qkv, const float* __restrict__ table16,
                                                           const float* __restrict__ logit_scale, const int* __restrict__ valid,
                                                           const T* __restrict__ out, const T* __restrict__ dout,
                                                           const float* __restrict__ lse, T* __restrict__ dqkv,
                                                           float* __restrict__ dtable16, float* __restrict__ dlogit_scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ks = sm;
    float* Vs = Ks + AT_KT * HD;
    int* Ki = (int*)(Vs + AT_KT * HD);
    float* red = (float*)(Ki + AT_KT);       // [16]
    float* tab = red + 16;                   // MODE 0: [(2ws-1)^2]
    const int T2 = (2 * g.ws - 1) * (2 * g.ws - 1);
    float* dtab = tab + T2;                  // MODE 0: [(2ws-1)^2]
    const int h = blockIdx.y, bw = blockIdx.z, b = bw / g.nW, w = bw % g.nW;
    const int C = g.H * HD;
    const int64_t rs = 3 * (int64_t)C;
    const int nq = blockIdx.x * AT_QB + threadIdx.x;
    const bool qok = nq < g.N;
    const int nqc = qok ? nq : g.N - 1;
    const int64_t tq = attn_token(g, b, w, nqc);
    float q[HD], dq[HD], dO[HD];
    float qf, qinv = 0.f, tau = 1.f, delta = 0.f;
    int qi;
    {
        float ss = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            q[d] = ldf(qkv + tq * rs + h * HD + d); ss += q[d] * q[d]; dq[d] = 0.f;
            dO[d] = qok ? ldf(dout + tq * C + h * HD + d) : 0.f;
            delta += dO[d] * ldf(out + tq * C + h * HD + d);
        }
        if (MODE == 0) {
            tau = __expf(fminf(logit_scale[h], 4.605170185988092f));
            qinv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
            qf = tau * qinv;
            qi = attn_info(g, w, nqc);
            for (int i = threadIdx.x; i < T2; i += AT_QB) { tab[i] = table16[(int64_t)i * g.H + h]; dtab[i] = 0.f; }
        } else {
            qf = g.scale;
            qi = valid[b * g.N + nqc];
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) q[d] *= qf;      // q~ = tau * q^  (or q * scale)
    }
    const float L = lse[((int64_t)bw * g.H + h) * g.N + nqc];
    for (int k0 = 0; k0 < g.N; k0 += AT_KT) {
        __syncthreads();
        stage_rows<T, HD>(g, qkv, rs, C + h * HD, b, w, k0, Ks, MODE == 0, 1.0f);
        stage_rows<T, HD>(g, qkv, rs, 2 * C + h * HD, b, w, k0, Vs, false, 1.0f);
        if (threadIdx.x < AT_KT) {
            const int n = k0 + threadIdx.x;
            Ki[threadIdx.x] = n < g.N ? (MODE == 0 ? attn_info(g, w, n) : valid[b * g.N + n]) : 0;
        }
        __syncthreads();
        const int kn = min(AT_KT, g.N - k0);
        for (int j = 0; j < kn; ++j) {
            float s = dot_lds<HD>(q, Ks + j * HD);
            const int ki = Ki[j];
            int rel = 0;
            if (MODE == 0) {
                rel = attn_rel(g, qi, ki);
                s += tab[rel];
                if ((qi >> 16) != (ki >> 16)) s -= 100.0f;
            } else if (!(qi && ki)) s -= 10000.0f;
            const float p = __expf(s - L);
            const float dp = dot_lds<HD>(dO, Vs + j * HD);
            const float ds = qok ? p * (dp - delta) : 0.f;
            if (MODE == 0) atomicAdd(dtab + rel, ds);
            const float* kr = Ks + j * HD;
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, kr[d], dq[d]);
        }
    }
    // dq holds d(q~)
    if (MODE == 0) {
        // q^ = q~/tau ; d tau = sum dq~ . q^ ; d q^ = tau dq~ ; dq = (dq^ - q^ (q^.dq^)) * qinv
        float dt = 0.f, qd = 0.f;
        const float it = 1.0f / tau;
#pragma unroll
        for (int d = 0; d < HD; ++d) { const float qh = q[d] * it; dt += dq[d] * qh; }
        qd = dt * tau;                                    // q^ . dq^
        if (qok) {
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                const float qh = q[d] * it;
                stf(dqkv + tq * rs + h * HD + d, (tau * dq[d] - qh * qd) * qinv);
            }
        }
        dt = block_sum(qok ? dt : 0.f, red);
        if (threadIdx.x == 0 && logit_scale[h] < 4.605170185988092f) atomicAdd(dlogit_scale + h, dt * tau);
        __syncthreads();
        for (int i = threadIdx.x; i < T2; i += AT_QB) {
            const float v = dtab[i];
            if (v != 0.f) atomicAdd(dtable16 + (int64_t)i * g.H + h, v);
        }
    } else if (qok) {
#pragma unroll
        for (int d = 0; d < HD; ++d) stf(dqkv + tq * rs + h * HD + d, dq[d] * g.scale);
    }
}

// ------------------------------------------------------------------------------------ backward: dK, dV
template <typename T, int HD, int MODE>
__global__ __launch_bounds__(AT_QB) void attn_bwd_dkv_simple(AttnGeom g, const T* __restrict__ qkv, const float* __restrict__ table16,
                                                            const float* __restrict__ logit_scale, const int* __restrict__ valid,
                                                            const T* __restrict__ out, const T* __restrict__ dout,
                                                            const float* __restrict__ lse, T* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Qs = sm;                          // [AT_KT][HD]  q~
    float* Ds = Qs + AT_KT * HD;             // [AT_KT][HD]  dO
    float* Os = Ds + AT_KT * HD;             // [AT_KT][HD]  O (for delta)
    int* Qi = (int*)(Os + AT_KT * HD);       // [AT_KT]
    float* Ql = (float*)(Qi + AT_KT);        // [AT_KT] lse
    float* Qd = Ql + AT_KT;                  // [AT_KT] delta
    float* tab = Qd + AT_KT;                 // MODE 0
    const int h = blockIdx.y, bw = blockIdx.z, b = bw / g.nW, w = bw % g.nW;
    const int C = g.H * HD;
    const int64_t rs = 3 * (int64_t)C;
    const int nk = blockIdx.x * AT_QB + threadIdx.x;
    const bool kok = nk < g.N;
    const int nkc = kok ? nk : g.N - 1;
    const int64_t tk = attn_token(g, b, w, nkc);
    float k[HD], v[HD], dk[HD], dv[HD];
    float kinv = 1.f;
    int ki;
    {
        float ss = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            k[d] = ldf(qkv + tk * rs + C + h * HD + d); ss += k[d] * k[d];
            v[d] = ldf(qkv + tk * rs + 2 * C + h * HD + d);
            dk[d] = 0.f; dv[d] = 0.f;
        }
        if (MODE == 0) {
            kinv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int d = 0; d < HD; ++d) k[d] *= kinv;
            ki = attn_info(g, w, nkc);
            const int T2 = (2 * g.ws - 1) * (2 * g.ws - 1);
            for (int i = threadIdx.x; i < T2; i += AT_QB) tab[i] = table16[(int64_t)i * g.H + h];
        } else ki = valid[b * g.N + nkc];
    }
    const float qmul = MODE == 0 ? __expf(fminf(logit_scale[h], 4.605170185988092f)) : g.scale;
    for (int q0 = 0; q0 < g.N; q0 += AT_KT) {
        __syncthreads();
        stage_rows<T, HD>(g, qkv, rs, h * HD, b, w, q0, Qs, MODE == 0, qmul);
        stage_rows<T, HD>(g, dout, C, h * HD, b, w, q0, Ds, false, 1.0f);
        stage_rows<T, HD>(g, out, C, h * HD, b, w, q0, Os, false, 1.0f);
        if (threadIdx.x < AT_KT) {
            const int n = q0 + threadIdx.x;
            Qi[threadIdx.x] = n < g.N ? (MODE == 0 ? attn_info(g, w, n) : valid[b * g.N + n]) : 0;
            Ql[threadIdx.x] = n < g.N ? lse[((int64_t)bw * g.H + h) * g.N + n] : 0.f;
        }
        __syncthreads();
        if (threadIdx.x < AT_KT) {
            float dl = 0.f;
            for (int d = 0; d < HD; ++d) dl += Ds[threadIdx.x * HD + d] * Os[threadIdx.x * HD + d];
            Qd[threadIdx.x] = dl;
        }
        __syncthreads();
        const int qn = min(AT_KT, g.N - q0);
        for (int j = 0; j < qn; ++j) {
            float s = dot_lds<HD>(k, Qs + j * HD);
            const int qi = Qi[j];
            if (MODE == 0) {
                s += tab[attn_rel(g, qi, ki)];
                if ((qi >> 16) != (ki >> 16)) s -= 100.0f;
            } else if (!(qi && ki)) s -= 10000.0f;
            const float p = __expf(s - Ql[j]);
            const float* dr = Ds + j * HD;
            const float dp = dot_lds<HD>(v, dr);
            const float ds = p * (dp - Qd[j]);
            const float* qr = Qs + j * HD;
#pragma unroll
            for (int d = 0; d < HD; ++d) { dv[d] = fmaf(p, dr[d], dv[d]); dk[d] = fmaf(ds, qr[d], dk[d]); }
        }
    }
    if (kok) {
        if (MODE == 0) {
            float kd = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) kd += k[d] * dk[d];
#pragma unroll
            for (int d = 0; d < HD; ++d) stf(dqkv + tk * rs + C + h * HD + d, (dk[d] - k[d] * kd) * kinv);
        } else {
#pragma unroll
            for (int d = 0; d < HD; ++d) stf(dqkv + tk * rs + C + h * HD + d, dk[d]);
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) stf(dqkv + tk * rs + 2 * C + h * HD + d, dv[d]);
    }
}

// ------------------------------------------------------------------------------------ launchers
static int check_geom(const char* fn, int mode, int B, int H, int hd, int N, int nW, int res, int ws, int shift) {
    MV_CHECK_ARG(mode == 0 || mode == 1, "%s: mode %d", fn, mode);
    MV_CHECK_ARG(B > 0 && H > 0 && N > 0 && nW > 0, "%s: empty geometry", fn);
    MV_CHECK_ARG(hd == 32 || hd == 64, "%s: head_dim %d unsupported (32|64)", fn, hd);
    if (mode == 0) {
        MV_CHECK_ARG(ws > 0 && ws < 128 && res % ws == 0 && N == ws * ws && nW == (res / ws) * (res / ws),
                     "%s: window geometry res=%d ws=%d N=%d nW=%d", fn, res, ws, N, nW);
        MV_CHECK_ARG(shift >= 0 && shift < ws, "%s: shift %d", fn, shift);
    } else {
        MV_CHECK_ARG(nW == 1, "%s: pad mode needs nW=1", fn);
    }
    return 0;
}

#define ATTN_SWITCH(KERNEL, T, smem, ...)                                                                   \
    do {                                                                                                    \
        if (hd == 32 && mode == 0) hipLaunchKernelGGL((KERNEL<T, 32, 0>), grid, dim3(AT_QB), smem, stream, __VA_ARGS__); \
        else if (hd == 64 && mode == 0) hipLaunchKernelGGL((KERNEL<T, 64, 0>), grid, dim3(AT_QB), smem, stream, __VA_ARGS__); \
        else if (hd == 32) hipLaunchKernelGGL((KERNEL<T, 32, 1>), grid, dim3(AT_QB), smem, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<T, 64, 1>), grid, dim3(AT_QB), smem, stream, __VA_ARGS__);          \
    } while (0)

template <typename T>
static void launch_attn_fwd(const AttnGeom& g, int hd, int mode, dim3 grid, size_t smem, hipStream_t stream, const void* qkv,
                            const float* table16, const float* ls, const int* valid, void* out, float* lse) {
    ATTN_SWITCH(attn_fwd_simple, T, smem, g, (const T*)qkv, table16, ls, valid, (T*)out, lse);
}
template <typename T>
static void launch_attn_dq(const AttnGeom& g, int hd, int mode, dim3 grid, size_t smem, hipStream_t stream, const void* qkv,
                           const float* table16, const float* ls, const int* valid, const void* out, const void* dout,
                           const float* lse, void* dqkv, float* dtable16, float* dls) {
    ATTN_SWITCH(attn_bwd_dq_simple, T, smem, g, (const T*)qkv, table16, ls, valid, (const T*)out, (const T*)dout, lse,
                (T*)dqkv, dtable16, dls);
}
template <typename T>
static void launch_attn_dkv(const AttnGeom& g, int hd, int mode, dim3 grid, size_t smem, hipStream_t stream, const void* qkv,
                            const float* table16, const float* ls, const int* valid, const void* out, const void* dout,
                            const float* lse, void* dqkv) {
    ATTN_SWITCH(attn_bwd_dkv_simple, T, smem, g, (const T*)qkv, table16, ls, valid, (const T*)out, (const T*)dout, lse,
                (T*)dqkv);
}

static int set_big_lds(const void* fn, size_t smem) {
    if (smem > 65536) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess;
    return 0;
}

extern "C" int mvuld_attn_fwd_simple(int mode, int B, int H, int hd, int N, int nW, int res, int ws, int shift, float scale,
                                     const void* qkv, const float* table16, const float* logit_scale, const int* valid,
                                     void* out, float* lse, int dtype, hipStream_t stream) {
    if (check_geom("attn_fwd", mode, B, H, hd, N, nW, res, ws, shift)) return 1;
    MV_CHECK_ARG(qkv && out && lse && (mode == 1 ? valid != nullptr : (table16 && logit_scale)), "attn_fwd: null pointer");
    AttnGeom g{mode, B, H, N, nW, res, ws, shift, scale};
    const int T2 = mode == 0 ? (2 * ws - 1) * (2 * ws - 1) : 0;
    const size_t smem = (size_t)(2 * AT_KT * hd + AT_KT + T2) * 4;
    MV_CHECK_ARG(smem <= 65536, "attn_fwd: LDS %zu", smem);
    dim3 grid((unsigned)cdiv(N, AT_QB), H, B * nW);
    if (dtype == MVULD_F32) launch_attn_fwd<float>(g, hd, mode, grid, smem, stream, qkv, table16, logit_scale, valid, out, lse);
    else launch_attn_fwd<bf16>(g, hd, mode, grid, smem, stream, qkv, table16, logit_scale, valid, out, lse);
    MV_LAUNCH_CHECK("attn_fwd_simple");
    return 0;
}

// dqkv must be fully written by this call: dQ by the first pass, dK/dV by the second.
extern "C" int mvuld_attn_bwd_simple(int mode, int B, int H, int hd, int N, int nW, int res, int ws, int shift, float scale,
                                     const void* qkv, const float* table16, const float* logit_scale, const int* valid,
                                     const void* out, const void* dout, const float* lse, void* dqkv, float* dtable16,
                                     float* dlogit_scale, int dtype, hipStream_t stream) {
    if (check_geom("attn_bwd", mode, B, H, hd, N, nW, res, ws, shift)) return 1;
    MV_CHECK_ARG(qkv && out && dout && lse && dqkv, "attn_bwd: null pointer");
    MV_CHECK_ARG(mode == 1 ? valid != nullptr : (table16 && logit_scale && dtable16 && dlogit_scale), "attn_bwd: null pointer");
    AttnGeom g{mode, B, H, N, nW, res, ws, shift, scale};
    const int T2 = mode == 0 ? (2 * ws - 1) * (2 * ws - 1) : 0;
    dim3 grid((unsigned)cdiv(N, AT_QB), H, B * nW);
    const size_t smem_q = (size_t)(2 * AT_KT * hd + AT_KT + 16 + 2 * T2) * 4;
    const size_t smem_k = (size_t)(3 * AT_KT * hd + 3 * AT_KT + T2) * 4;
    MV_CHECK_ARG(smem_q <= 65536 && smem_k <= 65536, "attn_bwd: LDS %zu %zu", smem_q, smem_k);
    if (dtype == MVULD_F32) {
        launch_attn_dq<float>(g, hd, mode, grid, smem_q, stream, qkv, table16, logit_scale, valid, out, dout, lse, dqkv, dtable16, dlogit_scale);
        launch_attn_dkv<float>(g, hd, mode, grid, smem_k, stream, qkv, table16, logit_scale, valid, out, dout, lse, dqkv);
    } else {
        launch_attn_dq<bf16>(g, hd, mode, grid, smem_q, stream, qkv, table16, logit_scale, valid, out, dout, lse, dqkv, dtable16, dlogit_scale);
        launch_attn_dkv<bf16>(g, hd, mode, grid, smem_k, stream, qkv, table16, logit_scale, valid, out, dout, lse, dqkv);
    }
    MV_LAUNCH_CHECK("attn_bwd_simple");
    return 0;
}
