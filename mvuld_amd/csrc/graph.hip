// Sparse neighbour aggregation over the batched code-property graph (CSR by destination and by source)
// and the pad/truncate-to-100-nodes unbatching.
//
// Replaces dgl.nn.pytorch.GATConv's SDDMM (u_add_v), edge_softmax and SpMM (u_mul_e_sum) as called
// at GraphModel.py:167-170 (dgl-cu102==0.8.1, third party) and unbatch_features (GraphModel.py:30-54).
// One wave per (node, head): the 512 features of a head are 8 per lane, incoming edges are walked
// sequentially (in-degree ~4), softmax statistics by wave shuffles.  HBM-bound gather: every edge
// reads one 512-feature row of its source.  Backward w.r.t. source features walks the by-source CSR
// so the scatter becomes a gather (deterministic, no float atomics on features).
#include "common.h"

#define DISPATCH_T(dtype, CALL)                 \
    do {                                        \
        if ((dtype) == MVULD_F32) { typedef float T; CALL; } \
        else { typedef bf16 T; CALL; }          \
    } while (0)

#define GAT_MAXPL 8     // features per lane: O <= 512

__device__ __forceinline__ float leaky(float x, float slope) { return x > 0.f ? x : x * slope; }

// el[n,h] = sum_f ft[n,h,f]*al[h,f] ; er likewise
template <typename T>
__global__ __launch_bounds__(256) void gat_scores_fwd_k(const T* __restrict__ ft, const float* __restrict__ al, const float* __restrict__ ar,
                                                        float* __restrict__ el, float* __restrict__ er, int64_t NH, int H, int O) {
    const int lane = threadIdx.x & 63;
    for (int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); idx < NH; idx += (int64_t)gridDim.x * 4) {
        const int h = (int)(idx % H);
        float sl = 0.f, sr = 0.f;
        for (int f = lane; f < O; f += 64) {
            const float v = ldf(ft + idx * O + f);
            sl += v * al[h * O + f];
            sr += v * ar[h * O + f];
        }
        sl = wave_sum(sl); sr = wave_sum(sr);
        if (lane == 0) { el[idx] = sl; er[idx] = sr; }
    }
}

// per (dst,h): a = softmax_e leaky(el[src_e,h] + er[dst,h]); out[dst,h,:] = sum_e a_e ft[src_e,h,:] + bias[h,:]
template <typename T>
__global__ __launch_bounds__(256) void gat_aggregate_fwd_k(const T* __restrict__ ft, const float* __restrict__ el, const float* __restrict__ er,
                                                           const int* __restrict__ indptr, const int* __restrict__ srcs,
                                                           const float* __restrict__ bias, T* __restrict__ out, float* __restrict__ alpha,
                                                           int N, int H, int O, float slope) {
    const int lane = threadIdx.x & 63;
    const int64_t NH = (int64_t)N * H;
    for (int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); idx < NH; idx += (int64_t)gridDim.x * 4) {
        const int d = (int)(idx / H), h = (int)(idx % H);
        const int beg = indptr[d], end = indptr[d + 1];
        const float erd = er[idx];
        float mx = -INFINITY;
        for (int e = beg + lane; e < end; e += 64) mx = fmaxf(mx, leaky(el[(int64_t)srcs[e] * H + h] + erd, slope));
        mx = wave_max(mx);
        float sm = 0.f;
        for (int e = beg + lane; e < end; e += 64) sm += __expf(leaky(el[(int64_t)srcs[e] * H + h] + erd, slope) - mx);
        sm = wave_sum(sm);
        const float inv = end > beg ? 1.0f / sm : 0.f;
        float acc[GAT_MAXPL];
#pragma unroll
        for (int i = 0; i < GAT_MAXPL; ++i) acc[i] = 0.f;
        for (int e = beg; e < end; ++e) {
            const int s = srcs[e];
            const float a = __expf(leaky(el[(int64_t)s * H + h] + erd, slope) - mx) * inv;
            if (lane == 0 && alpha) alpha[(int64_t)e * H + h] = a;
            const T* fr = ft + ((int64_t)s * H + h) * O;
#pragma unroll
            for (int i = 0; i < GAT_MAXPL; ++i) {
                const int f = lane + 64 * i;
                if (f < O) acc[i] = fmaf(a, ldf(fr + f), acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < GAT_MAXPL; ++i) {
            const int f = lane + 64 * i;
            if (f < O) stf(out + idx * O + f, acc[i] + (bias ? bias[h * O + f] : 0.f));
        }
    }
}

// per (dst,h): da_e = dout[dst,h,:].ft[src_e,h,:]; S = sum a_e da_e; dlogit_e = a_e (da_e - S) leaky'(x_e); der[dst,h] = sum_e dlogit_e
template <typename T>
__global__ __launch_bounds__(256) void gat_bwd_dst_k(const T* __restrict__ dout, const T* __restrict__ ft, const float* __restrict__ el,
                                                     const float* __restrict__ er, const float* __restrict__ alpha,
                                                     const int* __restrict__ indptr, const int* __restrict__ srcs,
                                                     float* __restrict__ dlogit, float* __restrict__ der, int N, int H, int O, float slope) {
    const int lane = threadIdx.x & 63;
    const int64_t NH = (int64_t)N * H;
    for (int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); idx < NH; idx += (int64_t)gridDim.x * 4) {
        const int d = (int)(idx / H), h = (int)(idx % H);
        const int beg = indptr[d], end = indptr[d + 1];
        float g[GAT_MAXPL];
#pragma unroll
        for (int i = 0; i < GAT_MAXPL; ++i) { const int f = lane + 64 * i; g[i] = f < O ? ldf(dout + idx * O + f) : 0.f; }
        float S = 0.f;
        for (int e = beg; e < end; ++e) {
            const T* fr = ft + ((int64_t)srcs[e] * H + h) * O;
            float da = 0.f;
#pragma unroll
            for (int i = 0; i < GAT_MAXPL; ++i) { const int f = lane + 64 * i; if (f < O) da = fmaf(g[i], ldf(fr + f), da); }
            da = wave_sum(da);
            S += alpha[(int64_t)e * H + h] * da;
        }
        const float erd = er[idx];
        float sder = 0.f;
        for (int e = beg; e < end; ++e) {
            const int s = srcs[e];
            const T* fr = ft + ((int64_t)s * H + h) * O;
            float da = 0.f;
#pragma unroll
            for (int i = 0; i < GAT_MAXPL; ++i) { const int f = lane + 64 * i; if (f < O) da = fmaf(g[i], ldf(fr + f), da); }
            da = wave_sum(da);
            const float x = el[(int64_t)s * H + h] + erd;
            const float dl = alpha[(int64_t)e * H + h] * (da - S) * (x > 0.f ? 1.0f : slope);
            if (lane == 0) dlogit[(int64_t)e * H + h] = dl;
            sder += dl;
        }
        if (lane == 0) der[idx] = sder;
    }
}

// per (src,h): dft[src,h,:] = sum_{e out of src} a_e dout[dst_e,h,:] + del*al[h,:] + der[src,h]*ar[h,:],  del = sum_e dlogit_e
template <typename T>
__global__ __launch_bounds__(256) void gat_bwd_src_k(const T* __restrict__ dout, const float* __restrict__ alpha, const float* __restrict__ dlogit,
                                                     const float* __restrict__ der, const float* __restrict__ al, const float* __restrict__ ar,
                                                     const int* __restrict__ indptr_s, const int* __restrict__ dsts, const int* __restrict__ slots,
                                                     T* __restrict__ dft, float* __restrict__ del, int N, int H, int O) {
    const int lane = threadIdx.x & 63;
    const int64_t NH = (int64_t)N * H;
    for (int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); idx < NH; idx += (int64_t)gridDim.x * 4) {
        const int s = (int)(idx / H), h = (int)(idx % H);
        const int beg = indptr_s[s], end = indptr_s[s + 1];
        float acc[GAT_MAXPL];
#pragma unroll
        for (int i = 0; i < GAT_MAXPL; ++i) acc[i] = 0.f;
        float sdel = 0.f;
        for (int k = beg; k < end; ++k) {
            const int slot = slots[k];
            const float a = alpha[(int64_t)slot * H + h];
            sdel += dlogit[(int64_t)slot * H + h];
            const T* gr = dout + ((int64_t)dsts[k] * H + h) * O;
#pragma unroll
            for (int i = 0; i < GAT_MAXPL; ++i) { const int f = lane + 64 * i; if (f < O) acc[i] = fmaf(a, ldf(gr + f), acc[i]); }
        }
        const float dr = der[idx];
#pragma unroll
        for (int i = 0; i < GAT_MAXPL; ++i) {
            const int f = lane + 64 * i;
            if (f < O) stf(dft + idx * O + f, acc[i] + sdel * al[h * O + f] + dr * ar[h * O + f]);
        }
        if (lane == 0) del[idx] = sdel;
    }
}

// dal[h,f] += sum_n del[n,h] ft[n,h,f] ; dar[h,f] += sum_n der[n,h] ft[n,h,f]
template <typename T>
__global__ __launch_bounds__(256) void gat_attn_grad_k(const T* __restrict__ ft, const float* __restrict__ del, const float* __restrict__ der,
                                                       float* __restrict__ dal, float* __restrict__ dar, int N, int H, int O, int rows_per_block) {
    const int col = blockIdx.x * 256 + threadIdx.x;              // over H*O
    if (col >= H * O) return;
    const int h = col / O;
    const int n0 = blockIdx.y * rows_per_block, n1 = min(N, n0 + rows_per_block);
    float a = 0.f, b = 0.f;
    for (int n = n0; n < n1; ++n) {
        const float v = ldf(ft + (int64_t)n * H * O + col);
        a = fmaf(del[(int64_t)n * H + h], v, a);
        b = fmaf(der[(int64_t)n * H + h], v, b);
    }
    atomicAdd(dal + col, a);
    atomicAdd(dar + col, b);
}

extern "C" int mvuld_gat_scores_fwd(const void* ft, const float* al, const float* ar, float* el, float* er, int N, int H, int O, int dtype,
                                    hipStream_t stream) {
    MV_CHECK_ARG(ft && al && ar && el && er && N > 0 && H > 0 && O > 0, "gat_scores_fwd: bad args");
    const int64_t NH = (int64_t)N * H;
    const int grid = (int)min((int64_t)4096, cdiv(NH, 4));
    DISPATCH_T(dtype, hipLaunchKernelGGL(gat_scores_fwd_k<T>, dim3(grid), dim3(256), 0, stream, (const T*)ft, al, ar, el, er, NH, H, O));
    MV_LAUNCH_CHECK("gat_scores_fwd");
    return 0;
}

extern "C" int mvuld_gat_aggregate_fwd(const void* ft, const float* el, const float* er, const int* indptr_dst, const int* src_by_dst,
                                       const float* bias, void* out, float* alpha, int N, int E, int H, int O, float slope, int dtype,
                                       hipStream_t stream) {
    MV_CHECK_ARG(ft && el && er && indptr_dst && src_by_dst && out && N > 0 && H > 0 && O > 0 && O <= 64 * GAT_MAXPL,
                 "gat_aggregate_fwd: bad args (O<=512)");
    (void)E;
    const int grid = (int)min((int64_t)4096, cdiv((int64_t)N * H, 4));
    DISPATCH_T(dtype, hipLaunchKernelGGL(gat_aggregate_fwd_k<T>, dim3(grid), dim3(256), 0, stream, (const T*)ft, el, er, indptr_dst,
                                         src_by_dst, bias, (T*)out, alpha, N, H, O, slope));
    MV_LAUNCH_CHECK("gat_aggregate_fwd");
    return 0;
}

// Full GATConv sparse backward given dout: writes dft [N,H,O], accumulates dal/dar [H,O] (atomic);
// workspace: dlogit [E,H], der [N,H], del [N,H] (fp32, caller-owned).
extern "C" int mvuld_gat_aggregate_bwd(const void* dout, const void* ft, const float* el, const float* er, const float* alpha,
                                       const float* al, const float* ar, const int* indptr_dst, const int* src_by_dst,
                                       const int* indptr_src, const int* dst_by_src, const int* slot_by_src, void* dft, float* dal,
                                       float* dar, float* ws_dlogit, float* ws_der, float* ws_del, int N, int E, int H, int O, float slope,
                                       int dtype, hipStream_t stream) {
    MV_CHECK_ARG(dout && ft && el && er && alpha && al && ar && indptr_dst && src_by_dst && indptr_src && dst_by_src && slot_by_src && dft &&
                     dal && dar && ws_dlogit && ws_der && ws_del,
                 "gat_aggregate_bwd: null pointer");
    MV_CHECK_ARG(N > 0 && H > 0 && O > 0 && O <= 64 * GAT_MAXPL, "gat_aggregate_bwd: bad sizes");
    (void)E;
    const int grid = (int)min((int64_t)4096, cdiv((int64_t)N * H, 4));
    DISPATCH_T(dtype, hipLaunchKernelGGL(gat_bwd_dst_k<T>, dim3(grid), dim3(256), 0, stream, (const T*)dout, (const T*)ft, el, er, alpha,
                                         indptr_dst, src_by_dst, ws_dlogit, ws_der, N, H, O, slope));
    DISPATCH_T(dtype, hipLaunchKernelGGL(gat_bwd_src_k<T>, dim3(grid), dim3(256), 0, stream, (const T*)dout, alpha, ws_dlogit, ws_der, al, ar,
                                         indptr_src, dst_by_src, slot_by_src, (T*)dft, ws_del, N, H, O));
    const int rpb = 64;
    dim3 g2((unsigned)cdiv((int64_t)H * O, 256), (unsigned)cdiv(N, rpb));
    DISPATCH_T(dtype, hipLaunchKernelGGL(gat_attn_grad_k<T>, g2, dim3(256), 0, stream, (const T*)ft, ws_del, ws_der, dal, dar, N, H, O, rpb));
    MV_LAUNCH_CHECK("gat_aggregate_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------ unbatch: pad / truncate to max_node rows
// out[b,i,:] = i < min(n_b, max_node) ? h[off_b + i, :] : 0        (GraphModel.py:30-54)
template <typename T>
__global__ void segment_pad_fwd_k(const T* __restrict__ h, const int* __restrict__ off, T* __restrict__ out, int B, int maxn, int F) {
    const int64_t total = (int64_t)B * maxn * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % F), r = (int)((i / F) % maxn), b = (int)(i / ((int64_t)F * maxn));
        const int n = off[b + 1] - off[b];
        if (r < n) out[i] = h[((int64_t)off[b] + r) * F + f];
        else stf(out + i, 0.f);
    }
}
// dh[off_b + i, :] = i < max_node ? dout[b,i,:] : 0
template <typename T>
__global__ void segment_pad_bwd_k(const T* __restrict__ dout, const int* __restrict__ off, T* __restrict__ dh, int B, int maxn, int F,
                                  int64_t ntot) {
    const int64_t total = ntot * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % F);
        const int node = (int)(i / F);
        int lo = 0, hi = B;                                   // largest b with off[b] <= node
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= node) lo = mid; else hi = mid; }
        const int r = node - off[lo];
        if (r < maxn) dh[i] = dout[((int64_t)lo * maxn + r) * F + f];
        else stf(dh + i, 0.f);
    }
}
extern "C" int mvuld_segment_pad_fwd(const void* h, const int* node_offsets, void* out, int B, int maxn, int F, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(h && node_offsets && out && B > 0 && maxn > 0 && F > 0, "segment_pad_fwd: bad args");
    const int grid = (int)min((int64_t)4096, cdiv((int64_t)B * maxn * F, 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL(segment_pad_fwd_k<T>, dim3(grid), dim3(256), 0, stream, (const T*)h, node_offsets, (T*)out, B, maxn, F));
    MV_LAUNCH_CHECK("segment_pad_fwd");
    return 0;
}
extern "C" int mvuld_segment_pad_bwd(const void* dout, const int* node_offsets, void* dh, int B, int maxn, int F, int64_t total_nodes, int dtype,
                                     hipStream_t stream) {
    MV_CHECK_ARG(dout && node_offsets && dh && B > 0 && maxn > 0 && F > 0 && total_nodes > 0, "segment_pad_bwd: bad args");
    const int grid = (int)min((int64_t)4096, cdiv(total_nodes * F, 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL(segment_pad_bwd_k<T>, dim3(grid), dim3(256), 0, stream, (const T*)dout, node_offsets, (T*)dh, B, maxn, F, total_nodes));
    MV_LAUNCH_CHECK("segment_pad_bwd");
    return 0;
}
