// HBM-bound data-movement / reduction kernels of the hot path (f32|bf16 storage, fp32 math).
#include "common.h"

#define DISPATCH_T(dtype, CALL)                 \
    do {                                        \
        if ((dtype) == MVULD_F32) { typedef float T; CALL; } \
        else { typedef bf16 T; CALL; }          \
    } while (0)

// ------------------------------------------------------------------------------------ transpose
// dst[b][c][r] = src[b][r][c]; 64x64 tiles through LDS (padded), coalesced on both sides.
template <typename T>
__global__ __launch_bounds__(256) void transpose_k(const T* __restrict__ src, T* __restrict__ dst, int R, int C) {
    __shared__ float tile[64][65];
    const int64_t boff = (int64_t)blockIdx.z * R * C;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? ldf(src + boff + (int64_t)r * C + c) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) stf(dst + boff + (int64_t)c * R + r, tile[tx][i]);
    }
}

extern "C" int mvuld_transpose(const void* src, void* dst, int R, int C, int batch, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(src && dst && R > 0 && C > 0 && batch > 0, "transpose: bad args");
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)cdiv(R, 64), batch);
    MV_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "transpose: grid too large");
    DISPATCH_T(dtype, hipLaunchKernelGGL(transpose_k<T>, grid, dim3(256), 0, stream, (const T*)src, (T*)dst, R, C));
    MV_LAUNCH_CHECK("transpose");
    return 0;
}

// Many independent transposes in one launch (the transposed weight copies the dgrad GEMMs read, refreshed after every
// optimizer step): jobs[j] = {src, dst, R, C, first tile}; one 64x64 tile per workgroup, job found by binary search.
struct TransposeJob { const void* src; void* dst; int64_t R, C, tile0; };

template <typename T>
__global__ __launch_bounds__(256) void transpose_batched_k(const TransposeJob* __restrict__ jobs, int njobs) {
    __shared__ float tile[64][65];
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].tile0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TransposeJob jb = jobs[lo];
    const int R = (int)jb.R, C = (int)jb.C;
    const int t = (int)(blockIdx.x - jb.tile0), tc = (C + 63) / 64;
    const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
    const T* src = (const T*)jb.src;
    T* dst = (T*)jb.dst;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
        const int r = min(r0 + i, R - 1), c = min(c0 + tx, C - 1);
        tile[i][tx] = ldf(src + (int64_t)r * C + c);
    }
    __syncthreads();
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) stf(dst + (int64_t)c * R + r, tile[tx][i]);
    }
}

extern "C" int mvuld_transpose_batched(const void* jobs, int njobs, int64_t total_tiles, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(jobs && njobs > 0 && total_tiles > 0 && total_tiles < 2147483647LL, "transpose_batched: bad args");
    DISPATCH_T(dtype, hipLaunchKernelGGL(transpose_batched_k<T>, dim3((unsigned)total_tiles), dim3(256), 0, stream, (const TransposeJob*)jobs, njobs));
    MV_LAUNCH_CHECK("transpose_batched");
    return 0;
}

// ------------------------------------------------------------------------------------ column sum (bias grads)
// out[c] += sum_r x[r*ld + c]   (atomic accumulate into fp32)
template <typename T>
__global__ __launch_bounds__(256) void colsum_k(const T* __restrict__ x, int64_t ld, float* __restrict__ out, int64_t M, int N,
                                                int64_t rows_per_block) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (c < N)
        for (int64_t r = r0 + ty; r < r1; r += 4) s += ldf(x + r * ld + c);
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < N) atomicAdd(out + c, red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
}

// bf16, 16-byte loads: 8 lanes cover 64 columns of a row (one 128-byte line), 32 rows per pass
__global__ __launch_bounds__(256) void colsum_vec_k(const bf16* __restrict__ x, int64_t ld, float* __restrict__ out, int64_t M, int N,
                                                    int64_t rows_per_block) {
    __shared__ float red[32][65];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c0 = blockIdx.x * 64 + tx * 8;
    const int cc = min(c0, N - 8);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t r = r0 + ty; r < r1; r += 32) {
        const bf16x8 v = *(const bf16x8*)(x + r * ld + cc);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)v.v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
        const int c = blockIdx.x * 64 + threadIdx.x;
        if (c < N) atomicAdd(out + c, t);
    }
}

extern "C" int mvuld_colsum(const void* x, int64_t ld, float* out, int64_t M, int N, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(x && out && M > 0 && N > 0 && ld >= N, "colsum: bad args");
    int64_t nby = min((int64_t)512, cdiv(M, 64));
    const int64_t rpb = cdiv(M, nby);
    nby = cdiv(M, rpb);
    dim3 grid((unsigned)cdiv(N, 64), (unsigned)nby);
    if (dtype == MVULD_BF16 && N % 8 == 0 && ld % 8 == 0 && ((uintptr_t)x & 15) == 0) {
        hipLaunchKernelGGL(colsum_vec_k, grid, dim3(256), 0, stream, (const bf16*)x, ld, out, M, N, rpb);
        MV_LAUNCH_CHECK("colsum_vec");
        return 0;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(colsum_k<T>, grid, dim3(256), 0, stream, (const T*)x, ld, out, M, N, rpb));
    MV_LAUNCH_CHECK("colsum");
    return 0;
}

// ------------------------------------------------------------------------------------ activation backward
// mode 0: GELU(erf), ref = pre-activation.  mode 1: ELU(alpha=1), ref = forward output.
template <typename T>
__global__ void act_bwd_k(const T* __restrict__ dy, const T* __restrict__ ref, T* __restrict__ dx, int64_t n, int mode) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float r = ldf(ref + i);
        const float d = mode == 0 ? dgelu_erf(r) : (r > 0.f ? 1.0f : r + 1.0f);
        stf(dx + i, ldf(dy + i) * d);
    }
}
extern "C" int mvuld_act_bwd(const void* dy, const void* ref, void* dx, int64_t n, int mode, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(dy && ref && dx && n > 0 && (mode == 0 || mode == 1), "act_bwd: bad args");
    const int grid = (int)min((int64_t)4096, cdiv(n, 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL(act_bwd_k<T>, dim3(grid), dim3(256), 0, stream, (const T*)dy, (const T*)ref, (T*)dx, n, mode));
    MV_LAUNCH_CHECK("act_bwd");
    return 0;
}

// elementwise ELU forward (rarely needed outside GEMM epilogues)
template <typename T>
__global__ void elu_fwd_k(const T* __restrict__ x, T* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        stf(y + i, elu1(ldf(x + i)));
}
extern "C" int mvuld_elu_fwd(const void* x, void* y, int64_t n, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(x && y && n > 0, "elu_fwd: bad args");
    const int grid = (int)min((int64_t)4096, cdiv(n, 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL(elu_fwd_k<T>, dim3(grid), dim3(256), 0, stream, (const T*)x, (T*)y, n));
    MV_LAUNCH_CHECK("elu_fwd");
    return 0;
}

// ------------------------------------------------------------------------------------ casts / axpy
template <typename TI, typename TO>
__global__ void cast_k(const TI* __restrict__ x, TO* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        stf(y + i, ldf(x + i));
}
extern "C" int mvuld_cast(const void* x, int dtype_in, void* y, int dtype_out, int64_t n, hipStream_t stream) {
    MV_CHECK_ARG(x && y && n > 0, "cast: bad args");
    const int grid = (int)min((int64_t)8192, cdiv(n, 256));
    if (dtype_in == MVULD_F32 && dtype_out == MVULD_BF16) hipLaunchKernelGGL((cast_k<float, bf16>), dim3(grid), dim3(256), 0, stream, (const float*)x, (bf16*)y, n);
    else if (dtype_in == MVULD_BF16 && dtype_out == MVULD_F32) hipLaunchKernelGGL((cast_k<bf16, float>), dim3(grid), dim3(256), 0, stream, (const bf16*)x, (float*)y, n);
    else if (dtype_in == MVULD_F32) hipLaunchKernelGGL((cast_k<float, float>), dim3(grid), dim3(256), 0, stream, (const float*)x, (float*)y, n);
    else hipLaunchKernelGGL((cast_k<bf16, bf16>), dim3(grid), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, n);
    MV_LAUNCH_CHECK("cast");
    return 0;
}

// y[i] = a[i] + b[i]
template <typename T>
__global__ void add_k(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        stf(y + i, ldf(a + i) + ldf(b + i));
}
extern "C" int mvuld_add(const void* a, const void* b, void* y, int64_t n, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(a && b && y && n > 0, "add: bad args");
    const int grid = (int)min((int64_t)8192, cdiv(n, 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL(add_k<T>, dim3(grid), dim3(256), 0, stream, (const T*)a, (const T*)b, (T*)y, n));
    MV_LAUNCH_CHECK("add");
    return 0;
}

// y[i] = a[i] * b[i]   (feature product of the noGlobalImage ablation head, new_model.py:196; its backward is two more calls)
template <typename T>
__global__ void mul_k(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        stf(y + i, ldf(a + i) * ldf(b + i));
}
extern "C" int mvuld_mul(const void* a, const void* b, void* y, int64_t n, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(a && b && y && n > 0, "mul: bad args");
    const int grid = (int)min((int64_t)8192, cdiv(n, 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL(mul_k<T>, dim3(grid), dim3(256), 0, stream, (const T*)a, (const T*)b, (T*)y, n));
    MV_LAUNCH_CHECK("mul");
    return 0;
}

// ------------------------------------------------------------------------------------ dropout (counter-based, recomputable)
// y = x * keep(i) / (1-p), keep(i) = hash(seed, i) >= p.  Same call on the gradient in backward.
// seed_off (optional): one uint64 on the device added to the host seed -- a step counter that lives in device memory, so that a
// hipGraph replay of a captured training step draws fresh masks (the host seed is frozen into the captured kernel arguments).
template <typename T>
__global__ void dropout_k(const T* __restrict__ x, T* __restrict__ y, int64_t n, float p, uint64_t seed, const uint64_t* __restrict__ seed_off) {
    if (seed_off) seed += seed_off[0] * 0xD1B54A32D192ED03ULL;
    const float inv = 1.0f / (1.0f - p);
    const uint32_t thr = (uint32_t)(p * 4294967296.0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool keep = mix32(seed + (uint64_t)i * 0x9E3779B97F4A7C15ULL) >= thr;
        stf(y + i, keep ? ldf(x + i) * inv : 0.f);
    }
}
// the same mask, eight bf16 per lane through 16-byte accesses (n % 8 == 0, 16-byte aligned): the text encoder's hidden dropouts
__global__ void dropout_bf16x8_k(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t n8, float p, uint64_t seed,
                                 const uint64_t* __restrict__ seed_off) {
    if (seed_off) seed += seed_off[0] * 0xD1B54A32D192ED03ULL;
    const float inv = 1.0f / (1.0f - p);
    const uint32_t thr = (uint32_t)(p * 4294967296.0);
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n8; c += (int64_t)gridDim.x * blockDim.x) {
        bf16x8 v = *(const bf16x8*)(x + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool keep = mix32(seed + (uint64_t)(c * 8 + e) * 0x9E3779B97F4A7C15ULL) >= thr;
            v.v[e] = (bf16)(keep ? (float)v.v[e] * inv : 0.f);
        }
        *(bf16x8*)(y + c * 8) = v;
    }
}
extern "C" int mvuld_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, const uint64_t* seed_offset, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(x && y && n > 0 && p >= 0.f && p < 1.f, "dropout: bad args");
    if (dtype == MVULD_BF16 && n % 8 == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
        const int grid = (int)min((int64_t)8192, cdiv(n / 8, 256));
        hipLaunchKernelGGL(dropout_bf16x8_k, dim3(grid), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, n / 8, p, seed, seed_offset);
    } else {
        const int grid = (int)min((int64_t)8192, cdiv(n, 256));
        DISPATCH_T(dtype, hipLaunchKernelGGL(dropout_k<T>, dim3(grid), dim3(256), 0, stream, (const T*)x, (T*)y, n, p, seed, seed_offset));
    }
    MV_LAUNCH_CHECK("dropout");
    return 0;
}

// ------------------------------------------------------------------------------------ Swin patch embed im2col
// img f32 [B,3,S,S] -> cols T [B*(S/4)^2, 48], column = c*16 + ky*4 + kx  (Conv2d weight [E,3,4,4] flattened)
template <typename T>
__global__ void im2col4_k(const float* __restrict__ img, T* __restrict__ cols, int B, int S) {
    const int P = S / 4;
    const int64_t total = (int64_t)B * P * P * 12;            // one thread per (token, c, ky): 4 contiguous pixels
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ck = (int)(i % 12);
        const int64_t tok = i / 12;
        const int c = ck / 4, ky = ck % 4;
        const int px = (int)(tok % P), py = (int)((tok / P) % P), b = (int)(tok / ((int64_t)P * P));
        const float4 v = *(const float4*)(img + (((int64_t)b * 3 + c) * S + py * 4 + ky) * S + px * 4);
        T* o = cols + tok * 48 + c * 16 + ky * 4;
        stf(o, v.x); stf(o + 1, v.y); stf(o + 2, v.z); stf(o + 3, v.w);
    }
}
extern "C" int mvuld_im2col_patch4(const float* img, void* cols, int B, int S, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(img && cols && B > 0 && S > 0 && S % 4 == 0, "im2col_patch4: bad args");
    const int64_t total = (int64_t)B * (S / 4) * (S / 4) * 12;
    const int grid = (int)min((int64_t)8192, cdiv(total, 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL(im2col4_k<T>, dim3(grid), dim3(256), 0, stream, img, (T*)cols, B, S));
    MV_LAUNCH_CHECK("im2col_patch4");
    return 0;
}

// ------------------------------------------------------------------------------------ PatchMerging gather / its inverse
// forward: y[b, i, j, q*C + c] = x[b, 2i + (q&1), 2j + (q>>1), c]   q = 0..3  (x0,x1,x2,x3 of swin_transformer_v2.py:354-358)
// inverse (backward): the same index map, copying y -> x.
template <typename T>
__global__ void patch_merge_k(const T* __restrict__ src, T* __restrict__ dst, int B, int res, int C, int inverse) {
    const int h = res / 2;
    const int64_t total = (int64_t)B * h * h * 4 * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int q = (int)((i / C) % 4);
        const int64_t t = i / (4 * (int64_t)C);
        const int j = (int)(t % h), ii = (int)((t / h) % h), b = (int)(t / ((int64_t)h * h));
        const int64_t xi = (((int64_t)b * res + 2 * ii + (q & 1)) * res + 2 * j + (q >> 1)) * C + c;
        if (inverse) dst[xi] = src[i];
        else dst[i] = src[xi];
    }
}
// bf16, C % 8 == 0: one 16-byte chunk per thread (index arithmetic once per 8 channels)
__global__ __launch_bounds__(256) void patch_merge_vec_k(const bf16* __restrict__ src, bf16* __restrict__ dst, int B, int res, int C8, int inverse) {
    const int h = res / 2;
    const int64_t total = (int64_t)B * h * h * 4 * C8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        const int q = (int)((i / C8) % 4);
        const int64_t t = i / (4 * (int64_t)C8);
        const int j = (int)(t % h), ii = (int)((t / h) % h), b = (int)(t / ((int64_t)h * h));
        const int64_t xi = (((int64_t)b * res + 2 * ii + (q & 1)) * res + 2 * j + (q >> 1)) * C8 + c;
        if (inverse) ((bf16x8*)dst)[xi] = ((const bf16x8*)src)[i];
        else ((bf16x8*)dst)[i] = ((const bf16x8*)src)[xi];
    }
}
extern "C" int mvuld_patch_merge_gather(const void* src, void* dst, int B, int res, int C, int inverse, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(src && dst && B > 0 && res > 0 && res % 2 == 0 && C > 0, "patch_merge_gather: bad args");
    const int64_t total = (int64_t)B * res * res * C;
    if (dtype == MVULD_BF16 && C % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const int gridv = (int)min((int64_t)16384, cdiv(total / 8, 256));
        hipLaunchKernelGGL(patch_merge_vec_k, dim3(gridv), dim3(256), 0, stream, (const bf16*)src, (bf16*)dst, B, res, C / 8, inverse);
        MV_LAUNCH_CHECK("patch_merge_gather_vec");
        return 0;
    }
    const int grid = (int)min((int64_t)8192, cdiv(total, 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL(patch_merge_k<T>, dim3(grid), dim3(256), 0, stream, (const T*)src, (T*)dst, B, res, C, inverse));
    MV_LAUNCH_CHECK("patch_merge_gather");
    return 0;
}

// ------------------------------------------------------------------------------------ RoBERTa embeddings
// position ids = cumsum(ids != pad) * (ids != pad) + pad ; valid = ids != pad
__global__ void position_ids_k(const int64_t* __restrict__ ids, int* __restrict__ pos, int* __restrict__ valid, int B, int L, int pad) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int run = 0;
    for (int l = 0; l < L; ++l) {
        const int ok = ids[(int64_t)b * L + l] != pad;
        run += ok;
        pos[(int64_t)b * L + l] = ok ? run + pad : pad;
        valid[(int64_t)b * L + l] = ok;
    }
}
extern "C" int mvuld_position_ids(const int64_t* ids, int* pos, int* valid, int B, int L, int pad, hipStream_t stream) {
    MV_CHECK_ARG(ids && pos && valid && B > 0 && L > 0, "position_ids: bad args");
    hipLaunchKernelGGL(position_ids_k, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, stream, ids, pos, valid, B, L, pad);
    MV_LAUNCH_CHECK("position_ids");
    return 0;
}

template <typename T>
__global__ void embed_fwd_k(const int64_t* __restrict__ ids, const int* __restrict__ pos, const float* __restrict__ word,
                            const float* __restrict__ posw, const float* __restrict__ type0, T* __restrict__ out, int64_t ntok,
                            int H, int vocab, int maxpos) {
    for (int64_t t = blockIdx.x; t < ntok; t += gridDim.x) {
        int64_t w = ids[t];
        int p = pos[t];
        w = w < 0 ? 0 : (w >= vocab ? vocab - 1 : w);
        p = p < 0 ? 0 : (p >= maxpos ? maxpos - 1 : p);
        for (int c = threadIdx.x; c < H; c += blockDim.x)
            stf(out + t * H + c, word[w * H + c] + posw[(int64_t)p * H + c] + type0[c]);
    }
}
template <typename T>
__global__ void embed_bwd_k(const int64_t* __restrict__ ids, const int* __restrict__ pos, const T* __restrict__ dy,
                            float* __restrict__ dword, float* __restrict__ dposw, float* __restrict__ dtype0, int64_t ntok, int H,
                            int vocab, int maxpos) {
    for (int64_t t = blockIdx.x; t < ntok; t += gridDim.x) {
        int64_t w = ids[t];
        int p = pos[t];
        w = w < 0 ? 0 : (w >= vocab ? vocab - 1 : w);
        p = p < 0 ? 0 : (p >= maxpos ? maxpos - 1 : p);
        for (int c = threadIdx.x; c < H; c += blockDim.x) {
            const float g = ldf(dy + t * H + c);
            atomicAdd(dword + w * H + c, g);
            atomicAdd(dposw + (int64_t)p * H + c, g);
        }
    }
}
extern "C" int mvuld_embed_fwd(const int64_t* ids, const int* pos, const float* word, const float* posw, const float* type0,
                               void* out, int64_t ntok, int H, int vocab, int maxpos, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(ids && pos && word && posw && type0 && out && ntok > 0 && H > 0, "embed_fwd: bad args");
    const int grid = (int)min((int64_t)4096, ntok);
    DISPATCH_T(dtype, hipLaunchKernelGGL(embed_fwd_k<T>, dim3(grid), dim3(256), 0, stream, ids, pos, word, posw, type0, (T*)out, ntok, H, vocab, maxpos));
    MV_LAUNCH_CHECK("embed_fwd");
    return 0;
}
// dtype0 (token-type row 0) is the column sum of dy: call mvuld_colsum for it.
extern "C" int mvuld_embed_bwd(const int64_t* ids, const int* pos, const void* dy, float* dword, float* dposw, int64_t ntok, int H,
                               int vocab, int maxpos, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(ids && pos && dy && dword && dposw && ntok > 0 && H > 0, "embed_bwd: bad args");
    const int grid = (int)min((int64_t)4096, ntok);
    DISPATCH_T(dtype, hipLaunchKernelGGL(embed_bwd_k<T>, dim3(grid), dim3(256), 0, stream, ids, pos, (const T*)dy, dword, dposw, nullptr, ntok, H, vocab, maxpos));
    MV_LAUNCH_CHECK("embed_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------ (masked) mean pool over tokens
// out[b,c] = sum_l x[b,l,c]*m[b,l] / sum_l m[b,l]   (m == null: plain mean; AdaptiveAvgPool1d / unixcoder.py:37)
template <typename T>
__global__ __launch_bounds__(256) void mean_pool_fwd_k(const T* __restrict__ x, const int* __restrict__ valid, T* __restrict__ out,
                                                       int L, int C) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f, n = 0.f;
    for (int l = 0; l < L; ++l) {
        const float m = valid ? (float)valid[b * L + l] : 1.0f;
        s += ldf(x + ((int64_t)b * L + l) * C + c) * m;
        n += m;
    }
    stf(out + (int64_t)b * C + c, s / n);
}
template <typename T>
__global__ __launch_bounds__(256) void mean_pool_bwd_k(const T* __restrict__ dout, const int* __restrict__ valid, T* __restrict__ dx,
                                                       int L, int C) {
    __shared__ float red[4];
    const int b = blockIdx.z, l = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    float n = 0.f;                                  // number of valid tokens of this row: block-wide sum
    if (valid) for (int i = threadIdx.x; i < L; i += 256) n += valid[b * L + i];
    n = wave_sum(n);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
    __syncthreads();
    const float cnt = valid ? red[0] + red[1] + red[2] + red[3] : (float)L;
    if (c >= C) return;
    const float m = valid ? (float)valid[b * L + l] : 1.0f;
    stf(dx + ((int64_t)b * L + l) * C + c, ldf(dout + (int64_t)b * C + c) * m / cnt);
}
extern "C" int mvuld_mean_pool_fwd(const void* x, const int* valid, void* out, int B, int L, int C, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(x && out && B > 0 && L > 0 && C > 0, "mean_pool_fwd: bad args");
    dim3 grid((unsigned)cdiv(C, 256), B);
    DISPATCH_T(dtype, hipLaunchKernelGGL(mean_pool_fwd_k<T>, grid, dim3(256), 0, stream, (const T*)x, valid, (T*)out, L, C));
    MV_LAUNCH_CHECK("mean_pool_fwd");
    return 0;
}
extern "C" int mvuld_mean_pool_bwd(const void* dout, const int* valid, void* dx, int B, int L, int C, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(dout && dx && B > 0 && L > 0 && C > 0 && L <= 65535 && B <= 65535, "mean_pool_bwd: bad args");
    dim3 grid((unsigned)cdiv(C, 256), L, B);
    DISPATCH_T(dtype, hipLaunchKernelGGL(mean_pool_bwd_k<T>, grid, dim3(256), 0, stream, (const T*)dout, valid, (T*)dx, L, C));
    MV_LAUNCH_CHECK("mean_pool_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------ l2norm over the node axis + mean over nodes
// g [B,Nn,C]: nrm[b,c] = sqrt(sum_i g^2) (no eps, GraphModel.py:74-79), hf[b,c] = mean_i g/nrm  (:201-204)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_mean_fwd_k(const T* __restrict__ g, T* __restrict__ hf, float* __restrict__ ssum,
                                                         float* __restrict__ snrm, int Nn, int C) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f, q = 0.f;
    for (int i = 0; i < Nn; ++i) { const float v = ldf(g + ((int64_t)b * Nn + i) * C + c); s += v; q += v * v; }
    const float nrm = sqrtf(q);
    ssum[(int64_t)b * C + c] = s; snrm[(int64_t)b * C + c] = nrm;
    stf(hf + (int64_t)b * C + c, s / (nrm * Nn));
}
template <typename T>
__global__ __launch_bounds__(256) void l2norm_mean_bwd_k(const T* __restrict__ g, const T* __restrict__ dhf, const float* __restrict__ ssum,
                                                         const float* __restrict__ snrm, T* __restrict__ dg, int Nn, int C) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float d = ldf(dhf + (int64_t)b * C + c) / Nn, s = ssum[(int64_t)b * C + c], nrm = snrm[(int64_t)b * C + c];
    const float a = d / nrm, bb = d * s / (nrm * nrm * nrm);
    for (int i = 0; i < Nn; ++i) {
        const int64_t o = ((int64_t)b * Nn + i) * C + c;
        stf(dg + o, a - bb * ldf(g + o));
    }
}
extern "C" int mvuld_l2norm_mean_fwd(const void* g, void* hf, float* ssum, float* snrm, int B, int Nn, int C, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(g && hf && ssum && snrm && B > 0 && Nn > 0 && C > 0, "l2norm_mean_fwd: bad args");
    dim3 grid((unsigned)cdiv(C, 256), B);
    DISPATCH_T(dtype, hipLaunchKernelGGL(l2norm_mean_fwd_k<T>, grid, dim3(256), 0, stream, (const T*)g, (T*)hf, ssum, snrm, Nn, C));
    MV_LAUNCH_CHECK("l2norm_mean_fwd");
    return 0;
}
extern "C" int mvuld_l2norm_mean_bwd(const void* g, const void* dhf, const float* ssum, const float* snrm, void* dg, int B, int Nn, int C,
                                     int dtype, hipStream_t stream) {
    MV_CHECK_ARG(g && dhf && ssum && snrm && dg && B > 0 && Nn > 0 && C > 0, "l2norm_mean_bwd: bad args");
    dim3 grid((unsigned)cdiv(C, 256), B);
    DISPATCH_T(dtype, hipLaunchKernelGGL(l2norm_mean_bwd_k<T>, grid, dim3(256), 0, stream, (const T*)g, (const T*)dhf, ssum, snrm, (T*)dg, Nn, C));
    MV_LAUNCH_CHECK("l2norm_mean_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------ softmax + cross-entropy (fp32 logits)
// loss += mean_b -log softmax(logits[b])[target[b]] * loss_scale ; probs = softmax ; dlogits = (probs - onehot) * loss_scale / B
__global__ void ce_k(const float* __restrict__ logits, const int64_t* __restrict__ target, float* __restrict__ loss,
                     float* __restrict__ probs, float* __restrict__ dlogits, int B, int K, float loss_scale) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, logits[b * K + k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(logits[b * K + k] - m);
    const int t = (int)target[b];
    for (int k = 0; k < K; ++k) {
        const float p = expf(logits[b * K + k] - m) / s;
        if (probs) probs[b * K + k] = p;
        if (dlogits) dlogits[b * K + k] = (p - (k == t ? 1.f : 0.f)) * loss_scale / B;
    }
    atomicAdd(loss, (logf(s) + m - logits[b * K + t]) * loss_scale / B);
}
extern "C" int mvuld_cross_entropy(const float* logits, const int64_t* target, float* loss, float* probs, float* dlogits, int B, int K,
                                   float loss_scale, hipStream_t stream) {
    MV_CHECK_ARG(logits && target && loss && B > 0 && K > 0, "cross_entropy: bad args");
    hipLaunchKernelGGL(ce_k, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, stream, logits, target, loss, probs, dlogits, B, K, loss_scale);
    MV_LAUNCH_CHECK("cross_entropy");
    return 0;
}

// soft targets (timm SoftTargetCrossEntropy / LabelSmoothingCrossEntropy: main.py:136-140): loss += mean_b sum_k -t[b,k] log softmax(x[b])[k]
// * loss_scale ; dlogits = (probs * sum_k t[b,k] - t[b]) * loss_scale / B.  `smoothing` > 0 with integer targets (target_i != null):
// t = one-hot * (1 - smoothing) + smoothing / K  (LabelSmoothingCrossEntropy: nll * confidence + mean(-logprobs) * smoothing).
__global__ void ce_soft_k(const float* __restrict__ logits, const float* __restrict__ target, const int64_t* __restrict__ target_i, float smoothing,
                          float* __restrict__ loss, float* __restrict__ probs, float* __restrict__ dlogits, int B, int K, float loss_scale) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, logits[b * K + k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(logits[b * K + k] - m);
    const float lz = logf(s) + m;
    const int ti = target_i ? (int)target_i[b] : -1;
    float tsum = 0.f, l = 0.f;
    for (int k = 0; k < K; ++k) {
        const float t = target_i ? ((k == ti ? 1.0f - smoothing : 0.f) + smoothing / K) : target[b * K + k];
        tsum += t;
        l += t * (lz - logits[b * K + k]);
    }
    for (int k = 0; k < K; ++k) {
        const float p = expf(logits[b * K + k] - m) / s;
        const float t = target_i ? ((k == ti ? 1.0f - smoothing : 0.f) + smoothing / K) : target[b * K + k];
        if (probs) probs[b * K + k] = p;
        if (dlogits) dlogits[b * K + k] = (p * tsum - t) * loss_scale / B;
    }
    atomicAdd(loss, l * loss_scale / B);
}
extern "C" int mvuld_cross_entropy_soft(const float* logits, const float* target, const int64_t* target_i, float smoothing, float* loss,
                                        float* probs, float* dlogits, int B, int K, float loss_scale, hipStream_t stream) {
    MV_CHECK_ARG(logits && (target || target_i) && loss && B > 0 && K > 0, "cross_entropy_soft: bad args");
    MV_CHECK_ARG(smoothing >= 0.f && smoothing < 1.f, "cross_entropy_soft: 0 <= smoothing < 1");
    hipLaunchKernelGGL(ce_soft_k, dim3((unsigned)cdiv(B, 64)), dim3(64), 0, stream, logits, target, target_i, smoothing, loss, probs, dlogits, B, K,
                       loss_scale);
    MV_LAUNCH_CHECK("cross_entropy_soft");
    return 0;
}

// ------------------------------------------------------------------------------------ Mixup / CutMix, batch mode (timm.data.Mixup, main.py:268-269)
// y[b] = lam * x[b] + (1 - lam) * x[B-1-b]   (mixup)   or   x[b] with the box [yl, yh) x [xl, xh) taken from x[B-1-b]   (cutmix);
// images [B, C, H, W] contiguous, 8 elements per lane where the row length allows.  The soft targets go with it:
// t[b] = lam * smooth_onehot(target[b]) + (1 - lam) * smooth_onehot(target[B-1-b]),  on = 1 - smoothing + off, off = smoothing / K.
template <typename T>
__global__ __launch_bounds__(256) void mixup_k(const T* __restrict__ x, T* __restrict__ y, int B, int64_t per, int H, int W, float lam, int cutmix,
                                               int yl, int yh, int xl, int xh) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)B * per;
    if (i >= total) return;
    const int b = (int)(i / per);
    const int64_t r = i - (int64_t)b * per;
    const int64_t j = (int64_t)(B - 1 - b) * per + r;
    if (cutmix) {
        const int w = (int)(r % W), h = (int)((r / W) % H);
        y[i] = (h >= yl && h < yh && w >= xl && w < xh) ? x[j] : x[i];
    } else {
        y[i] = (T)(lam * (float)x[i] + (1.0f - lam) * (float)x[j]);
    }
}
__global__ void mixup_target_k(const int64_t* __restrict__ target, float* __restrict__ out, int B, int K, float lam, float smoothing) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K, k = i % K;
    const float off = smoothing / K, on = 1.0f - smoothing + off;
    const float a = (int)target[b] == k ? on : off, c = (int)target[B - 1 - b] == k ? on : off;
    out[i] = a * lam + c * (1.0f - lam);
}
extern "C" int mvuld_mixup_batch(const void* x, void* y, const int64_t* target, float* soft_target, int B, int C, int H, int W, int K, float lam,
                                 int cutmix, int yl, int yh, int xl, int xh, float smoothing, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(x && y && x != y && B > 0 && C > 0 && H > 0 && W > 0, "mixup_batch: bad args (out of place)");
    MV_CHECK_ARG(lam >= 0.f && lam <= 1.f && (!cutmix || (0 <= yl && yl <= yh && yh <= H && 0 <= xl && xl <= xh && xh <= W)), "mixup_batch: lam / box");
    const int64_t per = (int64_t)C * H * W, total = (int64_t)B * per;
    const unsigned grid = (unsigned)cdiv(total, 256);
    if (dtype == MVULD_F32)
        hipLaunchKernelGGL(mixup_k<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, (float*)y, B, per, H, W, lam, cutmix, yl, yh, xl, xh);
    else
        hipLaunchKernelGGL(mixup_k<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, B, per, H, W, lam, cutmix, yl, yh, xl, xh);
    if (target && soft_target) {
        MV_CHECK_ARG(K > 0 && smoothing >= 0.f && smoothing < 1.f, "mixup_batch: classes / smoothing");
        hipLaunchKernelGGL(mixup_target_k, dim3((unsigned)cdiv((int64_t)B * K, 64)), dim3(64), 0, stream, target, soft_target, B, K, lam, smoothing);
    }
    MV_LAUNCH_CHECK("mixup_batch");
    return 0;
}

// The same with one parameter row per sample (timm's "elem" and "pair" modes): params[b] = {lam, cutmix, yl, yh, xl, xh} as six floats.
// The partner is still B-1-b; in "pair" mode the host writes the same row for b and B-1-b, in "elem" mode every sample has its own.
template <typename T>
__global__ __launch_bounds__(256) void mixup_rows_k(const T* __restrict__ x, T* __restrict__ y, int B, int64_t per, int H, int W,
                                                    const float* __restrict__ params) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * per) return;
    const int b = (int)(i / per);
    const int64_t r = i - (int64_t)b * per;
    const int64_t j = (int64_t)(B - 1 - b) * per + r;
    const float* p = params + 6 * b;
    const float lam = p[0];
    if (p[1] != 0.f) {
        const int w = (int)(r % W), h = (int)((r / W) % H);
        y[i] = (h >= (int)p[2] && h < (int)p[3] && w >= (int)p[4] && w < (int)p[5]) ? x[j] : x[i];
    } else if (lam == 1.0f) {
        y[i] = x[i];
    } else {
        y[i] = (T)(lam * (float)x[i] + (1.0f - lam) * (float)x[j]);
    }
}
__global__ void mixup_target_rows_k(const int64_t* __restrict__ target, float* __restrict__ out, int B, int K, const float* __restrict__ params,
                                    float smoothing) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K, k = i % K;
    const float lam = params[6 * b];
    const float off = smoothing / K, on = 1.0f - smoothing + off;
    const float a = (int)target[b] == k ? on : off, c = (int)target[B - 1 - b] == k ? on : off;
    out[i] = a * lam + c * (1.0f - lam);
}
extern "C" int mvuld_mixup_rows(const void* x, void* y, const int64_t* target, float* soft_target, int B, int C, int H, int W, int K,
                                const float* params, float smoothing, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(x && y && x != y && params && B > 0 && C > 0 && H > 0 && W > 0, "mixup_rows: bad args (out of place)");
    const int64_t per = (int64_t)C * H * W, total = (int64_t)B * per;
    const unsigned grid = (unsigned)cdiv(total, 256);
    if (dtype == MVULD_F32) hipLaunchKernelGGL(mixup_rows_k<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, (float*)y, B, per, H, W, params);
    else hipLaunchKernelGGL(mixup_rows_k<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, B, per, H, W, params);
    if (target && soft_target) {
        MV_CHECK_ARG(K > 0 && smoothing >= 0.f && smoothing < 1.f, "mixup_rows: classes / smoothing");
        hipLaunchKernelGGL(mixup_target_rows_k, dim3((unsigned)cdiv((int64_t)B * K, 64)), dim3(64), 0, stream, target, soft_target, B, K, params, smoothing);
    }
    MV_LAUNCH_CHECK("mixup_rows");
    return 0;
}

// ------------------------------------------------------------------------------------ Swin continuous position bias table
// table16[i,h] = 16*sigmoid( W2[h,:] . relu(W1 . coords[i] + b1) )   coords [T2,2], W1 [512,2], b1 [512], W2 [H,512]
// (swin_transformer_v2.py:159-163).  hidden [T2,512] is kept for the backward.
__global__ __launch_bounds__(256) void cpb_fwd_k(const float* __restrict__ coords, const float* __restrict__ W1, const float* __restrict__ b1,
                                                 const float* __restrict__ W2, float* __restrict__ hidden, float* __restrict__ table16,
                                                 int T2, int H) {
    __shared__ float hid[512];
    const int i = blockIdx.x;
    const float cy = coords[2 * i], cx = coords[2 * i + 1];
    for (int j = threadIdx.x; j < 512; j += 256) {
        const float v = fmaxf(W1[2 * j] * cy + W1[2 * j + 1] * cx + b1[j], 0.f);
        hid[j] = v;
        hidden[(int64_t)i * 512 + j] = v;
    }
    __syncthreads();
    // one wave per head (4 in flight): 512 products as 8 per lane + a wave reduction, no block barriers
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int h = wv; h < H; h += 4) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s = fmaf(hid[lane + 64 * k], W2[h * 512 + lane + 64 * k], s);
        s = wave_sum(s);
        if (lane == 0) table16[(int64_t)i * H + h] = 16.0f / (1.0f + __expf(-s));
    }
}
// dz[i,h] = dtable16 * t16*(1 - t16/16);  dW2[h,j] += dz*hid[i,j];  dhid[j] = sum_h dz*W2[h,j] (relu gate);
// dW1[j,:] += dhid*coords[i];  db1[j] += dhid.   One block per chunk of CPB_ROWS table rows; thread <-> hidden
// units j and j+256; per-block partials leave through fp32 atomics.
#define CPB_ROWS 16
#define CPB_MAXH 32
// One block per chunk of CPB_ROWS table rows; thread <-> hidden units j and j+256.  The per-block sums leave either as one row of
// `part` ([blocks][HH*512 + 1536]: dW2 | dW1 | db1, summed by cpb_bwd_reduce_k) or, without a workspace, through fp32 atomics
// (~1 M contended atomics per launch: 4x slower).
template <int HH>
__global__ __launch_bounds__(256) void cpb_bwd_k(const float* __restrict__ coords, const float* __restrict__ W2, const float* __restrict__ hidden,
                                                 const float* __restrict__ table16, const float* __restrict__ dtable16,
                                                 float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2, int T2, int H,
                                                 float* __restrict__ part) {
    __shared__ float dz[CPB_ROWS][HH];
    __shared__ float cy[CPB_ROWS], cx[CPB_ROWS];
    const int i0 = blockIdx.x * CPB_ROWS;
    const int nrow = min(CPB_ROWS, T2 - i0);
    for (int e = threadIdx.x; e < CPB_ROWS * HH; e += 256) {
        const int r = e / HH, h = e % HH;
        float v = 0.f;
        if (r < nrow && h < H) {
            const float t = table16[(int64_t)(i0 + r) * H + h];
            v = dtable16[(int64_t)(i0 + r) * H + h] * t * (1.0f - t * 0.0625f);
        }
        dz[r][h] = v;
    }
    if (threadIdx.x < CPB_ROWS) {
        const int r = threadIdx.x;
        cy[r] = r < nrow ? coords[2 * (i0 + r)] : 0.f;
        cx[r] = r < nrow ? coords[2 * (i0 + r) + 1] : 0.f;
    }
    __syncthreads();
    float* prow = part ? part + (size_t)blockIdx.x * (HH * 512 + 1536) : nullptr;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int j = threadIdx.x + half * 256;
        float w2[HH], acc2[HH];
#pragma unroll
        for (int h = 0; h < HH; ++h) { w2[h] = h < H ? W2[h * 512 + j] : 0.f; acc2[h] = 0.f; }
        float a0 = 0.f, a1 = 0.f, ab = 0.f;
        float hv[CPB_ROWS];
#pragma unroll
        for (int r = 0; r < CPB_ROWS; ++r) hv[r] = hidden[(int64_t)min(i0 + r, T2 - 1) * 512 + j];      // all rows in flight at once
#pragma unroll
        for (int r = 0; r < CPB_ROWS; ++r) {
            float dh = 0.f;
#pragma unroll
            for (int h = 0; h < HH; ++h) { acc2[h] = fmaf(dz[r][h], hv[r], acc2[h]); dh = fmaf(dz[r][h], w2[h], dh); }
            if (hv[r] > 0.f) { a0 += dh * cy[r]; a1 += dh * cx[r]; ab += dh; }          // rows >= nrow carry dz = 0 -> dh = 0
        }
        if (prow) {
#pragma unroll
            for (int h = 0; h < HH; ++h) prow[h * 512 + j] = acc2[h];
            prow[HH * 512 + 2 * j] = a0; prow[HH * 512 + 2 * j + 1] = a1; prow[HH * 512 + 1024 + j] = ab;
        } else {
#pragma unroll
            for (int h = 0; h < HH; ++h)
                if (h < H) atomicAdd(dW2 + h * 512 + j, acc2[h]);
            atomicAdd(dW1 + 2 * j, a0); atomicAdd(dW1 + 2 * j + 1, a1); atomicAdd(db1 + j, ab);
        }
    }
}
// column sums of part [nblk][HH*512 + 1536] into dW2 [H*512] | dW1 [1024] | db1 [512]
__global__ __launch_bounds__(256) void cpb_bwd_reduce_k(const float* __restrict__ part, int nblk, int HH, int H, float* __restrict__ dW1,
                                                        float* __restrict__ db1, float* __restrict__ dW2) {
    const int W = HH * 512 + 1536;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= W) return;
    const int per = (nblk + gridDim.y - 1) / gridDim.y;                 // blockIdx.y = slab of partial rows
    const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = b0; b < b1; b += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { const float t = part[(size_t)min(b + u, nblk - 1) * W + c]; a[u] += (b + u < b1) ? t : 0.f; }
    }
    const float v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    if (c < HH * 512) { if (c / 512 < H) atomicAdd(dW2 + c, v); }
    else if (c < HH * 512 + 1024) atomicAdd(dW1 + c - HH * 512, v);
    else atomicAdd(db1 + c - HH * 512 - 1024, v);
}
extern "C" int mvuld_cpb_table_fwd(const float* coords, const float* W1, const float* b1, const float* W2, float* hidden, float* table16,
                                   int T2, int H, hipStream_t stream) {
    MV_CHECK_ARG(coords && W1 && b1 && W2 && hidden && table16 && T2 > 0 && H > 0, "cpb_table_fwd: bad args");
    hipLaunchKernelGGL(cpb_fwd_k, dim3(T2), dim3(256), 0, stream, coords, W1, b1, W2, hidden, table16, T2, H);
    MV_LAUNCH_CHECK("cpb_table_fwd");
    return 0;
}
/* bytes of fp32 scratch mvuld_cpb_table_bwd wants (one row of HH*512+1536 partial sums per block of CPB_ROWS table rows) */
extern "C" int64_t mvuld_cpb_table_bwd_workspace_bytes(int T2, int H) {
    const int HH = H <= 4 ? 4 : (H <= 8 ? 8 : (H <= 16 ? 16 : 32));
    return (int64_t)cdiv(T2 > 0 ? T2 : 0, CPB_ROWS) * (HH * 512 + 1536) * 4;
}
extern "C" int mvuld_cpb_table_bwd(const float* coords, const float* W2, const float* hidden, const float* table16, const float* dtable16,
                                   float* dW1, float* db1, float* dW2, int T2, int H, float* ws, int64_t ws_bytes, hipStream_t stream) {
    MV_CHECK_ARG(coords && W2 && hidden && table16 && dtable16 && dW1 && db1 && dW2 && T2 > 0 && H > 0 && H <= CPB_MAXH, "cpb_table_bwd: bad args (H<=32)");
    const int nblk = (int)cdiv(T2, CPB_ROWS);
    const int HH = H <= 4 ? 4 : (H <= 8 ? 8 : (H <= 16 ? 16 : 32));
    float* part = (ws && ws_bytes >= (int64_t)nblk * (HH * 512 + 1536) * 4) ? ws : nullptr;
#define CPB_BWD(HV) hipLaunchKernelGGL(cpb_bwd_k<HV>, dim3(nblk), dim3(256), 0, stream, coords, W2, hidden, table16, dtable16, dW1, db1, dW2, T2, H, part)
    if (HH == 4) CPB_BWD(4); else if (HH == 8) CPB_BWD(8); else if (HH == 16) CPB_BWD(16); else CPB_BWD(32);
#undef CPB_BWD
    if (part) hipLaunchKernelGGL(cpb_bwd_reduce_k, dim3((unsigned)cdiv(HH * 512 + 1536, 256), 8), dim3(256), 0, stream, part, nblk, HH, H, dW1, db1, dW2);
    MV_LAUNCH_CHECK("cpb_table_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------ fp32 -> 3 x bf16 split (near-fp32 GEMM on the bf16 matrix cores)
// x = hi + lo (hi = bf16(x), lo = bf16(x - hi)); a.b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi with fp32 accumulation (error ~2^-16).
// dst row = [P0 | P1 | P2], each Kp wide (Kp >= K, zero padded):  mode 0 (A operand): hi, lo, hi ; mode 1 (B operand): hi, hi, lo.
__global__ void split3_k(const float* __restrict__ src, int64_t ld, bf16* __restrict__ dst, int64_t rows, int K, int Kp, int mode) {
    const int64_t total = rows * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / Kp;
        const int k = (int)(i % Kp);
        const float x = k < K ? src[r * ld + k] : 0.f;
        const bf16 hi = (bf16)x;
        const bf16 lo = (bf16)(x - (float)hi);
        bf16* d = dst + r * 3 * Kp + k;
        d[0] = hi;
        d[Kp] = mode == 0 ? lo : hi;
        d[2 * Kp] = mode == 0 ? hi : lo;
    }
}
extern "C" int mvuld_split_bf16x3(const float* src, int64_t ld, void* dst, int64_t rows, int K, int Kp, int mode, hipStream_t stream) {
    MV_CHECK_ARG(src && dst && rows > 0 && K > 0 && Kp >= K && Kp % 8 == 0 && (mode == 0 || mode == 1), "split_bf16x3: bad args");
    const int grid = (int)min((int64_t)4096, cdiv(rows * Kp, 256));
    hipLaunchKernelGGL(split3_k, dim3(grid), dim3(256), 0, stream, src, ld, (bf16*)dst, rows, K, Kp, mode);
    MV_LAUNCH_CHECK("split_bf16x3");
    return 0;
}

// ------------------------------------------------------------------------------------ packed (pad-free) token sequences
// The reference pads every function / source line to 512 tokens (unixcoder.py:56-68, data_list.py:293-299) and masks the pad
// keys; pad QUERY rows are computed and thrown away (the sentence vector is a masked mean, unixcoder.py:37).  Packing keeps
// only the non-pad tokens: row cu[b] + r of the packed matrices is the r-th non-pad token of sequence b.
// One wave per sequence: ballot + prefix popcount over 64 positions at a time.  pos = HF create_position_ids_from_input_ids.
__global__ __launch_bounds__(64) void pack_tokens_k(const int64_t* __restrict__ ids, const int* __restrict__ cu, int64_t* __restrict__ ids_p,
                                                    int* __restrict__ pos_p, int* __restrict__ rowmap, int L, int pad) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int base = cu[b], cap = cu[b + 1] - base;
    int run = 0;
    for (int l0 = 0; l0 < L; l0 += 64) {
        const int l = l0 + lane;
        const int64_t id = l < L ? ids[(int64_t)b * L + l] : pad;
        const bool ok = l < L && id != pad;
        const unsigned long long m = __ballot(ok);
        const int rank = run + __popcll(m & ((1ull << lane) - 1ull));
        if (ok && rank < cap) {                    // rank < cap always holds when cu was built from these ids
            ids_p[base + rank] = id;
            pos_p[base + rank] = rank + 1 + pad;
            rowmap[base + rank] = b * L + l;
        }
        run += __popcll(m);
    }
}
extern "C" int mvuld_pack_tokens(const int64_t* ids, const int* cu, int64_t* ids_packed, int* pos_packed, int* rowmap, int B, int L, int pad,
                                 hipStream_t stream) {
    MV_CHECK_ARG(ids && cu && ids_packed && pos_packed && rowmap && B > 0 && L > 0, "pack_tokens: bad args");
    hipLaunchKernelGGL(pack_tokens_k, dim3(B), dim3(64), 0, stream, ids, cu, ids_packed, pos_packed, rowmap, L, pad);
    MV_LAUNCH_CHECK("pack_tokens");
    return 0;
}

// out[b] = mean of rows cu[b] .. cu[b+1]-1 (the sentence vector over a packed sequence); backward spreads dout[b] / len_b
template <typename T>
__global__ __launch_bounds__(256) void segment_mean_fwd_k(const T* __restrict__ x, const int* __restrict__ cu, T* __restrict__ out, int C) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = cu[b], r1 = cu[b + 1];
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += ldf(x + (int64_t)r * C + c);
    stf(out + (int64_t)b * C + c, s / (float)(r1 - r0));           // an empty sequence gives NaN, as the reference's 0 / 0 does
}
template <typename T>
__global__ __launch_bounds__(256) void segment_mean_bwd_k(const T* __restrict__ dout, const int* __restrict__ cu, T* __restrict__ dx, int C) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = cu[b], r1 = cu[b + 1];
    const float v = ldf(dout + (int64_t)b * C + c) / (float)(r1 - r0);
    for (int r = r0; r < r1; ++r) stf(dx + (int64_t)r * C + c, v);
}
// bf16, C % 8 == 0: a workgroup = 32 column groups of 8 x 8 row lanes; every lane sums rows r0 + ry, r0 + ry + 8, ... with independent 16-byte
// loads (the scalar loop above is one dependent 2-byte load per row: 126 us for 32 x 512 x 768 against 9 us here), the 8 row lanes meet in LDS
__global__ __launch_bounds__(256) void segment_mean_fwd_vec_k(const bf16* __restrict__ x, const int* __restrict__ cu, bf16* __restrict__ out, int C) {
    __shared__ float red[8][32][8];
    const int b = blockIdx.y, cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = (blockIdx.x * 32 + cx) * 8;
    const int r0 = cu[b], r1 = cu[b + 1];
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (int r = r0 + ry; r < r1; r += 8) {
            const uint4 u = *(const uint4*)(x + (int64_t)r * C + c);
            const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[2 * k] += __uint_as_float(w[k] << 16); s[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u); }
        }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[ry][cx][k] = s[k];
    __syncthreads();
    if (ry == 0 && c < C) {
        const float inv = 1.0f / (float)(r1 - r0);          // an empty sequence gives NaN (0 * inf), as the reference's 0 / 0 does
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += red[j][cx][k];
            out[(int64_t)b * C + c + k] = (bf16)(t * inv);
        }
    }
}
extern "C" int mvuld_segment_mean_fwd(const void* x, const int* cu, void* out, int B, int C, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(x && cu && out && B > 0 && C > 0, "segment_mean_fwd: bad args");
    if (dtype == MVULD_BF16 && C % 8 == 0 && (((uintptr_t)x) & 15) == 0) {
        hipLaunchKernelGGL(segment_mean_fwd_vec_k, dim3((unsigned)cdiv(C, 256), B), dim3(256), 0, stream, (const bf16*)x, cu, (bf16*)out, C);
        MV_LAUNCH_CHECK("segment_mean_fwd");
        return 0;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(segment_mean_fwd_k<T>, dim3((unsigned)cdiv(C, 256), B), dim3(256), 0, stream, (const T*)x, cu, (T*)out, C));
    MV_LAUNCH_CHECK("segment_mean_fwd");
    return 0;
}
extern "C" int mvuld_segment_mean_bwd(const void* dout, const int* cu, void* dx, int B, int C, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(dout && cu && dx && B > 0 && C > 0, "segment_mean_bwd: bad args");
    DISPATCH_T(dtype, hipLaunchKernelGGL(segment_mean_bwd_k<T>, dim3((unsigned)cdiv(C, 256), B), dim3(256), 0, stream, (const T*)dout, cu, (T*)dx, C));
    MV_LAUNCH_CHECK("segment_mean_bwd");
    return 0;
}

// row gather / scatter through a row map: scatter = 0: dst[t] = src[map[t]] ; scatter = 1: dst[map[t]] = src[t]   (t < T, C % 8 == 0 bf16 / % 4 f32)
template <typename T>
__global__ __launch_bounds__(256) void rows_map_k(const T* __restrict__ src, const int* __restrict__ map, T* __restrict__ dst, int64_t Trows, int C,
                                                  int scatter) {
    constexpr int V = 16 / sizeof(T);
    const int cpr = C / V;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Trows * cpr) return;
    const int64_t t = i / cpr;
    const int c = (int)(i % cpr) * V;
    const int64_t m = map[t];
    const int64_t s = scatter ? t : m, d = scatter ? m : t;
    *(uint4*)(dst + d * C + c) = *(const uint4*)(src + s * C + c);
}
extern "C" int mvuld_rows_map(const void* src, const int* map, void* dst, int64_t rows, int C, int scatter, int dtype, hipStream_t stream) {
    MV_CHECK_ARG(src && map && dst && rows > 0 && C > 0 && C % (dtype == MVULD_F32 ? 4 : 8) == 0, "rows_map: bad args");
    const int cpr = C / (dtype == MVULD_F32 ? 4 : 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL(rows_map_k<T>, dim3((unsigned)cdiv(rows * cpr, 256)), dim3(256), 0, stream, (const T*)src, map, (T*)dst, rows, C, scatter));
    MV_LAUNCH_CHECK("rows_map");
    return 0;
}

// y[i] = x[i] * s[0]  (s = one fp32 on the device): the upstream gradient of the scalar loss applied to dlogits without a host
// round trip (autograd of CrossEntropyLoss under loss * k / a sum of losses: main_bigvul.py:331-333)
__global__ void scale_by_dev_k(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * s[0];
}
extern "C" int mvuld_scale_by_dev(const float* x, const float* s, float* y, int64_t n, hipStream_t stream) {
    MV_CHECK_ARG(x && s && y && n > 0, "scale_by_dev: bad args");
    hipLaunchKernelGGL(scale_by_dev_k, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, x, s, y, n);
    MV_LAUNCH_CHECK("scale_by_dev");
    return 0;
}

// DropPath (stochastic depth, timm semantics: swin_transformer_v2.py:301,304 via timm.models.layers.DropPath): per (block, sample)
// keep / (1 - rate) factors drawn on the device from the counter hash -- out[k*B + b] = hash(seed, k*B + b) >= rate[k] ? 1/(1-rate[k]) : 0.
// No host RNG, no host -> device copy in the step.
__global__ void droppath_scales_k(const float* __restrict__ rates, float* __restrict__ out, int nblk, int B, uint64_t seed,
                                  const uint64_t* __restrict__ seed_off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblk * B) return;
    if (seed_off) seed += seed_off[0] * 0xD1B54A32D192ED03ULL;
    const float p = rates[i / B];
    const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
    const bool keep = mix32(seed + (uint64_t)i * 0x9E3779B97F4A7C15ULL) >= thr;
    out[i] = keep ? 1.0f / (1.0f - p) : 0.f;
}
extern "C" int mvuld_droppath_scales(const float* rates, float* out, int nblk, int B, uint64_t seed, const uint64_t* seed_offset, hipStream_t stream) {
    MV_CHECK_ARG(rates && out && nblk > 0 && B > 0, "droppath_scales: bad args");
    hipLaunchKernelGGL(droppath_scales_k, dim3((unsigned)cdiv((int64_t)nblk * B, 256)), dim3(256), 0, stream, rates, out, nblk, B, seed, seed_offset);
    MV_LAUNCH_CHECK("droppath_scales");
    return 0;
}

// p[0] += inc  (the device-resident step counter behind `seed_offset`: one launch per training step, inside the captured graph)
__global__ void counter_add_k(uint64_t* p, uint64_t inc) { p[0] += inc; }
extern "C" int mvuld_counter_add(uint64_t* counter, uint64_t inc, hipStream_t stream) {
    MV_CHECK_ARG(counter, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_k, dim3(1), dim3(1), 0, stream, counter, inc);
    MV_LAUNCH_CHECK("counter_add");
    return 0;
}

// ------------------------------------------------------------------------------------ fp8 (OCP e4m3) quantisation, per-tensor scale
// scale = max|x| / 448 (the largest finite e4m3 value), q = round_to_e4m3(x / scale): the operands of the fp8 forward GEMMs
// (BASELINE config 5: QKV / FFN GEMMs of the two encoders; swin_transformer_v2.py:146-152,177,26-32, HF RobertaModel dense layers).
// Two launches: per-block |x| maxima (deterministic: no float atomics), then the conversion, which folds those maxima into `scale`
// (v_cvt_pk_fp8_f32, eight values per lane, 16-byte loads and 8-byte stores).
template <typename T>
__global__ __launch_bounds__(256) void absmax_k(const T* __restrict__ x, int64_t n, float* __restrict__ partials) {
    __shared__ float red[4];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(ldf(x + i)));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void absmax_bf16x8_k(const bf16* __restrict__ x, int64_t n8, float* __restrict__ partials) {
    __shared__ float red[4];
    float m = 0.f;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n8; c += (int64_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *(const bf16x8*)(x + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)v.v[e]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// every block folds the (<= 1024) per-block maxima itself (4 KB out of L2) instead of a third launch; block 0 publishes the scale
template <typename T>
__global__ __launch_bounds__(256) void quant_e4m3_k(const T* __restrict__ x, int64_t n8, const float* __restrict__ partials, int nblk,
                                                    float* __restrict__ scale, uint2* __restrict__ out) {
    __shared__ float red[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) m = fmaxf(m, partials[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const float sc = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-12f) * (1.0f / 448.0f);
    if (blockIdx.x == 0 && threadIdx.x == 0) scale[0] = sc;
    const float inv = 1.0f / sc;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n8; c += (int64_t)gridDim.x * blockDim.x) {
        float f[8];
        if constexpr (sizeof(T) == 2) {
            const bf16x8 v = *(const bf16x8*)(x + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)v.v[e] * inv;
        } else {
            const float4 v0 = *(const float4*)(x + c * 8), v1 = *(const float4*)(x + c * 8 + 4);
            f[0] = v0.x * inv; f[1] = v0.y * inv; f[2] = v0.z * inv; f[3] = v0.w * inv;
            f[4] = v1.x * inv; f[5] = v1.y * inv; f[6] = v1.z * inv; f[7] = v1.w * inv;
        }
        unsigned lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        out[c] = make_uint2(lo, hi);
    }
}
extern "C" int mvuld_quant_e4m3(const void* x, int64_t n, int dtype, void* out, float* scale_out, float* partials, hipStream_t stream) {
    MV_CHECK_ARG(x && out && scale_out && partials && n > 0 && n % 8 == 0 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0,
                 "quant_e4m3: bad args (n % 8 == 0, 16-byte aligned, 1024-float scratch)");
    const int grid = (int)min((int64_t)1024, cdiv(n, 2048));
    if (dtype == MVULD_BF16) hipLaunchKernelGGL(absmax_bf16x8_k, dim3(grid), dim3(256), 0, stream, (const bf16*)x, n / 8, partials);
    else hipLaunchKernelGGL(absmax_k<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, n, partials);
    const int grid2 = (int)min((int64_t)4096, cdiv(n / 8, 512));
    DISPATCH_T(dtype, hipLaunchKernelGGL(quant_e4m3_k<T>, dim3(grid2), dim3(256), 0, stream, (const T*)x, n / 8, partials, grid, scale_out, (uint2*)out));
    MV_LAUNCH_CHECK("quant_e4m3");
    return 0;
}

// delayed scaling: the {scale, amax} pairs the fused e4m3 emitters (layernorm_fwd_q8, the fp8 GEMM's GELU epilogue) read and fold into
__global__ void fp8_roll_scales_k(float* __restrict__ state, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float amax = state[2 * i + 1];
    if (amax > 0.f) { state[2 * i] = amax * (1.0f / 448.0f); state[2 * i + 1] = 0.f; }
}
extern "C" int mvuld_fp8_roll_scales(float* state, int n, hipStream_t stream) {
    MV_CHECK_ARG(state && n > 0, "fp8_roll_scales: bad args");
    hipLaunchKernelGGL(fp8_roll_scales_k, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, state, n);
    MV_LAUNCH_CHECK("fp8_roll_scales");
    return 0;
}

// ------------------------------------------------------------------------------------ batched e4m3 quantisation of many fp32 tensors
// The fp8 forward GEMMs need every QKV / FFN weight requantised after each optimizer step: 102 tensors x (absmax + convert) was 204
// launches.  One job table (like transpose_batched) and three launches: per-block |x| maxima of all tensors, one block per tensor to
// fold its maxima into its scale, one conversion pass.  Each block covers Q_CHUNK elements of one tensor.
struct QuantJob { const float* src; uint8_t* dst; float* scale; int64_t n; int64_t blk0; };
#define Q_CHUNK 8192
__device__ __forceinline__ QuantJob q_job_of(const QuantJob* __restrict__ jobs, int njobs, int64_t blk) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= blk) lo = mid; else hi = mid - 1;
    }
    return jobs[lo];
}
__global__ __launch_bounds__(256) void quant_batched_absmax_k(const QuantJob* __restrict__ jobs, int njobs, float* __restrict__ partials) {
    __shared__ float red[4];
    const QuantJob jb = q_job_of(jobs, njobs, blockIdx.x);
    const int64_t e0 = (blockIdx.x - jb.blk0) * Q_CHUNK, e1 = min(jb.n, e0 + Q_CHUNK);
    float m = 0.f;
    for (int64_t i = e0 + threadIdx.x * 4; i < e1; i += 1024) {
        const float4 v = *(const float4*)(jb.src + i);            // n % 8 == 0 and chunks of 8192: whole float4s
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void quant_batched_scale_k(const QuantJob* __restrict__ jobs, int njobs, const float* __restrict__ partials) {
    __shared__ float red[4];
    const QuantJob jb = jobs[blockIdx.x];
    const int64_t nblk = (jb.n + Q_CHUNK - 1) / Q_CHUNK;
    float m = 0.f;
    for (int64_t i = threadIdx.x; i < nblk; i += 256) m = fmaxf(m, partials[jb.blk0 + i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) jb.scale[0] = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-12f) * (1.0f / 448.0f);
}
__global__ __launch_bounds__(256) void quant_batched_convert_k(const QuantJob* __restrict__ jobs, int njobs) {
    const QuantJob jb = q_job_of(jobs, njobs, blockIdx.x);
    const int64_t e0 = (blockIdx.x - jb.blk0) * Q_CHUNK, e1 = min(jb.n, e0 + Q_CHUNK);
    const float inv = 1.0f / jb.scale[0];
    for (int64_t i = e0 + threadIdx.x * 8; i < e1; i += 2048) {
        const float4 v0 = *(const float4*)(jb.src + i), v1 = *(const float4*)(jb.src + i + 4);
        unsigned lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v0.x * inv, v0.y * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v0.z * inv, v0.w * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v1.x * inv, v1.y * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v1.z * inv, v1.w * inv, hi, true);
        *(uint2*)(jb.dst + i) = make_uint2(lo, hi);
    }
}
// jobs: device array of {src fp32, dst e4m3, scale fp32[1], n (% 8 == 0), blk0 = first block of the tensor (prefix sum of
// ceil(n / 8192))}; partials: total_blocks floats of scratch.  Same result as mvuld_quant_e4m3 per tensor.
extern "C" int mvuld_quant_e4m3_batched(const void* jobs, int njobs, int64_t total_blocks, float* partials, hipStream_t stream) {
    MV_CHECK_ARG(jobs && partials && njobs > 0 && total_blocks > 0 && total_blocks < 2147483647LL, "quant_e4m3_batched: bad args");
    hipLaunchKernelGGL(quant_batched_absmax_k, dim3((unsigned)total_blocks), dim3(256), 0, stream, (const QuantJob*)jobs, njobs, partials);
    hipLaunchKernelGGL(quant_batched_scale_k, dim3(njobs), dim3(256), 0, stream, (const QuantJob*)jobs, njobs, (const float*)partials);
    hipLaunchKernelGGL(quant_batched_convert_k, dim3((unsigned)total_blocks), dim3(256), 0, stream, (const QuantJob*)jobs, njobs);
    MV_LAUNCH_CHECK("quant_e4m3_batched");
    return 0;
}

// ------------------------------------------------------------------------------------ SwinV2 qkv bias = (q_bias, 0, v_bias), all blocks at once
// swin_transformer_v2.py:147-150 concatenates (q_bias, zeros, v_bias) in every forward; the parameters change once per optimizer
// step, so the 24 packed [3C] buffers are rebuilt by ONE launch after the step instead of two small copies per block on the
// forward's critical chain of kernels.
struct QkvBiasJob { const float* q; const float* v; float* dst; int64_t C; };
__global__ __launch_bounds__(256) void qkv_bias_pack_batched_k(const QkvBiasJob* __restrict__ jobs) {
    const QkvBiasJob jb = jobs[blockIdx.x];
    const int C = (int)jb.C;
    for (int i = threadIdx.x; i < 3 * C; i += 256) jb.dst[i] = i < C ? jb.q[i] : (i < 2 * C ? 0.f : jb.v[i - 2 * C]);
}
extern "C" int mvuld_qkv_bias_pack_batched(const void* jobs, int njobs, hipStream_t stream) {
    MV_CHECK_ARG(jobs && njobs > 0, "qkv_bias_pack_batched: bad args");
    hipLaunchKernelGGL(qkv_bias_pack_batched_k, dim3(njobs), dim3(256), 0, stream, (const QkvBiasJob*)jobs);
    MV_LAUNCH_CHECK("qkv_bias_pack_batched");
    return 0;
}

// ------------------------------------------------------------------------------------ continuous-position-bias tables of ALL blocks in one launch
// The table of a SwinV2 block (16 * sigmoid(cpb_mlp(relative_coords_table)), swin_transformer_v2.py:159-163) depends on parameters
// only: it is rebuilt for every block by one launch after each optimizer step instead of one launch per block on the forward's chain
// of kernels (and, in inference, once instead of every forward).  Same arithmetic as cpb_fwd_k.
struct CpbJob { const float* coords; const float* W1; const float* b1; const float* W2; float* hidden; float* table16; int64_t T2, H, row0; };
__global__ __launch_bounds__(256) void cpb_fwd_batched_k(const CpbJob* __restrict__ jobs, int njobs) {
    __shared__ float hid[512];
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].row0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const CpbJob jb = jobs[lo];
    const int i = (int)(blockIdx.x - jb.row0), H = (int)jb.H;
    const float cy = jb.coords[2 * i], cx = jb.coords[2 * i + 1];
    for (int j = threadIdx.x; j < 512; j += 256) {
        const float v = fmaxf(jb.W1[2 * j] * cy + jb.W1[2 * j + 1] * cx + jb.b1[j], 0.f);
        hid[j] = v;
        jb.hidden[(int64_t)i * 512 + j] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int h = wv; h < H; h += 4) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s = fmaf(hid[lane + 64 * k], jb.W2[h * 512 + lane + 64 * k], s);
        s = wave_sum(s);
        if (lane == 0) jb.table16[(int64_t)i * H + h] = 16.0f / (1.0f + __expf(-s));
    }
}
// jobs: device array of {coords [T2,2], W1 [512,2], b1 [512], W2 [H,512], hidden [T2,512] (out), table16 [T2,H] (out), T2, H, row0 =
// first block index of the job (prefix sum of T2)}; total_rows = sum of T2.
extern "C" int mvuld_cpb_table_fwd_batched(const void* jobs, int njobs, int64_t total_rows, hipStream_t stream) {
    MV_CHECK_ARG(jobs && njobs > 0 && total_rows > 0 && total_rows < 2147483647LL, "cpb_table_fwd_batched: bad args");
    hipLaunchKernelGGL(cpb_fwd_batched_k, dim3((unsigned)total_rows), dim3(256), 0, stream, (const CpbJob*)jobs, njobs);
    MV_LAUNCH_CHECK("cpb_table_fwd_batched");
    return 0;
}
