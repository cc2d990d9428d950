// Shared helpers for libmvuld_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef __bf16 bf16;

enum { MVULD_F32 = 0, MVULD_BF16 = 1, MVULD_FP8 = 2 };   // FP8 = OCP e4m3fn (gfx950's v_cvt_pk_fp8_f32 / fp8 MFMA format)

// thread-local last error text (mvuld_last_error)
void mvuld_set_error(const char* fmt, ...);

#define MV_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            mvuld_set_error(__VA_ARGS__);  \
            return 1;                      \
        }                                  \
    } while (0)

#define MV_LAUNCH_CHECK(name)                                                      \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            mvuld_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return 2;                                                              \
        }                                                                          \
    } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers ------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float ldf(const T* p) { return (float)(*p); }
template <typename T>
__device__ __forceinline__ void stf(T* p, float v) { *p = (T)v; }

// counter-based 32-bit hash of a 64-bit counter: the element masks of mvuld_dropout (and of the LayerNorm that fuses it)
__device__ __forceinline__ uint32_t mix32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// saturate to the e4m3 range before v_cvt_pk_fp8_f32 (an out-of-range input converts to NaN, not to the largest value)
__device__ __forceinline__ float q_clamp(float x) { return __builtin_amdgcn_fmed3f(x, -448.0f, 448.0f); }
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` holds >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : (__expf(x) - 1.0f); }

// 8 x bf16 <-> 8 x float through one 16-byte access
struct alignas(16) bf16x8 { bf16 v[8]; };
struct alignas(8) bf16x4 { bf16 v[4]; };
