// Shared by the GEMM translation units of libmvuld_hip.so (gemm.hip, gemm_p256.hip).
#pragma once
#include "common.h"

// EPI_GELU_DG / EPI_MUL_AUX (round 3): the training pair of the FFN.  The forward product writes gelu'(pre-activation) to aux instead of
// the pre-activation itself (same bytes; the erf and the exponential are the ones GELU needs anyway: + 2 instructions per element), and
// the data-gradient product of the second Linear multiplies by it -- a one-instruction epilogue where EPI_MUL_DGELU recomputes erf + exp
// (17 issue slots per element: a quarter of a K = 512 tile's time, section 9b item 1b of DESIGN.md).
enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_GELU = 2, EPI_ELU = 3, EPI_MUL_DGELU = 4, EPI_MUL_DELU = 5, EPI_ADD_AUX = 6, EPI_GELU_DG = 7, EPI_MUL_AUX = 8 };
__host__ __device__ __forceinline__ constexpr bool epi_reads_aux(int e) { return e >= EPI_MUL_DGELU && e != EPI_GELU_DG; }
__host__ __device__ __forceinline__ constexpr bool epi_writes_aux(int e) { return e == EPI_GELU || e == EPI_GELU_DG; }
enum { OUT_STORE = 0, OUT_ACCUM = 1, OUT_ATOMIC = 2 };

struct GemmArgs {
    const void* A; const void* B; void* C;
    int64_t lda, ldb, ldc, sA, sB, sC;     // leading dims and batch strides, in elements
    int M, N, K, batch, splitk;
    const float* bias;                     // [N] or null
    void* aux; int64_t ldaux, sAux;        // pre-activation (GELU out / dGELU in) or ELU output (dELU in); dtype of C
    float alpha; int epi; int out_mode;
    const float* scale_a; const float* scale_b;   // fp8 operands: per-tensor dequantisation scales (one fp32 each, on the device), or null
    // fp8 GELU epilogue: also (or, with C == nullptr, only) emit the activation as e4m3 under the scale at q_scale[0], and fold
    // max|value| into q_amax[0] (atomicMax on the bit pattern of a non-negative float) for the next step's scale
    void* q_out = nullptr; int64_t ldq = 0; const float* q_scale = nullptr; unsigned* q_amax = nullptr;
    // gemm_nt_f32x3_k only: operand stored transposed -- A as [K, M] (element (m, k) at A[k * lda + m]), B as [K, N]
    int ta = 0, tb = 0;
};

typedef bf16 __attribute__((ext_vector_type(8))) bf16x8_t;
typedef float __attribute__((ext_vector_type(4))) f32x4_t;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4_t;

// erf with |error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26): 1 v_rcp + 1 v_exp + ~10 plain VALU, against ~40 for ocml's erff.
// Used where the result is rounded to bf16 anyway (bf16-output GEMM epilogues); the fp32 parity mode keeps erff.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return copysignf(fmaf(-p * t, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_fast(float x) {
    const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
    return fmaf(x, pdf, cdf);
}

// gelu(x) and gelu'(x) from one erf / one exponential; gl is gelu_fast(x) bit for bit
__device__ __forceinline__ void gelu_dgelu_fast(float x, float& gl, float& dg) {
    const float u = x * 0.70710678118654752f, ax = fabsf(u);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);      // exp(-x^2 / 2)
    const float erf = copysignf(fmaf(-p * t, e, 1.0f), u);
    gl = 0.5f * x * (1.0f + erf);
    dg = fmaf(x, 0.39894228040143268f * e, 0.5f * (1.0f + erf));
}
template <typename TO> __device__ __forceinline__ void gelu_dgelu_t(float x, float& gl, float& dg) {
    if constexpr (sizeof(TO) == 2) gelu_dgelu_fast(x, gl, dg);
    else { gl = gelu_erf(x); dg = dgelu_erf(x); }
}

template <typename TO> __device__ __forceinline__ float gelu_t(float x) {
    if constexpr (sizeof(TO) == 2) return gelu_fast(x); else return gelu_erf(x);
}
template <typename TO> __device__ __forceinline__ float dgelu_t(float x) {
    if constexpr (sizeof(TO) == 2) return dgelu_fast(x); else return dgelu_erf(x);
}

// persistent 256 x 256 NT kernel (gemm_p256.hip); returns 0 when it took the launch, -1 when the shape is not its to take
int mvuld_gemm_nt_p256_try(const GemmArgs& g, int dtype_out, hipStream_t stream);
// the same kernel on OCP e4m3 operands (v_mfma_f32_16x16x32_fp8_fp8): 0 = launched, -1 = shape not eligible (K % 64, K >= 256, N % 8)
int mvuld_gemm_nt_p256_fp8(const GemmArgs& g, hipStream_t stream);

// 256 x 256-tile weight-gradient kernel (gemm_tn256.hip); `ws` = slab area (no ticket block); 0 = took the launch, -1 = not its shape
int64_t mvuld_gemm_tn256_workspace_bytes(int M, int N, int K);
int mvuld_gemm_tn256_try(const void* dY, int64_t ldy, const void* X, int64_t ldx, float* dW, int64_t ldw, int M, int N, int K, float* dbias,
                         void* ws, int64_t ws_bytes, hipStream_t stream);
