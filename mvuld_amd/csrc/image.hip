// Image ingestion on the device: the reference's evaluation transform (data/build.py:146-168) -- PIL bicubic resize to S x S,
// ToTensor, Normalize -- as two byte kernels, bit-exact with Pillow's 8-bit resampler (Resample.c): a horizontal and a vertical
// separable pass with 22-bit fixed-point coefficients (tables built on the host exactly as Pillow's precompute_coeffs does),
// rounding to uint8 after each pass; the vertical pass also converts to float, normalises and writes channel-major (CHW) rows for
// the patch embedding.  HBM-bound byte work: 3 x H x W bytes in, 3 x S x S floats out per image.
#include "common.h"
#include <cstdint>

#define IMG_PRECISION_BITS 22

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> IMG_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// out[b][y][xo][c] = clip8(2^21 + sum_t in[b][y][x0(xo) + t][c] * kk[xo][t])
__global__ __launch_bounds__(256) void image_resize_h_u8_k(const uint8_t* __restrict__ in, int H, int W, const int* __restrict__ bounds,
                                                            const int* __restrict__ kk, int ksize, int Wo, uint8_t* __restrict__ out) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, b = blockIdx.z;
    if (xo >= Wo) return;
    const int x0 = bounds[2 * xo], n = bounds[2 * xo + 1];
    const uint8_t* row = in + ((int64_t)b * H + y) * W * 3 + (int64_t)x0 * 3;
    const int* k = kk + (int64_t)xo * ksize;
    int a0 = 1 << (IMG_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < n; ++t) {
        const int w = k[t];
        a0 += row[3 * t] * w; a1 += row[3 * t + 1] * w; a2 += row[3 * t + 2] * w;
    }
    uint8_t* o = out + (((int64_t)b * H + y) * Wo + xo) * 3;
    o[0] = (uint8_t)clip8(a0); o[1] = (uint8_t)clip8(a1); o[2] = (uint8_t)clip8(a2);
}

// u8 = clip8(2^21 + sum_t in[b][y0(yo) + t][xo][c] * kk[yo][t]);  out[b][c][yo][xo] = (u8 / 255 - mean[c]) / std[c]
// (u8_out, optional: the resized image itself, [b][yo][xo][c], for parity checks against PIL)
template <typename TO>
__global__ __launch_bounds__(256) void image_resize_v_norm_k(const uint8_t* __restrict__ in, int Hin, int Wo, const int* __restrict__ bounds,
                                                              const int* __restrict__ kk, int ksize, int Ho, TO* __restrict__ out,
                                                              uint8_t* __restrict__ u8_out, float m0, float m1, float m2, float s0, float s1,
                                                              float s2) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;
    const int yo = blockIdx.y, b = blockIdx.z;
    if (xo >= Wo) return;
    const int y0 = bounds[2 * yo], n = bounds[2 * yo + 1];
    const uint8_t* col = in + (((int64_t)b * Hin + y0) * Wo + xo) * 3;
    const int* k = kk + (int64_t)yo * ksize;
    int a0 = 1 << (IMG_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < n; ++t) {
        const int w = k[t];
        const uint8_t* p = col + (int64_t)t * Wo * 3;
        a0 += p[0] * w; a1 += p[1] * w; a2 += p[2] * w;
    }
    const int v0 = clip8(a0), v1 = clip8(a1), v2 = clip8(a2);
    if (u8_out) {
        uint8_t* u = u8_out + (((int64_t)b * Ho + yo) * Wo + xo) * 3;
        u[0] = (uint8_t)v0; u[1] = (uint8_t)v1; u[2] = (uint8_t)v2;
    }
    const int64_t plane = (int64_t)Ho * Wo;
    TO* o = out + (int64_t)b * 3 * plane + (int64_t)yo * Wo + xo;
    // torchvision: img.float().div(255) then sub(mean).div(std), all in fp32
    o[0] = (TO)(__fdiv_rn(__fdiv_rn((float)v0, 255.0f) - m0, s0));
    o[plane] = (TO)(__fdiv_rn(__fdiv_rn((float)v1, 255.0f) - m1, s1));
    o[2 * plane] = (TO)(__fdiv_rn(__fdiv_rn((float)v2, 255.0f) - m2, s2));
}

// images [B][H][W][3] uint8 (RGB, device) -> out [B][3][Ho][Wo] (fp32 or bf16).  bounds_* int32 [n_out][2], kk_* int32 [n_out][ksize_*]
// (Pillow's precompute_coeffs + normalize_coeffs_8bpc for that axis); tmp = B*H*Wo*3 bytes of scratch (unused when W == Wo and
// bounds_h is null: the horizontal pass is skipped, as Pillow does); u8_out optional [B][Ho][Wo][3].
extern "C" int mvuld_image_resize_bicubic_normalize(const void* images, int B, int H, int W, const int* bounds_h, const int* kk_h, int ksize_h,
                                                    const int* bounds_v, const int* kk_v, int ksize_v, int Ho, int Wo, void* tmp, void* out,
                                                    int out_dtype, void* u8_out, float mean_r, float mean_g, float mean_b, float std_r, float std_g, float std_b,
                                                    hipStream_t stream) {
    MV_CHECK_ARG(images && out && bounds_v && kk_v && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && ksize_v > 0,
                 "image_resize_bicubic_normalize: bad args");
    MV_CHECK_ARG(bounds_h ? (kk_h && tmp && ksize_h > 0) : W == Wo, "image_resize_bicubic_normalize: no horizontal tables but W != Wo");
    MV_CHECK_ARG(Ho <= 65535 && H <= 65535 && B <= 65535, "image_resize_bicubic_normalize: grid limits");
    const uint8_t* src = (const uint8_t*)images;
    if (bounds_h) {
        hipLaunchKernelGGL(image_resize_h_u8_k, dim3((unsigned)cdiv(Wo, 256), H, B), dim3(256), 0, stream, src, H, W, bounds_h, kk_h, ksize_h, Wo,
                           (uint8_t*)tmp);
        src = (const uint8_t*)tmp;
    }
    if (out_dtype == MVULD_BF16)
        hipLaunchKernelGGL(image_resize_v_norm_k<bf16>, dim3((unsigned)cdiv(Wo, 256), Ho, B), dim3(256), 0, stream, src, H, Wo, bounds_v, kk_v, ksize_v,
                           Ho, (bf16*)out, (uint8_t*)u8_out, mean_r, mean_g, mean_b, std_r, std_g, std_b);
    else
        hipLaunchKernelGGL(image_resize_v_norm_k<float>, dim3((unsigned)cdiv(Wo, 256), Ho, B), dim3(256), 0, stream, src, H, Wo, bounds_v, kk_v, ksize_v,
                           Ho, (float*)out, (uint8_t*)u8_out, mean_r, mean_g, mean_b, std_r, std_g, std_b);
    MV_LAUNCH_CHECK("image_resize_bicubic_normalize");
    return 0;
}
