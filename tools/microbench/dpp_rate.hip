// Issue cost of DPP controls on gfx950, one wave per SIMD: cycles per instruction in a loop of 64 INDEPENDENT v_add_f32_dpp (8 chains x 8)
// and of 64 DEPENDENT ones, for row_shr:1, wave_shr:1 and a plain v_add_f32.  s_memtime around the loop (shader cycles).
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/dpp_rate.hip -o tools/microbench/dpp_rate && tools/microbench/dpp_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    if (CTRL == 0) return v;
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL ? CTRL : 0x111, 0xf, 0xf, true));
}
template <int CTRL, bool DEP>
__global__ void k(const float* in, float* out, long long* cyc, int iters) {
    float x[8], a[8];
    for (int i = 0; i < 8; ++i) { x[i] = in[threadIdx.x + 64 * i]; a[i] = x[i] * 0.5f; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (DEP) a[0] = x[i] + dpp<CTRL>(a[0]);
                else a[i] = x[i] + dpp<CTRL>(a[i]);
            }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CTRL, bool DEP>
void run(const char* name, float* din, float* dout, long long* dc) {
    const int iters = 1000;
    hipLaunchKernelGGL((k<CTRL, DEP>), dim3(1), dim3(64), 0, 0, din, dout, dc, iters);
    long long c;
    hipMemcpy(&c, dc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-34s %6.2f cycles per instruction\n", name, (double)c / (iters * 64.0));
}
int main() {
    float *din, *dout; long long* dc;
    hipMalloc(&din, 64 * 8 * 4); hipMalloc(&dout, 64 * 4 * 4); hipMalloc(&dc, 8);
    hipMemset(din, 0, 64 * 8 * 4);
    run<0, false>("v_add_f32 independent", din, dout, dc);
    run<0, true>("v_add_f32 dependent", din, dout, dc);
    run<0x111, false>("v_add_f32_dpp row_shr:1 independent", din, dout, dc);
    run<0x111, true>("v_add_f32_dpp row_shr:1 dependent", din, dout, dc);
    run<0x138, false>("v_add_f32_dpp wave_shr:1 independent", din, dout, dc);
    run<0x138, true>("v_add_f32_dpp wave_shr:1 dependent", din, dout, dc);
    run<0x142, false>("v_add_f32_dpp row_bcast:15 indep", din, dout, dc);
    return 0;
}
