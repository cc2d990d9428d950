// Does gfx950 execute the GFX9 whole-wave DPP shifts (wave_shl:1 = 0x130, wave_rol:1 = 0x134, wave_shr:1 = 0x138, wave_ror:1 = 0x13C)?
// The assembler accepts them; this prints what each control delivers to every lane so that the fused attention backward can rely on it.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/dpp_wave_shift.hip -o tools/microbench/dpp_wave_shift && tools/microbench/dpp_wave_shift
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL, bool BC>
__global__ void k(const int* in, int* out) {
    const int v = in[threadIdx.x];
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xf, 0xf, BC);
}
template <int CTRL, bool BC>
void run(const char* name, int* din, int* dout) {
    int h[64];
    hipLaunchKernelGGL((k<CTRL, BC>), dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-22s", name);
    for (int i = 0; i < 64; ++i) printf(" %d", h[i]);
    printf("\n");
}
int main() {
    int h[64], *din, *dout;
    for (int i = 0; i < 64; ++i) h[i] = 100 + i;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(h));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    run<0x130, false>("wave_shl:1", din, dout);
    run<0x130, true>("wave_shl:1 bound_ctrl", din, dout);
    run<0x138, false>("wave_shr:1", din, dout);
    run<0x138, true>("wave_shr:1 bound_ctrl", din, dout);
    run<0x134, false>("wave_rol:1", din, dout);
    run<0x13C, false>("wave_ror:1", din, dout);
    run<0x101, true>("row_shl:1 bound_ctrl", din, dout);
    run<0x111, true>("row_shr:1 bound_ctrl", din, dout);
    return 0;
}
