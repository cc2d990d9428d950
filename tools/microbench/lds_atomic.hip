// Rate of LDS float atomics (ds_add_f32, no return) on gfx950: lanes per clock per CU for conflict-free and same-address patterns,
// next to ds_write_b32 / ds_read_b32 of the same shape.  hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic && ./lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>   // 0 atomic distinct, 1 atomic 4-way same address, 2 atomic all lanes of a 16-group same address, 3 plain write, 4 plain read, 5 atomic distinct stride 2 banks
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    __shared__ float buf[16384];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) buf[i] = 0.f;
    __syncthreads();
    int a;
    if (MODE == 0 || MODE == 3 || MODE == 4) a = wave * 64 + lane;
    else if (MODE == 1) a = wave * 64 + (lane & 15) - 4 * (lane >> 4) + 16;        // lanes (c, fg) and (c+4, fg+1) coincide
    else if (MODE == 2) a = wave * 64 + (lane >> 4);
    else a = (wave * 64 + lane) * 2;
    float acc = 0.f;
    const float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = (a + u * 1024 * (MODE == 5 ? 2 : 1)) & 16383;
            if (MODE <= 2 || MODE == 5) atomicAdd(buf + idx, v);
            else if (MODE == 3) ((volatile float*)buf)[idx] = v;
            else acc += ((volatile float*)buf)[idx];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = buf[16] + acc;
}

int main() {
    float* d; hipMalloc(&d, 4096 * 4);
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"ds_add_f32 distinct addresses", "ds_add_f32 4 lanes per address", "ds_add_f32 16 lanes per address", "ds_write_b32 distinct", "ds_read_b32 distinct", "ds_add_f32 distinct, stride 2"};
    for (int mode = 0; mode < 6; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 0, 0, d, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 0, 0, d, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 0, 0, d, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(1024), 0, 0, d, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(1024), 0, 0, d, iters); break;
                default: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(1024), 0, 0, d, iters); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double lane_ops = (double)iters * 8 * 1024;          // per CU (one block per CU)
        printf("%-36s %8.3f ms  %6.2f lanes/ns/CU  (~%5.1f lanes/clk at 2.1 GHz)\n", names[mode], ms, lane_ops / (ms * 1e6), lane_ops / (ms * 1e6) / 2.1);
    }
    return 0;
}
