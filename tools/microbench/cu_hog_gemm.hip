// What a persistent GEMM grid costs when some CUs are not its own -- and what the dynamic tile walk recovers.
// A "hog" kernel (H workgroups, 160 KB of LDS each, spinning for a fixed time: the footprint of a collective's channels or of another
// stream's persistent grid) is launched on one stream; while it runs, the bf16 product is launched on another stream through the
// library's C ABI (mvuld_gemm_nt) and timed with events, static walk vs dynamic walk (mvuld_set_gemm_dynamic_tiles).
//   hipcc --offload-arch=gfx950 -O3 -I include tools/microbench/cu_hog_gemm.hip -o build_variants/cu_hog_gemm -ldl
//   ./build_variants/cu_hog_gemm mvuld_amd/libmvuld_hip.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "mvuld_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void hog_k(unsigned long long ticks, unsigned* sink) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (lds[threadIdx.x] == 7) sink[0] = 1;
}

typedef int (*gemm_fn)(const void*, int64_t, int64_t, const void*, int64_t, int64_t, void*, int64_t, int64_t, int, int, int, int, const float*, int, void*,
                       int64_t, int64_t, float, int, int, int, int, int, mvuld_stream_t);
typedef int (*set_fn)(int);

int main(int argc, char** argv) {
    void* lib = dlopen(argc > 1 ? argv[1] : "mvuld_amd/libmvuld_hip.so", RTLD_NOW);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
    gemm_fn gemm = (gemm_fn)dlsym(lib, "mvuld_gemm_nt");
    set_fn set_dyn = (set_fn)dlsym(lib, "mvuld_set_gemm_dynamic_tiles");
    if (!gemm || !set_dyn) { printf("symbols missing\n"); return 1; }
    CK(hipFuncSetAttribute((const void*)hog_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int shapes[][3] = {{25088, 1536, 512}, {25088, 2048, 512}, {25088, 512, 2048}, {16384, 2304, 768}, {100352, 384, 128}};
    hipStream_t sh, sg;
    CK(hipStreamCreate(&sh)); CK(hipStreamCreate(&sg));
    hipEvent_t e0, e1, h0, h1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&h0)); CK(hipEventCreate(&h1));
    float hog_ms = 0, end_ms = 0;
    double base = 0;
    unsigned* sink; CK(hipMalloc(&sink, 4));
    printf("%-22s %5s | %9s %9s %9s | %s\n", "M x N x K", "hogs", "static us", "dyn 1 us", "dyn 2 us", "dyn1/static dyn2/static (ideal = unhogged static x 256 / (256 - hogs))");
    for (auto& sp : shapes) {
        const int M = sp[0], N = sp[1], K = sp[2];
        void *A, *B, *C; float* bias;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&bias, N * 4));
        CK(hipMemset(A, 0x3c, (size_t)M * K * 2)); CK(hipMemset(B, 0x3c, (size_t)N * K * 2)); CK(hipMemset(bias, 0, N * 4));
        for (int hogs : {0, 8, 16, 32, 64}) {
            double t[3];
            for (int dyn = 0; dyn < 3; ++dyn) {
                set_dyn(dyn);
                std::vector<float> ms;
                for (int it = 0; it < 12; ++it) {
                    // the hog holds its CUs for ~2 ms (s_memtime ticks at the shader clock here), long enough to cover the 5 back-to-back products timed beside it
                    CK(hipEventRecord(h0, sh));
                    if (hogs) hipLaunchKernelGGL(hog_k, dim3(hogs), dim3(256), 160 * 1024, sh, 4400000ull, sink);
                    CK(hipEventRecord(h1, sh));
                    hipLaunchKernelGGL(hog_k, dim3(1), dim3(64), 1024, sg, 66000ull, sink);      // 30 us: lets the hog take its CUs first
                    CK(hipEventRecord(e0, sg));
                    for (int r = 0; r < 5; ++r)
                        if (gemm(A, K, 0, B, K, 0, C, N, 0, M, N, K, 1, bias, MVULD_EPI_BIAS, nullptr, 0, 0, 1.0f, MVULD_OUT_STORE, 1, MVULD_BF16, MVULD_BF16, 0,
                                 (mvuld_stream_t)sg) != 0) { printf("gemm failed\n"); return 1; }
                    CK(hipEventRecord(e1, sg));
                    CK(hipDeviceSynchronize());
                    float v; CK(hipEventElapsedTime(&v, e0, e1));
                    if (it >= 2) ms.push_back(v / 5 * 1000);
                    CK(hipEventElapsedTime(&v, h0, h1)); hog_ms = v;
                    CK(hipEventElapsedTime(&v, h0, e1)); end_ms = v;
                }
                std::sort(ms.begin(), ms.end());
                t[dyn] = ms[ms.size() / 2];
            }
            if (hogs == 0) base = t[0];
            char nm[64]; snprintf(nm, sizeof nm, "%d x %d x %d", M, N, K);
            printf("%-22s %5d | %9.1f %9.1f %9.1f | %.2f %.2f   (ideal %.2f; hog ran %.2f ms)\n", nm, hogs, t[0], t[1], t[2], t[1] / t[0], t[2] / t[0], base * 256.0 / (256 - hogs) / t[0], hog_ms);
        }
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(bias));
    }
    set_dyn(0);
    return 0;
}
