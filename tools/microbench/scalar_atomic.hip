// Does gfx950 execute scalar memory atomics (s_atomic_add ... glc: the return value lands in an SGPR and is counted in lgkmcnt, not in vmcnt)?
// A persistent kernel whose vector-memory queue is full of LDS-DMA and stores can then claim work without touching that queue.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/scalar_atomic.hip -o /tmp/scalar_atomic && /tmp/scalar_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void claim_k(unsigned* counter, unsigned* out, unsigned long long* cyc) {
    unsigned r, one = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(counter), "0"(one) : "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = r; cyc[blockIdx.x] = t1 - t0; }
}
int main() {
    const int n = 1 << 16;
    unsigned *c, *o; unsigned long long* t;
    hipMalloc(&c, 4); hipMalloc(&o, n * 4); hipMalloc(&t, n * 8);
    hipMemset(c, 0, 4);
    hipLaunchKernelGGL(claim_k, dim3(n), dim3(64), 0, 0, c, o, t);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<unsigned> h(n); std::vector<unsigned long long> ht(n); unsigned fin;
    hipMemcpy(h.data(), o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t, n * 8, hipMemcpyDeviceToHost); hipMemcpy(&fin, c, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = fin == (unsigned)n;
    for (int i = 0; i < n; ++i) ok = ok && h[i] == (unsigned)i;
    std::sort(ht.begin(), ht.end());
    printf("scalar atomics: final %u (expected %d), tickets unique and dense: %s; round trip (s_memtime ticks, 100 MHz): median %llu, p99 %llu\n", fin, n, ok ? "yes" : "NO",
           ht[n / 2], ht[n * 99 / 100]);
    return ok ? 0 : 2;
}
