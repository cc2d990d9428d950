// CSR index of a batched graph, built on the device (graph half: what DGL builds lazily on the host for
// GATConv's update_all -- GraphModel.py:171-176 over dgl.batch(...) graphs).  Same content and ORDER as the host builder
// (mvuld_amd/graph.py: BatchedGraph.index): edges grouped by destination keep their edge-id order (stable sort), likewise by
// source, plus for every edge in by-source order its slot in the by-destination ordering.  The sort is rocPRIM's stable LSD radix
// sort through hipCUB (a ROCm library: this is plumbing around the path, not a hot kernel); the rest are small gather kernels.
#include "common.h"
#include <cstdint>
#include <hipcub/hipcub.hpp>

__global__ void gi_keys_k(const int64_t* __restrict__ v, int E, unsigned* __restrict__ keys, int* __restrict__ ids) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) { keys[e] = (unsigned)v[e]; if (ids) ids[e] = e; }
}
// indptr[n] = number of sorted keys < n  (lower bound), n = 0 .. N
__global__ void gi_indptr_k(const unsigned* __restrict__ sorted, int E, int N, int* __restrict__ indptr) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n > N) return;
    int lo = 0, hi = E;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sorted[mid] < (unsigned)n) lo = mid + 1; else hi = mid;
    }
    indptr[n] = lo;
}
// other_by[i] = other[order[i]];  pos[order[i]] = i (when pos != null);  slot[i] = pos_in[order[i]] (when slot != null)
__global__ void gi_gather_k(const int64_t* __restrict__ other, const int* __restrict__ order, int E, int* __restrict__ other_by,
                            int* __restrict__ pos, const int* __restrict__ pos_in, int* __restrict__ slot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const int e = order[i];
    other_by[i] = (int)other[e];
    if (pos) pos[e] = i;
    if (slot) slot[i] = pos_in[e];
}

static size_t gi_sort_temp_bytes(int E) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr, (int*)nullptr, E);
    return bytes;
}
static inline size_t gi_al(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int64_t mvuld_graph_csr_workspace_bytes(int E) {
    if (E <= 0) return 0;
    return (int64_t)(6 * gi_al((size_t)E * 4) + gi_al(gi_sort_temp_bytes(E)));
}

// src, dst: int64 [E] on the device (node ids < N).  Outputs int32: indptr_dst / indptr_src [N + 1], src_by_dst / dst_by_src /
// slot_by_src [E].  ws: mvuld_graph_csr_workspace_bytes(E) bytes of scratch.
extern "C" int mvuld_graph_csr_build(const int64_t* src, const int64_t* dst, int E, int N, int* indptr_dst, int* src_by_dst, int* indptr_src,
                                     int* dst_by_src, int* slot_by_src, void* ws, int64_t ws_bytes, hipStream_t stream) {
    MV_CHECK_ARG(src && dst && E > 0 && N > 0 && indptr_dst && src_by_dst && indptr_src && dst_by_src && slot_by_src, "graph_csr_build: bad args");
    MV_CHECK_ARG(ws && ws_bytes >= mvuld_graph_csr_workspace_bytes(E), "graph_csr_build: workspace too small");
    char* w = (char*)ws;
    const size_t seg = gi_al((size_t)E * 4);
    unsigned* keys = (unsigned*)w;
    unsigned* keys_sorted = (unsigned*)(w + seg);
    int* ids = (int*)(w + 2 * seg);
    int* order_d = (int*)(w + 3 * seg);
    int* order_s = (int*)(w + 4 * seg);
    int* pos_in_d = (int*)(w + 5 * seg);
    void* temp = w + 6 * seg;
    size_t temp_bytes = gi_sort_temp_bytes(E);
    const dim3 blk(256), grdE((unsigned)cdiv(E, 256)), grdN((unsigned)cdiv(N + 1, 256));
    // by destination
    hipLaunchKernelGGL(gi_keys_k, grdE, blk, 0, stream, dst, E, keys, ids);
    if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_sorted, ids, order_d, E, 0, 32, stream) != hipSuccess) {
        mvuld_set_error("graph_csr_build: radix sort failed");
        return 1;
    }
    hipLaunchKernelGGL(gi_indptr_k, grdN, blk, 0, stream, keys_sorted, E, N, indptr_dst);
    hipLaunchKernelGGL(gi_gather_k, grdE, blk, 0, stream, src, order_d, E, src_by_dst, pos_in_d, (const int*)nullptr, (int*)nullptr);
    // by source
    hipLaunchKernelGGL(gi_keys_k, grdE, blk, 0, stream, src, E, keys, (int*)nullptr);
    if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_sorted, ids, order_s, E, 0, 32, stream) != hipSuccess) {
        mvuld_set_error("graph_csr_build: radix sort failed");
        return 1;
    }
    hipLaunchKernelGGL(gi_indptr_k, grdN, blk, 0, stream, keys_sorted, E, N, indptr_src);
    hipLaunchKernelGGL(gi_gather_k, grdE, blk, 0, stream, dst, order_s, E, dst_by_src, (int*)nullptr, (const int*)pos_in_d, slot_by_src);
    MV_LAUNCH_CHECK("graph_csr_build");
    return 0;
}
