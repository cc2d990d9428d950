// Persistent 128 x 256-tile NT GEMM whose EPILOGUE RUNS UNDER THE NEXT TILE'S MAIN LOOP:  C = epi(alpha * A . B^T + bias), bf16 -> bf16.
//
// Why: in gemm_p256.hip both waves of a SIMD reach a tile's epilogue together (one barrier per k-step keeps the workgroup in lockstep), so
// the matrix cores idle while the GELU math and the stores run -- a third of every K <= 768 GELU product (fc1 + GELU
// 93 us, 63 us without its epilogue).  A 256-row tile has no registers for a second accumulator set; a 128-row tile has (2 x 64), and at
// N >= 1536 its main loop costs the same per FLOP as the 256-row one (measured with the epilogue compiled out: 68.6 vs 67.7 us).  So:
//   * the tile loop is unrolled by two over accumulator sets acc[0] / acc[1]: while tile t+1 accumulates into one set, tile t's finished
//     set is turned into output IN PIECES, one (16-row fragment, 32-column half) chunk per k-step, placed after the step's MFMAs are
//     issued -- the chunk's VALU work and its one or two 16-byte-per-lane stores overlap the matrix cores of both waves of the SIMD;
//   * the same 4-stage LDS-DMA ring as gemm_p256.hip runs across tiles (three k-steps always in flight, source-side XOR swizzle, operands
//     swapped so a lane owns consecutive output columns, permlane16_swap -> 16-byte row segments, no LDS in the epilogue);
//   * vmcnt is counted at RUN time: the wave keeps the number of vector-memory operations it has issued and the count right after each
//     ring stage's loads; the wait in front of stage s allows exactly (issued - mark[s]) younger operations to stay in flight, whatever mix
//     of DMA loads, bias DMA and chunk stores that is.  A wave whose sub-tile touches the matrix edge does not count its (predicated)
//     stores: under-counting only waits longer;
//   * per-tile bias slices ride the DMA queue into four 1 KiB LDS slots (tile t's slice is read during tile t+1).
// Shapes: K % 32 == 0, K >= 128, N % 8 == 0; epilogues NONE / BIAS / GELU (+ pre-activation out).  M and N tails: DMA rows clamped, stores
// predicated.
//
// MEASURED RESULT (round 2): correct and bit-repeatable, and SLOWER than gemm_p256.hip, so it is OFF by default
// (MVULD_GEMM_P128D / mvuld_set_gemm_p128d_mode: 0 never = default, 1 rule, 2 whenever legal; the tests force 2).  On 25088 x 2048 x 512
// + GELU: 94 us (p256) vs 127 us here; with the chunks' stores compiled out 86.5 us, with the chunks compiled out 83.8 us.  So the
// interleaved VALU work is nearly free (+3 %), as intended -- it is the stores that cost: vmcnt retires in issue order, every DMA load
// issued behind a store is "landed" for the counted wait only once that store has been acknowledged, and under this load an
// acknowledgement takes ~4 k-steps.  The 256-row kernel's burst of stores at a tile boundary pays that once per tile; spreading the
// stores through the loop pays it every step.  Hiding the stores needs completion tracking for the operand loads that does not share a
// counter with them (a different wave, or a landed-flag in LDS), which the register file (2 x 256 per SIMD, all taken) does not allow here.
#include "gemm_common.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

#ifndef P128D_X
#define P128D_X 0      // timing experiments: bit 0 = chunks without their stores, bit 1 = without the chunks altogether
#endif
#define D_BM 128
#define D_BN 256
#define D_STAGE_BYTES 24576                      // A 128 x 64 B + B 256 x 64 B
#define D_A_BYTES 8192
#define D_BIAS_OFF (4 * D_STAGE_BYTES)
#define D_LDS_BYTES (4 * D_STAGE_BYTES + 4096)

__device__ __forceinline__ int d_off(int row, int ch) { return row * 64 + ((ch ^ ((4 - ((row >> 2) & 3)) & 3)) << 4); }
__device__ __forceinline__ unsigned d_pk(float a, float b) {
    typedef bf16 __attribute__((ext_vector_type(2))) bf16x2_t;
    bf16x2_t v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
// s_waitcnt vmcnt(n) for a run-time n: the immediate has to be a constant, so one arm per value; n above the table waits as for the
// table's last entry (fewer operations allowed in flight than could be: safe)
__device__ __forceinline__ void d_wait_vm(int n) {
#define D_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        D_W(0) D_W(1) D_W(2) D_W(3) D_W(4) D_W(5) D_W(6) D_W(7) D_W(8) D_W(9) D_W(10) D_W(11) D_W(12) D_W(13) D_W(14) D_W(15) D_W(16)
        D_W(17) D_W(18) D_W(19) D_W(20) D_W(21) D_W(22) D_W(23)
        default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    }
#undef D_W
}

template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm_nt_bf16_p128d(GemmArgs g, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* glb_vp;
    constexpr int NI = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;            // 2 x 4 waves, 64 (m) x 64 (n) each
    const int fr = lane & 15, fg = lane >> 4;
    const int nt = tiles_m * tiles_n;
    const int G = gridDim.x, bx = blockIdx.x;
    const int nk = g.K / 32;
    const int my_tiles = bx < nt ? (nt - bx + G - 1) / G : 0;
    const int total = my_tiles * nk;
    const char* A = (const char*)g.A;
    const char* B = (const char*)g.B;
    bf16* C = (bf16*)g.C;
    bf16* aux = (bf16*)g.aux;
    const int nst = (C ? 1 : 0) + ((EPI == EPI_GELU && aux) ? 1 : 0);      // store instructions per chunk

    auto tile_of = [&](int ord, int& m0, int& n0) __attribute__((always_inline)) {      // same XCD-contiguous order as gemm_p256.hip
        const int p = bx + ord * G;
        const int q = nt >> 3, r = nt & 7, x = p & 7, i = p >> 3;
        const int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
        m0 = (t / tiles_n) * D_BM;
        n0 = (t % tiles_n) * D_BN;
    };

    // ---- DMA issue stream, three k-steps ahead of the compute stream, across tiles.  Per k-step a wave issues A piece `wave` (16 rows) and
    // B pieces 2 wave, 2 wave + 1; wave 0 also the tile's bias slice at every tile switch.
    const char* ga;
    const char* gb[2];
    int iss_ord = 0, iss_kt = 0, issued = 0;
    int nvm = 0;                                         // vector-memory operations this wave has issued so far (scalar)
    int q0 = 0, q1 = 0, q2 = 0;                          // nvm right after the loads of the three k-steps in flight (q0 = the oldest)
    auto setup_ptrs = [&](int ord) __attribute__((always_inline)) {
        int m0, n0;
        tile_of(ord, m0, n0);
        const int rowa = wave * 16 + (lane >> 2);
        ga = A + (int64_t)min(m0 + rowa, g.M - 1) * g.lda * 2 + ((lane & 3) ^ ((4 - ((rowa >> 2) & 3)) & 3)) * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rowb = (wave * 2 + i) * 16 + (lane >> 2);
            gb[i] = B + (int64_t)min(n0 + rowb, g.N - 1) * g.ldb * 2 + ((lane & 3) ^ ((4 - ((rowb >> 2) & 3)) & 3)) * 16;
        }
        if (g.bias && wave == 0) {
            const int c = n0 + 4 * lane;
            __builtin_amdgcn_global_load_lds((glb_vp)(g.bias + (c < g.N ? c : 0)), (lds_vp)(smem + D_BIAS_OFF + (ord & 3) * 1024), 16, 0, 0);
            ++nvm;
        }
    };
    auto issue_one = [&]() __attribute__((always_inline)) {
        if (issued < total) {
            char* st = smem + (issued & 3) * D_STAGE_BYTES;
            const int k0 = iss_kt * 64;                  // bytes
            __builtin_amdgcn_global_load_lds((glb_vp)(ga + k0), (lds_vp)(st + wave * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)(gb[0] + k0), (lds_vp)(st + D_A_BYTES + (wave * 2) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)(gb[1] + k0), (lds_vp)(st + D_A_BYTES + (wave * 2 + 1) * 1024), 16, 0, 0);
            nvm += 3;
            q2 = nvm;
            ++issued;
            if (++iss_kt == nk) {
                iss_kt = 0;
                if (++iss_ord < my_tiles) setup_ptrs(iss_ord);
            }
        }
    };
    if (my_tiles > 0) setup_ptrs(0);
    issue_one(); q0 = q2;
    issue_one(); q1 = q2;
    issue_one();

    int oa[NI], ob[4];
#pragma unroll
    for (int i = 0; i < NI; ++i) oa[i] = d_off(wr * 64 + i * 16 + fr, fg);
#pragma unroll
    for (int j = 0; j < 4; ++j) ob[j] = D_A_BYTES + d_off(wc * 64 + j * 16 + fr, fg);

    f32x4_t acc[2][NI][4];
    const int cps = nk >= 8 ? 1 : 2;                     // epilogue chunks per k-step (8 chunks per tile; nk >= 4)
    int cs = 0;

    // one (fragment row i, column half jp) chunk of the finished tile `pord` held in acc[P]
    const unsigned lo_c = (unsigned)((fr * (int)g.ldc + (fg & 1) * 16 + (fg >> 1) * 8) * 2);       // lane part of every C store address
    const unsigned lo_x = (unsigned)((fr * (int)g.ldaux + (fg & 1) * 16 + (fg >> 1) * 8) * 2);
    int pm0 = 0, pn0 = 0;                                // origin of the finished tile whose accumulators are being written out
    auto chunk = [&](auto Ptag, auto Itag, auto Jtag, int pord) __attribute__((always_inline)) {
        constexpr int P = decltype(Ptag)::value, i = decltype(Itag)::value, jp = decltype(Jtag)::value;
        const int nw = pn0 + wc * 64, mw = pm0 + wr * 64;
        constexpr int j0 = 2 * jp;
        float b4[2][4];
        if (g.bias) {
            const unsigned ba = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + D_BIAS_OFF + (pord & 3) * 1024 + (wc * 64 + j0 * 16 + 4 * fg) * 4);
            f32x4_t t0, t1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t0), "=&v"(t1) : "v"(ba) : "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) { b4[0][r] = t0[r]; b4[1][r] = t1[r]; }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[0][r] = b4[1][r] = 0.f;
        }
        // addresses: a scalar 64-bit base per chunk + one 32-bit lane offset shared by all chunks (keeps 64-bit lane pointers out of
        // the register file: with them the GELU instance spilled, and a spill reload drains the DMA queue)
        const int row = mw + i * 16 + fr;
        const int col = nw + (j0 + (fg & 1)) * 16 + (fg >> 1) * 8;
        const bool ok = (P128D_X & 1) ? false : (row < g.M && col < g.N);
        // The finished accumulators are loop invariants of the k-loop this chunk sits in: without the opaque pass-through hipcc hoists
        // the whole epilogue's arithmetic in front of the loop and spills it (a spill reload drains the DMA queue with vmcnt(0)).
        f32x4_t a[2] = {acc[P][i][j0], acc[P][i][j0 + 1]};
        asm volatile("" : "+v"(a[0]), "+v"(a[1]));
        unsigned lc = lo_c, lx = lo_x;
        asm volatile("" : "+v"(lc), "+v"(lx));
        float v[2][4], pre[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = g.alpha * a[h][r] + b4[h][r];
                pre[h][r] = t;
                if (EPI == EPI_GELU) t = gelu_fast(t);
                v[h][r] = t;
            }
        // stores: scalar 64-bit base + 32-bit lane offset, written as such (hipcc builds 64-bit lane addresses otherwise and keeps them).
        // The s_nop covers the 5 wait states a VMEM instruction needs after a VALU write (v_readlane of a spilled SGPR) of its base
        // SGPRs: the hazard recogniser does not look inside inline asm.
        if (C) {
            const char* pc = (const char*)C + ((int64_t)(mw + i * 16) * g.ldc + nw + j0 * 16) * 2;
            const auto s0 = __builtin_amdgcn_permlane16_swap(d_pk(v[0][0], v[0][1]), d_pk(v[1][0], v[1][1]), false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(d_pk(v[0][2], v[0][3]), d_pk(v[1][2], v[1][3]), false, false);
            const u32x4_t d = {s0[0], s1[0], s0[1], s1[1]};
            if (ok) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2" ::"v"(lc), "v"(d), "s"(pc) : "memory");
        }
        if (EPI == EPI_GELU && aux) {
            const char* px = (const char*)aux + ((int64_t)(mw + i * 16) * g.ldaux + nw + j0 * 16) * 2;
            const auto s0 = __builtin_amdgcn_permlane16_swap(d_pk(pre[0][0], pre[0][1]), d_pk(pre[1][0], pre[1][1]), false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(d_pk(pre[0][2], pre[0][3]), d_pk(pre[1][2], pre[1][3]), false, false);
            const u32x4_t d = {s0[0], s1[0], s0[1], s1[1]};
            if (ok) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2" ::"v"(lx), "v"(d), "s"(px) : "memory");
        }
        // only a wave whose whole sub-tile is inside the matrix is sure to have issued these stores
        const bool interior = (mw + 64 <= g.M) && (nw + 64 <= g.N);
        if (interior && !(P128D_X & 1)) nvm += nst;
    };
    auto chunk_c = [&](auto Ptag, int c, int pord) __attribute__((always_inline)) {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        switch (c) {
            case 0: chunk(Ptag, I0{}, I0{}, pord); break;
            case 1: chunk(Ptag, I0{}, I1{}, pord); break;
            case 2: chunk(Ptag, I1{}, I0{}, pord); break;
            case 3: chunk(Ptag, I1{}, I1{}, pord); break;
            case 4: chunk(Ptag, I2{}, I0{}, pord); break;
            case 5: chunk(Ptag, I2{}, I1{}, pord); break;
            case 6: chunk(Ptag, I3{}, I0{}, pord); break;
            default: chunk(Ptag, I3{}, I1{}, pord); break;
        }
    };

    // main loop of tile `ord` into acc[P]; tile ord - 1 (if any) leaves acc[1 - P] chunk by chunk
    auto run_tile = [&](auto Ptag, int ord) __attribute__((always_inline)) {
        constexpr int P = decltype(Ptag)::value;
        using Q = std::integral_constant<int, 1 - P>;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[P][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt, ++cs) {
            d_wait_vm(nvm - q0);                         // everything up to and including step cs's loads has landed
            __builtin_amdgcn_s_barrier();
            q0 = q1; q1 = q2;
            issue_one();                                 // step cs + 3 (sets q2)
            // the chunk's stores below must stay behind these loads in the queue: the counts above depend on it
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const char* st = smem + (cs & 3) * D_STAGE_BYTES;
            bf16x8_t fa[NI], fb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8_t*)(st + ob[j]);
#pragma unroll
            for (int i = 0; i < NI; ++i) fa[i] = *(const bf16x8_t*)(st + oa[i]);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[P][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[P][i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (ord > 0 && !(P128D_X & 2)) {
                const int c0 = kt * cps;
                if (c0 < 8) chunk_c(Q{}, c0, ord - 1);
                if (cps == 2 && c0 + 1 < 8) chunk_c(Q{}, c0 + 1, ord - 1);
            }
        }
        tile_of(ord, pm0, pn0);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int ord = 0;
    while (ord < my_tiles) {
        run_tile(P0{}, ord);
        ++ord;
        if (ord >= my_tiles) break;
        run_tile(P1{}, ord);
        ++ord;
    }
    // the last tile's accumulators: nothing left to hide them under
    if (my_tiles > 0) {
        if ((my_tiles - 1) & 1) {
            for (int c = 0; c < 8; ++c) chunk_c(P1{}, c, my_tiles - 1);
        } else {
            for (int c = 0; c < 8; ++c) chunk_c(P0{}, c, my_tiles - 1);
        }
    }
}

static int d_num_cus() {
    static const int n = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        return v > 0 ? v : 256;
    }();
    return n;
}

template <int EPI>
static void d_launch(const GemmArgs& g, hipStream_t stream) {
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p128d<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, D_LDS_BYTES);
        return true;
    }();
    (void)attr;
    const int tiles_m = (int)cdiv(g.M, D_BM), tiles_n = (int)cdiv(g.N, D_BN);
    const int nt = tiles_m * tiles_n;
    const int grid = nt < d_num_cus() ? nt : d_num_cus();
    hipLaunchKernelGGL((gemm_nt_bf16_p128d<EPI>), dim3(grid), dim3(512), D_LDS_BYTES, stream, g, tiles_m, tiles_n);
}

static std::atomic<int> g_p128d_mode{-1};
static int p128d_mode() {
    int m = g_p128d_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("MVULD_GEMM_P128D");
        m = e ? atoi(e) : 0;
        if (m < 0 || m > 2) m = 0;
        g_p128d_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
extern "C" int mvuld_set_gemm_p128d_mode(int mode) {
    MV_CHECK_ARG(mode >= 0 && mode <= 2, "set_gemm_p128d_mode: mode must be 0, 1 or 2");
    g_p128d_mode.store(mode, std::memory_order_relaxed);
    return 0;
}

// 0 = launched; -1 = not this kernel's shape (the caller goes on to the 256-row kernel)
int mvuld_gemm_nt_p128d_try(const GemmArgs& g, hipStream_t stream) {
    const int mode = p128d_mode();
    if (mode == 0 || g.batch != 1 || g.splitk != 1 || g.out_mode != OUT_STORE) return -1;
    if (g.epi != EPI_NONE && g.epi != EPI_BIAS && g.epi != EPI_GELU) return -1;
    if (g.K % 32 != 0 || g.K < 128 || g.N % 8 != 0 || g.ldc % 8 != 0 || !g.C || (((uintptr_t)g.C) & 15) != 0) return -1;
    if (g.aux && (g.epi != EPI_GELU || g.ldaux % 8 != 0 || (((uintptr_t)g.aux) & 15) != 0)) return -1;
    if (g.bias && (((uintptr_t)g.bias) & 15) != 0) return -1;
    if (mode == 1) {
        // wide outputs behind short contractions: the epilogue is a third of such a product and this kernel hides it; long contractions
        // and narrow outputs keep the 256-row tile (fewer L2 -> LDS bytes per FLOP)
        if (g.N < 1536 || g.N % 256 != 0 || g.K > 1024) return -1;
        if (cdiv(g.M, D_BM) * cdiv(g.N, D_BN) < 2 * (int64_t)d_num_cus()) return -1;
    }
    if (g.epi == EPI_GELU) d_launch<EPI_GELU>(g, stream);
    else d_launch<EPI_BIAS>(g, stream);
    return 0;
}
