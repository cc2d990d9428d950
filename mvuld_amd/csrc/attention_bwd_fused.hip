// Fused single-pass backward of SwinV2 window attention on the CDNA4 matrix cores (swin_transformer_v2.py:140-179, 245-268 backward).
// Its own translation unit: built with -mllvm -amdgpu-mfma-vgpr-form (Makefile) so that the score products' results land in arch VGPRs,
// where the VALU works on them, while the resident dK^T / dV^T / dQ^T tiles are accumulated by inline-asm MFMAs in the accumulator file.
#include "attention_common.h"

// MODE 0, head_dim 32, window side ws <= 28 with ws % 4 == 0 (SwinV2-base stages 0-2: 28 x 28 windows): dQ, dK, dV AND the bias-table
// gradient from ONE recomputation of S and dP per (query, key) pair -- the three-pass backward above recomputes the scores, the
// exponentials and dP once per pass (dQ 4.2 + dK/dV 5.3 + bias table 5.5 ms of the round-3 step).  Structure (cdna_hip_programming.md,
// "Attention backward" and "An accumulator tile as the next MFMA's operand"):
//   * tiles are aligned to IMAGE ROWS of the window: a block = one query row x one key row, 32 x 32 slots of which ws x ws are real
//     (v_mfma_f32_32x32x16_bf16).  That costs (32 / 28)^2 of padded scores and buys: the relative-position offset of a block is ONE dy,
//     so a lane's bias words are consecutive table words of one table row, the vertical mask region is block-uniform, and the
//     bias-table gradient of a block is a diagonal fold of the dS tile into ONE table row.
//   * one workgroup of 4 waves (one per SIMD, up to 512 registers each) per (window, head).  Wave w OWNS key rows 7w .. 7w+6: their
//     dK^T and dV^T tiles stay in 224 accumulator registers for the whole sweep, so dK / dV need no sum across waves.  All waves sweep
//     the query rows together; the q-side tile of a row (q~, dO, -lse, -delta; delta = rowsum(dO o O) is formed here) is fetched one
//     row ahead and shared through a double-buffered 4.25 KB LDS tile.
//   * the key is on the LANE in S = Q~.K^T and dP = dO.V^T (rows = queries in the registers), so P and dS are, converted to bf16 pairs,
//     directly the B operands of dV^T += dO^T.P and dK^T += Q~^T.dS; only dS crosses LDS, once, wave-privately (8-byte stores of
//     register quads into a [key][query] image, ds_read_b64_tr_b16 back), for dQ^T += K^T.dS^T.  -lse and the bias are the initial
//     accumulator of S, -delta that of dP.
//   * dQ of a query row is the sum of the four waves' partial tiles: 4.5 KB per wave through LDS, summed by slices (8 queries per wave),
//     which also carry the cosine-normalisation backward and d(logit_scale).
//   * d(bias table): dB[dy][dx] = sum of dS over pairs with that offset.  Within a block dx = qx - kx: the four queries of a register
//     quad are folded along the diagonal with whole-wave DPP shifts (wave_shr:1, tools/microbench/dpp_wave_shift.hip: gfx950 executes
//     the GFX9 whole-wave shifts) -- 3 shifted adds per quad, no LDS -- and because a wave's key rows are consecutive, block (qy, ky)
//     and block (qy + 1, ky + 1) share dy: the folded quads ride a 7-block diagonal chain in registers and touch the LDS table once
//     per step instead of once per block.
typedef float __attribute__((ext_vector_type(16))) f32x16_t;
#ifndef AF_X
#define AF_X 0              // timing experiments (wrong results): 1 = no blocks, 2 = no dQ slice epilogue, 3 = no q-side fetch / commit, 4 = no table-gradient fold / flush
#endif
#if AF_X == 9
__device__ long long af_stamps[4][32];          // in-kernel stamps (shader cycles) of workgroup 0, step 5: cdna_hip_programming.md section 7
#define AF_STAMP(i) do { if (blockIdx.x == 0 && qy == 5 && lane == 0) af_stamps[wave][i] = __builtin_amdgcn_s_memtime(); } while (0)
#define AF_KSTAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) af_stamps[threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int mvuld_debug_af_stamps(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(af_stamps), sizeof(af_stamps)) == hipSuccess ? 0 : 1; }
#else
#define AF_STAMP(i) do { } while (0)
#define AF_KSTAMP(i) do { } while (0)
#endif
#define AF_RPW 7            // key rows per wave
#define AF_WAVES 4
#define AF_NEG 384          // words of -inf behind the bias table: a padding-key lane walks (AF_RPW - 1) table rows + 32 words of it per step
#define AF_STG 2304         // bytes per wave: [q 8 x 64][dO 8 x 64][O 8 x 64][512 unused][lse 64 x 4]
#define AF_NODY (-1000000)  // "no table-gradient row pending"
#define AF_PQ 36            // row stride (words) of a dQ partial tile: 8 consecutive rows of a 16-byte store hit 32 different banks
// [rows][32] bf16 image with 64-byte rows; the 16-byte chunk c of a row whose position inside its window row (or tile) is x sits at
// chunk c ^ ((x >> 2) & 3): 32 consecutive rows of a 16-byte fragment read, and the 4-row blocks of a transposed read, are conflict free
__device__ __forceinline__ int af_off(int row, int x, int chunk) { return row * 32 + ((chunk ^ ((x >> 2) & 3)) << 3); }
__device__ __forceinline__ bf16x8_t af_tr(const bf16* img, int offA, int offB) {
    typedef __attribute__((address_space(3))) bf16x4_t* lds_p;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(img + offA));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(img + offB));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8_t af_trp(const bf16* pa, const bf16* pb) {
    typedef __attribute__((address_space(3))) bf16x4_t* lds_p;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)pa);
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)pb);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// token row of window position (y, x) of window (wy, wx) of sample b: the roll / partition index map without its divisions
__device__ __forceinline__ int64_t af_token(const AttnGeom& g, int ws, int b, int wy, int wx, int y, int x) {
    int oy = wy * ws + y + g.shift, ox = wx * ws + x + g.shift;
    if (oy >= g.res) oy -= g.res;
    if (ox >= g.res) ox -= g.res;
    return ((int64_t)b * g.res + oy) * g.res + ox;
}
// LDS-DMA (global -> LDS, no VGPR in between), issued from inline asm so that hipcc neither counts it nor drains it: a DMA the compiler can
// see gets s_waitcnt vmcnt(0) in front of the next LDS read it can see -- here the first fragment read of the block loop, i.e. the whole fetch
// latency at the top of every step (what the register-staged prefetch of the first form cost once its registers were spilled: 22 % of the
// kernel).  Lane l's 16 (4) bytes land at lds_dst + 16 l (4 l); the issuing wave waits for them itself (af_dma_wait) before it reads them.
typedef __attribute__((address_space(3))) char* af_lds_cp;
__device__ __forceinline__ unsigned af_lds_addr(const void* p) { return (unsigned)(uintptr_t)(af_lds_cp)(char*)p; }
__device__ __forceinline__ void af_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void af_dma4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void af_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// sum over the 8 lanes l & ~7 .. l | 7, VALU only: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
__device__ __forceinline__ float af_sum8(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ float af_sum4(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
    return v;
}
// whole-wave shift by one lane away from lane 0 (wave_shr:1): lane l receives lane l - 1, lane 0 receives 0
__device__ __forceinline__ float af_shr1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
// acc (a resident tile in the ACCUMULATOR registers) += A0.B0 + A1.B1 (two 16-deep k-steps; operands in arch VGPRs).  Inline asm because the
// builtin's register class is a per-function choice of hipcc (-amdgpu-mfma-vgpr-form puts every builtin MFMA on arch VGPRs, which is what
// the score products want; without it every MFMA result lives in AGPRs and each score costs three v_accvgpr moves).  s_nop 1: the packed
// bf16 operands may have been written by the VALU converts immediately in front (hipcc pads nothing inside an asm statement); the chain
// D -> C of the two products needs no padding.  Readers of `acc` other than these statements must go through af_settle.
// SAFE (the run-time-geometry instantiation): the statements sit inside wave-uniform branches there (a wave may own fewer than AF_RPW key rows),
// and at the join of a branch hipcc may COPY a tile (v_accvgpr_mov) right behind the statement that is still writing it -- it does not know
// the statement is an MFMA -- so the tile is settled inside the statement.  The WS = 28 instantiations have no branch around them.
template <bool SAFE>
__device__ __forceinline__ void af_mfma2_acc(f32x16_t& acc, const bf16x8_t& a0, const bf16x8_t& b0, const bf16x8_t& a1, const bf16x8_t& b1) {
    if (SAFE)
        asm("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %3, %4, %0\n\ts_nop 15\n\ts_nop 7"
            : "+a"(acc)
            : "v"(__builtin_bit_cast(u32x4_t, a0)), "v"(__builtin_bit_cast(u32x4_t, b0)), "v"(__builtin_bit_cast(u32x4_t, a1)), "v"(__builtin_bit_cast(u32x4_t, b1)));
    else
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %3, %4, %0"
            : "+a"(acc)
            : "v"(__builtin_bit_cast(u32x4_t, a0)), "v"(__builtin_bit_cast(u32x4_t, b0)), "v"(__builtin_bit_cast(u32x4_t, a1)), "v"(__builtin_bit_cast(u32x4_t, b1)));
}
// the same with the A operands in the accumulator file too (the transposed q-side fragments of a step: read by these statements only)
template <bool SAFE>
__device__ __forceinline__ void af_mfma2_acc_aa(f32x16_t& acc, const u32x4_t& a0, const bf16x8_t& b0, const u32x4_t& a1, const bf16x8_t& b1) {
    if (SAFE)
        asm("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %3, %4, %0\n\ts_nop 15\n\ts_nop 7"
            : "+a"(acc)
            : "a"(a0), "v"(__builtin_bit_cast(u32x4_t, b0)), "a"(a1), "v"(__builtin_bit_cast(u32x4_t, b1)));
    else
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %3, %4, %0"
            : "+a"(acc)
            : "a"(a0), "v"(__builtin_bit_cast(u32x4_t, b0)), "a"(a1), "v"(__builtin_bit_cast(u32x4_t, b1)));
}
// one k-step: acc += A.B
template <bool SAFE>
__device__ __forceinline__ void af_mfma1_acc_aa(f32x16_t& acc, const u32x4_t& a0, const u32x4_t& b0) {
    if (SAFE) asm("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7" : "+a"(acc) : "a"(a0), "v"(b0));
    else asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a0), "v"(b0));
}
// 18+ wait states between the last MFMA that wrote the tile and whatever the compiler does with it next (v_accvgpr_read)
__device__ __forceinline__ void af_settle(f32x16_t& acc) { asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc)); }
__host__ __device__ inline size_t af_lds_bytes(int ws) {
    const int N = ws * ws, W2 = 2 * ws - 1;
    return (size_t)2 * (N - ws + 32) * 64                    // K^, V images
         + (size_t)(((W2 * W2 + 64 + 3) & ~3) + AF_NEG) * 4  // bias table (log2 units) + slack behind the last row (16-byte multiple) + the -inf rows of padding keys
         + (size_t)AF_WAVES * AF_STG                         // LDS-DMA landing zone of the next query row: per wave 8 rows of q, dO, O and their lse
         + (size_t)2 * (32 * 32 * 2 + 32 * 4)                // raw q and 1 / |q| of two query rows (the dQ slice's normalisation backward)
         + (size_t)2 * (2 * 32 * 32 * 2 + 2 * 32 * 4)        // q-side tiles of two query rows: q~, dO, -lse, -delta
         + (size_t)AF_WAVES * 32 * 32 * 2                    // dS transposition images
         + (size_t)AF_WAVES * 32 * AF_PQ * 4                 // dQ partial tiles
         + 64;
}

// WS > 0: the window side as a compile-time constant (28: every row / table offset becomes an immediate of the LDS instructions, every wave owns
// exactly AF_RPW key rows); WS = 0: any ws <= 28 with ws % 4 == 0 at run time (the small geometries of the tests)
// PIPE: block a + 1's LDS reads are issued before block a's vector work (32 more live registers: the masked instantiation, which also holds
// the 16 registers of the x mask, runs without it)
// XM3 (WS > 0, shift == WS / 2: what SwinV2 uses): the x part of the shift mask from THREE registers instead of sixteen.  In a window of the last
// column the positions split at s = ws - shift into two regions; a pair is masked when its query and its key lie on different sides.  For a
// lane (one key) that is a function of the query's side only, and a register's query 8 g + 4 half + i lies on one side for both lane halves
// except where s falls between them: m0 = the lane's addend for low queries, m1 for high ones, m2 = (half ? m1 : m0) for the straddling registers.
template <bool MASK, int WS, bool PIPE = !MASK, bool XM3 = false>
__global__ __launch_bounds__(64 * AF_WAVES) void attn_bwd_fused_win_k(AttnGeom g, const bf16* __restrict__ qkv, const float* __restrict__ table16,
                                                                     const float* __restrict__ logit_scale, const bf16* __restrict__ outp,
                                                                     const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                                     bf16* __restrict__ dqkv, float* __restrict__ part_out,
                                                                     float* __restrict__ dlogit_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HD = 32;
    const int ws = WS ? WS : g.ws, N = ws * ws, W2 = 2 * ws - 1, T2 = W2 * W2;
    const int KR = N - ws + 32;                                   // image rows: the last key row is read 32 slots wide
    bf16* Ki = (bf16*)smem;
    bf16* Vi = Ki + (size_t)KR * 32;
    float* tab = (float*)(Vi + (size_t)KR * 32);                  // [T2 + 64 (+ pad)] then negrow[64]
    float* negrow = tab + ((T2 + 64 + 3) & ~3);
    bf16* Qs = (bf16*)(negrow + AF_NEG);                          // [2][32][32]
    bf16* Ds = Qs + 2 * 1024;                                     // [2][32][32]
    float* Nl = (float*)(Ds + 2 * 1024);                          // [2][32]  -lse * log2(e)   (padding queries: -inf)
    float* Nd = Nl + 64;                                          // [2][32]  -delta
    bf16* Tb = (bf16*)(Nd + 64);                                  // [waves][32 keys][32 queries]
    float* Pq = (float*)(Tb + AF_WAVES * 1024);                   // [waves][32 queries][AF_PQ]
    char* Stg = (char*)(Pq + AF_WAVES * 32 * AF_PQ);              // [waves][AF_STG]
    bf16* Rq = (bf16*)(Stg + AF_WAVES * AF_STG);                  // [2][32][32] raw q, row-major 64-byte rows
    float* Rn = (float*)(Rq + 2 * 1024);                          // [2][32] 1 / |q|
    float* red = Rn + 64;                                         // [16]

    const int bwh = am_xcd_order(blockIdx.x, gridDim.x);
    const int h = bwh % g.H, bw = bwh / g.H, b = bw / g.nW, w = bw % g.nW;
    const int C = g.H * HD;
    const int64_t rs = 3 * (int64_t)C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, hh = lane >> 5;
    const int64_t lse0 = ((int64_t)bw * g.H + h) * N;
    float* pout = part_out + (size_t)bwh * T2;

    if (am_dropped(g, b)) {                  // d(out) of this sample is zero: dQ = dK = dV = 0 and no share of the table gradient
        for (int i = tid; i < N * 4; i += blockDim.x) {
            const int64_t t = am_token(g, b, w, i >> 2);
            bf16* o = dqkv + t * rs + h * HD + (i & 3) * 8;
            *(uint4*)o = make_uint4(0, 0, 0, 0);
            *(uint4*)(o + C) = make_uint4(0, 0, 0, 0);
            *(uint4*)(o + 2 * C) = make_uint4(0, 0, 0, 0);
        }
        for (int i = tid; i < T2; i += blockDim.x) pout[i] = 0.f;
        return;
    }
    // the table gradient of this (window, head) is summed into its own partial table with float atomics (no return value: nothing waits for
    // them; 60 consecutive words per wave and step): zeroed here, ordered before the first atomic by the barrier behind the staging
    for (int i = tid; i < T2; i += blockDim.x) pout[i] = 0.f;

    AF_KSTAMP(16);
    const float tau = __expf(fminf(logit_scale[h], LN100));
    const int nwx = g.res / ws, wy = w / nwx, wx = w % nwx;
    // ---- stage K^ (normalised) and V.  Everything global of the prologue goes out in ONE burst: K and V rows by LDS-DMA straight into their
    // swizzled images (a DMA instruction fills 16 rows x 64 bytes in lane order, so the swizzle is applied to the SOURCE chunk a lane fetches:
    // LDS slot sl of row n takes source chunk sl ^ ((x >> 2) & 3) -- an involution), the head's bias table as strided loads into registers, the
    // first query row (below).  One wait, then the keys are normalised in place.  (First form: four register-staged batches of K / V, then the
    // table, then the first row -- six global round trips in a row: 15 us of the 160 us a (window, head) takes at stage 2.)
    {
        const unsigned ki_a = __builtin_amdgcn_readfirstlane(af_lds_addr(Ki)), vi_a = __builtin_amdgcn_readfirstlane(af_lds_addr(Vi));
        const int slot = lane & 3;
        for (int j = wave; j < N / 16; j += AF_WAVES) {              // N = ws^2 with ws % 4 == 0: whole 16-row pieces
            const int n = 16 * j + (lane >> 2), y = n / ws, x = n - y * ws;
            const bf16* p = qkv + af_token(g, ws, b, wy, wx, y, x) * rs + C + h * HD + ((slot ^ ((x >> 2) & 3)) << 3);
            af_dma16(p, ki_a + j * 1024);
            af_dma16(p + C, vi_a + j * 1024);
        }
        for (int i = N * 4 + tid; i < KR * 4; i += blockDim.x) {     // the rows behind the window (read 32 slots wide by the last key row): zeros
            *(uint4*)(Ki + i * 8) = make_uint4(0, 0, 0, 0);
            *(uint4*)(Vi + i * 8) = make_uint4(0, 0, 0, 0);
        }
    }
    {
        constexpr int TU = 13;                                       // table words per thread at ws = 28: (55^2 + 64) / 256
        float tv[TU];
        const int nt = (T2 + 64 + (int)blockDim.x - 1) / (int)blockDim.x;
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int i = tid + u * (int)blockDim.x;
            tv[u] = (u < nt && i < T2) ? table16[(int64_t)i * g.H + h] : 0.f;
        }
        for (int i = tid + TU * (int)blockDim.x; i < T2; i += blockDim.x) tab[i] = table16[(int64_t)i * g.H + h] * LOG2E;      // (larger tables: none here)
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int i = tid + u * (int)blockDim.x;
            if (i < T2 + 64) tab[i] = tv[u] * LOG2E;
        }
    }
    for (int i = tid; i < AF_NEG; i += blockDim.x) negrow[i] = NEG_BIG;

    // ---- q-side tile of one query row.  Wave w fetches (LDS-DMA) and prepares rows 8 w .. 8 w + 7: lane (row = lane >> 3, pc = lane & 7), pc 0-3
    // = the four 16-byte chunks of q, 4-7 = those of dO (and O, for delta = rowsum(dO o O))
    // (their per-lane address arithmetic takes an OPAQUE copy of the lane index: hoisted out of the step loop it would be a dozen more
    //  live registers, i.e. spills, and a scratch reload inside the loop shares vmcnt with the DMA)
    char* stg = Stg + wave * AF_STG;
    const unsigned stg_a = __builtin_amdgcn_readfirstlane(af_lds_addr(stg));
    auto q_issue = [&](int qy) {
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const int hh = lane >> 5;
        // instruction 1: lanes 0-31 = (row lane >> 2, chunk lane & 3) of q, lanes 32-63 the same of dO; instruction 2: of O (both halves);
        // instruction 3: the rows' lse (lane & 7 = row)
        const int x1 = min(wave * 8 + ((lane & 31) >> 2), ws - 1), ch = lane & 3;
        const int64_t t1 = af_token(g, ws, b, wy, wx, qy, x1);
        const bf16* p1 = hh == 0 ? qkv + t1 * rs + h * HD + ch * 8 : dout + t1 * C + h * HD + ch * 8;
        af_dma16(p1, stg_a);
        af_dma16(outp + t1 * C + h * HD + ch * 8, stg_a + 1024);
        af_dma4(lse + lse0 + qy * ws + min(wave * 8 + (lane & 7), ws - 1), stg_a + 2048);
    };
    auto q_commit = [&](int buf) {                              // after af_dma_wait() of the issuing wave (the same wave)
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        const int r = lane >> 3, pc = lane & 7, px = wave * 8 + r;
        const bool pv = px < ws;
        U8 a, o;
        a.u = *(const uint4*)(stg + (pc < 4 ? 0 : 512) + r * 64 + (pc & 3) * 16);
        o.u = *(const uint4*)(stg + (pc < 4 ? 0 : 1024) + r * 64 + (pc & 3) * 16);
        float x[8], acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[e] = (float)a.e[e]; acc += x[e] * (float)o.e[e]; }           // q: sum q^2; dO: sum dO * O
        acc = af_sum4(acc);
        U8 w8;
        if (pc < 4) {
            const float inv = 1.0f / fmaxf(sqrtf(acc), 1e-12f);
            const float sc = pv ? tau * LOG2E * inv : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) w8.e[e] = (bf16)(x[e] * sc);
            *(uint4*)(Qs + buf * 1024 + af_off(px, px, pc)) = w8.u;
            *(uint4*)(Rq + buf * 1024 + px * 32 + pc * 8) = a.u;
            if (pc == 0) {
                Nl[buf * 32 + px] = pv ? -((const float*)(stg + 2048))[r] * LOG2E : NEG_BIG;
                Rn[buf * 32 + px] = inv;
            }
        } else {
            w8.u = pv ? a.u : make_uint4(0, 0, 0, 0);
            *(uint4*)(Ds + buf * 1024 + af_off(px, px, pc - 4)) = w8.u;
            if (pc == 4) Nd[buf * 32 + px] = pv ? -acc : 0.f;
        }
    };
    AF_KSTAMP(17);
    q_issue(0);
    af_dma_wait();
    __syncthreads();                                             // every wave's K / V pieces have landed
    for (int c = tid; c < N * 4; c += blockDim.x) {              // k^ = k / |k| in place: the four slots of a row sit in four neighbouring lanes
        U8 xk;
        xk.u = *(const uint4*)(Ki + c * 8);
        float f[8], ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { f[e] = (float)xk.e[e]; ss += f[e] * f[e]; }
        ss = af_sum4(ss);
        const float sc = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int e = 0; e < 8; ++e) xk.e[e] = (bf16)(f[e] * sc);
        *(uint4*)(Ki + c * 8) = xk.u;
    }
    q_commit(0);
    __syncthreads();
    AF_KSTAMP(18);

    // ---- per-lane fragment offsets inside a 32-row image (bf16 elements)
    const int qq = (lane & 15) >> 2, pp = lane & 3, grp = (lane >> 4) & 1;
    int rowf[2], trq[2][2], trk[2][2];
    auto troff = [&](int rowbase) { const int row = rowbase + qq; return af_off(row, row, 2 * grp + (pp >> 1)) + 4 * (pp & 1); };
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        rowf[s] = af_off(r31, r31, 2 * s + hh);
        trq[s][0] = troff(16 * s + 4 * hh); trq[s][1] = troff(16 * s + 8 + 4 * hh);          // k order of an accumulator tile
        trk[s][0] = troff(16 * s + 8 * hh); trk[s][1] = troff(16 * s + 8 * hh + 4);          // natural k order
    }
    const bool wmask = MASK && am_window_masked(g, w);
    // The blocks across a shifted window's vertical mask split (am_ysplit of the three-pass kernels) are NOT skipped here: a wave-uniform branch
    // around the accumulating MFMA statements makes hipcc copy accumulator tiles at the join while they are still being written (see
    // af_mfma2_acc); the masked instantiation serves 2 of the step's 22 window-attention launches.
    constexpr bool yskip = false;
    f32x16_t xm;                                                   // x part of the shift mask: -100 where the regions of (qx, kx) differ
#pragma unroll
    for (int r = 0; r < 16; ++r) xm[r] = 0.f;
    float xm3[3] = {0.f, 0.f, 0.f};
    if (MASK && wmask && !XM3) {
        const int rk = am_rid(g, wx * ws + min(r31, ws - 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qx = (r & 3) + 8 * (r >> 2) + 4 * hh;
            xm[r] = am_rid(g, wx * ws + min(qx, ws - 1)) != rk ? -100.0f * LOG2E : 0.f;
        }
    }
    if (MASK && wmask && XM3 && wx == nwx - 1) {
        const bool khigh = min(r31, ws - 1) >= ws - g.shift;
        xm3[0] = khigh ? -100.0f * LOG2E : 0.f;
        xm3[1] = khigh ? 0.f : -100.0f * LOG2E;
        xm3[2] = hh ? xm3[1] : xm3[0];
    }
    const bool kpad = r31 >= ws;
    const float* blane = tab + (ws - 1 - min(r31, ws - 1)) + 4 * hh;                            // + (dy + ws - 1) * W2 per block
    const float* bneg = negrow + 4 * hh;
    bf16* Tw = Tb + wave * 1024;
    float* Pw = Pq + wave * 32 * AF_PQ;
    const int ky0 = wave * AF_RPW;
    // per-lane fragment pointers of the wave's FIRST key row: block a adds a * ws * 32 elements (WS = 28: an immediate of the LDS instruction)
    const bf16* Kw = Ki + ky0 * ws * 32;
    const bf16* Vw = Vi + ky0 * ws * 32;
    const bf16* kp_row[2] = {Kw + rowf[0], Kw + rowf[1]};
    const bf16* vp_row[2] = {Vw + rowf[0], Vw + rowf[1]};
    const bf16* kp_tr[2][2] = {{Kw + trk[0][0], Kw + trk[0][1]}, {Kw + trk[1][0], Kw + trk[1][1]}};
    const bf16* tp_tr[2][2] = {{Tw + trk[0][0], Tw + trk[0][1]}, {Tw + trk[1][0], Tw + trk[1][1]}};
    bf16* tp_wr[4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) tp_wr[gq] = Tw + af_off(r31, r31, gq) + 4 * hh;
    const int nrow = WS == 4 * AF_RPW ? AF_RPW : max(0, min(AF_RPW, ws - ky0));               // key rows this wave owns (wave-uniform)

    f32x16_t dk[AF_RPW], dv[AF_RPW];
    f32x4_t R[AF_RPW];                                            // folded dS quads on their diagonal chains
#pragma unroll
    for (int a = 0; a < AF_RPW; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[a][r] = 0.f; dv[a][r] = 0.f; }
        R[a] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    // table gradient row dy += the folded quads: quad gq of lane (kx', half), kx' <= ws + 2, belongs to column c = 8 gq + 4 half + 3 - kx' + ws - 1.
    // Columns of different (gq, half) overlap across lanes, so the eight pieces are first gathered per COLUMN (lane c collects its up to
    // eight contributors with ds_bpermute: independent, one LDS latency) and the row of the workgroup's partial table takes one float atomic
    // per column, without return: the wave does not wait for it.
    auto gather = [&](const f32x4_t& v) {
        float gsum = 0.f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kx = 8 * gq + 4 * half + ws + 2 - lane;
                const float t = __int_as_float(__builtin_amdgcn_ds_bpermute(((half << 5) | (kx & 31)) << 2, __float_as_int(v[gq])));
                gsum += (kx >= 0 && kx < ws + 3) ? t : 0.f;
            }
        return gsum;
    };
    auto flush = [&](float gsum, int dy) {
        if (lane < W2) unsafeAtomicAdd(pout + (dy + ws - 1) * W2 + lane, gsum);
    };

    float dtau_part = 0.f;
    // the chain that ended in the previous step: gathered there, added to the partial table at the top of this step -- an atomic issued at the
    // end of a step would still be counted in vmcnt when the step waits for its LDS-DMA (measured with in-kernel stamps: ~2 000 cycles)
    float gpend = 0.f;
    int dypend = AF_NODY;
    // One block = (query row qy) x (key row ky0 + a).  The seven blocks of a step run as a software pipeline -- one wave per SIMD hides no
    // latency by itself (first form of this kernel: half of all wave cycles parked in s_waitcnt, profiles/r04_fused_attn_counters.csv):
    //   L(a)  issue the LDS reads of block a: 16 bias words, two K^ and two V row fragments
    //   S(a)  score init (bias - lse [+ mask]) and the four MFMAs of S and dP
    //   X(a)  exp2 / dS / bf16 pairs, the diagonal fold, dV^T and dK^T MFMAs, dS^T quads -> LDS
    //   Qi(a) issue the transposed reads of dS^T and K^T;  Qm(a) the two dQ^T MFMAs
    // in the order  L(a+1) X(a) Qi(a) S(a+1) Qm(a):  block a+1's reads fly under block a's vector work, and its score MFMAs are already in
    // the matrix pipe while the wave waits for its own dS^T to come back from LDS.
    struct Blk { f32x16_t b; bf16x8_t kf[2], vf[2]; };
    // q-side fragments of a step: row fragments of q~ and dO (A operands of S and dP), their transposes (A operands of dK^T and dV^T, in the
    // accumulator file), -lse and -delta per register.  Loaded right behind the barrier that publishes the row's tile -- i.e. during the
    // previous step's dQ slice epilogue -- not at the top of the step
    bf16x8_t qa[2], da[2];
    u32x4_t qT[2], dT[2];
    f32x16_t nl, ndl;
    auto load_frags = [&](int buf) {
        const bf16* Qc = Qs + buf * 1024;
        const bf16* Dc = Ds + buf * 1024;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qa[s] = *(const bf16x8_t*)(Qc + rowf[s]);
            da[s] = *(const bf16x8_t*)(Dc + rowf[s]);
            qT[s] = __builtin_bit_cast(u32x4_t, af_tr(Qc, trq[s][0], trq[s][1]));
            dT[s] = __builtin_bit_cast(u32x4_t, af_tr(Dc, trq[s][0], trq[s][1]));
        }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4_t a4 = *(const f32x4_t*)(Nl + buf * 32 + 8 * gq + 4 * hh);
            const f32x4_t b4 = *(const f32x4_t*)(Nd + buf * 32 + 8 * gq + 4 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) { nl[4 * gq + i] = a4[i]; ndl[4 * gq + i] = b4[i]; }
        }
    };
    for (int qy = 0; qy < ws; ++qy) {
        const int cur = qy & 1;
        AF_STAMP(0);
        int lane_e = threadIdx.x & 63;
        asm volatile("" : "+v"(lane_e));
        const int xq = wave * 8 + (lane_e >> 3), dc = lane_e & 7;  // query position in the row, dims 4 dc .. 4 dc + 3
        load_frags(qy & 1);                                       // (their LDS latency passes under the address arithmetic of the fetch below)
        if (dypend > AF_NODY) flush(gpend, dypend);
        if (qy + 1 < ws && AF_X != 3) q_issue(qy + 1);
        AF_STAMP(1);
        f32x16_t dq;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[r] = 0.f;
        const int rq = (MASK && wmask) ? am_rid(g, wy * ws + qy) : 0;
        auto ydiff_of = [&](int a) { return MASK && wmask && am_rid(g, wy * ws + ky0 + a) != rq; };
        auto live = [&](int a) { return a < nrow && !(ydiff_of(a) && yskip); };          // wave-uniform
        // bias words of block a: table row dy + ws - 1 = qy - ky0 - a + ws - 1.  One lane base per step, pointing at the row of the wave's LAST
        // key row (the lowest address), so that every block's row is a non-negative (WS = 28: immediate) offset; padding-key lanes point into
        // the -inf words instead and walk them by the same offsets
        const float* bstep = kpad ? bneg : blane + (qy - ky0 - (nrow - 1) + ws - 1) * W2;
        auto stage_Lb = [&](int a, Blk& L) {                        // the 16 bias words
            const float* bl = bstep + (nrow - 1 - a) * W2;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int i = 0; i < 4; ++i) L.b[4 * gq + i] = bl[8 * gq + i];
        };
        auto stage_Lkv = [&](int a, Blk& L) {                       // the K^ and V row fragments
#pragma unroll
            for (int s = 0; s < 2; ++s) { L.kf[s] = *(const bf16x8_t*)(kp_row[s] + a * ws * 32); L.vf[s] = *(const bf16x8_t*)(vp_row[s] + a * ws * 32); }
        };
        auto stage_S = [&](int a, const Blk& L, f32x16_t& sc, f32x16_t& dp) {
            sc = L.b + nl;
            if (MASK && wmask) {
                const bool yd = ydiff_of(a);
                if constexpr (XM3) {
                    constexpr int SX = WS / 2;                      // queries 8 g + 4 half + i >= SX are the high side
                    const float mm[3] = {yd ? -100.0f * LOG2E : xm3[0], yd ? -100.0f * LOG2E : xm3[1], yd ? -100.0f * LOG2E : xm3[2]};
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = (r & 3) + 8 * (r >> 2);
                        sc[r] += c >= SX ? mm[1] : (c + 4 >= SX ? mm[2] : mm[0]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[r] += yd ? -100.0f * LOG2E : xm[r];
                }
            }
            // both k-steps of S first: the exponentials of the block then run under the dP (and the previous block's dQ) products
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[0], L.kf[0], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[1], L.kf[1], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[0], L.vf[0], ndl, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[1], L.vf[1], dp, 0, 0, 0);
        };
        AF_STAMP(2);
        Blk L[2];
        f32x16_t sc, dp;
#if AF_X == 1
        if (false)
#endif
        if (live(0)) { stage_Lb(0, L[0]); stage_Lkv(0, L[0]); stage_S(0, L[0], sc, dp); }
        f32x4_t prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < AF_RPW; ++a) {
            if (a < nrow && AF_X != 1) {                                        // wave-uniform
                const int ky = ky0 + a;
                const bool on = live(a), on1 = a + 1 < AF_RPW && live(a + 1);
                if (PIPE && on1) stage_Lkv(a + 1, L[(a + 1) & 1]);
                f32x4_t F = {0.f, 0.f, 0.f, 0.f};
                bf16x8_t k0, t0, k1, t1;
                if (on) {
                    u32x4_t pw[2], dw[2];
                    f32x16_t ds;
                    // two halves (queries 16 s .. 16 s + 15 = registers 8 s .. 8 s + 7 = k-step s of the dV^T / dK^T products): the first half's
                    // quads go to LDS and its two MFMAs into the matrix pipe while the VALU works on the second half
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
#pragma unroll
                        for (int r = 8 * s; r < 8 * s + 8; r += 2) {
                            const f32x2_t p2 = am_exp2((f32x2_t){sc[r], sc[r + 1]});
                            const f32x2_t d2 = p2 * (f32x2_t){dp[r], dp[r + 1]};
                            ds[r] = d2[0]; ds[r + 1] = d2[1];
                            pw[s][(r >> 1) & 3] = am_pk(p2);
                            dw[s][(r >> 1) & 3] = am_pk(d2);
                        }
                        // dS^T through LDS: register quad gq (queries 8 gq + 4 half ..+3) of key r31 -> [key][query] image
                        *(uint2*)tp_wr[2 * s] = make_uint2(dw[s][0], dw[s][1]);
                        *(uint2*)tp_wr[2 * s + 1] = make_uint2(dw[s][2], dw[s][3]);
                        af_mfma1_acc_aa<WS == 0>(dv[a], dT[s], pw[s]);
                        af_mfma1_acc_aa<WS == 0>(dk[a], qT[s], dw[s]);
                    }
                    asm volatile("" ::: "memory");
                    // every LDS request of the block's tail goes out NOW, in one burst: the transposed reads of dS^T (behind the quads' stores: the
                    // LDS executes a wave's operations in order) and of K^T, and block a + 1's bias words; the diagonal fold below (16 dependent
                    // DPP instructions, no memory) covers their latency.  Left to hipcc the transposed reads sit directly in front of their wait.
                    k0 = af_trp(kp_tr[0][0] + a * ws * 32, kp_tr[0][1] + a * ws * 32); t0 = af_trp(tp_tr[0][0], tp_tr[0][1]);
                    k1 = af_trp(kp_tr[1][0] + a * ws * 32, kp_tr[1][1] + a * ws * 32); t1 = af_trp(tp_tr[1][0], tp_tr[1][1]);
                    if (on1) stage_Lb(a + 1, L[(a + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    // diagonal fold of the quads: lane kx' ends up with the sum over i of ds[4 gq + i] of lane kx' - (3 - i), i.e. with the pairs
                    // of offset dx = 8 gq + 4 half + 3 - kx'.  The values move towards the (at least three, ws <= 28) padding-key lanes behind
                    // the row, whose own dS is exactly 0 -- they also keep the halves apart: nothing real crosses from lane 31 into lane 32
                    // (the four quads level by level: a DPP operand written by the instruction in front costs two wait states)
                    if (AF_X != 4) {
                        f32x4_t u;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) u[gq] = ds[4 * gq + 1] + af_shr1(ds[4 * gq]);
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) u[gq] = ds[4 * gq + 2] + af_shr1(u[gq]);
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) F[gq] = ds[4 * gq + 3] + af_shr1(u[gq]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!PIPE && on1) stage_Lkv(a + 1, L[(a + 1) & 1]);
                if (on1) stage_S(a + 1, L[(a + 1) & 1], sc, dp);
                if (on) af_mfma2_acc<WS == 0>(dq, k0, t0, k1, t1);
                const f32x4_t t = R[a];
                R[a] = prev + F;
                prev = t;
                if (a == nrow - 1 && AF_X != 4) { gpend = gather(R[a]); dypend = qy - ky; }          // the chain ends at the wave's last key row
                AF_STAMP(3 + a);
            }
            // the next query row's tile (fetched by LDS-DMA since the top of the step) is normalised and published HERE, in the middle of the
            // blocks: its LDS latencies and ~90 vector instructions then fall into the MFMA latencies of the block pipeline instead of
            // standing alone in front of the barrier
            if (a == 3 && qy + 1 < ws && AF_X != 3) { af_dma_wait(); q_commit(cur ^ 1); }
        }
        // ---- dQ of this query row: partial tile -> LDS (lane = query r31, registers = dims (r & 3) + 8 (r >> 2) + 4 half)
        af_settle(dq);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *(f32x4_t*)(Pw + r31 * AF_PQ + 8 * gq + 4 * hh) = (f32x4_t){dq[4 * gq], dq[4 * gq + 1], dq[4 * gq + 2], dq[4 * gq + 3]};
        AF_STAMP(10);
        AF_STAMP(11);
        __syncthreads();
        AF_STAMP(12);
        if (AF_X != 2) {
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ww = 0; ww < AF_WAVES; ++ww) v += *(const f32x4_t*)(Pq + (ww * 32 + xq) * AF_PQ + 4 * dc);
            const bool qv = xq < ws;
            U4 rawq;
            rawq.u = *(const uint2*)(Rq + cur * 1024 + xq * 32 + 4 * dc);     // raw q and 1 / |q| of this row: left by q_commit a step ago
            const float qinv = Rn[cur * 32 + xq];
            float qh[4], dot = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { qh[e] = (float)rawq.e[e] * qinv; dot += v[e] * qh[e]; }
            if (qv) dtau_part += dot;
            dot = af_sum8(dot) * tau;                              // q^ . d(q^)
            if (qv) {
                const int64_t tqr = af_token(g, ws, b, wy, wx, qy, xq);
                U4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o.e[e] = (bf16)((tau * v[e] - qh[e] * dot) * qinv);
                *(uint2*)(dqkv + tqr * rs + h * HD + 4 * dc) = o.u;
            }
        }
        AF_STAMP(13);
        __syncthreads();
        AF_STAMP(14);
    }
    AF_KSTAMP(19);
    // ---- the chains still open after the last query row (dy of chain a: ws - 1 - ky)
    if (dypend > AF_NODY) flush(gpend, dypend);
#pragma unroll
    for (int a = 0; a < AF_RPW; ++a)
        if (a < nrow - 1) flush(gather(R[a]), ws - 1 - (ky0 + a));
    // ---- dK, dV of the wave's key rows: lane = key position r31, registers = dims (r & 3) + 8 (r >> 2) + 4 half.
    // The raw keys of ALL the wave's rows (normalisation backward) are requested first, in one burst: one global round trip per workgroup
    // instead of one per key row (seven dependent ones were 10 us of the 150 us a (window, head) takes at stage 2)
    U4 kraw[AF_RPW][4];
    int64_t ktok[AF_RPW];
#pragma unroll
    for (int a = 0; a < AF_RPW; ++a) {
        ktok[a] = af_token(g, ws, b, wy, wx, min(ky0 + a, ws - 1), min(r31, ws - 1));
        const bf16* kp = qkv + ktok[a] * rs + C + h * HD + 4 * hh;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) kraw[a][gq].u = *(const uint2*)(kp + 8 * gq);
    }
#pragma unroll
    for (int a = 0; a < AF_RPW; ++a) {
        // (their last MFMA retired a whole dQ-slice epilogue and two barriers ago: an ordering fence for hipcc, no wait states)
        asm volatile("" : "+a"(dk[a]));
        asm volatile("" : "+a"(dv[a]));
        if (a < nrow) {
            const int64_t t = ktok[a];
            float kh[16], ss = 0.f;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { kh[4 * gq + e] = (float)kraw[a][gq].e[e]; ss += kh[4 * gq + e] * kh[4 * gq + e]; }
            }
            ss += __shfl_xor(ss, 32, 64);
            const float kinv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { kh[r] *= kinv; dk[a][r] *= LN2; dot += dk[a][r] * kh[r]; }     // dk was accumulated against q~ * log2(e)
            dot += __shfl_xor(dot, 32, 64);
            if (!kpad) {
                bf16* o0 = dqkv + t * rs + C + h * HD + 4 * hh;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    U4 ok, ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ok.e[e] = (bf16)((dk[a][4 * gq + e] - kh[4 * gq + e] * dot) * kinv);
                        ov.e[e] = (bf16)dv[a][4 * gq + e];
                    }
                    *(uint2*)(o0 + 8 * gq) = ok.u;
                    *(uint2*)(o0 + C + 8 * gq) = ov.u;
                }
            }
        }
    }
    AF_KSTAMP(20);
    dtau_part = wave_sum(dtau_part);
    if (lane == 0) red[wave] = dtau_part;
    __syncthreads();
    if (tid == 0 && logit_scale[h] < LN100) {
        float t = 0.f;
        for (int i = 0; i < AF_WAVES; ++i) t += red[i];
        atomicAdd(dlogit_scale + h, t * tau);
    }
}


// ------------------------------------------------------------------------------------------------ launcher (called by mvuld_attn_bwd_mfma)
int af_supported(int hd, int ws) { return hd == 32 && ws >= 4 && ws <= 4 * AF_RPW && (ws & 3) == 0 && af_lds_bytes(ws) <= 160 * 1024; }

int af_launch(const AttnGeom& g, int shift, int64_t groups, const void* qkv, const float* table16, const float* logit_scale, const void* out,
              const void* dout, const float* lse, void* dqkv, float* ws_part, float* dlogit_scale, hipStream_t stream) {
    const size_t bytes = af_lds_bytes(g.ws);
#define AM_FUSED(MASKV, WSV, PIPEV, XM3V)                                                                                                \
    do {                                                                                                                 \
        if (hipFuncSetAttribute((const void*)attn_bwd_fused_win_k<MASKV, WSV, PIPEV, XM3V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { \
            mvuld_set_error("attn_bwd_fused_win_k: hipFuncSetAttribute(%zu) failed", bytes);                            \
            return 1;                                                                                                    \
        }                                                                                                                \
        hipLaunchKernelGGL((attn_bwd_fused_win_k<MASKV, WSV, PIPEV, XM3V>), dim3((unsigned)groups), dim3(64 * AF_WAVES), bytes, stream, g, (const bf16*)qkv, \
                           table16, logit_scale, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, ws_part, dlogit_scale);            \
    } while (0)
    if (g.ws == 28 && shift == 0) AM_FUSED(false, 28, true, false);
    else if (g.ws == 28 && shift == 14) AM_FUSED(true, 28, true, true);
    else AM_FUSED(true, 0, false, false); // the generic instantiation: masked form only (it also serves unshifted blocks)
#undef AM_FUSED
    return 0;
}
