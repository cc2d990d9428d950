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
#define AF_RPW 7            // key rows per wave
#define AF_WAVES 4
#define AF_DS 60            // row stride (words) of the LDS table gradient: indices 8g + 4h - kx + ws - 1 <= 58
#define AF_NEG 384          // words of -inf behind the bias table: a padding-key lane walks (AF_RPW - 1) table rows + 32 words of it per step
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
// 18+ wait states between the last MFMA that wrote the tile and whatever the compiler does with it next (v_accvgpr_read)
__device__ __forceinline__ void af_settle(f32x16_t& acc) { asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc)); }
__host__ __device__ inline size_t af_lds_bytes(int ws) {
    const int N = ws * ws, W2 = 2 * ws - 1;
    return (size_t)2 * (N - ws + 32) * 64                    // K^, V images
         + (size_t)(((W2 * W2 + 64 + 3) & ~3) + AF_NEG) * 4  // bias table (log2 units) + slack behind the last row (16-byte multiple) + the -inf rows of padding keys
         + (size_t)W2 * AF_DS * 4                            // table gradient
         + (size_t)2 * (2 * 32 * 32 * 2 + 2 * 32 * 4)        // q-side tiles of two query rows: q~, dO, -lse, -delta
         + (size_t)AF_WAVES * 32 * 32 * 2                    // dS transposition images
         + (size_t)AF_WAVES * 32 * AF_PQ * 4                 // dQ partial tiles
         + 64;
}

// WS > 0: the window side as a compile-time constant (28: every row / table offset becomes an immediate of the LDS instructions, every wave owns
// exactly AF_RPW key rows); WS = 0: any ws <= 28 with ws % 4 == 0 at run time (the small geometries of the tests)
// PIPE: block a + 1's LDS reads are issued before block a's vector work (32 more live registers: the masked instantiation, which also holds
// the 16 registers of the x mask, runs without it)
template <bool MASK, int WS, bool PIPE = !MASK>
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
    float* dtab = negrow + AF_NEG;                                // [W2][AF_DS]
    bf16* Qs = (bf16*)(dtab + W2 * AF_DS);                        // [2][32][32]
    bf16* Ds = Qs + 2 * 1024;                                     // [2][32][32]
    float* Nl = (float*)(Ds + 2 * 1024);                          // [2][32]  -lse * log2(e)   (padding queries: -inf)
    float* Nd = Nl + 64;                                          // [2][32]  -delta
    bf16* Tb = (bf16*)(Nd + 64);                                  // [waves][32 keys][32 queries]
    float* Pq = (float*)(Tb + AF_WAVES * 1024);                   // [waves][32 queries][AF_PQ]
    float* red = Pq + AF_WAVES * 32 * AF_PQ;                      // [16]

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

    const float tau = __expf(fminf(logit_scale[h], LN100));
    const int nwx = g.res / ws, wy = w / nwx, wx = w % nwx;
    // ---- stage K^ (normalised) and V; chunk c of position n: threads 4n .. 4n+3
    for (int c0 = tid; c0 < KR * 4; c0 += 4 * blockDim.x) {
        U8 xk[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * blockDim.x, n = c >> 2;
            xk[u].u = make_uint4(0, 0, 0, 0); xv[u].u = xk[u].u;
            if (n < N) {
                const bf16* p = qkv + am_token(g, b, w, n) * rs + C + h * HD + (c & 3) * 8;
                xk[u].u = *(const uint4*)p;
                xv[u].u = *(const uint4*)(p + C);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * blockDim.x, n = c >> 2;
            float f[8], ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[e] = (float)xk[u].e[e]; ss += f[e] * f[e]; }
            ss += __shfl_xor(ss, 1, 64);
            ss += __shfl_xor(ss, 2, 64);
            const float sc = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int e = 0; e < 8; ++e) xk[u].e[e] = (bf16)(f[e] * sc);
            if (n < KR) {
                const int o = af_off(n, n < N ? n % ws : 0, c & 3);
                *(uint4*)(Ki + o) = xk[u].u;
                *(uint4*)(Vi + o) = xv[u].u;
            }
        }
    }
    for (int i = tid; i < T2 + 64; i += blockDim.x) tab[i] = i < T2 ? table16[(int64_t)i * g.H + h] * LOG2E : 0.f;
    for (int i = tid; i < AF_NEG; i += blockDim.x) negrow[i] = NEG_BIG;
    for (int i = tid; i < W2 * AF_DS; i += blockDim.x) dtab[i] = 0.f;

    // ---- q-side tile of one query row: thread (px = position in the row, pc = chunk role: 0-3 q, 4-7 dO / O)
    const int px = tid >> 3, pc = tid & 7;
    const bool pv = px < ws;
    struct QFetch { U8 a, o; float l; };
    auto q_issue = [&](int qy, QFetch& f) {
        const int n = qy * ws + min(px, ws - 1);
        const int64_t t = af_token(g, ws, b, wy, wx, qy, min(px, ws - 1));
        const bf16* pa = pc < 4 ? qkv + t * rs + h * HD + pc * 8 : dout + t * C + h * HD + (pc - 4) * 8;
        const bf16* po = pc < 4 ? pa : outp + t * C + h * HD + (pc - 4) * 8;
        f.a.u = *(const uint4*)pa;                                // unconditional loads on valid addresses (clamped lanes re-read a neighbour)
        f.o.u = *(const uint4*)po;
        f.l = lse[lse0 + n];
    };
    auto q_commit = [&](const QFetch& f, int buf) {
        float x[8], acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[e] = (float)f.a.e[e]; acc += x[e] * (float)f.o.e[e]; }       // q: sum q^2; dO: sum dO * O
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        U8 o;
        if (pc < 4) {
            const float sc = pv ? tau * LOG2E / fmaxf(sqrtf(acc), 1e-12f) : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.e[e] = (bf16)(x[e] * sc);
            *(uint4*)(Qs + buf * 1024 + af_off(px, px, pc)) = o.u;
            if (pc == 0) Nl[buf * 32 + px] = pv ? -f.l * LOG2E : NEG_BIG;
        } else {
            o.u = pv ? f.a.u : make_uint4(0, 0, 0, 0);
            *(uint4*)(Ds + buf * 1024 + af_off(px, px, pc - 4)) = o.u;
            if (pc == 4) Nd[buf * 32 + px] = pv ? -acc : 0.f;
        }
    };
    {
        QFetch f0;
        q_issue(0, f0);
        q_commit(f0, 0);
    }
    __syncthreads();

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
    if (MASK && wmask) {
        const int rk = am_rid(g, wx * ws + min(r31, ws - 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qx = (r & 3) + 8 * (r >> 2) + 4 * hh;
            xm[r] = am_rid(g, wx * ws + min(qx, ws - 1)) != rk ? -100.0f * LOG2E : 0.f;
        }
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
    // eight contributors with ds_bpermute: independent, one LDS latency) and the row takes ONE read-modify-write -- eight dependent
    // read-modify-writes cost eight LDS round trips per step with nothing to hide them behind.
    auto flush = [&](const f32x4_t& v, int dy) {
        float gsum = 0.f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kx = 8 * gq + 4 * half + ws + 2 - lane;
                const float t = __int_as_float(__builtin_amdgcn_ds_bpermute(((half << 5) | (kx & 31)) << 2, __float_as_int(v[gq])));
                gsum += (kx >= 0 && kx < ws + 3) ? t : 0.f;
            }
        if (lane < AF_DS) dtab[(dy + ws - 1) * AF_DS + lane] += gsum;
    };

    float dtau_part = 0.f;
    QFetch nf;
    // One block = (query row qy) x (key row ky0 + a).  The seven blocks of a step run as a software pipeline -- one wave per SIMD hides no
    // latency by itself (first form of this kernel: half of all wave cycles parked in s_waitcnt, profiles/r04_fused_attn_counters.csv):
    //   L(a)  issue the LDS reads of block a: 16 bias words, two K^ and two V row fragments
    //   S(a)  score init (bias - lse [+ mask]) and the four MFMAs of S and dP
    //   X(a)  exp2 / dS / bf16 pairs, the diagonal fold, dV^T and dK^T MFMAs, dS^T quads -> LDS
    //   Qi(a) issue the transposed reads of dS^T and K^T;  Qm(a) the two dQ^T MFMAs
    // in the order  L(a+1) X(a) Qi(a) S(a+1) Qm(a):  block a+1's reads fly under block a's vector work, and its score MFMAs are already in
    // the matrix pipe while the wave waits for its own dS^T to come back from LDS.
    struct Blk { f32x16_t b; bf16x8_t kf[2], vf[2]; };
    for (int qy = 0; qy < ws; ++qy) {
        const int cur = qy & 1;
        // raw q of this row for the dQ slice this lane finishes after the step (normalisation backward): fetched a whole step ahead
        const int xq = wave * 8 + (lane >> 3), dc = lane & 7;      // query position in the row, dims 4 dc .. 4 dc + 3
        const int64_t tqr = af_token(g, ws, b, wy, wx, qy, min(xq, ws - 1));
        U4 rawq;
        rawq.u = *(const uint2*)(qkv + tqr * rs + h * HD + 4 * dc);
        if (qy + 1 < ws) q_issue(qy + 1, nf);
        const bf16* Qc = Qs + cur * 1024;
        const bf16* Dc = Ds + cur * 1024;
        bf16x8_t qa[2], da[2];
        u32x4_t qT[2], dT[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qa[s] = *(const bf16x8_t*)(Qc + rowf[s]);
            da[s] = *(const bf16x8_t*)(Dc + rowf[s]);
            qT[s] = __builtin_bit_cast(u32x4_t, af_tr(Qc, trq[s][0], trq[s][1]));
            dT[s] = __builtin_bit_cast(u32x4_t, af_tr(Dc, trq[s][0], trq[s][1]));
        }
        f32x16_t nl, ndl;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4_t a4 = *(const f32x4_t*)(Nl + cur * 32 + 8 * gq + 4 * hh);
            const f32x4_t b4 = *(const f32x4_t*)(Nd + cur * 32 + 8 * gq + 4 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) { nl[4 * gq + i] = a4[i]; ndl[4 * gq + i] = b4[i]; }
        }
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
        auto stage_L = [&](int a, Blk& L) {
            const int ky = ky0 + a;
            const float* bl = bstep + (nrow - 1 - a) * W2;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int i = 0; i < 4; ++i) L.b[4 * gq + i] = bl[8 * gq + i];
#pragma unroll
            for (int s = 0; s < 2; ++s) { L.kf[s] = *(const bf16x8_t*)(kp_row[s] + a * ws * 32); L.vf[s] = *(const bf16x8_t*)(vp_row[s] + a * ws * 32); }
        };
        auto stage_S = [&](int a, const Blk& L, f32x16_t& sc, f32x16_t& dp) {
            sc = L.b + nl;
            if (MASK && wmask) {
                const bool yd = ydiff_of(a);
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] += yd ? -100.0f * LOG2E : xm[r];
            }
            dp = ndl;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[s], L.kf[s], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[s], L.vf[s], dp, 0, 0, 0);
            }
        };
        Blk L[2];
        f32x16_t sc, dp;
        if (live(0)) { stage_L(0, L[0]); stage_S(0, L[0], sc, dp); }
        f32x4_t prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < AF_RPW; ++a) {
            if (a < nrow) {                                        // wave-uniform
                const int ky = ky0 + a;
                const bool on = live(a), on1 = a + 1 < AF_RPW && live(a + 1);
                if (PIPE && on1) stage_L(a + 1, L[(a + 1) & 1]);
                f32x4_t F = {0.f, 0.f, 0.f, 0.f};
                bf16x8_t k0, t0, k1, t1;
                if (on) {
                    u32x4_t pw[2], dw[2];
                    f32x16_t ds;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2_t p2 = am_exp2((f32x2_t){sc[r], sc[r + 1]});
                        const f32x2_t d2 = p2 * (f32x2_t){dp[r], dp[r + 1]};
                        ds[r] = d2[0]; ds[r + 1] = d2[1];
                        pw[r >> 3][(r >> 1) & 3] = am_pk(p2);
                        dw[r >> 3][(r >> 1) & 3] = am_pk(d2);
                    }
                    // dS^T through LDS: register quad gq (queries 8 gq + 4 half ..+3) of key r31 -> [key][query] image
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *(uint2*)tp_wr[gq] = make_uint2(dw[gq >> 1][(gq & 1) * 2], dw[gq >> 1][(gq & 1) * 2 + 1]);
                    asm volatile("" ::: "memory");
                    af_mfma2_acc_aa<WS == 0>(dv[a], dT[0], __builtin_bit_cast(bf16x8_t, pw[0]), dT[1], __builtin_bit_cast(bf16x8_t, pw[1]));
                    af_mfma2_acc_aa<WS == 0>(dk[a], qT[0], __builtin_bit_cast(bf16x8_t, dw[0]), qT[1], __builtin_bit_cast(bf16x8_t, dw[1]));
                    // diagonal fold of the quads: lane kx' ends up with the sum over i of ds[4 gq + i] of lane kx' - (3 - i), i.e. with the pairs
                    // of offset dx = 8 gq + 4 half + 3 - kx'.  The values move towards the (at least three, ws <= 28) padding-key lanes behind
                    // the row, whose own dS is exactly 0 -- they also keep the halves apart: nothing real crosses from lane 31 into lane 32
                    // (the four quads level by level: a DPP operand written by the instruction in front costs two wait states)
                    {
                        f32x4_t u;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) u[gq] = ds[4 * gq + 1] + af_shr1(ds[4 * gq]);
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) u[gq] = ds[4 * gq + 2] + af_shr1(u[gq]);
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) F[gq] = ds[4 * gq + 3] + af_shr1(u[gq]);
                    }
                    k0 = af_trp(kp_tr[0][0] + a * ws * 32, kp_tr[0][1] + a * ws * 32); t0 = af_trp(tp_tr[0][0], tp_tr[0][1]);
                    k1 = af_trp(kp_tr[1][0] + a * ws * 32, kp_tr[1][1] + a * ws * 32); t1 = af_trp(tp_tr[1][0], tp_tr[1][1]);
                    asm volatile("" ::: "memory");
                }
                if (!PIPE && on1) stage_L(a + 1, L[(a + 1) & 1]);
                if (on1) stage_S(a + 1, L[(a + 1) & 1], sc, dp);
                if (on) af_mfma2_acc<WS == 0>(dq, k0, t0, k1, t1);
                const f32x4_t t = R[a];
                R[a] = prev + F;
                prev = t;
                if (a == nrow - 1) flush(R[a], qy - ky);          // the chain ends at the wave's last key row
            }
        }
        // ---- dQ of this query row: partial tile -> LDS (lane = query r31, registers = dims (r & 3) + 8 (r >> 2) + 4 half)
        af_settle(dq);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *(f32x4_t*)(Pw + r31 * AF_PQ + 8 * gq + 4 * hh) = (f32x4_t){dq[4 * gq], dq[4 * gq + 1], dq[4 * gq + 2], dq[4 * gq + 3]};
        if (qy + 1 < ws) q_commit(nf, cur ^ 1);
        __syncthreads();
        {
            f32x4_t v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ww = 0; ww < AF_WAVES; ++ww) v += *(const f32x4_t*)(Pq + (ww * 32 + xq) * AF_PQ + 4 * dc);
            const bool qv = xq < ws;
            float qh[4], ss = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { qh[e] = (float)rawq.e[e]; ss += qh[e] * qh[e]; }
            ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
            const float qinv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { qh[e] *= qinv; dot += v[e] * qh[e]; }
            if (qv) dtau_part += dot;
            dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 4, 64);
            dot *= tau;                                            // q^ . d(q^)
            if (qv) {
                U4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o.e[e] = (bf16)((tau * v[e] - qh[e] * dot) * qinv);
                *(uint2*)(dqkv + tqr * rs + h * HD + 4 * dc) = o.u;
            }
        }
        __syncthreads();
    }
    // ---- the chains still open after the last query row (dy of chain a: ws - 1 - ky)
#pragma unroll
    for (int a = 0; a < AF_RPW; ++a)
        if (a < nrow - 1) flush(R[a], ws - 1 - (ky0 + a));
    // ---- dK, dV of the wave's key rows: lane = key position r31, registers = dims (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
    for (int a = 0; a < AF_RPW; ++a) {
        af_settle(dk[a]);
        af_settle(dv[a]);
        if (a < nrow) {
            const int ky = ky0 + a;
            const int64_t t = af_token(g, ws, b, wy, wx, ky, min(r31, ws - 1));
            const bf16* kp = qkv + t * rs + C + h * HD + 4 * hh;
            float kh[16], ss = 0.f;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                U4 x;
                x.u = *(const uint2*)(kp + 8 * gq);
#pragma unroll
                for (int e = 0; e < 4; ++e) { kh[4 * gq + e] = (float)x.e[e]; ss += kh[4 * gq + e] * kh[4 * gq + e]; }
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
    dtau_part = wave_sum(dtau_part);
    if (lane == 0) red[wave] = dtau_part;
    __syncthreads();
    for (int i = tid; i < T2; i += blockDim.x) pout[i] = dtab[(i / W2) * AF_DS + i % W2];
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
#define AM_FUSED(MASKV, WSV)                                                                                                \
    do {                                                                                                                 \
        if (hipFuncSetAttribute((const void*)attn_bwd_fused_win_k<MASKV, WSV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { \
            mvuld_set_error("attn_bwd_fused_win_k: hipFuncSetAttribute(%zu) failed", bytes);                            \
            return 1;                                                                                                    \
        }                                                                                                                \
        hipLaunchKernelGGL((attn_bwd_fused_win_k<MASKV, WSV>), dim3((unsigned)groups), dim3(64 * AF_WAVES), bytes, stream, g, (const bf16*)qkv, \
                           table16, logit_scale, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, ws_part, dlogit_scale);            \
    } while (0)
    if (g.ws == 28) { if (shift > 0) AM_FUSED(true, 28); else AM_FUSED(false, 28); }
    else AM_FUSED(true, 0);               // the generic instantiation: masked form only (it also serves unshifted blocks)
#undef AM_FUSED
    return 0;
}
