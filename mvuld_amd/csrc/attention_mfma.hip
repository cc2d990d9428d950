// Fused attention on the CDNA4 matrix cores (bf16 storage, fp32 accumulate): forward, dQ pass, dK/dV pass.
// Same contract as the VALU kernels in attention.hip (MODE 0 SwinV2 window attention with cosine scores +
// continuous position bias + shift mask and the roll/partition index map; MODE 1 pad-masked attention).
//
// Tiling (v_mfma_f32_16x16x32_bf16; one workgroup of 8 waves per (window|sequence, head), staged ONCE):
//   forward / dQ : a wave walks 16-QUERY tiles.  S^T = K_tile . Q^T puts the query on the lane (column) and 4 keys per
//     16x16 tile in its registers, so the softmax statistics are lane-local up to a 4-lane reduce (two v_permlane*_swap,
//     no LDS), and two S^T tiles (32 keys) ARE the B operand of the next product (O^T = V^T . P^T, dQ^T = K^T . dS^T)
//     with no cross-lane movement -- the k-slot permutation (slot (g,j<4) <-> key 4g+j, slot (g,j>=4) <-> key 16+4g+j-4)
//     is applied to the A operand instead, which is read TRANSPOSED straight out of the row-major LDS image with
//     ds_read_b64_tr_b16 (gfx950's 4x16 hardware transpose read).
//   dK/dV : a wave walks 16-KEY tiles; S = Q_tile . K^T and dP = dO_tile . V^T put the key on the lane, and P / dS are
//     the B operands of dV^T = dO^T . P and dK^T = Q~^T . dS (A = transposed reads of the dO / Q~ images).
// LDS holds K (normalised for MODE 0) and V row-major (forward, dQ pass) or Q~ and dO (dK/dV pass): 2 x N x (hd+8) bf16
// (128 KB at N=784, hd=32), plus per-row info, and for MODE 0 the head's (2w-1)^2 fp32 bias table (+ its gradient):
// up to ~155 KB of the 160 KB.  The shift mask comes from 3x3 region ids: no [N,N] tensor ever exists.  P is recomputed
// in backward from the saved log-sum-exp; delta = rowsum(dO * O) comes from a small pre-kernel.  d(bias table) is
// accumulated with LDS float atomics per workgroup and flushed with one global atomic per touched entry.
#include "attention_common.h"
// LDS image of a [rows][HD] bf16 operand tile, read two ways: 16-byte fragments of 16 consecutive rows (ds_read_b128) and transposed
// 8-byte pieces of 4 x 4 consecutive rows (ds_read_b64_tr_b16: 16 lanes = 4 rows x 32 bytes).
//   PAD (HD = 64, and the bias-gradient pass): rows padded to HD + 8 elements.  At HD = 64 (144 B = 36 banks) both patterns are
//        conflict free; at HD = 32 (80 B = 20 banks) four consecutive rows of a transposed read span 68 banks: 2-way conflict.
//   SWZ (HD = 32): 64-byte rows, the four 16-byte chunks of a row XOR-ed with ((row >> 1 & 1) << 1 | (row >> 2 & 1)): eight
//        consecutive rows put one chunk in eight different bank groups, four consecutive rows put a chunk PAIR in four different
//        ones; a row and row + 16 share the swizzle, so the "+ 16 rows" immediate offsets survive.  20 % less LDS, too.
template <int HD, bool SWZ> struct AmTile { static constexpr int LD = SWZ ? HD : HD + 8; };
template <int HD, bool SWZ>
__device__ __forceinline__ int am_off(int row, int chunk) {
    if (SWZ) return row * HD + ((chunk ^ ((((row >> 1) & 1) << 1) | ((row >> 2) & 1))) << 3);
    return row * (HD + 8) + (chunk << 3);
}

template <int HD, bool SWZ = (HD == 32)>
__device__ __forceinline__ void stage_tile(const AttnGeom& g, const bf16* __restrict__ base, int64_t rowstride, int coloff, int b,
                                           int w, int n0, int rows, bf16* rm, bool normalize, float mul) {
    constexpr int CPR = HD / 8;
    const int total = rows * CPR;
    for (int c0 = threadIdx.x; c0 < total; c0 += 4 * blockDim.x) {
        U8 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * blockDim.x;
            const int n = n0 + c / CPR;
            x[u].u = make_uint4(0, 0, 0, 0);
            if (c < total && n < g.N) x[u].u = *(const uint4*)(base + am_token(g, b, w, n) * rowstride + coloff + (c % CPR) * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * blockDim.x;
            if (normalize || mul != 1.0f) {
                float f[8];
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { f[e] = (float)x[u].e[e]; ss += f[e] * f[e]; }
                float sc = mul;
                if (normalize) {
#pragma unroll
                    for (int o = 1; o < CPR; o <<= 1) ss += __shfl_xor(ss, o, 64);
                    sc = mul / fmaxf(sqrtf(ss), 1e-12f);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) x[u].e[e] = (bf16)(f[e] * sc);
            }
            if (c < total) *(uint4*)(rm + am_off<HD, SWZ>(c / CPR, c % CPR)) = x[u].u;
        }
    }
}

// A operand = transposed 16(dims d0..d0+15) x 32(row slots) fragment of a row-major image rm[rows][ld]:
// k-slot (g, j<4) <-> row r0+4g+j, (g, j>=4) <-> row r0+16+4g+j-4.  ds_read_b64_tr_b16: within each group of 16 lanes,
// lane 4q+p supplies the address of block row q, columns 4p..4p+3; lane i receives column i of the 4 rows.
template <int HD, bool SWZ = (HD == 32)>
__device__ __forceinline__ bf16x8_t read_tr(const bf16* rm, int d0, int r0, int lane) {
    typedef __attribute__((address_space(3))) bf16x4_t* lds_p;
    const int fg = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#if AM_X == 3
    return *(const bf16x8_t*)(rm + am_off<HD, SWZ>(r0 + (lane & 15), (d0 >> 3) + (fg & 1)));
#endif
    const bf16* a0 = rm + am_off<HD, SWZ>(r0 + 4 * fg + q, (d0 >> 3) + (pp >> 1)) + 4 * (pp & 1);
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)a0);
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0 + 16 * AmTile<HD, SWZ>::LD));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ------------------------------------------------------------------------------------------------ forward
// Scores live in log2 units (log2(e) is folded into q~, the bias table and the mask constants) so the softmax uses bare
// v_exp_f32.  Per-key info word (Kinfo): MODE 0: 4*(iy*(2w-1)+ix) | region << 16 (| AM_PAD), MODE 1: valid (| AM_PAD).

// Score fix-ups.  The continuous position bias does not cost a VALU add per score: it is the INITIAL ACCUMULATOR of the S = K.Q^T
// MFMA (am_bias4).  A lane's four accumulator rows are four consecutive window positions n0 .. n0+3 with n0 % 4 == 0; when the
// window side is a multiple of 4 (G4) they lie in one window row, so their table entries are adjacent words: one address and two
// ds_read2_b32 instead of four address computations and four ds_read_b32.  tabx + offset: the forward / dQ passes (keys on the rows)
// keep the table reversed so that the words ascend with the key; the dK/dV pass (queries on the rows) uses it as it is.
template <int MODE, bool MASK, bool TAIL, bool G4>
__device__ __forceinline__ f32x4_t am_bias4(const int (&ki)[4], const char* tabx) {
    f32x4_t b = {0.f, 0.f, 0.f, 0.f};
#if AM_X == 2
    return b;
#endif
    if (MODE == 0) {
        if (G4) {
            const float* a = (const float*)(tabx + ((MASK || TAIL) ? (ki[0] & 0xffff) : ki[0]));
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = a[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = *(const float*)(tabx + ((MASK || TAIL) ? (ki[r] & 0xffff) : ki[r]));
        }
    }
    return b;
}
// what is left per score after the MFMA (s already holds q.k + bias): shift mask / pad mask / tail padding
template <int MODE, bool MASK, bool TAIL>
__device__ __forceinline__ float am_mask(float s, int ki, int regq, int vq) {
    float v = s;
    if (MODE == 0) {
        if (MASK) v = (((ki >> 16) & 0xff) != regq) ? v - 100.0f * LOG2E : v;
    } else {
        v = (vq & ki & 1) ? s : s - 10000.0f * LOG2E;
    }
    if (TAIL) v = (ki & AM_PAD) ? NEG_BIG : v;
    return v;
}

// one block of NT 16-key tiles (NT = 4: 64 keys, NT = 2: 32 keys) of the online-softmax forward.  The softmax denominator rides
// the matrix cores too: lacc = ones . P^T accumulates sum_k p~ (the bf16-rounded p that also multiplies V) next to O^T.
template <int HD, int MODE, bool MASK, bool TAIL, int NT, bool G4, bool DROP = false>
__device__ __forceinline__ void am_fwd_block(const bf16* __restrict__ Ks, const bf16* __restrict__ Vs, const int* __restrict__ Kinfo, int kb,
                                             const bf16x8_t (&qf)[HD / 32], const char* tabq, int regq, int vq, int lane, float& m,
                                             f32x4_t& lacc, f32x4_t (&oacc)[HD / 16], const bf16x8_t& ones, unsigned ebase = 0,
                                             unsigned dseed = 0, unsigned dthr = 0) {
    const int fc = lane & 15, fg = lane >> 4;
    f32x4_t s[NT];
    int ki[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int4 inf = *(const int4*)(Kinfo + kb + 16 * t + 4 * fg);
        ki[t][0] = inf.x; ki[t][1] = inf.y; ki[t][2] = inf.z; ki[t][3] = inf.w;
        s[t] = am_bias4<MODE, MASK, TAIL, G4>(ki[t], tabq);
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks)
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)(Ks + am_off<HD, HD == 32>(kb + 16 * t + fc, ks * 4 + fg)), qf[ks], s[t], 0, 0, 0);
    }
    float bm = NEG_BIG;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE != 0 || MASK || TAIL) s[t][r] = am_mask<MODE, MASK, TAIL>(s[t][r], ki[t][r], regq, vq);
            bm = fmaxf(bm, s[t][r]);
        }
    bm = max4g(bm);
    const float mn = fmaxf(m, bm);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    u32x4_t pw[NT / 2], puw[DROP ? NT / 2 : 1];      // pw multiplies V (dropped-out when DROP); the denominator sums the undropped puw
    const f32x2_t mn2 = {mn, mn};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            const f32x2_t p = am_exp2((f32x2_t){s[t][2 * hp], s[t][2 * hp + 1]} - mn2);
            if (DROP) {           // ebase = row * ceil(N/2); the kept probabilities are scaled by 1/(1-p) once, at the end (O * dinv / l)
                const unsigned u = am_pk(p);
                puw[t >> 1][(t & 1) * 2 + hp] = u;
                pw[t >> 1][(t & 1) * 2 + hp] = u & am_keep2(ebase + (unsigned)((kb >> 1) + 8 * t + 2 * fg + hp), dseed, dthr);
            } else {
                pw[t >> 1][(t & 1) * 2 + hp] = am_pk(p);
            }
        }
    bf16x8_t pb[NT / 2], pu[DROP ? NT / 2 : 1];
#pragma unroll
    for (int pr = 0; pr < NT / 2; ++pr) {
        pb[pr] = __builtin_bit_cast(bf16x8_t, pw[pr]);
        if (DROP) pu[pr] = __builtin_bit_cast(bf16x8_t, puw[pr]);
    }
    m = mn;
    lacc *= alpha;
#pragma unroll
    for (int pr = 0; pr < NT / 2; ++pr) lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, DROP ? pu[pr] : pb[pr], lacc, 0, 0, 0);
#pragma unroll
    for (int d = 0; d < HD / 16; ++d) {
        oacc[d] *= alpha;
#pragma unroll
        for (int pr = 0; pr < NT / 2; ++pr)
            oacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(read_tr<HD>(Vs, d * 16, kb + 32 * pr, lane), pb[pr], oacc[d], 0, 0, 0);
    }
}

// grid.x = B*nW*H*qsplit: workgroup (bwh, part) stages K,V of (window, head) once and walks the q tiles  part, part+qsplit, ...
template <int HD, int MODE, bool MASK>
__global__ __launch_bounds__(1024) void attn_fwd_mfma_k(AttnGeom g, const bf16* __restrict__ qkv, const float* __restrict__ table16,
                                                       const float* __restrict__ logit_scale, const int* __restrict__ valid,
                                                       bf16* __restrict__ out, float* __restrict__ lse, int Npad, int qsplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KLD = AmTile<HD, HD == 32>::LD;
    bf16* Ks = (bf16*)smem;                       // [Npad][KLD]  (normalised for MODE 0)
    bf16* Vs = Ks + (size_t)Npad * KLD;           // [Npad][KLD]
    int* Kinfo = (int*)(Vs + (size_t)Npad * KLD); // [Npad]
    float* tab = (float*)(Kinfo + Npad);          // MODE 0: [(2ws-1)^2], in log2 units
    int bwh, part;
    am_part(g, qsplit, bwh, part, qsplit);
    const int h = bwh % g.H, bw = bwh / g.H, b = bw / g.nW, w = bw % g.nW;
    const int C = g.H * HD;
    const int64_t rs = 3 * (int64_t)C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fc = lane & 15, fg = lane >> 4;
    const int64_t lse0 = ((int64_t)bw * g.H + h) * g.N;          // lse rows keep the launch's N as their stride
    const unsigned NL = (unsigned)g.N;
    const int Np = am_localize(g, b, Npad);
    if (Np == 0) return;
    if (MODE == 0 && am_dropped(g, b)) {                     // workgroup-uniform: zeros where this workgroup would have written
        const int ntile0 = (g.N + 15) / 16;
        for (int qt = part + qsplit * wave; qt < ntile0; qt += qsplit * (blockDim.x >> 6)) {
            const int nq = qt * 16 + fc;
            if (nq < g.N) {
                const int64_t t = am_token(g, b, w, nq);
#pragma unroll
                for (int d = 0; d < HD / 16; ++d) *(uint2*)(out + t * C + h * HD + d * 16 + 4 * fg) = make_uint2(0, 0);
                if (fg == 0) lse[lse0 + nq] = 0.f;
            }
        }
        return;
    }
    const bool drop = MODE == 1 && g.drop_inv != 1.0f;
    const unsigned dseed = drop ? am_seed(g) : 0u;

    stage_tile<HD>(g, qkv, rs, C + h * HD, b, w, 0, Np, Ks, MODE == 0, 1.0f);
    stage_tile<HD>(g, qkv, rs, 2 * C + h * HD, b, w, 0, Np, Vs, false, 1.0f);
    for (int i = threadIdx.x; i < Np; i += blockDim.x) {
        const int inf = am_info(g, valid, b, w, i);
        Kinfo[i] = MODE == 0 ? (((inf & 0xffff) << 2) | (inf & ~0xffff)) : inf;
    }
    int C0 = 0;
    float tau = 1.f;
    const int T2 = MODE == 0 ? (2 * g.ws - 1) * (2 * g.ws - 1) : 0;
    if (MODE == 0) {
        // stored REVERSED: entry (bq - ok) of the table sits at word T2-1-bq + ok, so the entries of consecutive keys ascend
        for (int i = threadIdx.x; i < T2; i += blockDim.x) tab[i] = table16[(int64_t)(T2 - 1 - i) * g.H + h] * LOG2E;
        C0 = T2 - 1 - ((g.ws - 1) * (2 * g.ws - 1) + (g.ws - 1));
        tau = __expf(fminf(logit_scale[h], LN100));
    }
    __syncthreads();

    const int ntile = (g.N + 15) / 16;
    const int nfull64 = (g.N / 64) * 64;          // keys covered by pad-free 64-key blocks
    const bool g4 = MODE == 0 && (g.ws & 3) == 0; // four consecutive window positions share a row: adjacent bias-table words
    const bool wmask = MASK && am_window_masked(g, w);
    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    for (int qt = part + qsplit * wave; qt < ntile; qt += qsplit * (blockDim.x >> 6)) {
        const int nq = qt * 16 + fc;
        const bool qok = nq < g.N;
        const int nqc = qok ? nq : g.N - 1;
        const int64_t tq = am_token(g, b, w, nqc);
        const int qinf = am_info(g, valid, b, w, nqc);
        const char* tabq = (const char*)tab + 4 * (C0 - (qinf & 0xffff));
        const int regq = (qinf >> 16) & 0xff, vq = qinf;
        bf16x8_t qf[HD / 32];
        {
            float f[HD / 32][8];
            float ss = 0.f;
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
                U8 x;
                x.u = *(const uint4*)(qkv + tq * rs + h * HD + ks * 32 + fg * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) { f[ks][e] = (float)x.e[e]; ss += f[ks][e] * f[ks][e]; }
            }
            float sc = g.scale * LOG2E;
            if (MODE == 0) sc = tau * LOG2E / fmaxf(sqrtf(sum4g(ss)), 1e-12f);
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[ks][e] = (bf16)(f[ks][e] * sc);
        }
        float m = -INFINITY;
        f32x4_t lacc = {0.f, 0.f, 0.f, 0.f};
        f32x4_t oacc[HD / 16];
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) oacc[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        int kb = 0;
        if (MODE == 1 && drop) {
            const unsigned eb = ((unsigned)lse0 + (unsigned)nqc) * ((NL + 1) >> 1);           // (b, h, q) row -> first key-pair counter
            for (; kb < nfull64; kb += 64)
                am_fwd_block<HD, MODE, MASK, false, 4, false, MODE == 1>(Ks, Vs, Kinfo, kb, qf, tabq, regq, vq, lane, m, lacc, oacc, ones, eb, dseed, g.drop_thr);
            for (; kb < Np; kb += 32)
                am_fwd_block<HD, MODE, MASK, true, 2, false, MODE == 1>(Ks, Vs, Kinfo, kb, qf, tabq, regq, vq, lane, m, lacc, oacc, ones, eb, dseed, g.drop_thr);
        } else {
#define AM_FWD_SWEEP(MK)                                                                                                                       \
            if (g4) {                                                                                                                          \
                for (; kb < nfull64; kb += 64) am_fwd_block<HD, MODE, MK, false, 4, true>(Ks, Vs, Kinfo, kb, qf, tabq, regq, vq, lane, m, lacc, oacc, ones); \
                for (; kb < Np; kb += 32) am_fwd_block<HD, MODE, MK, true, 2, true>(Ks, Vs, Kinfo, kb, qf, tabq, regq, vq, lane, m, lacc, oacc, ones);      \
            } else {                                                                                                                           \
                for (; kb < nfull64; kb += 64) am_fwd_block<HD, MODE, MK, false, 4, false>(Ks, Vs, Kinfo, kb, qf, tabq, regq, vq, lane, m, lacc, oacc, ones); \
                for (; kb < Np; kb += 32) am_fwd_block<HD, MODE, MK, true, 2, false>(Ks, Vs, Kinfo, kb, qf, tabq, regq, vq, lane, m, lacc, oacc, ones);     \
            }
            if (MASK && wmask) { AM_FWD_SWEEP(true) } else { AM_FWD_SWEEP(false) }
#undef AM_FWD_SWEEP
        }
        const float l = lacc[0];             // every row of ones . P^T is the same sum over ALL keys: no cross-lane reduce
        if (qok) {
            const float inv = drop ? g.drop_inv / l : 1.0f / l;
#pragma unroll
            for (int d = 0; d < HD / 16; ++d) {
                U4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o.e[r] = (bf16)(oacc[d][r] * inv);
                *(uint2*)(out + tq * C + h * HD + d * 16 + 4 * fg) = o.u;
            }
            if (fg == 0) lse[lse0 + nq] = (m + __log2f(l)) * LN2;
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ d logit_scale)
// one block of NT key tiles: dS^T = P^T o (dP^T - delta), dQ^T += K^T . dS^T   (scores in log2 units, gradients in natural units)
// QT query tiles per wave share every LDS read of the block -- the K and V fragments of the two score products and the transposed
// K operand of the dQ product (the passes are LDS-bound: section 9b of DESIGN.md); only the bias words are per (query, key).
template <int HD, int MODE, bool MASK, bool TAIL, int NT, bool G4, bool DROP = false, int QT = 1>
__device__ __forceinline__ void am_dq_block(const bf16* __restrict__ Ks, const bf16* __restrict__ Vs, const int* __restrict__ Kinfo, int kb,
                                            const bf16x8_t (&qf)[QT][HD / 32], const bf16x8_t (&dof)[QT][HD / 32], const char* const (&tabq)[QT],
                                            const int (&regq)[QT], const int (&vq)[QT], const float (&L2q)[QT], const f32x4_t (&negD)[QT],
                                            int lane, f32x4_t (&dq)[QT][HD / 16], const unsigned (&ebase)[QT], unsigned dseed = 0,
                                            unsigned dthr = 0, float dinv = 1.f) {
    const int fc = lane & 15, fg = lane >> 4;
    f32x4_t s[QT][NT], dp[QT][NT];
    int ki[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int4 inf = *(const int4*)(Kinfo + kb + 16 * t + 4 * fg);
        ki[t][0] = inf.x; ki[t][1] = inf.y; ki[t][2] = inf.z; ki[t][3] = inf.w;
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            s[q][t] = am_bias4<MODE, MASK, TAIL, G4>(ki[t], tabq[q]);   // bias and -delta are accumulator inits, not VALU ops
            dp[q][t] = DROP ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : negD[q];    // with dropout the mask sits between dO.V^T and -delta
        }
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
            const int o = am_off<HD, HD == 32>(kb + 16 * t + fc, ks * 4 + fg);
            const bf16x8_t kfr = *(const bf16x8_t*)(Ks + o), vfr = *(const bf16x8_t*)(Vs + o);
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                s[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[q][ks], s[q][t], 0, 0, 0);
                dp[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, dof[q][ks], dp[q][t], 0, 0, 0);
            }
        }
    }
    u32x4_t dsw[QT][NT / 2];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const f32x2_t L2 = {L2q[q], L2q[q]};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
                f32x2_t sv = {s[q][t][2 * hp], s[q][t][2 * hp + 1]}, dpv = {dp[q][t][2 * hp], dp[q][t][2 * hp + 1]};
                if (MODE != 0 || MASK || TAIL) {                      // padding keys -> NEG_BIG -> p = 0
                    sv[0] = am_mask<MODE, MASK, TAIL>(sv[0], ki[t][2 * hp], regq[q], vq[q]);
                    sv[1] = am_mask<MODE, MASK, TAIL>(sv[1], ki[t][2 * hp + 1], regq[q], vq[q]);
                }
                if (DROP) {           // dS = P o (mask o dP / (1-p) - delta): the keep mask of the pair, widened to the two fp32 lanes
                    const unsigned mk = am_keep2(ebase[q] + (unsigned)((kb >> 1) + 8 * t + 2 * fg + hp), dseed, dthr);
                    dpv[0] = fmaf(__uint_as_float(__float_as_uint(dpv[0]) & (unsigned)((int)(mk << 16) >> 16)), dinv, negD[q][0]);
                    dpv[1] = fmaf(__uint_as_float(__float_as_uint(dpv[1]) & (unsigned)((int)mk >> 16)), dinv, negD[q][0]);
                }
                dsw[q][t >> 1][(t & 1) * 2 + hp] = am_pk(am_exp2(sv - L2) * dpv);
            }
    }
#pragma unroll
    for (int d = 0; d < HD / 16; ++d)
#pragma unroll
        for (int pr = 0; pr < NT / 2; ++pr) {
            const bf16x8_t kt = read_tr<HD>(Ks, d * 16, kb + 32 * pr, lane);
#pragma unroll
            for (int q = 0; q < QT; ++q)
                dq[q][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, __builtin_bit_cast(bf16x8_t, dsw[q][pr]), dq[q][d], 0, 0, 0);
        }
}

template <int HD, int MODE, bool MASK>
__global__ __launch_bounds__((HD == 32 || MODE == 1) ? 1024 : 512) void attn_bwd_dq_mfma_k(AttnGeom g, const bf16* __restrict__ qkv, const float* __restrict__ table16,
                                                          const float* __restrict__ logit_scale, const int* __restrict__ valid,
                                                          const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                          float* __restrict__ dlogit_scale, bf16* __restrict__ qt_out, int Npad, int qsplit,
                                                          const bf16* __restrict__ outp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KLD = AmTile<HD, HD == 32>::LD;
    bf16* Ks = (bf16*)smem;                       // [Npad][KLD]
    bf16* Vs = Ks + (size_t)Npad * KLD;           // [Npad][KLD]
    int* Kinfo = (int*)(Vs + (size_t)Npad * KLD); // [Npad]
    float* red = (float*)(Kinfo + Npad);          // [16]
    float* tab = red + 16;                        // MODE 0: [T2], log2 units
    const int T2 = MODE == 0 ? (2 * g.ws - 1) * (2 * g.ws - 1) : 0;
    int bwh, part;
    am_part(g, qsplit, bwh, part, qsplit);
    const int h = bwh % g.H, bw = bwh / g.H, b = bw / g.nW, w = bw % g.nW;
    const int C = g.H * HD;
    const int64_t rs = 3 * (int64_t)C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fc = lane & 15, fg = lane >> 4;
    const int64_t lse0 = ((int64_t)bw * g.H + h) * g.N;
    const unsigned NL = (unsigned)g.N;
    const int Np = am_localize(g, b, Npad);
    if (Np == 0) return;
    if (MODE == 0 && am_dropped(g, b)) {                     // d(out) of this sample is zero: so is dQ (workgroup-uniform)
        constexpr int QT0 = HD == 32 ? 2 : 1;
        const int ntile0 = (g.N + 15) / 16, nitem0 = (ntile0 + QT0 - 1) / QT0;
        for (int item = part + qsplit * wave; item < nitem0; item += qsplit * (blockDim.x >> 6))
#pragma unroll
            for (int q = 0; q < QT0; ++q) {
                const int nq = (item * QT0 + q) * 16 + fc;
                if (item * QT0 + q < ntile0 && nq < g.N) {
                    const int64_t t = am_token(g, b, w, nq);
#pragma unroll
                    for (int d = 0; d < HD / 16; ++d) *(uint2*)(dqkv + t * rs + h * HD + d * 16 + 4 * fg) = make_uint2(0, 0);
                }
            }
        return;
    }
    const bool drop = MODE == 1 && g.drop_inv != 1.0f;
    const unsigned dseed = drop ? am_seed(g) : 0u;

    stage_tile<HD>(g, qkv, rs, C + h * HD, b, w, 0, Np, Ks, MODE == 0, 1.0f);
    stage_tile<HD>(g, qkv, rs, 2 * C + h * HD, b, w, 0, Np, Vs, false, 1.0f);
    for (int i = threadIdx.x; i < Np; i += blockDim.x) {
        const int inf = am_info(g, valid, b, w, i);
        Kinfo[i] = MODE == 0 ? (((inf & 0xffff) << 2) | (inf & ~0xffff)) : inf;
    }
    int C0 = 0;
    float tau = 1.f;
    if (MODE == 0) {
        for (int i = threadIdx.x; i < T2; i += blockDim.x) tab[i] = table16[(int64_t)(T2 - 1 - i) * g.H + h] * LOG2E;     // reversed, as in the forward
        C0 = T2 - 1 - ((g.ws - 1) * (2 * g.ws - 1) + (g.ws - 1));
        tau = __expf(fminf(logit_scale[h], LN100));
    }
    __syncthreads();

    float dtau_part = 0.f;
    const int ntile = (g.N + 15) / 16;
    const int nfull64 = (g.N / 64) * 64;
    const bool g4 = MODE == 0 && (g.ws & 3) == 0;
    const bool wmask = MASK && am_window_masked(g, w);
    const int ysp = am_ysplit(g, w, tau);
    // MODE 0 (hd = 32 windows): two query tiles per wave and 32-key blocks; the text encoder keeps one tile and 64-key blocks
    // (its K/V image at hd = 64 leaves no registers for a second tile's accumulators)
    constexpr int QT = (MODE == 0 && HD == 32) ? 2 : 1;
    const int nitem = (ntile + QT - 1) / QT;
    const int nfull32 = (g.N / 32) * 32;
    for (int item = part + qsplit * wave; item < nitem; item += qsplit * (blockDim.x >> 6)) {
        bool qok[QT], tok[QT];
        int nqc[QT], regq[QT], vq[QT];
        int64_t tq[QT];
        const char* tabq[QT];
        float L2q[QT];
        f32x4_t negD[QT];
        unsigned eb[QT];
        bf16x8_t qf[QT][HD / 32], dof[QT][HD / 32];
        f32x4_t dq[QT][HD / 16];
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const int qt = item * QT + q;
            tok[q] = qt < ntile;                                   // wave-uniform: an odd tile count leaves the last slot empty
            const int nq = (tok[q] ? qt : ntile - 1) * 16 + fc;
            qok[q] = tok[q] && nq < g.N;
            nqc[q] = nq < g.N ? nq : g.N - 1;
            tq[q] = am_token(g, b, w, nqc[q]);
            const int qinf = am_info(g, valid, b, w, nqc[q]);
            tabq[q] = (const char*)tab + 4 * (C0 - (qinf & 0xffff));
            regq[q] = (qinf >> 16) & 0xff;
            vq[q] = qinf;
            float f[HD / 32][8];
            float ss = 0.f, dsum = 0.f;
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
                U8 x, y, o;
                x.u = *(const uint4*)(qkv + tq[q] * rs + h * HD + ks * 32 + fg * 8);
                y.u = qok[q] ? *(const uint4*)(dout + tq[q] * C + h * HD + ks * 32 + fg * 8) : make_uint4(0, 0, 0, 0);
                o.u = *(const uint4*)(outp + tq[q] * C + h * HD + ks * 32 + fg * 8);
                dof[q][ks] = y.v;
#pragma unroll
                for (int e = 0; e < 8; ++e) { f[ks][e] = (float)x.e[e]; ss += f[ks][e] * f[ks][e]; dsum += (float)y.e[e] * (float)o.e[e]; }
            }
            // delta = rowsum(dO o O) of this query: computed here (the pass needs dO anyway) and published for the dK/dV and
            // bias-table passes, which run after this kernel -- no separate launch on the backward's critical chain
            const float Dq = sum4g(dsum);                           // padding queries: dO = 0 => delta = 0 => dS = 0
            if (qok[q] && fg == 0) delta[tq[q] * g.H + h] = Dq;
            float sc = g.scale * LOG2E;
            if (MODE == 0) sc = tau * LOG2E / fmaxf(sqrtf(sum4g(ss)), 1e-12f);
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[q][ks][e] = (bf16)(f[ks][e] * sc);
            if (MODE == 0 && qt_out && qok[q]) {          // q~ * log2(e), re-used by the bias-table gradient pass
#pragma unroll
                for (int ks = 0; ks < HD / 32; ++ks) { U8 o; o.v = qf[q][ks]; *(uint4*)(qt_out + tq[q] * C + h * HD + ks * 32 + fg * 8) = o.u; }
            }
            L2q[q] = lse[lse0 + nqc[q]] * LOG2E;
            negD[q] = (f32x4_t){-Dq, -Dq, -Dq, -Dq};
            eb[q] = ((unsigned)lse0 + (unsigned)nqc[q]) * ((NL + 1) >> 1);
#pragma unroll
            for (int d = 0; d < HD / 16; ++d) dq[q][d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        int kb = 0;
        const int q0 = item * QT * 16, q1 = min(q0 + QT * 16, g.N);              // wave-uniform
        if (QT == 2) {
#define AM_DQ2_SWEEP(MK)                                                                                                                       \
            if (g4) {                                                                                                                          \
                for (; kb < nfull32; kb += 32) { if (MK && am_yskip(ysp, q0, q1, kb, kb + 32)) continue;                                       \
                    am_dq_block<HD, MODE, MK, false, 2, true, false, QT>(Ks, Vs, Kinfo, kb, qf, dof, tabq, regq, vq, L2q, negD, lane, dq, eb); } \
                for (; kb < Np; kb += 32) { if (MK && am_yskip(ysp, q0, q1, kb, g.N)) continue;                                                \
                    am_dq_block<HD, MODE, MK, true, 2, true, false, QT>(Ks, Vs, Kinfo, kb, qf, dof, tabq, regq, vq, L2q, negD, lane, dq, eb); } \
            } else {                                                                                                                           \
                for (; kb < nfull32; kb += 32) { if (MK && am_yskip(ysp, q0, q1, kb, kb + 32)) continue;                                       \
                    am_dq_block<HD, MODE, MK, false, 2, false, false, QT>(Ks, Vs, Kinfo, kb, qf, dof, tabq, regq, vq, L2q, negD, lane, dq, eb); } \
                for (; kb < Np; kb += 32) { if (MK && am_yskip(ysp, q0, q1, kb, g.N)) continue;                                                \
                    am_dq_block<HD, MODE, MK, true, 2, false, false, QT>(Ks, Vs, Kinfo, kb, qf, dof, tabq, regq, vq, L2q, negD, lane, dq, eb); } \
            }
            if (MASK && wmask) { AM_DQ2_SWEEP(true) } else { AM_DQ2_SWEEP(false) }
#undef AM_DQ2_SWEEP
        } else if (MODE == 1 && drop) {
            for (; kb < nfull64; kb += 64)
                am_dq_block<HD, MODE, MASK, false, 4, false, MODE == 1, QT>(Ks, Vs, Kinfo, kb, qf, dof, tabq, regq, vq, L2q, negD, lane, dq, eb, dseed, g.drop_thr, g.drop_inv);
            for (; kb < Np; kb += 32)
                am_dq_block<HD, MODE, MASK, true, 2, false, MODE == 1, QT>(Ks, Vs, Kinfo, kb, qf, dof, tabq, regq, vq, L2q, negD, lane, dq, eb, dseed, g.drop_thr, g.drop_inv);
        } else {
#define AM_DQ_SWEEP(MK)                                                                                                                        \
            if (g4) {                                                                                                                          \
                for (; kb < nfull64; kb += 64) am_dq_block<HD, MODE, MK, false, 4, true, false, QT>(Ks, Vs, Kinfo, kb, qf, dof, tabq, regq, vq, L2q, negD, lane, dq, eb); \
                for (; kb < Np; kb += 32) am_dq_block<HD, MODE, MK, true, 2, true, false, QT>(Ks, Vs, Kinfo, kb, qf, dof, tabq, regq, vq, L2q, negD, lane, dq, eb);      \
            } else {                                                                                                                           \
                for (; kb < nfull64; kb += 64) am_dq_block<HD, MODE, MK, false, 4, false, false, QT>(Ks, Vs, Kinfo, kb, qf, dof, tabq, regq, vq, L2q, negD, lane, dq, eb); \
                for (; kb < Np; kb += 32) am_dq_block<HD, MODE, MK, true, 2, false, false, QT>(Ks, Vs, Kinfo, kb, qf, dof, tabq, regq, vq, L2q, negD, lane, dq, eb);     \
            }
            if (MASK && wmask) { AM_DQ_SWEEP(true) } else { AM_DQ_SWEEP(false) }
#undef AM_DQ_SWEEP
        }
        // dq[q][d][r] = d(q~)[query fc][dim d*16 + 4*fg + r]   (q~ = tau * q^ in natural units)
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            if (MODE == 0) {
                float qh[HD / 16][4];
                float ss = 0.f;
#pragma unroll
                for (int d = 0; d < HD / 16; ++d) {
                    U4 x;
                    x.u = *(const uint2*)(qkv + tq[q] * rs + h * HD + d * 16 + 4 * fg);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { qh[d][r] = (float)x.e[r]; ss += qh[d][r] * qh[d][r]; }
                }
                const float qinv = 1.0f / fmaxf(sqrtf(sum4g(ss)), 1e-12f);
                float dot = 0.f;
#pragma unroll
                for (int d = 0; d < HD / 16; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { qh[d][r] *= qinv; dot += dq[q][d][r] * qh[d][r]; }
                if (qok[q]) dtau_part += dot;
                dot = sum4g(dot) * tau;                   // q^ . d(q^)
                if (qok[q]) {
#pragma unroll
                    for (int d = 0; d < HD / 16; ++d) {
                        U4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o.e[r] = (bf16)((tau * dq[q][d][r] - qh[d][r] * dot) * qinv);
                        *(uint2*)(dqkv + tq[q] * rs + h * HD + d * 16 + 4 * fg) = o.u;
                    }
                }
            } else if (qok[q]) {
#pragma unroll
                for (int d = 0; d < HD / 16; ++d) {
                    U4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o.e[r] = (bf16)(dq[q][d][r] * g.scale);
                    *(uint2*)(dqkv + tq[q] * rs + h * HD + d * 16 + 4 * fg) = o.u;
                }
            }
        }
    }
    if (MODE == 0) {
        dtau_part = wave_sum(dtau_part);
        if (lane == 0) red[wave] = dtau_part;
        __syncthreads();
        if (threadIdx.x == 0 && logit_scale[h] < LN100) {
            float t = 0.f;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
            atomicAdd(dlogit_scale + h, t * tau);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: d(bias table)  (MODE 0)
// dB[dy,dx] = sum over windows and over (q,k) with (yq-yk, xq-xk) = (dy,dx) of dS[q,k].  Per-element LDS float atomics run at
// ~1 lane/clk/CU and cost 2.5x the whole rest of the backward, so this pass walks the pairs in an order that makes the table
// entry of every (lane, register) LOOP-INVARIANT: tiles are aligned to image rows of the window (a q tile = up to 16 tokens of
// one row, a key block = one row padded to 32 slots), a wave owns one (dy, q-part) and sweeps yq with yk = yq - dy, so dS is
// summed in 8 registers per lane and only the final sums touch LDS / global atomics (~14x fewer atomics at w = 28).
// DPP row shifts inside a 16-lane row, zero fill: shl n -> lane i reads lane i+n, shr n -> lane i reads lane i-n.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// Work item of a wave = (q part of <=16 x positions, group of G vertical offsets dy in [0, ws)).  Each dy is swept together
// with its complement dy - ws: image row yq pairs with key row yq - dy (+ ws while yq < dy), so every (item, yq) is exactly
// G row tiles (16 queries x 32 key slots) and the q-side operands of a row are fetched once for all G of them.  The table
// entry of a register is loop invariant between the two switches of a dy, so dS accumulates in registers; a flush folds the
// 4-register diagonals of a lane row with DPP shifts before the LDS atomics (3x fewer of them).
// KT = 16-slot key sub-tiles per image row: 2 covers windows up to 32 wide, 1 (ws <= 16) drops the all-padding second half.
template <int HD, int G, int KT>
__global__ __launch_bounds__(G >= 4 ? 512 : 1024) void attn_bwd_dbias_mfma_k(AttnGeom g, const bf16* __restrict__ qkv, const bf16* __restrict__ qt,
                                                             const float* __restrict__ table16, const bf16* __restrict__ dout,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             float* __restrict__ dtable16, float* __restrict__ part_out, int Npad, int split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KLD = HD + 8;
    bf16* Ks = (bf16*)smem;                       // [Npad][KLD] normalised keys
    bf16* Vs = Ks + (size_t)Npad * KLD;           // [Npad][KLD]
    int* Ktok = (int*)(Vs + (size_t)Npad * KLD);  // [Npad] token index of every window position
    float* tab = (float*)(Ktok + Npad);           // [T2]  log2 units
    const int W2 = 2 * g.ws - 1, T2 = W2 * W2;
    float* dtab = tab + T2;                       // [T2]
    const int bid = am_xcd_order(blockIdx.x, gridDim.x);
    const int part = bid % split, bwh = bid / split;
    const int h = bwh % g.H, bw = bwh / g.H, b = bw / g.nW, w = bw % g.nW;
    const int C = g.H * HD;
    const int64_t rs = 3 * (int64_t)C;
    const int lane = threadIdx.x & 63, fc = lane & 15, fg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // item / dy / key-row arithmetic stays scalar
    const int ws = g.ws;

    const bool dropped = am_dropped(g, b);                   // workgroup-uniform: dS of this sample is zero, its table share too
    if (!dropped) {
        stage_tile<HD, false>(g, qkv, rs, C + h * HD, b, w, 0, Npad, Ks, true, 1.0f);          // padded rows: b128 reads only here
        stage_tile<HD, false>(g, qkv, rs, 2 * C + h * HD, b, w, 0, Npad, Vs, false, 1.0f);
        for (int i = threadIdx.x; i < Npad; i += blockDim.x) Ktok[i] = (int)am_token(g, b, w, min(i, g.N - 1));
    }
    for (int i = threadIdx.x; i < T2; i += blockDim.x) { tab[i] = dropped ? 0.f : table16[(int64_t)i * g.H + h] * LOG2E; dtab[i] = 0.f; }
    __syncthreads();

    const int nqp = (ws + 15) / 16;               // q parts per image row
    const int qw = (ws + nqp - 1) / nqp;          // tokens per q part (<= 16)
    const int ngrp = (ws + G - 1) / G;
    const int nitem = ngrp * nqp;
    const int nwx = g.res / ws, wy = w / nwx, wx = w % nwx;
    const bf16* qt_h = qt + h * HD + fg * 8;
    const bf16* do_h = dout + h * HD + fg * 8;
    const float* lse_h = lse + ((int64_t)bw * g.H + h) * g.N;
    const float* dl_h = delta + h;
    for (int item = part + split * wave; item < (dropped ? 0 : nitem); item += split * (blockDim.x >> 6)) {
        const int grp = item / nqp, qp = item % nqp;
        const int xq = qp * qw + fc;
        const bool qv = fc < qw && xq < ws;
        const int xqc = min(xq, ws - 1);
        const int rxq = g.shift > 0 ? am_rid(g, wx * ws + xqc) : 0;
        // per register: key slot x = 16 t + 4 fg + r' (r = 4 t + r'); bit r of xmask = other x region, of pmask = beyond the row
        unsigned xmask = 0, pmask = 0;
#pragma unroll
        for (int r = 0; r < 4 * KT; ++r) {
            const int xk = (r >> 2) * 16 + 4 * fg + (r & 3);
            if (g.shift > 0 && am_rid(g, wx * ws + min(xk, ws - 1)) != rxq) xmask |= 1u << r;
            if (xk >= ws) pmask |= 1u << r;
        }
        // bias[r] carries the table entry, the x part of the shift mask (-100) and, for slots beyond the row, -inf.
        // (A pair that differs in both the x and the y region gets -200 instead of the reference's -100: both underflow
        //  exp() to exactly 0 in fp32 against scores bounded by tau + 16.)
        auto load_bias = [&](float* bj, int dyv) {
#pragma unroll
            for (int r = 0; r < 4 * KT; ++r) {
                const int xkc = min((r >> 2) * 16 + 4 * fg + (r & 3), ws - 1);
                float v = tab[(dyv + ws - 1) * W2 + (xqc - xkc + ws - 1)];
                v = ((xmask >> r) & 1) ? v - 100.0f * LOG2E : v;
                bj[r] = ((pmask >> r) & 1) ? NEG_BIG : v;
            }
        };
        // dtab[dy][xq - xk] += acc: lane fc, register r' of tile t lands in column fc - r' + cb(t); the four registers of a
        // diagonal sit in lanes fc..fc+3 of the row, so shl-folds give columns m = fc (16 lanes) and shr-folds m = fc - 3 < 0.
        auto flush = [&](f32x2_t* aj, int dyv) {
            float* drow = dtab + (dyv + ws - 1) * W2;
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                float a[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = (qv && !((pmask >> (4 * t + r)) & 1)) ? aj[2 * t + (r >> 1)][r & 1] : 0.f;
                const float sm = a[0] + dpp_row<0x101>(a[1]) + dpp_row<0x102>(a[2]) + dpp_row<0x103>(a[3]);
                const float sn = a[3] + dpp_row<0x111>(a[2]) + dpp_row<0x112>(a[1]);
                const int cb = qp * qw - 16 * t - 4 * fg + ws - 1;
                const int cm = cb + fc, cn = cb + fc - 3;
                if (cm >= 0 && cm < W2 && sm != 0.f) atomicAdd(drow + cm, sm);
                if (fc < 3 && cn >= 0 && cn < W2 && sn != 0.f) atomicAdd(drow + cn, sn);
            }
#pragma unroll
            for (int r = 0; r < 2 * KT; ++r) aj[r] = f32x2_t{0.f, 0.f};
        };
        f32x2_t acc[G][2 * KT];                        // register r of tile t = acc[.][2 t + (r >> 1)][r & 1]
        float bias[G][4 * KT];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int dyj = grp * G + j;
#pragma unroll
            for (int r = 0; r < 2 * KT; ++r) acc[j][r] = f32x2_t{0.f, 0.f};
            load_bias(bias[j], dyj == 0 ? 0 : min(dyj, ws - 1) - ws);
        }
        const int ka0 = min(fc, ws - 1), ka1 = min(16 + fc, ws - 1);      // A rows: key slot (t, fc) of image row yk
        const int kb0 = ka0 * KLD + fg * 8, kb1 = ka1 * KLD + fg * 8;
        // q-side operands of one image row: q~*log2e (from the dQ pass), dO, lse*log2e, delta.  Loads are unconditional
        // (clamped lanes read a valid neighbour; their columns are never flushed).
        struct QRow { U8 q, d; float L2, D; };
        auto fetch = [&](int yq, QRow& o) {
            const int nq = yq * ws + xqc;
            const int64_t tq = Ktok[nq];
            o.q.u = *(const uint4*)(qt_h + tq * C);
            o.d.u = *(const uint4*)(do_h + tq * C);
            o.L2 = lse_h[nq];
            o.D = dl_h[tq * g.H];
        };
        QRow cur, nxt;
        fetch(0, cur);
        for (int yq = 0; yq < ws; ++yq) {
#if AM_X == 5
            if (yq == 0) fetch(1, nxt);
#else
            fetch(min(yq + 1, ws - 1), nxt);
#endif
            // a dy switches from its complement (dy - ws) to itself when the sweep reaches image row dy
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int dyj = grp * G + j;                               // wave-uniform
                if (yq == dyj && dyj > 0 && dyj < ws) { flush(acc[j], dyj - ws); load_bias(bias[j], dyj); }
            }
            // G row tiles in one basic block (no branches: the scheduler overlaps one tile's LDS reads and MFMAs with the
            // previous tile's exp/fma).  The score MFMA starts from the bias, the dP MFMA from -delta; groups past the
            // window edge (dy >= ws) compute on a clamped row and are never flushed.
            const float L2 = cur.L2 * LOG2E;
            const f32x4_t nD = {-cur.D, -cur.D, -cur.D, -cur.D};
            const int rq = g.shift > 0 ? am_rid(g, wy * ws + yq) : 0;
            bf16x8_t fk0[2], fk1[2], fv0[2], fv1[2];
            auto frags = [&](int j, int slot) {
                const int dyc = min(grp * G + j, ws - 1);
                const int yk = yq - dyc + (yq < dyc ? ws : 0);
                const int ko = yk * ws * KLD;                               // scalar
                fk0[slot] = *(const bf16x8_t*)(Ks + ko + kb0); fv0[slot] = *(const bf16x8_t*)(Vs + ko + kb0);
                if constexpr (KT == 2) { fk1[slot] = *(const bf16x8_t*)(Ks + ko + kb1); fv1[slot] = *(const bf16x8_t*)(Vs + ko + kb1); }
            };
#if AM_X == 4
            if (yq == 0) { frags(0, 0); frags(1, 1); }
#else
            frags(0, 0);
#endif
#pragma unroll
            for (int j = 0; j < G; ++j) {
#if AM_X != 4
                if (j + 1 < G) frags(j + 1, (j + 1) & 1);                  // next tile's LDS reads fly under this tile's math
#endif
                const int dyc = min(grp * G + j, ws - 1);
                const int yk = yq - dyc + (yq < dyc ? ws : 0);
                const bool ydiff = g.shift > 0 && am_rid(g, wy * ws + yk) != rq;
                const float c2 = (ydiff ? -100.0f * LOG2E : 0.f) - L2;
                // register PAIRS of the MFMA results, spelled out: left to itself the vectoriser pairs registers of different
                // results and pays two v_mov per v_pk_fma_f32 to line them up (a quarter of the block's instructions)
                const f32x2_t c22 = {c2, c2};
                auto tile = [&](const bf16x8_t& fk, const bf16x8_t& fv, const float* bj, f32x2_t* aj) {
                    const f32x4_t b = {bj[0], bj[1], bj[2], bj[3]};
                    const f32x4_t sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, cur.q.v, b, 0, 0, 0);
                    const f32x4_t dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, cur.d.v, nD, 0, 0, 0);
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) {
                        const f32x2_t x = f32x2_t{sc[2 * hp], sc[2 * hp + 1]} + c22;
                        const f32x2_t e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                        aj[hp] = __builtin_elementwise_fma(e, f32x2_t{dp[2 * hp], dp[2 * hp + 1]}, aj[hp]);
                    }
                };
                tile(fk0[j & 1], fv0[j & 1], bias[j], acc[j]);
                if constexpr (KT == 2) tile(fk1[j & 1], fv1[j & 1], bias[j] + 4, acc[j] + 2);
            }
            // first use of the prefetched row pinned behind the tiles (keeps its vmcnt wait out of the MFMA block)
            asm volatile("" : "+v"(nxt.q.v), "+v"(nxt.d.v), "+v"(nxt.L2), "+v"(nxt.D));
            cur = nxt;
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int dyj = grp * G + j;
            if (dyj < ws) flush(acc[j], dyj);
        }
    }
    __syncthreads();
    if (part_out) {                                   // per-workgroup partial table, summed by attn_dbias_reduce_k
        float* o = part_out + (size_t)bid * T2;
        for (int i = threadIdx.x; i < T2; i += blockDim.x) o[i] = dtab[i];
        return;
    }
    for (int i = threadIdx.x; i < T2; i += blockDim.x) {
        const float v = dtab[i];
        if (v != 0.f) atomicAdd(dtable16 + (int64_t)i * g.H + h, v);
    }
}

// dtable16[i][h] += sum over (bw, p) of part[((bw * H + h) * split + p)][i]; blockIdx = (entry block, head, bw slab)
__global__ __launch_bounds__(256) void attn_dbias_reduce_k(const float* __restrict__ part, float* __restrict__ dtable16, int T2, int H, int BW,
                                                           int split) {
    const int i = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
    const int per = (BW + gridDim.z - 1) / gridDim.z;
    const int b0 = blockIdx.z * per, b1 = min(BW, b0 + per);
    if (i >= T2) return;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int n = (b1 - b0) * split;                  // partials of this slab: rows ((bw * H + h) * split + p), p fastest
    auto at = [&](int kk) { const int bw = b0 + kk / split, p = kk % split; return part[((size_t)(bw * H + h) * split + p) * T2 + i]; };
    for (int k = 0; k < n; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { const float t = at(min(k + u, n - 1)); a[u] += (k + u < n) ? t : 0.f; }
    }
    const float v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    if (v != 0.f) atomicAdd(dtable16 + (int64_t)i * H + h, v);
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// per-query info word Qi: MODE 0: 4*(iy*(2w-1)+ix + C0) | region << 16 (| AM_PAD); MODE 1: valid (| AM_PAD).  tabk = (char*)tab - 4*bk.
// Qd holds MINUS delta: it is the initial accumulator of the dP = dO.V^T product, as the bias is of S = Q~.K^T.
template <int HD, int MODE, bool MASK, bool TAIL, int NT, bool G4, bool DROP = false>
__device__ __forceinline__ void am_dkv_block(const bf16* __restrict__ Qs, const bf16* __restrict__ Ds, const int* __restrict__ Qi,
                                             const float* __restrict__ Ql, const float* __restrict__ Qd, int qb, const bf16x8_t (&kf)[HD / 32],
                                             const bf16x8_t (&vf)[HD / 32], const char* tabk, int regk, int vk, int lane,
                                             f32x4_t (&dk)[HD / 16], f32x4_t (&dv)[HD / 16], unsigned ebase = 0, unsigned NLh = 0,
                                             unsigned dseed = 0, unsigned dthr = 0, float dinv = 1.f) {
    const int fc = lane & 15, fg = lane >> 4;
    f32x4_t s[NT], dp[NT];
    int qi[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int4 inf = *(const int4*)(Qi + qb + 16 * t + 4 * fg);
        qi[t][0] = inf.x; qi[t][1] = inf.y; qi[t][2] = inf.z; qi[t][3] = inf.w;
        s[t] = am_bias4<MODE, MASK, TAIL, G4>(qi[t], tabk);
        const f32x4_t nD = *(const f32x4_t*)(Qd + qb + 16 * t + 4 * fg);
        dp[t] = DROP ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : nD;
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
            const int o = am_off<HD, HD == 32>(qb + 16 * t + fc, ks * 4 + fg);
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)(Qs + o), kf[ks], s[t], 0, 0, 0);
            dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)(Ds + o), vf[ks], dp[t], 0, 0, 0);
        }
    }
    u32x4_t pw[NT / 2], dsw[NT / 2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4_t l4 = *(const f32x4_t*)(Ql + qb + 16 * t + 4 * fg);
        unsigned H4[4];
        if (DROP) {
            // The hash of (query row, key pair) serves the lanes of keys 2j and 2j + 1 (neighbours: the key is the lane): each computes the
            // hashes of two of the tile's four query rows and they swap; ebase = (b, h) row block * ceil(N/2) + this lane's key pair.
            const unsigned par = lane & 1;
            const unsigned c0 = ebase + (unsigned)(qb + 16 * t + 4 * fg + 2 * par) * NLh;
            const unsigned a = am_hash(c0 ^ dseed), b = am_hash((c0 + NLh) ^ dseed);
            const unsigned pa = am_swap1(a), pb = am_swap1(b);
            H4[0] = par ? pa : a; H4[1] = par ? pb : b; H4[2] = par ? a : pa; H4[3] = par ? b : pb;
        }
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            f32x2_t sv = {s[t][2 * hp], s[t][2 * hp + 1]};
            const f32x2_t L = {l4[2 * hp], l4[2 * hp + 1]}, dpv = {dp[t][2 * hp], dp[t][2 * hp + 1]};
            if (MODE != 0 || MASK || TAIL) {
                sv[0] = am_mask<MODE, MASK, TAIL>(sv[0], qi[t][2 * hp], regk, vk);
                sv[1] = am_mask<MODE, MASK, TAIL>(sv[1], qi[t][2 * hp + 1], regk, vk);
            }
            const f32x2_t p = am_exp2(sv + L);                                  // L = -lse; padding queries -> NEG_BIG -> 0
            if (DROP) {
                // kept probabilities are scaled by 1/(1-p) once, on dV at the end; dS = P o (mask o dP / (1-p) - delta)
                f32x2_t pd, ds;
                const unsigned sh = (lane & 1) * 16, thr15m1 = dthr & 0xffffu;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = 2 * hp + e;
                    const bool keep = ((H4[r] >> sh) & 0x7fffu) > thr15m1;
                    const float nd = Qd[qb + 16 * t + 4 * fg + r];
                    pd[e] = keep ? p[e] : 0.f;
                    ds[e] = p[e] * fmaf(keep ? dpv[e] : 0.f, dinv, nd);
                }
                pw[t >> 1][(t & 1) * 2 + hp] = am_pk(pd);
                dsw[t >> 1][(t & 1) * 2 + hp] = am_pk(ds);
            } else {
                pw[t >> 1][(t & 1) * 2 + hp] = am_pk(p);
                dsw[t >> 1][(t & 1) * 2 + hp] = am_pk(p * dpv);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < HD / 16; ++d)
#pragma unroll
        for (int pr = 0; pr < NT / 2; ++pr) {
            dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(read_tr<HD>(Ds, d * 16, qb + 32 * pr, lane), __builtin_bit_cast(bf16x8_t, pw[pr]), dv[d], 0, 0, 0);
            dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(read_tr<HD>(Qs, d * 16, qb + 32 * pr, lane), __builtin_bit_cast(bf16x8_t, dsw[pr]), dk[d], 0, 0, 0);
        }
}

template <int HD, int MODE, bool MASK>
__global__ __launch_bounds__((HD == 32 || MODE == 1) ? 1024 : 512) void attn_bwd_dkv_mfma_k(AttnGeom g, const bf16* __restrict__ qkv, const float* __restrict__ table16,
                                                           const float* __restrict__ logit_scale, const int* __restrict__ valid,
                                                           const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, bf16* __restrict__ dqkv, int Npad, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KLD = AmTile<HD, HD == 32>::LD;
    bf16* Qs = (bf16*)smem;                       // [Npad][KLD]   q~ * log2(e)
    bf16* Ds = Qs + (size_t)Npad * KLD;           // [Npad][KLD]   dO
    float* Ql = (float*)(Ds + (size_t)Npad * KLD);// [Npad] lse * log2(e)
    float* Qd = Ql + Npad;                        // [Npad] delta
    int* Qi = (int*)(Qd + Npad);                  // [Npad] info
    float* tab = (float*)(Qi + Npad);             // MODE 0: [T2], log2 units
    int bwh, part;
    am_part(g, ksplit, bwh, part, ksplit);
    const int h = bwh % g.H, bw = bwh / g.H, b = bw / g.nW, w = bw % g.nW;
    const int C = g.H * HD;
    const int64_t rs = 3 * (int64_t)C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fc = lane & 15, fg = lane >> 4;
    const int64_t lse0 = ((int64_t)bw * g.H + h) * g.N;
    const unsigned NL = (unsigned)g.N;
    const int Np = am_localize(g, b, Npad);
    if (Np == 0) return;
    if (MODE == 0 && am_dropped(g, b)) {                     // d(out) of this sample is zero: so are dK and dV (workgroup-uniform)
        const int ntile0 = (g.N + 15) / 16;
        for (int kt = part + ksplit * wave; kt < ntile0; kt += ksplit * (blockDim.x >> 6)) {
            const int nk = kt * 16 + fc;
            if (nk < g.N) {
                const int64_t t = am_token(g, b, w, nk);
#pragma unroll
                for (int d = 0; d < HD / 16; ++d) {
                    *(uint2*)(dqkv + t * rs + C + h * HD + d * 16 + 4 * fg) = make_uint2(0, 0);
                    *(uint2*)(dqkv + t * rs + 2 * C + h * HD + d * 16 + 4 * fg) = make_uint2(0, 0);
                }
            }
        }
        return;
    }
    const bool drop = MODE == 1 && g.drop_inv != 1.0f;
    const unsigned dseed = drop ? am_seed(g) : 0u;
    int C0 = 0;
    float qmul = g.scale;
    const int T2 = MODE == 0 ? (2 * g.ws - 1) * (2 * g.ws - 1) : 0;
    if (MODE == 0) {
        for (int i = threadIdx.x; i < T2; i += blockDim.x) tab[i] = table16[(int64_t)i * g.H + h] * LOG2E;
        C0 = (g.ws - 1) * (2 * g.ws - 1) + (g.ws - 1);
        qmul = __expf(fminf(logit_scale[h], LN100));
    }
    stage_tile<HD>(g, qkv, rs, h * HD, b, w, 0, Np, Qs, MODE == 0, qmul * LOG2E);
    stage_tile<HD>(g, dout, C, h * HD, b, w, 0, Np, Ds, false, 1.0f);
    for (int i = threadIdx.x; i < Np; i += blockDim.x) {
        const int inf = am_info(g, valid, b, w, i);
        Qi[i] = MODE == 0 ? ((((inf & 0xffff) + C0) << 2) | (inf & ~0xffff)) : inf;
        float L = 0.f, D = 0.f;
        if (i < g.N) {
            L = lse[lse0 + i] * LOG2E;
            D = delta[am_token(g, b, w, i) * g.H + h];
        }
        Ql[i] = -L;                // negated, like delta: the block adds it (v_pk_add_f32)
        Qd[i] = -D;
    }
    __syncthreads();

    const bool g4 = MODE == 0 && (g.ws & 3) == 0;
    const bool wmask = MASK && am_window_masked(g, w);
    const int ysp = am_ysplit(g, w, MODE == 0 ? qmul : 0.f);
    const int ntile = (g.N + 15) / 16;
    const int nfull64 = (g.N / 64) * 64;
    for (int kt = part + ksplit * wave; kt < ntile; kt += ksplit * (blockDim.x >> 6)) {
        const int nk = kt * 16 + fc;
        const bool kok = nk < g.N;
        const int nkc = kok ? nk : g.N - 1;
        const int64_t tk = am_token(g, b, w, nkc);
        const int kinf = am_info(g, valid, b, w, nkc);          // clamped key: its column is simply not stored
        const char* tabk = (const char*)tab - 4 * (kinf & 0xffff);
        const int regk = (kinf >> 16) & 0xff, vk = kinf;
        bf16x8_t kf[HD / 32], vf[HD / 32];
        {
            float f[HD / 32][8];
            float ss = 0.f;
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
                U8 x, y;
                x.u = *(const uint4*)(qkv + tk * rs + C + h * HD + ks * 32 + fg * 8);
                y.u = *(const uint4*)(qkv + tk * rs + 2 * C + h * HD + ks * 32 + fg * 8);
                vf[ks] = y.v;
#pragma unroll
                for (int e = 0; e < 8; ++e) { f[ks][e] = (float)x.e[e]; ss += f[ks][e] * f[ks][e]; }
            }
            const float sc = MODE == 0 ? 1.0f / fmaxf(sqrtf(sum4g(ss)), 1e-12f) : 1.0f;
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) kf[ks][e] = (bf16)(f[ks][e] * sc);
        }
        f32x4_t dk[HD / 16], dv[HD / 16];
#pragma unroll
        for (int d = 0; d < HD / 16; ++d) { dk[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[d] = dk[d]; }
        int qb = 0;
        const int k0 = kt * 16, k1 = min(k0 + 16, g.N);                         // this wave's keys (wave-uniform)
        if (MODE == 1 && drop) {
            const unsigned NLh = (NL + 1) >> 1;
            const unsigned eb = (unsigned)lse0 * NLh + ((unsigned)nk >> 1);      // (b, h) row block + this lane's key PAIR (unclamped: neighbours agree); + q * NLh per query
            for (; qb < nfull64; qb += 64)
                am_dkv_block<HD, MODE, MASK, false, 4, false, MODE == 1>(Qs, Ds, Qi, Ql, Qd, qb, kf, vf, tabk, regk, vk, lane, dk, dv, eb, NLh, dseed, g.drop_thr, g.drop_inv);
            for (; qb < Np; qb += 32)
                am_dkv_block<HD, MODE, MASK, true, 2, false, MODE == 1>(Qs, Ds, Qi, Ql, Qd, qb, kf, vf, tabk, regk, vk, lane, dk, dv, eb, NLh, dseed, g.drop_thr, g.drop_inv);
        } else {
#define AM_DKV_SWEEP(MK)                                                                                                                       \
            if (g4) {                                                                                                                          \
                for (; qb < nfull64; qb += 64) { if (MK && am_yskip(ysp, qb, qb + 64, k0, k1)) continue;                                       \
                    am_dkv_block<HD, MODE, MK, false, 4, true>(Qs, Ds, Qi, Ql, Qd, qb, kf, vf, tabk, regk, vk, lane, dk, dv); }               \
                for (; qb < Np; qb += 32) { if (MK && am_yskip(ysp, qb, g.N, k0, k1)) continue;                                               \
                    am_dkv_block<HD, MODE, MK, true, 2, true>(Qs, Ds, Qi, Ql, Qd, qb, kf, vf, tabk, regk, vk, lane, dk, dv); }                \
            } else {                                                                                                                           \
                for (; qb < nfull64; qb += 64) { if (MK && am_yskip(ysp, qb, qb + 64, k0, k1)) continue;                                       \
                    am_dkv_block<HD, MODE, MK, false, 4, false>(Qs, Ds, Qi, Ql, Qd, qb, kf, vf, tabk, regk, vk, lane, dk, dv); }              \
                for (; qb < Np; qb += 32) { if (MK && am_yskip(ysp, qb, g.N, k0, k1)) continue;                                               \
                    am_dkv_block<HD, MODE, MK, true, 2, false>(Qs, Ds, Qi, Ql, Qd, qb, kf, vf, tabk, regk, vk, lane, dk, dv); }               \
            }
            if (MASK && wmask) { AM_DKV_SWEEP(true) } else { AM_DKV_SWEEP(false) }
#undef AM_DKV_SWEEP
        }
        // dv[d][r], dk[d][r]: dim d*16 + 4*fg + r of key fc; dk was accumulated against q~ * log2(e)
        if (kok) {
            const float dvs = drop ? g.drop_inv : 1.0f;          // dropout: dV = (mask o P)^T dO / (1-p), the scale applied once here
#pragma unroll
            for (int d = 0; d < HD / 16; ++d) {
                U4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o.e[r] = (bf16)(dv[d][r] * dvs);
                *(uint2*)(dqkv + tk * rs + 2 * C + h * HD + d * 16 + 4 * fg) = o.u;
            }
        }
        if (MODE == 0) {
            float kh[HD / 16][4];
            float ss = 0.f;
#pragma unroll
            for (int d = 0; d < HD / 16; ++d) {
                U4 x;
                x.u = *(const uint2*)(qkv + tk * rs + C + h * HD + d * 16 + 4 * fg);
#pragma unroll
                for (int r = 0; r < 4; ++r) { kh[d][r] = (float)x.e[r]; ss += kh[d][r] * kh[d][r]; }
            }
            const float kinv = 1.0f / fmaxf(sqrtf(sum4g(ss)), 1e-12f);
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < HD / 16; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) { kh[d][r] *= kinv; dk[d][r] *= LN2; dot += dk[d][r] * kh[d][r]; }
            dot = sum4g(dot);
            if (kok) {
#pragma unroll
                for (int d = 0; d < HD / 16; ++d) {
                    U4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o.e[r] = (bf16)((dk[d][r] - kh[d][r] * dot) * kinv);
                    *(uint2*)(dqkv + tk * rs + C + h * HD + d * 16 + 4 * fg) = o.u;
                }
            }
        } else if (kok) {
#pragma unroll
            for (int d = 0; d < HD / 16; ++d) {
                U4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o.e[r] = (bf16)(dk[d][r] * LN2);
                *(uint2*)(dqkv + tk * rs + C + h * HD + d * 16 + 4 * fg) = o.u;
            }
        }
    }
}

// ================================================================================================ window fast path (round 3)
// MODE 0, head_dim 32, window side a multiple of 4 (SwinV2-base stages 0-2: 28 x 28 windows).  Same arithmetic, same block sizes and
// the same accumulation order as the kernels above -- outputs are BIT-IDENTICAL (tests compare the two) -- but the passes were bound by
// the LDS (profiles/r03_base_attn_counters.csv: LDS array busy 0.50-0.80 of the CU's cycles, 37-40 % of that bank conflicts), so the LDS
// work per score is cut:
//   * bias words: the table is kept THREE times, copy c shifted by c words, S3 bytes apart.  A table word index w is encoded as
//     8 * (w >> 1) + (w & 1) * S3; the sum of a query-side and a key-side encoding then addresses copy (parity + parity) at an 8-byte
//     aligned word pair -- the carry of the two parities lands in the copy index instead of breaking the alignment -- so a lane's four
//     adjacent words are two ds_read_b64 (2 LDS cycles each, 64 banks) instead of two ds_read2_b32 (4 cycles each, 32 banks), with
//     one v_add for the address as before.  S3 / 4 = 22 mod 64 keeps the copies' bank ranges apart.
//   * the encoded offsets of a block's key groups are one 16-byte read per lane (Koff[block][fg][tile]) instead of four info words;
//   * all bias reads of a block are issued back to back in ONE asm statement behind one wait (the compiled loops above hold one
//     lgkmcnt(0) per tile: DESIGN section 9b item 6);
//   * forward: two query tiles per wave share every K fragment, transposed V read and offset word (as the dQ pass already did);
//     dK/dV: two key tiles per wave share the Q~ / dO fragments, their transposed reads and the per-query lse / delta / offset words.
typedef __attribute__((address_space(3))) char* lds_cp;
__device__ __forceinline__ unsigned aw_lds(const void* p) { return (unsigned)(uintptr_t)(lds_cp)(char*)p; }
__host__ __device__ inline int aw_stride_bytes(int T2) {
    int s = (T2 + 3) / 2 * 2;            // words: room for the copy shifted by two, even
    while ((s & 63) != 22) s += 2;
    return s * 4;
}
__device__ __forceinline__ int aw_enc(int w, int S3) { return ((w >> 1) << 3) + (w & 1) * S3; }

// three shifted copies of the head's table in log2 units; REV: entry i holds table[T2 - 1 - i] (forward / dQ: keys on the rows)
template <bool REV>
__device__ __forceinline__ void aw_fill_table(float* tab3, const float* __restrict__ table16, int T2, int H, int h, int S3) {
    const int W = S3 >> 2;
    for (int i = threadIdx.x; i < 3 * W; i += blockDim.x) {
        const int c = i / W, j = i - c * W + c;
        tab3[i] = j < T2 ? table16[(int64_t)(REV ? T2 - 1 - j : j) * H + h] * LOG2E : 0.f;
    }
}
// Koff[blk][fg][t] = encoded table offset of window position blk*64 + 16 t + 4 fg (positions >= N: 0, a safe in-range read)
__device__ __forceinline__ void aw_fill_off(int* Koff, int nblk, int N, int ws, int S3) {
    const int W2 = 2 * ws - 1;
    for (int i = threadIdx.x; i < nblk * 16; i += blockDim.x) {
        const int n = (i >> 4) * 64 + (i & 3) * 16 + ((i >> 2) & 3) * 4;
        Koff[i] = n < N ? aw_enc((n / ws) * W2 + n % ws, S3) : 0;
    }
}

// bias words of NB (tile, tile) pairs: per pair two 8-byte reads, all issued back to back, one wait; outputs valid on return
__device__ __forceinline__ void aw_bias_read(const unsigned (&a)[2], f32x4_t (&b)[2]) {
    f32x2_t l0, h0, l1, h1;
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %5\n\tds_read_b64 %3, %5 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1) : "v"(a[0]), "v"(a[1]) : "memory");
    b[0] = __builtin_shufflevector(l0, h0, 0, 1, 2, 3);
    b[1] = __builtin_shufflevector(l1, h1, 0, 1, 2, 3);
}
__device__ __forceinline__ void aw_bias_read(const unsigned (&a)[4], f32x4_t (&b)[4]) {
    f32x2_t l0, h0, l1, h1, l2, h2, l3, h3;
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %9\n\tds_read_b64 %3, %9 offset:8\n\t"
                 "ds_read_b64 %4, %10\n\tds_read_b64 %5, %10 offset:8\n\tds_read_b64 %6, %11\n\tds_read_b64 %7, %11 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
    b[0] = __builtin_shufflevector(l0, h0, 0, 1, 2, 3);
    b[1] = __builtin_shufflevector(l1, h1, 0, 1, 2, 3);
    b[2] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3);
    b[3] = __builtin_shufflevector(l3, h3, 0, 1, 2, 3);
}
__device__ __forceinline__ void aw_bias_read(const unsigned (&a)[8], f32x4_t (&b)[8]) {
    f32x2_t l[8], h[8];
    asm volatile("ds_read_b64 %0, %16\n\tds_read_b64 %1, %16 offset:8\n\tds_read_b64 %2, %17\n\tds_read_b64 %3, %17 offset:8\n\t"
                 "ds_read_b64 %4, %18\n\tds_read_b64 %5, %18 offset:8\n\tds_read_b64 %6, %19\n\tds_read_b64 %7, %19 offset:8\n\t"
                 "ds_read_b64 %8, %20\n\tds_read_b64 %9, %20 offset:8\n\tds_read_b64 %10, %21\n\tds_read_b64 %11, %21 offset:8\n\t"
                 "ds_read_b64 %12, %22\n\tds_read_b64 %13, %22 offset:8\n\tds_read_b64 %14, %23\n\tds_read_b64 %15, %23 offset:8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1]), "=&v"(l[2]), "=&v"(h[2]), "=&v"(l[3]), "=&v"(h[3]),
                   "=&v"(l[4]), "=&v"(h[4]), "=&v"(l[5]), "=&v"(h[5]), "=&v"(l[6]), "=&v"(h[6]), "=&v"(l[7]), "=&v"(h[7])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = __builtin_shufflevector(l[i], h[i], 0, 1, 2, 3);
}

// encoded offsets of the NT key (query) groups of the block at position kb: Koff[kb / 64][fg][t], NT = 2 takes half of the 16-byte row
template <int NT>
__device__ __forceinline__ void aw_load_off(const int* __restrict__ Koff, int kb, int fg, int (&ko)[NT]) {
    if constexpr (NT == 4) {
        const int4 v = *(const int4*)(Koff + (kb >> 6) * 16 + fg * 4);
        ko[0] = v.x; ko[1] = v.y; ko[2] = v.z; ko[3] = v.w;
    } else {
        const int2 v = *(const int2*)(Koff + (kb >> 6) * 16 + fg * 4 + ((kb >> 5) & 1) * 2);
        ko[0] = v.x; ko[1] = v.y;
    }
}

// ------------------------------------------------------------------------------------------------ forward, two query tiles per wave
#define AW_LAZY_TH 6.0f
template <bool MASK, bool TAIL, int NT, int QT, bool LAZY>
__device__ __forceinline__ void aw_fwd_block(const bf16* __restrict__ Ks, const bf16* __restrict__ Vs, const int* __restrict__ Kinfo,
                                             const int* __restrict__ Koff, int kb, const bf16x8_t (&qf)[QT], const unsigned (&tabq)[QT],
                                             const int (&regq)[QT], int lane, float (&m)[QT], f32x4_t (&lacc)[QT], f32x4_t (&oacc)[QT][2],
                                             const bf16x8_t& ones) {
    const int fc = lane & 15, fg = lane >> 4;
    int ko[NT];
    aw_load_off<NT>(Koff, kb, fg, ko);
    unsigned a[QT * NT];
    f32x4_t s[QT * NT];
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
        for (int t = 0; t < NT; ++t) a[q * NT + t] = tabq[q] + (unsigned)ko[t];
    aw_bias_read(a, s);
    int ki[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (MASK || TAIL) {
            const int4 inf = *(const int4*)(Kinfo + kb + 16 * t + 4 * fg);
            ki[t][0] = inf.x; ki[t][1] = inf.y; ki[t][2] = inf.z; ki[t][3] = inf.w;
        }
        const bf16x8_t kfr = *(const bf16x8_t*)(Ks + am_off<32, true>(kb + 16 * t + fc, fg));
#pragma unroll
        for (int q = 0; q < QT; ++q) s[q * NT + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[q], s[q * NT + t], 0, 0, 0);
    }
    bf16x8_t pb[QT][NT / 2];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        float bm = NEG_BIG;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (MASK || TAIL) s[q * NT + t][r] = am_mask<0, MASK, TAIL>(s[q * NT + t][r], ki[t][r], regq[q], 0);
                bm = fmaxf(bm, s[q * NT + t][r]);
            }
        if (LAZY) {
            // Deferred maximum: the reference point m of a query moves only when some score of the block exceeds it by more than
            // 2^AW_LAZY_TH (wave-uniform test on the lanes' own maxima); until then p = 2^(s - m) <= 2^AW_LAZY_TH needs no cross-lane
            // reduction and the accumulators no rescale (the passes are bound by instruction issue: profiles/r03_win_attn_counters.csv).
            if (__builtin_amdgcn_ballot_w64(bm > m[q] + AW_LAZY_TH) != 0) {
                const float mn = fmaxf(m[q], max4g(bm));
                const float alpha = __builtin_amdgcn_exp2f(m[q] - mn);
                m[q] = mn;
                lacc[q] *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d) oacc[q][d] *= alpha;
            }
        } else {
            const float mn = fmaxf(m[q], max4g(bm));
            const float alpha = __builtin_amdgcn_exp2f(m[q] - mn);
            m[q] = mn;
            lacc[q] *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d) oacc[q][d] *= alpha;
        }
        u32x4_t pw[NT / 2];
        const f32x2_t mn2 = {m[q], m[q]};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int hp = 0; hp < 2; ++hp)
                pw[t >> 1][(t & 1) * 2 + hp] = am_pk(am_exp2((f32x2_t){s[q * NT + t][2 * hp], s[q * NT + t][2 * hp + 1]} - mn2));
#pragma unroll
        for (int pr = 0; pr < NT / 2; ++pr) {
            pb[q][pr] = __builtin_bit_cast(bf16x8_t, pw[pr]);
            lacc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[q][pr], lacc[q], 0, 0, 0);
        }
    }
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int pr = 0; pr < NT / 2; ++pr) {
            const bf16x8_t vt = read_tr<32>(Vs, d * 16, kb + 32 * pr, lane);
#pragma unroll
            for (int q = 0; q < QT; ++q) oacc[q][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pb[q][pr], oacc[q][d], 0, 0, 0);
        }
}

template <bool MASK, bool LAZY>
__global__ __launch_bounds__(1024) void attn_fwd_win_k(AttnGeom g, const bf16* __restrict__ qkv, const float* __restrict__ table16,
                                                      const float* __restrict__ logit_scale, bf16* __restrict__ out, float* __restrict__ lse,
                                                      int Npad, int qsplit, int S3) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HD = 32, QT = 2;
    bf16* Ks = (bf16*)smem;                       // [Npad][32] swizzled, normalised keys
    bf16* Vs = Ks + (size_t)Npad * HD;            // [Npad][32]
    int* Kinfo = (int*)(Vs + (size_t)Npad * HD);  // [Npad]  region ids / padding flags (masked and tail blocks only)
    const int nblk = (Npad + 63) >> 6;
    int* Koff = Kinfo + Npad;                     // [nblk][4][4]
    float* tab3 = (float*)(Koff + nblk * 16);     // 3 x S3 bytes, log2 units, reversed
    int bwh, part;
    am_part(g, qsplit, bwh, part, qsplit);
    const int h = bwh % g.H, bw = bwh / g.H, b = bw / g.nW, w = bw % g.nW;
    const int C = g.H * HD;
    const int64_t rs = 3 * (int64_t)C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fc = lane & 15, fg = lane >> 4;
    const int64_t lse0 = ((int64_t)bw * g.H + h) * g.N;
    const int Np = Npad;
    if (am_dropped(g, b)) {                                  // workgroup-uniform: zeros where this workgroup would have written
        const int ntile0 = (g.N + 15) / 16, nitem0 = (ntile0 + QT - 1) / QT;
        for (int item = part + qsplit * wave; item < nitem0; item += qsplit * (blockDim.x >> 6))
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const int nq = (item * QT + q) * 16 + fc;
                if (item * QT + q < ntile0 && nq < g.N) {
                    const int64_t t = am_token(g, b, w, nq);
                    *(uint2*)(out + t * C + h * HD + 4 * fg) = make_uint2(0, 0);
                    *(uint2*)(out + t * C + h * HD + 16 + 4 * fg) = make_uint2(0, 0);
                    if (fg == 0) lse[lse0 + nq] = 0.f;
                }
            }
        return;
    }

    stage_tile<HD>(g, qkv, rs, C + h * HD, b, w, 0, Np, Ks, true, 1.0f);
    stage_tile<HD>(g, qkv, rs, 2 * C + h * HD, b, w, 0, Np, Vs, false, 1.0f);
    for (int i = threadIdx.x; i < Np; i += blockDim.x) Kinfo[i] = am_info(g, nullptr, b, w, i);
    const int W2 = 2 * g.ws - 1, T2 = W2 * W2;
    aw_fill_off(Koff, nblk, g.N, g.ws, S3);
    aw_fill_table<true>(tab3, table16, T2, g.H, h, S3);
    const int C0 = T2 - 1 - ((g.ws - 1) * W2 + (g.ws - 1));
    const float tau = __expf(fminf(logit_scale[h], LN100));
    __syncthreads();

    const int ntile = (g.N + 15) / 16, nitem = (ntile + QT - 1) / QT;
    const int nfull64 = (g.N / 64) * 64;
    const bool wmask = MASK && am_window_masked(g, w);
    const int ysp = am_ysplit(g, w, tau);
    const unsigned tab0 = aw_lds(tab3);
    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    for (int item = part + qsplit * wave; item < nitem; item += qsplit * (blockDim.x >> 6)) {
        bool qok[QT];
        int nqv[QT], regq[QT];
        int64_t tq[QT];
        unsigned tabq[QT];
        bf16x8_t qf[QT];
        float m[QT];
        f32x4_t lacc[QT], oacc[QT][2];
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const int qt = item * QT + q;
            const bool tok = qt < ntile;                         // wave-uniform: an odd tile count leaves the last slot a duplicate
            const int nq = (tok ? qt : ntile - 1) * 16 + fc;
            qok[q] = tok && nq < g.N;
            nqv[q] = nq;
            const int nqc = nq < g.N ? nq : g.N - 1;
            tq[q] = am_token(g, b, w, nqc);
            const int qinf = am_info(g, nullptr, b, w, nqc);
            tabq[q] = tab0 + (unsigned)aw_enc(C0 - (qinf & 0xffff), S3);
            regq[q] = (qinf >> 16) & 0xff;
            float f[8];
            float ss = 0.f;
            U8 x;
            x.u = *(const uint4*)(qkv + tq[q] * rs + h * HD + fg * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[e] = (float)x.e[e]; ss += f[e] * f[e]; }
            const float sc = tau * LOG2E / fmaxf(sqrtf(sum4g(ss)), 1e-12f);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[q][e] = (bf16)(f[e] * sc);
            m[q] = -INFINITY;
            lacc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            oacc[q][0] = lacc[q]; oacc[q][1] = lacc[q];
        }
        int kb = 0;
        if (MASK && wmask) {
            const int q0 = item * QT * 16, q1 = min(q0 + QT * 16, g.N);          // wave-uniform
            for (; kb < nfull64; kb += 64) {
                if (am_yskip(ysp, q0, q1, kb, kb + 64)) continue;
                aw_fwd_block<true, false, 4, QT, LAZY>(Ks, Vs, Kinfo, Koff, kb, qf, tabq, regq, lane, m, lacc, oacc, ones);
            }
            for (; kb < Np; kb += 32) {
                if (am_yskip(ysp, q0, q1, kb, g.N)) continue;
                aw_fwd_block<true, true, 2, QT, LAZY>(Ks, Vs, Kinfo, Koff, kb, qf, tabq, regq, lane, m, lacc, oacc, ones);
            }
        } else {
            for (; kb < nfull64; kb += 64) aw_fwd_block<false, false, 4, QT, LAZY>(Ks, Vs, Kinfo, Koff, kb, qf, tabq, regq, lane, m, lacc, oacc, ones);
            for (; kb < Np; kb += 32) aw_fwd_block<false, true, 2, QT, LAZY>(Ks, Vs, Kinfo, Koff, kb, qf, tabq, regq, lane, m, lacc, oacc, ones);
        }
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const float l = lacc[q][0];
            if (qok[q]) {
                const float inv = 1.0f / l;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    U4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o.e[r] = (bf16)(oacc[q][d][r] * inv);
                    *(uint2*)(out + tq[q] * C + h * HD + d * 16 + 4 * fg) = o.u;
                }
                if (fg == 0) lse[lse0 + nqv[q]] = (m[q] + __log2f(l)) * LN2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
static int am_check(const char* fn, int mode, int B, int H, int hd, int N, int nW, int res, int ws, int shift) {
    MV_CHECK_ARG(mode == 0 || mode == 1 || mode == 2, "%s: mode %d", fn, mode);
    MV_CHECK_ARG(B > 0 && H > 0 && N > 0 && nW > 0, "%s: empty geometry", fn);
    MV_CHECK_ARG(hd == 32 || hd == 64, "%s: head_dim %d unsupported (32|64)", fn, hd);
    if (mode == 0) {
        MV_CHECK_ARG(ws > 0 && ws < 128 && res % ws == 0 && N == ws * ws && nW == (res / ws) * (res / ws), "%s: window geometry", fn);
        MV_CHECK_ARG(shift >= 0 && shift < ws, "%s: shift %d", fn, shift);
    } else {
        MV_CHECK_ARG(nW == 1, "%s: pad mode needs nW=1", fn);
        MV_CHECK_ARG(mode == 1 || res > 0, "%s: packed mode passes the total token count in `res`", fn);
    }
    MV_CHECK_ARG((int64_t)B * nW * H * 8 < 2147483647LL, "%s: grid", fn);
    return 0;
}

template <typename F>
static int am_set_lds(F fn, size_t bytes, const char* name) {
    MV_CHECK_ARG(bytes <= 160 * 1024, "%s: needs %zu bytes of LDS (> 160 KiB): sequence / window too long for this kernel", name, bytes);
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        mvuld_set_error("%s: hipFuncSetAttribute(%zu) failed", name, bytes);
        return 1;
    }
    return 0;
}

// enough workgroups to fill 256 CUs a few times over when (windows x heads) alone is small
static int am_split(int64_t groups, int ntile) {
    // Query / key tiles of one (window, head) are split over several workgroups only while there are fewer workgroups than
    // ~2 per CU: every split re-stages the whole K/V (or Q~/dO) tile.  (1024 measured 14-25 % slower than 512 on the 512-group
    // stage-2 layers and the 384-group text encoder.)
    static const int target = [] { const char* e = getenv("MVULD_ATTN_SPLIT_TARGET"); return e ? atoi(e) : 512; }();      // once, thread-safe
    int s = 1;
    while (groups * s < target && s * 2 * 8 <= ntile) s *= 2;
    return s;
}

// (Round 3 measured a "tail split" of the partial last round of (window x head) groups -- the text encoder's 384 groups on 256 CUs -- neutral:
// forward 73.3 vs 73.8 us, backward 177.9 vs 179.9 us, step 60.4 vs 60.2 ms; its knob left the library in round 4, the `whole` field of
// AttnGeom and am_part's handling of it remain for the record: tools/experiments/README.md.)
// Skipping of the tiles across a shifted window's vertical mask split (am_ysplit): MVULD_ATTN_YSKIP / mvuld_set_attn_yskip, 1 = on (default),
// 0 = every pair is computed.  Bit-identical either way while tau <= 22 (tests); a head past that bound computes every pair regardless.
static std::atomic<int> g_am_yskip{-1};
extern "C" int mvuld_set_attn_yskip(int on) {
    g_am_yskip.store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}
static int am_yskip_on() {
    int v = g_am_yskip.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("MVULD_ATTN_YSKIP");
        v = (!e || atoi(e) != 0) ? 1 : 0;
        g_am_yskip.store(v, std::memory_order_relaxed);
    }
    return v;
}
// Forward window fast path (attn_fwd_win_k) for MODE 0, head_dim 32, window side % 4 == 0.  MVULD_ATTN_WIN / mvuld_set_attn_win:
// 1 (default) = fast path with the deferred maximum, 2 = fast path on the general kernel's exact schedule (bit-identical to it: test),
// 0 = the general kernel.
static std::atomic<int> g_am_win{-1};
extern "C" int mvuld_set_attn_win(int mode) {
    g_am_win.store(mode < 0 ? 0 : (mode > 2 ? 2 : mode), std::memory_order_relaxed);
    return 0;
}
static int am_win_mode() {
    int v = g_am_win.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("MVULD_ATTN_WIN");
        v = e ? atoi(e) : 1;
        v = v < 0 ? 0 : (v > 2 ? 2 : v);
        g_am_win.store(v, std::memory_order_relaxed);
    }
    return v;
}
// Fused single-pass window backward (attn_bwd_fused_win_k): MVULD_ATTN_BWD_FUSED / mvuld_set_attn_bwd_fused, 1 (default) = on where it
// applies (mode 0, head_dim 32, ws % 4 == 0, ws <= 28), 0 = the three-pass backward (dQ, dK/dV, bias table).
static std::atomic<int> g_am_fused{-1};
extern "C" int mvuld_set_attn_bwd_fused(int on) {
    g_am_fused.store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}
static int am_fused_on() {
    int v = g_am_fused.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("MVULD_ATTN_BWD_FUSED");
        v = (!e || atoi(e) != 0) ? 1 : 0;
        g_am_fused.store(v, std::memory_order_relaxed);
    }
    return v;
}
/* 1 when mvuld_attn_bwd_mfma takes this geometry with the fused kernel: ONE call with passes = 3 then does dQ, dK, dV and the table gradient
 * (a call with passes = 2 alone is a no-op) */
extern "C" int mvuld_attn_bwd_fused_active(int mode, int hd, int ws) {
    return (mode == 0 && af_supported(hd, ws) && am_fused_on()) ? 1 : 0;
}
static void am_plan(AttnGeom& g, int64_t groups, int ntile, int& split, unsigned& grid) {
    split = am_split(groups, ntile);
    g.whole = 0;
    grid = (unsigned)(groups * split);
}

#define AM_LAUNCH(KERNEL, HDV, MODEV, bytes, ...)                                                \
    do {                                                                                          \
        if (am_set_lds(KERNEL<HDV, MODEV>, bytes, #KERNEL)) return 1;                             \
        hipLaunchKernelGGL((KERNEL<HDV, MODEV>), grid, dim3((HDV == 32 || MODEV == 1) ? 1024 : 512), bytes, stream, __VA_ARGS__);    \
    } while (0)

#define AM_DISPATCH(KERNEL, bytes, ...)                                      \
    do {                                                                      \
        if (hd == 32 && mode == 0) AM_LAUNCH(KERNEL, 32, 0, bytes, __VA_ARGS__);      \
        else if (hd == 64 && mode == 0) AM_LAUNCH(KERNEL, 64, 0, bytes, __VA_ARGS__); \
        else if (hd == 32) AM_LAUNCH(KERNEL, 32, 1, bytes, __VA_ARGS__);              \
        else AM_LAUNCH(KERNEL, 64, 1, bytes, __VA_ARGS__);                            \
    } while (0)

static void am_set_dropout(AttnGeom& g, int mode, float p, uint64_t seed, const uint64_t* seed_offset) {
    g.drop_thr = 0; g.drop_seed = 0; g.drop_inv = 1.f; g.drop_off = seed_offset;
    if (mode >= 1 && p > 0.f) {
        const unsigned thr15 = (unsigned)((double)p * 32768.0 + 0.5);       // a key is dropped when its 15-bit hash field < thr15
        if (thr15 == 0) return;                                             // p < 2^-16: no dropout
        g.drop_thr = (thr15 - 1) | ((thr15 - 1) << 16);
        g.drop_seed = (unsigned)(seed ^ (seed >> 32));
        g.drop_inv = 32768.0f / (float)(32768u - thr15);                    // 1 / (exact keep probability)
    }
}

extern "C" int mvuld_attn_fwd_mfma(int mode, int B, int H, int hd, int N, int nW, int res, int ws, int shift, float scale,
                                   const void* qkv, const float* table16, const float* logit_scale, const int* valid, void* out,
                                   float* lse, float attn_drop_p, uint64_t drop_seed, const uint64_t* seed_offset, const float* sample_scale, int dtype,
                                   hipStream_t stream) {
    if (am_check("attn_fwd_mfma", mode, B, H, hd, N, nW, res, ws, shift)) return 1;
    MV_CHECK_ARG(dtype == MVULD_BF16, "attn_fwd_mfma: bf16 storage only");
    MV_CHECK_ARG(qkv && out && lse && (mode >= 1 ? valid != nullptr : (table16 && logit_scale)), "attn_fwd_mfma: null pointer");
    MV_CHECK_ARG(attn_drop_p >= 0.f && attn_drop_p < 1.f && (attn_drop_p == 0.f || mode >= 1), "attn_fwd_mfma: attention dropout is a mode 1 / 2 feature, 0 <= p < 1");
    AttnGeom g{mode, B, H, N, nW, res, ws, shift, scale, nullptr, 0, 0, 0, 1.f, nullptr};
    g.yskip = am_yskip_on();
    g.sscale = mode == 0 ? sample_scale : nullptr;
    am_set_dropout(g, mode, attn_drop_p, drop_seed, seed_offset);
    if (mode == 2) { g.mode = 1; g.cu = valid; valid = nullptr; mode = 1; }      // packed sequences: `valid` carries cu_seqlens [B + 1]
    const int Npad = (N + 31) / 32 * 32;
    const int T2 = mode == 0 ? (2 * ws - 1) * (2 * ws - 1) : 0;
    const int ld = hd == 32 ? 32 : hd + 8;          // AmTile: swizzled 64-byte rows at hd = 32, padded rows at 64
    const size_t bytes = (size_t)2 * Npad * ld * 2 + (size_t)Npad * 4 + (size_t)T2 * 4;
    int qsplit;
    unsigned nwg;
    am_plan(g, (int64_t)B * nW * H, (N + 15) / 16, qsplit, nwg);
    dim3 grid(nwg);
    // 16 waves (4 per SIMD) hide the LDS / MFMA latencies of the score loop twice as well as 8; MVULD_ATTN_FWD_THREADS overrides
    static const int fwd_threads = [] { const char* e = getenv("MVULD_ATTN_FWD_THREADS"); const int v = e ? atoi(e) : 1024; return v > 0 ? v : 1024; }();
#define AM_FWD(HDV, MODEV, MASKV)                                                                              \
    do {                                                                                                        \
        if (am_set_lds(attn_fwd_mfma_k<HDV, MODEV, MASKV>, bytes, "attn_fwd_mfma_k")) return 1;                 \
        hipLaunchKernelGGL((attn_fwd_mfma_k<HDV, MODEV, MASKV>), grid, dim3(fwd_threads), bytes, stream, g, (const bf16*)qkv, table16, \
                           logit_scale, valid, (bf16*)out, lse, Npad, qsplit);                                  \
    } while (0)
    const int S3 = aw_stride_bytes(T2);
    const size_t bytes_win = (size_t)2 * Npad * 64 + (size_t)Npad * 4 + (size_t)((Npad + 63) / 64) * 64 + (size_t)3 * S3;
#define AM_FWD_WIN(MASKV, LZ)                                                                                   \
    do {                                                                                                        \
        if (am_set_lds(attn_fwd_win_k<MASKV, LZ>, bytes_win, "attn_fwd_win_k")) return 1;                       \
        hipLaunchKernelGGL((attn_fwd_win_k<MASKV, LZ>), grid, dim3(1024), bytes_win, stream, g, (const bf16*)qkv, table16, logit_scale, \
                           (bf16*)out, lse, Npad, qsplit, S3);                                                  \
    } while (0)
    const int win = am_win_mode();
    if (mode == 0 && hd == 32 && (ws & 3) == 0 && bytes_win <= 160 * 1024 && win == 1) { if (shift > 0) AM_FWD_WIN(true, true); else AM_FWD_WIN(false, true); }
    else if (mode == 0 && hd == 32 && (ws & 3) == 0 && bytes_win <= 160 * 1024 && win == 2) { if (shift > 0) AM_FWD_WIN(true, false); else AM_FWD_WIN(false, false); }
    else if (mode == 0 && hd == 32) { if (shift > 0) AM_FWD(32, 0, true); else AM_FWD(32, 0, false); }
    else if (mode == 0) { if (shift > 0) AM_FWD(64, 0, true); else AM_FWD(64, 0, false); }
    else if (hd == 32) AM_FWD(32, 1, false);
    else AM_FWD(64, 1, false);
    MV_LAUNCH_CHECK("attn_fwd_mfma");
    return 0;
}

// workspaces (caller-owned): delta [tokens * H] fp32; qt [tokens, H*hd] bf16 (MODE 0: normalised, scaled queries).  dqkv is fully written.
/* bytes of fp32 scratch (ws_part) the bias-table gradient pass of mvuld_attn_bwd_mfma wants: one (2ws-1)^2 table per workgroup */
extern "C" int64_t mvuld_attn_bwd_mfma_workspace_bytes(int mode, int B, int H, int nW, int ws) {
    if (mode != 0) return 0;
    const int64_t groups = (int64_t)B * nW * H;
    const int64_t T2 = (int64_t)(2 * ws - 1) * (2 * ws - 1);
    return (groups > 512 ? groups : 512) * T2 * 4;          // the pass splits small grids up to < 512 workgroups
}
extern "C" int mvuld_attn_bwd_mfma(int mode, int B, int H, int hd, int N, int nW, int res, int ws, int shift, float scale,
                                   const void* qkv, const float* table16, const float* logit_scale, const int* valid,
                                   const void* out, const void* dout, const float* lse, void* dqkv, float* dtable16,
                                   float* dlogit_scale, float* ws_delta, void* ws_qt, float* ws_part, int64_t ws_part_bytes, int passes,
                                   float attn_drop_p, uint64_t drop_seed, const uint64_t* seed_offset, const float* sample_scale, int dtype,
                                   hipStream_t stream) {
    if (am_check("attn_bwd_mfma", mode, B, H, hd, N, nW, res, ws, shift)) return 1;
    MV_CHECK_ARG(passes >= 1 && passes <= 3, "attn_bwd_mfma: passes is a mask of 1 (delta + dQ + dK/dV) and 2 (bias-table gradient)");
    MV_CHECK_ARG(dtype == MVULD_BF16, "attn_bwd_mfma: bf16 storage only");
    MV_CHECK_ARG(qkv && out && dout && lse && dqkv && ws_delta, "attn_bwd_mfma: null pointer");
    MV_CHECK_ARG(mode >= 1 ? valid != nullptr : (table16 && logit_scale && dtable16 && dlogit_scale && ws_qt), "attn_bwd_mfma: null pointer");
    MV_CHECK_ARG(attn_drop_p >= 0.f && attn_drop_p < 1.f && (attn_drop_p == 0.f || mode >= 1), "attn_bwd_mfma: attention dropout is a mode 1 / 2 feature, 0 <= p < 1");
    AttnGeom g{mode, B, H, N, nW, res, ws, shift, scale, nullptr, 0, 0, 0, 1.f, nullptr};
    g.yskip = am_yskip_on();
    g.sscale = mode == 0 ? sample_scale : nullptr;
    am_set_dropout(g, mode, attn_drop_p, drop_seed, seed_offset);
    int64_t ntok = (int64_t)B * nW * N;
    if (mode == 2) { g.mode = 1; g.cu = valid; valid = nullptr; mode = 1; ntok = res; }
    const int Npad = (N + 31) / 32 * 32;
    const int T2 = mode == 0 ? (2 * ws - 1) * (2 * ws - 1) : 0;
    (void)ntok;          // delta = rowsum(dO o O) is computed (and written to ws_delta) by the dQ kernel
    if (mvuld_attn_bwd_fused_active(mode, hd, ws)) {
        if (!(passes & 1)) return 0;                     // the table gradient came with the fused pass
        const int64_t groups = (int64_t)B * nW * H;
        MV_CHECK_ARG(ws_part && ws_part_bytes >= groups * T2 * 4, "attn_bwd_mfma: the fused backward needs ws_part of %lld bytes (mvuld_attn_bwd_mfma_workspace_bytes)",
                     (long long)(groups * T2 * 4));
        if (af_launch(g, shift, groups, qkv, table16, logit_scale, out, dout, lse, dqkv, ws_part, dlogit_scale, stream)) return 1;
        hipLaunchKernelGGL(attn_dbias_reduce_k, dim3(cdiv(T2, 256), H, max(1, min(16, B * nW / 8))), dim3(256), 0, stream, ws_part, dtable16, T2, H,
                           B * nW, 1);
        MV_LAUNCH_CHECK("attn_bwd_mfma(fused)");
        return 0;
    }
    int split;
    unsigned nwg;
    am_plan(g, (int64_t)B * nW * H, (N + 15) / 16, split, nwg);
    dim3 grid(nwg);
    const int ld = hd == 32 ? 32 : hd + 8;
    const size_t bytes_q = (size_t)2 * Npad * ld * 2 + (size_t)Npad * 4 + 64 + (size_t)T2 * 4;
    const size_t bytes_k = (size_t)2 * Npad * ld * 2 + (size_t)Npad * 12 + (size_t)T2 * 4;
#define AM_BWD(HDV, MODEV, MASKV)                                                                                       \
    do {                                                                                                                 \
        if (am_set_lds(attn_bwd_dq_mfma_k<HDV, MODEV, MASKV>, bytes_q, "attn_bwd_dq_mfma_k")) return 1;                  \
        hipLaunchKernelGGL((attn_bwd_dq_mfma_k<HDV, MODEV, MASKV>), grid, dim3((HDV == 32 || MODEV == 1) ? 1024 : 512), bytes_q, stream, g, \
                           (const bf16*)qkv, table16, logit_scale, valid, (const bf16*)dout, lse, ws_delta, (bf16*)dqkv,  \
                           dlogit_scale, (bf16*)ws_qt, Npad, split, (const bf16*)out);                                                \
        if (am_set_lds(attn_bwd_dkv_mfma_k<HDV, MODEV, MASKV>, bytes_k, "attn_bwd_dkv_mfma_k")) return 1;                 \
        hipLaunchKernelGGL((attn_bwd_dkv_mfma_k<HDV, MODEV, MASKV>), grid, dim3((HDV == 32 || MODEV == 1) ? 1024 : 512), bytes_k, stream, g, \
                           (const bf16*)qkv, table16, logit_scale, valid, (const bf16*)dout, lse, ws_delta, (bf16*)dqkv,  \
                           Npad, split);                                                                                  \
    } while (0)
    if (passes & 1) {
        if (mode == 0 && hd == 32) { if (shift > 0) AM_BWD(32, 0, true); else AM_BWD(32, 0, false); }
        else if (mode == 0) { if (shift > 0) AM_BWD(64, 0, true); else AM_BWD(64, 0, false); }
        else if (hd == 32) AM_BWD(32, 1, false);
        else AM_BWD(64, 1, false);
    }
    if (mode == 0 && (passes & 2)) {
        MV_CHECK_ARG(hd == 32 && ws <= 32, "attn_bwd_mfma: the bias-table gradient pass covers head_dim 32 and windows up to 32x32");
        const size_t bytes = (size_t)2 * Npad * (hd + 8) * 2 + (size_t)Npad * 4 + (size_t)2 * T2 * 4;
        // dy per work item: 4 (ws = 28 -> 7 groups x 2 q parts = 14 items on 8 waves, 250 VGPRs) or 2 (28 items on 16 waves, 128 VGPRs)
        static const int dbias_g = [] { const char* e = getenv("MVULD_ATTN_DBIAS_G"); return e ? atoi(e) : 4; }();
        const int DG = dbias_g == 2 ? 2 : 4;
        const int items = ((ws + DG - 1) / DG) * ((ws + 15) / 16);
        int sp = 1;
        while ((int64_t)B * nW * H * sp < 256 && sp * 2 <= items) sp *= 2;
        float* part = (ws_part && ws_part_bytes >= (int64_t)B * nW * H * sp * T2 * 4) ? ws_part : nullptr;
        // waves per workgroup = work items per workgroup (ws = 14: 4 items; 8 waves would leave half of them idle and the
        // small tile footprint lets several workgroups share a CU instead)
        const int per_wg = (items + sp - 1) / sp;
#define AM_DBIAS(GV, KTV, MAXW)                                                                                          \
    do {                                                                                                                 \
        if (am_set_lds(attn_bwd_dbias_mfma_k<32, GV, KTV>, bytes, "attn_bwd_dbias_mfma_k")) return 1;                    \
        hipLaunchKernelGGL((attn_bwd_dbias_mfma_k<32, GV, KTV>), dim3(B * nW * H * sp), dim3(64 * max(1, min(MAXW, per_wg))), bytes, \
                           stream, g, (const bf16*)qkv, (const bf16*)ws_qt, table16, (const bf16*)dout, lse, ws_delta, dtable16, \
                           part, Npad, sp);                                                                              \
    } while (0)
        if (DG == 4) { if (ws <= 16) AM_DBIAS(4, 1, 8); else AM_DBIAS(4, 2, 8); }
        else { if (ws <= 16) AM_DBIAS(2, 1, 16); else AM_DBIAS(2, 2, 16); }
#undef AM_DBIAS
        if (part)
            hipLaunchKernelGGL(attn_dbias_reduce_k, dim3(cdiv(T2, 256), H, max(1, min(16, B * nW * sp / 8))), dim3(256), 0, stream, part, dtable16, T2, H,
                               B * nW, sp);
    }
    MV_LAUNCH_CHECK("attn_bwd_mfma");
    return 0;
}
