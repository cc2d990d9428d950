// Shared by the attention translation units (attention_mfma.hip, attention_bwd_fused.hip): the geometry descriptor, the roll / partition
// index map, mask regions, dropout hash and small vector helpers.  gfx950 only.
#pragma once
#include "common.h"
#include <stdlib.h>
#include <atomic>

struct AttnGeom {
    int mode, B, H, N, nW, res, ws, shift;
    float scale;
    const int* cu;          // MODE 1, packed (varlen) sequences: cu[b] .. cu[b+1] are the token rows of sequence b; null = dense [B, N]
    int64_t tok0;           // MODE 1: first token row of the workgroup's sequence (set inside the kernels)
    // MODE 1, attention-probability dropout (HF attention_probs_dropout_prob): keep(b,h,q,k) = hash(seed, element) >= thr, kept
    // probabilities scaled by inv = 1/(1-p); thr = 0 switches it off.  Counter-based: the backward passes regenerate the mask.
    unsigned drop_thr, drop_seed;
    float drop_inv;
    const uint64_t* drop_off;   // optional device-resident step counter mixed into the seed (hipGraph replays draw fresh masks)
    // Tail balancing (am_plan): the first `whole` workgroups (in launch order) take one (window, head) each, the workgroups behind them
    // split the remaining ones `split` ways; whole = 0: every (window, head) is split `split` ways.
    int whole = 0;
    // MODE 0, shifted windows: skip the key blocks (query blocks) that lie wholly across a window's vertical mask split from the wave's
    // queries (keys) -- every pair in them carries the -100 of swin_transformer_v2.py:245-268 (am_ysplit); 0 = compute them as every other pair
    int yskip = 0;
    // MODE 0, optional [B] per-sample scale of the residual branch this attention belongs to (DropPath, swin_transformer_v2.py:301): a sample
    // whose scale is exactly 0 contributes nothing downstream (forward: its output is multiplied by 0; backward: its d(out) IS 0), so its
    // workgroups write zeros and return instead of computing them
    const float* sscale = nullptr;
};
__device__ __forceinline__ bool am_dropped(const AttnGeom& g, int b) { return g.mode == 0 && g.sscale != nullptr && g.sscale[b] == 0.f; }
__device__ __forceinline__ unsigned am_seed(const AttnGeom& g) {
    return g.drop_off ? g.drop_seed ^ (unsigned)(g.drop_off[0] * 0x9E3779B97F4A7C15ULL >> 32) : g.drop_seed;
}
// Round 3: ONE 32-bit hash serves the two keys of an aligned pair (2j, 2j+1) of a query row -- the counter is row * ceil(N/2) + j, the
// even key takes bits 0..14, the odd key bits 16..30, and a key is kept when its 15-bit field >= thr15 = round(p * 2^15) (keep
// probability exactly 1 - thr15 / 2^15, the scale is its reciprocal).  The two 32-bit multiplies of the hash were most of the text
// encoder's attention VALU work (301 vs 108 vector instructions per 64-key forward block with / without dropout).  drop_thr carries
// (thr15 - 1) in both halves: the packed 16-bit subtraction (thr15 - 1) - field is negative exactly where the key is kept, and its
// sign, smeared over the half by a packed arithmetic shift, is the keep mask of the packed bf16 probability pair.
__device__ __forceinline__ unsigned am_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
typedef short __attribute__((ext_vector_type(2))) s16x2_t;
__device__ __forceinline__ unsigned am_keep2(unsigned pairctr, unsigned seed, unsigned thr2m1) {          // 0xFFFF in the halves that are kept
    const unsigned f = am_hash(pairctr ^ seed) & 0x7FFF7FFFu;
    const s16x2_t d = __builtin_bit_cast(s16x2_t, thr2m1) - __builtin_bit_cast(s16x2_t, f);
    return __builtin_bit_cast(unsigned, d >> (s16x2_t){15, 15});
}
// quad_perm [1, 0, 3, 2]: the value of the lane's neighbour (lane ^ 1)
__device__ __forceinline__ unsigned am_swap1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); }

typedef bf16 __attribute__((ext_vector_type(8))) bf16x8_t;
typedef bf16 __attribute__((ext_vector_type(4))) bf16x4_t;
typedef float __attribute__((ext_vector_type(4))) f32x4_t;
// Explicit two-wide fp32 math for the softmax fix-ups: left to itself the SLP vectoriser pairs elements (1,2),(3,4).. of an
// accumulator quad, and every v_pk_mul_f32 then costs two v_mov to build its operand pair plus v_alignbit / v_perm to re-pack the
// bf16 fragment (28 of the 98 VALU instructions of a 64-key dQ block).  Register pairs (0,1),(2,3) of an MFMA result are aligned.
typedef float __attribute__((ext_vector_type(2))) f32x2_t;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4_t;
typedef bf16 __attribute__((ext_vector_type(2))) bf16x2_t;
__device__ __forceinline__ unsigned am_pk(f32x2_t v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)); }
#ifndef AM_X
#define AM_X 0      // timing experiments (tools/attn_variants.sh): 1 = no exp, 2 = no bias reads, 3 = no transposed reads; bias-table pass: 4 = K/V fragments read once per item, 5 = q-side rows fetched once per item
#endif
__device__ __forceinline__ f32x2_t am_exp2(f32x2_t v) {
#if AM_X == 1
    return v * 0.001f;
#else
    return (f32x2_t){__builtin_amdgcn_exp2f(v[0]), __builtin_amdgcn_exp2f(v[1])};
#endif
}

#define LN100 4.605170185988092f
#define NEG_BIG -1.0e30f

__device__ __forceinline__ int64_t am_token(const AttnGeom& g, int b, int w, int n) {
    if (g.mode == 1) return g.tok0 + n;
    const int nwx = g.res / g.ws;
    const int sy = (w / nwx) * g.ws + n / g.ws, sx = (w % nwx) * g.ws + n % g.ws;
    int oy = sy + g.shift, ox = sx + g.shift;
    if (oy >= g.res) oy -= g.res;
    if (ox >= g.res) ox -= g.res;
    return ((int64_t)b * g.res + oy) * g.res + ox;
}
__device__ __forceinline__ int am_rid(const AttnGeom& g, int s) { return s < g.res - g.ws ? 0 : (s < g.res - g.shift ? 1 : 2); }
// A shifted window mixes mask regions only in the last row / column of windows (where the rolled image wraps); every other window
// is one region (id 0) and takes the unmasked loops: 9 of the 16 windows of stage 0, 1 of the 4 of stage 1.
__device__ __forceinline__ bool am_window_masked(const AttnGeom& g, int w) {
    const int nwx = g.res / g.ws;
    return g.shift > 0 && ((w / nwx) == nwx - 1 || (w % nwx) == nwx - 1);
}
// The last ROW of windows of a shifted block holds two vertical mask regions: window rows below ws - shift come from the bottom of the
// image, the rest from its (rolled-in) top, and the -100 on every pair across the split leaves them exp2(-144 + (s - m)) of the row's
// largest probability.  With cosine logits |q.k| <= tau and a bias in (0, 16) that is below 2^-57 for tau <= 22: under the fp32
// resolution of every accumulator it would be added to, so whole tiles of such pairs are skipped -- same bits out -- while a head whose tau
// has grown past the bound keeps computing them (as the reference's finite -100 demands).  Returns the split as a token index of the
// window's row-major order, or INT_MAX (no split in this window / tau too large / switched off).  The horizontal split of the last
// COLUMN of windows cannot be skipped tile-wise: every 16-token tile of a row-major window holds tokens of both of its sides.
#define AM_YSKIP_TAU 22.0f
__device__ __forceinline__ int am_ysplit(const AttnGeom& g, int w, float tau) {
    const int nwx = g.res / g.ws;
    if (g.mode != 0 || g.shift <= 0 || !g.yskip || !(tau <= AM_YSKIP_TAU) || (w / nwx) != nwx - 1) return 0x7fffffff;
    return (g.ws - g.shift) * g.ws;
}
// query tokens [q0, q1) and key tokens [k0, k1) on opposite sides of the split
__device__ __forceinline__ bool am_yskip(int ys, int q0, int q1, int k0, int k1) { return (q1 <= ys && k0 >= ys) || (q0 >= ys && k1 <= ys); }
// per-token info word.  MODE 0: (iy*(2ws-1)+ix) | region << 16 ; MODE 1: validity in bit 0.  Bit 30 marks a padding row
// (all other fields then hold safe in-range values, so the hot loops stay branch-free).
#define AM_PAD (1 << 30)
__device__ __forceinline__ int am_info(const AttnGeom& g, const int* __restrict__ valid, int b, int w, int n) {
    if (n >= g.N) return AM_PAD;
    if (g.mode == 1) return (valid == nullptr || valid[g.tok0 + n]) ? 1 : 0;
    const int nwx = g.res / g.ws;
    const int iy = n / g.ws, ix = n % g.ws;
    int reg = 0;
    if (g.shift > 0) reg = am_rid(g, (w / nwx) * g.ws + iy) * 3 + am_rid(g, (w % nwx) * g.ws + ix);
    return (iy * (2 * g.ws - 1) + ix) | (reg << 16);
}

// MODE 1: point the geometry at the workgroup's sequence.  Dense: rows b*N .. b*N+N-1.  Packed: rows cu[b] .. cu[b+1]-1, g.N becomes
// the sequence's own length (<= the launch's N, which sizes LDS and the lse rows).  Returns the extent (multiple of 32) the
// staging and key / query loops run over; 0 = empty sequence.
__device__ __forceinline__ int am_localize(AttnGeom& g, int b, int Npad) {
    if (g.mode != 1) return Npad;
    if (g.cu == nullptr) { g.tok0 = (int64_t)b * g.N; return Npad; }
    g.tok0 = g.cu[b];
    g.N = g.cu[b + 1] - g.cu[b];
    return min(Npad, (g.N + 31) / 32 * 32);
}

// reductions over the 4 lanes that share (lane & 15): lanes l, l^16, l^32, l^48 -- VALU only (v_permlane16/32_swap)
__device__ __forceinline__ float sum4g(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float max4g(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

union U8 { uint4 u; bf16x8_t v; bf16 e[8]; };
union U4 { uint2 u; bf16x4_t v; bf16 e[4]; };

// Stage `rows` token rows (HD wide, from column `coloff` of a [tokens, rowstride] matrix) starting at n0 into the
// row-major LDS image rm[rows][HD+8], with optional L2 normalisation and scale.  Rows >= N are zero.  Four 16-byte loads
// are kept in flight per thread.
// XCD-aware order: workgroups b, b+8, ... share an XCD and its L2.  Giving each XCD a contiguous run of (window, head, part)
// items keeps the heads of one window -- which read interleaved 64/128-byte column slices of the same qkv rows -- on one L2.
__device__ __forceinline__ int am_xcd_order(int bid, int total) {
    const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// (launch-order block index) -> (window x head index, part, parts): XCD-contiguous order inside the unsplit and inside the split range
__device__ __forceinline__ void am_part(const AttnGeom& g, int split, int& bwh, int& part, int& parts) {
    const int bx = blockIdx.x, total = gridDim.x;
    if (g.whole > 0) {
        if (bx < g.whole) { bwh = am_xcd_order(bx, g.whole); part = 0; parts = 1; return; }
        const int r = am_xcd_order(bx - g.whole, total - g.whole);
        bwh = g.whole + r / split; part = r % split; parts = split;
        return;
    }
    const int bid = am_xcd_order(bx, total);
    part = bid % split; bwh = bid / split; parts = split;
}

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// attention_bwd_fused.hip: the fused single-pass window backward (mode 0, head_dim 32, ws % 4 == 0, ws <= 28)
int af_supported(int hd, int ws);
int af_launch(const AttnGeom& g, int shift, int64_t groups, const void* qkv, const float* table16, const float* logit_scale, const void* out,
              const void* dout, const float* lse, void* dqkv, float* ws_part, float* dlogit_scale, hipStream_t stream);
