// GEMM family of libmvuld_hip.so:  C[b] = epilogue(alpha * A[b] . B[b]^T)   ("NT": both operands K-contiguous)
//
//   * gemm_nt_mfma_bf16 : bf16 operands, fp32 accumulate on the CDNA4 matrix cores
//     (v_mfma_f32_16x16x32_bf16), 128x128x64 tiles, 4 waves (2x2, 64x64 per wave),
//     register-staged double-buffered LDS with an XOR chunk swizzle, LDS-staged
//     row-major epilogue (bias / GELU / ELU / activation-derivative / split-K atomics).
//   * gemm_nt_simple    : any (f32|bf16) operands, VALU fp32 FMA, 64x64x16 tiles.  The
//     fp32 parity path and the odd-shaped GEMMs of the head (K or ld not 16-byte friendly).
//
// Every Linear / Conv1d(k=1) / Conv2d(4x4,s4) / batched bmm of the hot path lands here:
//   Swin qkv/proj/fc1/fc2/reduction/patch-embed (swin_transformer_v2.py:150,177,27-30,361,490),
//   RoBERTa q/k/v/out/intermediate/output dense, GATConv fc, head Linear layers
//   (GraphModel.py:153-209) and Rs_GCN's 1x1 convs + theta^T.phi / R.g (Rs_GCN.py:57-70).
#include "common.h"
#include <stdlib.h>

#include "gemm_common.h"
#include <atomic>

template <typename TO>
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, TO* C, TO* aux, int row, int col, float acc) {
    float v = g.alpha * acc;
    if (g.bias) v += g.bias[col];
    const int64_t ci = (int64_t)row * g.ldc + col;
    const int64_t ai = (int64_t)row * g.ldaux + col;
    switch (g.epi) {
        case EPI_GELU:
            if (aux) stf(aux + ai, v);
            v = gelu_erf(v);
            break;
        case EPI_GELU_DG:
            if (aux) stf(aux + ai, dgelu_erf(v));
            v = gelu_erf(v);
            break;
        case EPI_MUL_AUX: v *= ldf(aux + ai); break;
        case EPI_ELU: v = elu1(v); break;
        case EPI_MUL_DGELU: v *= dgelu_erf(ldf(aux + ai)); break;
        case EPI_MUL_DELU: { const float y = ldf(aux + ai); v *= (y > 0.f ? 1.0f : y + 1.0f); } break;
        case EPI_ADD_AUX: v += ldf(aux + ai); break;
        default: break;
    }
    if (g.out_mode == OUT_ATOMIC) {
        if constexpr (sizeof(TO) == 4) atomicAdd((float*)(C + ci), v);
    } else if (g.out_mode == OUT_ACCUM) {
        stf(C + ci, ldf(C + ci) + v);
    } else {
        stf(C + ci, v);
    }
}

// ------------------------------------------------------------------------------------ simple
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void gemm_nt_simple(GemmArgs g) {
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    const int bz = blockIdx.z;
    const int b = bz / g.splitk, ks = bz % g.splitk;
    const TI* A = (const TI*)g.A + (int64_t)b * g.sA;
    const TI* B = (const TI*)g.B + (int64_t)b * g.sB;
    TO* C = (TO*)g.C + (int64_t)b * g.sC;
    TO* aux = g.aux ? (TO*)g.aux + (int64_t)b * g.sAux : nullptr;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int lr = t >> 2, lk = (t & 3) * 4;
    const int ktiles = (g.K + 15) / 16;
    const int per = (ktiles + g.splitk - 1) / g.splitk;
    const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);
    float acc[4][4] = {};
    for (int kt = kt0; kt < kt1; ++kt) {
        const int k0 = kt * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + lk + j;
            const int ra = m0 + lr, rb = n0 + lr;
            As[lk + j][lr] = (ra < g.M && k < g.K) ? ldf(A + (int64_t)ra * g.lda + k) : 0.f;
            Bs[lk + j][lr] = (rb < g.N && k < g.K) ? ldf(B + (int64_t)rb * g.ldb + k) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], bb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; bb[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], bb[j], acc[i][j]);
        }
        __syncthreads();
    }
    if (kt0 >= kt1 && g.out_mode == OUT_ATOMIC) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = m0 + ty * 4 + i, c = n0 + tx * 4 + j;
            if (r < g.M && c < g.N) epilogue_store<TO>(g, C, aux, r, c, acc[i][j]);
        }
}

// ------------------------------------------------------------------------------------ MFMA bf16
#define GT_BM 128
#define GT_BN 128
#define GT_BK 64
#define GT_EPI_LD 68
// LDS: NBUF x {A,B} tiles of 16 KiB, and the epilogue's per-wave fp32 staging done in 2/NBUF... passes:
//   NBUF = 2: 64 KiB main, 4 x 64 x 68 x 4 = 69632 B epilogue  -> 2 workgroups / CU
//   NBUF = 1: 32 KiB main, epilogue in two 32-row halves (34816 B) -> 3 workgroups / CU (VGPR-limited): more tiles in
//             flight per CU, which is what the short-K shapes of this model need (the loop is L2-latency bound)
#define GT_LDS_BYTES_N(NBUF) ((NBUF) == 2 ? 69632 : 34816)

__device__ __forceinline__ int swz_off(int row, int chunk) {        // byte offset inside a [128][64] bf16 tile
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <typename TO, int NBUF, bool GLDS = false>
__global__ __launch_bounds__(256, NBUF == 2 ? 2 : 3) void gemm_nt_mfma_bf16(GemmArgs g, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;

    // XCD-aware tile order: blocks b, b+8, ... share an XCD (and its L2); give each XCD a
    // contiguous run of tiles, walked N-fastest so neighbours reuse the same A row panel.
    const int nt = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nt >> 3, r = nt & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * GT_BM, n0 = tn * GT_BN;
    const int bz = blockIdx.y;
    const int b = bz / g.splitk, ks = bz % g.splitk;
    const bf16* A = (const bf16*)g.A + (int64_t)b * g.sA;
    const bf16* B = (const bf16*)g.B + (int64_t)b * g.sB;
    TO* C = (TO*)g.C + (int64_t)b * g.sC;
    TO* aux = g.aux ? (TO*)g.aux + (int64_t)b * g.sAux : nullptr;

    const int ktiles = (g.K + GT_BK - 1) / GT_BK;
    const int per = (ktiles + g.splitk - 1) / g.splitk;
    const int kt0 = ks * per, kt1 = min(ktiles, kt0 + per);
    if (kt0 >= kt1 && g.out_mode == OUT_ATOMIC) return;

    // staging map: 1024 16-byte chunks per operand tile, 4 per thread
    int s_row[4], s_ch[4];
    const bf16* a_ptr[4];
    const bf16* b_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        s_row[i] = c >> 3; s_ch[i] = c & 7;
        a_ptr[i] = A + (int64_t)min(m0 + s_row[i], g.M - 1) * g.lda;
        b_ptr[i] = B + (int64_t)min(n0 + s_row[i], g.N - 1) * g.ldb;
    }
    u32x4_t ra[4], rb[4];
    // Loads are unconditional on clamped addresses and zeroed by a select afterwards: a predicated load makes the
    // compiler branch around it and drain vmcnt, which serialises the eight round trips of a k-step.
    // The zeroing select sits in stage_write, behind an empty asm that pins the first use of the loaded registers after the
    // MFMA block -- otherwise the scheduler hoists the select (and the vmcnt wait it needs) in front of the MFMAs.
    int k_loaded = 0;
    auto stage_load = [&](int kt) {
        const int k0 = kt * GT_BK;
        k_loaded = k0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = k0 + s_ch[i] * 8;
            const int kc = kk < g.K ? kk : 0;
            ra[i] = *(const u32x4_t*)(a_ptr[i] + kc);
            rb[i] = *(const u32x4_t*)(b_ptr[i] + kc);
        }
    };
    auto stage_write = [&](int buf) {
        char* sa = smem + buf * 32768;
        char* sb = sa + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4_t ta = ra[i], tb = rb[i];
            asm volatile("" : "+v"(ta), "+v"(tb));
            const bool ok = (k_loaded + s_ch[i] * 8) < g.K;
            const int off = swz_off(s_row[i], s_ch[i]);
            *(u32x4_t*)(sa + off) = ok ? ta : (u32x4_t){0u, 0u, 0u, 0u};
            *(u32x4_t*)(sb + off) = ok ? tb : (u32x4_t){0u, 0u, 0u, 0u};
        }
    };

    // GLDS (K % 64 == 0, double buffer): the tile goes global -> LDS by LDS-DMA, no VGPR staging and no ds_write.  One
    // instruction fills 1 KiB = 8 tile rows in lane order, so lane l fetches the chunk that the XOR swizzle stores in slot l & 7.
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* glb_vp;
    const bf16* ga_ptr[4];
    const bf16* gb_ptr[4];
    if (GLDS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ ((row >> 1) & 7);
            ga_ptr[i] = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + ch * 8;
            gb_ptr[i] = B + (int64_t)min(n0 + row, g.N - 1) * g.ldb + ch * 8;
        }
    }
    auto stage_glds = [&](int kt, int buf) {
        const int k0 = kt * GT_BK;
        char* sa = smem + buf * 32768;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_vp)(ga_ptr[i] + k0), (lds_vp)(sa + (wave * 4 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)(gb_ptr[i] + k0), (lds_vp)(sa + 16384 + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    if (kt0 < kt1) {
        if (GLDS) {
            stage_glds(kt0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            stage_load(kt0);
            if (NBUF == 2) stage_write(0);
        }
    }
    if (NBUF == 2) __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = (kt + 1) < kt1;
        if (NBUF == 1) {
            __syncthreads();                 // every wave is done reading the previous tile
            stage_write(0);
            __syncthreads();
        }
        if (more) { if (GLDS) stage_glds(kt + 1, cur ^ 1); else stage_load(kt + 1); }
        const char* sa = smem + cur * 32768;
        const char* sb = sa + 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *(const bf16x8_t*)(sa + swz_off(wr * 64 + i * 16 + fr, kk * 4 + fg));
                fb[i] = *(const bf16x8_t*)(sb + swz_off(wc * 64 + i * 16 + fr, kk * 4 + fg));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (NBUF == 2) {
            if (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA of tile kt+1 has landed
            else if (more) stage_write(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue: accumulators -> per-wave LDS tile -> row-major 16-byte-per-lane pass  (EPH halves of 64/EPH rows)
    constexpr int EPH = NBUF == 2 ? 1 : 2;
    constexpr int EROWS = 64 / EPH;
    float* ep = (float*)(smem + wave * (EROWS * GT_EPI_LD * 4));
#pragma unroll
  for (int half = 0; half < EPH; ++half) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4 / EPH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) ep[(i * 16 + fg * 4 + r) * GT_EPI_LD + j * 16 + fr] = acc[half * (4 / EPH) + i][j][r];
    __syncthreads();
    // row-major pass: 8 lanes x 8 columns cover a 64-wide row, 8 rows per pass, 16-byte (bf16) / 32-byte (f32) stores
    const int lc = (lane & 7) * 8;
    const int col = n0 + wc * 64 + lc;
    const bool vec_ok = (g.out_mode == OUT_STORE) && (g.N % 8 == 0) && (g.ldc % 8 == 0) && (g.epi < EPI_GELU || (g.ldaux % 8 == 0));
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
    if (g.bias) {                           // wave-uniform; the loads inside are unconditional on clamped columns
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = g.bias[min(col + e, g.N - 1)];
            bias8[e] = (col + e < g.N) ? t : 0.f;
        }
    }
#pragma unroll 2
    for (int p = 0; p < 8 / EPH; ++p) {
        const int lr = p * 8 + (lane >> 3);
        const int row = m0 + wr * 64 + half * EROWS + lr;
        const bool valid = row < g.M && col < g.N;
        const float4 va = *(const float4*)(ep + lr * GT_EPI_LD + lc);
        const float4 vb = *(const float4*)(ep + lr * GT_EPI_LD + lc + 4);
        float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
        if (vec_ok) {
            float ax[8];
            if (epi_reads_aux(g.epi)) {
                const TO* ap = aux + (int64_t)min(row, g.M - 1) * g.ldaux + (col < g.N ? col : 0);
                if constexpr (sizeof(TO) == 4) {
                    const float4 a0 = *(const float4*)ap, a1 = *(const float4*)(ap + 4);
                    ax[0] = a0.x; ax[1] = a0.y; ax[2] = a0.z; ax[3] = a0.w; ax[4] = a1.x; ax[5] = a1.y; ax[6] = a1.z; ax[7] = a1.w;
                } else {
                    const bf16x8 a = *(const bf16x8*)ap;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ax[e] = (float)a.v[e];
                }
            }
            float pre[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = g.alpha * v[e] + bias8[e];
                pre[e] = x;
                switch (g.epi) {
                    case EPI_GELU: x = gelu_t<TO>(x); break;
                    case EPI_GELU_DG: gelu_dgelu_t<TO>(x, x, pre[e]); break;      // aux <- gelu'(pre-activation)
                    case EPI_MUL_AUX: x *= ax[e]; break;
                    case EPI_ELU: x = elu1(x); break;
                    case EPI_MUL_DGELU: x *= dgelu_t<TO>(ax[e]); break;
                    case EPI_MUL_DELU: x *= (ax[e] > 0.f ? 1.0f : ax[e] + 1.0f); break;
                    case EPI_ADD_AUX: x += ax[e]; break;
                    default: break;
                }
                v[e] = x;
            }
            if (!valid) continue;
            TO* cp = C + (int64_t)row * g.ldc + col;
            if constexpr (sizeof(TO) == 4) {
                *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                if (epi_writes_aux(g.epi) && aux) {
                    float* qp = (float*)(aux + (int64_t)row * g.ldaux + col);
                    *(float4*)qp = make_float4(pre[0], pre[1], pre[2], pre[3]);
                    *(float4*)(qp + 4) = make_float4(pre[4], pre[5], pre[6], pre[7]);
                }
            } else {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.v[e] = (bf16)v[e];
                *(bf16x8*)cp = o;
                if (epi_writes_aux(g.epi) && aux) {
                    bf16x8 q;
#pragma unroll
                    for (int e = 0; e < 8; ++e) q.v[e] = (bf16)pre[e];
                    *(bf16x8*)(aux + (int64_t)row * g.ldaux + col) = q;
                }
            }
        } else if (valid) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (col + e < g.N) epilogue_store<TO>(g, C, aux, row, col + e, v[e]);
        }
    }
  }
}

// ------------------------------------------------------------------------------------ MFMA bf16, 256 x 256 tile, 4-stage LDS-DMA ring
// The 128^2 kernel above moves (128 + 128) x 64 operand elements from L2 per 128 x 128 x 64 MACs (64 FLOP per L2 byte) and tops
// out at the L2 -> CU bandwidth (~12 TB/s: halving its MFMAs, its LDS reads or its LDS writes each buys < 10 %).  This one is
// 128 FLOP per L2 byte: 8 waves (2 x 4), each 128 x 64 (8 x 4 accumulator fragments, 12 ds_read_b128 per 32 MFMAs), operands
// streamed global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, no VGPR staging) into a ring of four 32-deep stages, three
// tiles in flight behind counted `s_waitcnt vmcnt(N)` and ONE barrier per k-step.  128 KiB LDS -> one workgroup per CU.
// Needs K % 32 == 0; M / N tails by row clamping (discarded at the store).
#define G2_BM 256
#define G2_BN 256
#define G2_BK 32
#define G2_STAGE_BYTES 32768        // A 256 x 32 bf16 (16 KiB) + B 256 x 32 bf16 (16 KiB)
#define G2_LDS_BYTES (4 * G2_STAGE_BYTES)

// byte offset of 16-byte chunk `ch` (0..3) of row `row` inside a [256][32] bf16 tile: 64-byte rows, the chunk slot XORed with a
// function of the row group so that every 16-lane group of a ds_read_b128 fragment read covers all 64 banks
__device__ __forceinline__ int g2_off(int row, int ch) { return row * 64 + ((ch ^ ((4 - ((row >> 2) & 3)) & 3)) << 4); }

// (The opt-in 256 x 256 ring kernel that stood here -- gemm_nt_mfma_bf16_256, 1.09 PFLOP/s at 4096^3 but one workgroup per CU -- lost to the
// persistent kernel of gemm_p256.hip on every shape of the step and left the library in round 4: tools/experiments/gemm_nt_mfma_bf16_256.hip.txt.)
// The ring structure of gemm_nt_mfma_bf16_256 at the small tile: 4 waves x (64 x 64), three 32-deep stages of 16 KiB (48 KiB:
// 3 workgroups / CU like the default kernel), two tiles in flight per workgroup at all times, one barrier per k-step, no VGPR
// staging and no ds_write.  Meant for the short-K shapes where the default loop's single burst of loads per k-step leaves the
// L2 -> CU path idle most of the time.  K % 32 == 0.
#define G3_STAGE_BYTES 16384        // A 128 x 32 bf16 (8 KiB) + B 128 x 32 bf16 (8 KiB)
#define G3_LDS_BYTES (3 * G3_STAGE_BYTES)

template <typename TO>
__global__ __launch_bounds__(256, 3) void gemm_nt_mfma_bf16_ring128(GemmArgs g, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* glb_vp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int nt = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nt >> 3, r = nt & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * GT_BM, n0 = tn * GT_BN;
    const int b = blockIdx.y;
    const bf16* A = (const bf16*)g.A + (int64_t)b * g.sA;
    const bf16* B = (const bf16*)g.B + (int64_t)b * g.sB;
    TO* C = (TO*)g.C + (int64_t)b * g.sC;
    TO* aux = g.aux ? (TO*)g.aux + (int64_t)b * g.sAux : nullptr;
    const int nk = g.K / G2_BK;

    // LDS-DMA map: 8 pieces of 1 KiB (16 rows) per operand tile, wave w issues pieces 2w, 2w+1 of A and of B
    const bf16* ga[2];
    const bf16* gb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + (lane >> 2);
        const int ch = (lane & 3) ^ ((4 - ((row >> 2) & 3)) & 3);
        ga[i] = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + ch * 8;
        gb[i] = B + (int64_t)min(n0 + row, g.N - 1) * g.ldb + ch * 8;
    }
    auto issue = [&](int kt) {
        char* st = smem + (kt % 3) * G3_STAGE_BYTES;
        const int k0 = kt * G2_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((glb_vp)(ga[i] + k0), (lds_vp)(st + (wave * 2 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)(gb[i] + k0), (lds_vp)(st + 8192 + (wave * 2 + i) * 1024), 16, 0, 0);
        }
    };
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int oa[4], ob[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        oa[i] = g2_off(wr * 64 + i * 16 + fr, fg);
        ob[i] = 8192 + g2_off(wc * 64 + i * 16 + fr, fg);
    }
    if (0 < nk) issue(0);
    if (1 < nk) issue(1);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the one younger tile (4 loads) is still in flight
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // tile kt visible to everyone; everyone is done with tile kt-1
        if (kt + 2 < nk) issue(kt + 2);          // refills the stage tile kt-1 occupied
        const char* st = smem + stage * G3_STAGE_BYTES;
        stage = stage == 2 ? 0 : stage + 1;
        bf16x8_t fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = *(const bf16x8_t*)(st + ob[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *(const bf16x8_t*)(st + oa[i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue (as gemm_nt_mfma_bf16<TO, 1>): two 32-row halves of the wave's 64 x 64 block through a per-wave fp32 LDS tile
    float* ep = (float*)(smem + wave * (32 * GT_EPI_LD * 4));
    const int lc = (lane & 7) * 8;
    const int col = n0 + wc * 64 + lc;
    const bool vec_ok = (g.out_mode == OUT_STORE) && (g.N % 8 == 0) && (g.ldc % 8 == 0) && (g.epi < EPI_GELU || (g.ldaux % 8 == 0));
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
    if (g.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = g.bias[min(col + e, g.N - 1)];
            bias8[e] = (col + e < g.N) ? t : 0.f;
        }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        // the staging tile is private to the wave: one workgroup barrier (everyone is done reading the last stage), after that
        // only wave-level ordering (LDS instructions of a wave execute in order)
        if (qt == 0) __syncthreads();
        else __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep[(i * 16 + fg * 4 + r) * GT_EPI_LD + j * 16 + fr] = acc[qt * 2 + i][j][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll 2
        for (int p = 0; p < 4; ++p) {
            const int lr = p * 8 + (lane >> 3);
            const int row = m0 + wr * 64 + qt * 32 + lr;
            const bool valid = row < g.M && col < g.N;
            const float4 va = *(const float4*)(ep + lr * GT_EPI_LD + lc);
            const float4 vb = *(const float4*)(ep + lr * GT_EPI_LD + lc + 4);
            float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
            if (vec_ok) {
                float ax[8];
                if (epi_reads_aux(g.epi)) {
                    const TO* ap = aux + (int64_t)min(row, g.M - 1) * g.ldaux + (col < g.N ? col : 0);
                    if constexpr (sizeof(TO) == 4) {
                        const float4 a0 = *(const float4*)ap, a1 = *(const float4*)(ap + 4);
                        ax[0] = a0.x; ax[1] = a0.y; ax[2] = a0.z; ax[3] = a0.w; ax[4] = a1.x; ax[5] = a1.y; ax[6] = a1.z; ax[7] = a1.w;
                    } else {
                        const bf16x8 a = *(const bf16x8*)ap;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ax[e] = (float)a.v[e];
                    }
                }
                float pre[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = g.alpha * v[e] + bias8[e];
                    pre[e] = x;
                    switch (g.epi) {
                        case EPI_GELU: x = gelu_t<TO>(x); break;
                        case EPI_GELU_DG: gelu_dgelu_t<TO>(x, x, pre[e]); break;      // aux <- gelu'(pre-activation)
                        case EPI_MUL_AUX: x *= ax[e]; break;
                        case EPI_ELU: x = elu1(x); break;
                        case EPI_MUL_DGELU: x *= dgelu_t<TO>(ax[e]); break;
                        case EPI_MUL_DELU: x *= (ax[e] > 0.f ? 1.0f : ax[e] + 1.0f); break;
                        case EPI_ADD_AUX: x += ax[e]; break;
                        default: break;
                    }
                    v[e] = x;
                }
                if (!valid) continue;
                TO* cp = C + (int64_t)row * g.ldc + col;
                if constexpr (sizeof(TO) == 4) {
                    *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    if (epi_writes_aux(g.epi) && aux) {
                        float* qp = (float*)(aux + (int64_t)row * g.ldaux + col);
                        *(float4*)qp = make_float4(pre[0], pre[1], pre[2], pre[3]);
                        *(float4*)(qp + 4) = make_float4(pre[4], pre[5], pre[6], pre[7]);
                    }
                } else {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o.v[e] = (bf16)v[e];
                    *(bf16x8*)cp = o;
                    if (epi_writes_aux(g.epi) && aux) {
                        bf16x8 q;
#pragma unroll
                        for (int e = 0; e < 8; ++e) q.v[e] = (bf16)pre[e];
                        *(bf16x8*)(aux + (int64_t)row * g.ldaux + col) = q;
                    }
                }
            } else if (valid) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (col + e < g.N) epilogue_store<TO>(g, C, aux, row, col + e, v[e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ MFMA bf16, 256 x 128 tile, 3-stage LDS-DMA ring
// gemm_nt_mfma_bf16_ring128 with twice the rows: 8 waves (4 x 2) x (64 x 64), three 32-deep stages of 24 KiB (72 KiB: two
// workgroups = 16 waves per CU), 85 instead of 64 FLOP per L2 byte.  K % 32 == 0.
#define G4_BM 256
#define G4_STAGE_BYTES 24576        // A 256 x 32 bf16 (16 KiB) + B 128 x 32 bf16 (8 KiB)
#define G4_LDS_BYTES (3 * G4_STAGE_BYTES)

template <typename TO>
__global__ __launch_bounds__(512, 2) void gemm_nt_mfma_bf16_ring256x128(GemmArgs g, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* glb_vp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;        // 4 x 2 waves
    const int fr = lane & 15, fg = lane >> 4;
    const int nt = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nt >> 3, r = nt & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * G4_BM, n0 = tn * GT_BN;
    const int b = blockIdx.y;
    const bf16* A = (const bf16*)g.A + (int64_t)b * g.sA;
    const bf16* B = (const bf16*)g.B + (int64_t)b * g.sB;
    TO* C = (TO*)g.C + (int64_t)b * g.sC;
    TO* aux = g.aux ? (TO*)g.aux + (int64_t)b * g.sAux : nullptr;
    const int nk = g.K / G2_BK;

    // LDS-DMA map: pieces of 1 KiB (16 rows); wave w issues pieces 2w, 2w+1 of the A tile (16 pieces) and piece w of the B tile (8)
    const bf16* ga[2];
    const bf16* gb;
    {
        const int chs = lane & 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 16 + (lane >> 2);
            ga[i] = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + (chs ^ ((4 - ((row >> 2) & 3)) & 3)) * 8;
        }
        const int rowb = wave * 16 + (lane >> 2);
        gb = B + (int64_t)min(n0 + rowb, g.N - 1) * g.ldb + (chs ^ ((4 - ((rowb >> 2) & 3)) & 3)) * 8;
    }
    auto issue = [&](int kt) {
        char* st = smem + (kt % 3) * G4_STAGE_BYTES;
        const int k0 = kt * G2_BK;
        __builtin_amdgcn_global_load_lds((glb_vp)(ga[0] + k0), (lds_vp)(st + (wave * 2) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_vp)(ga[1] + k0), (lds_vp)(st + (wave * 2 + 1) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_vp)(gb + k0), (lds_vp)(st + 16384 + wave * 1024), 16, 0, 0);
    };
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int oa[4], ob[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        oa[i] = g2_off(wr * 64 + i * 16 + fr, fg);
        ob[i] = 16384 + g2_off(wc * 64 + i * 16 + fr, fg);
    }
    if (0 < nk) issue(0);
    if (1 < nk) issue(1);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the one younger tile (3 loads) is still in flight
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // tile kt visible to everyone; everyone is done with tile kt-1
        if (kt + 2 < nk) issue(kt + 2);          // refills the stage tile kt-1 occupied
        const char* st = smem + stage * G4_STAGE_BYTES;
        stage = stage == 2 ? 0 : stage + 1;
        bf16x8_t fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = *(const bf16x8_t*)(st + ob[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *(const bf16x8_t*)(st + oa[i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue (as gemm_nt_mfma_bf16<TO, 1>): two 32-row halves of the wave's 64 x 64 block through a per-wave fp32 LDS tile
    float* ep = (float*)(smem + wave * (32 * GT_EPI_LD * 4));
    const int lc = (lane & 7) * 8;
    const int col = n0 + wc * 64 + lc;
    const bool vec_ok = (g.out_mode == OUT_STORE) && (g.N % 8 == 0) && (g.ldc % 8 == 0) && (g.epi < EPI_GELU || (g.ldaux % 8 == 0));
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
    if (g.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = g.bias[min(col + e, g.N - 1)];
            bias8[e] = (col + e < g.N) ? t : 0.f;
        }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        // the staging tile is private to the wave: one workgroup barrier (everyone is done reading the last stage), after that
        // only wave-level ordering (LDS instructions of a wave execute in order)
        if (qt == 0) __syncthreads();
        else __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep[(i * 16 + fg * 4 + r) * GT_EPI_LD + j * 16 + fr] = acc[qt * 2 + i][j][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll 2
        for (int p = 0; p < 4; ++p) {
            const int lr = p * 8 + (lane >> 3);
            const int row = m0 + wr * 64 + qt * 32 + lr;
            const bool valid = row < g.M && col < g.N;
            const float4 va = *(const float4*)(ep + lr * GT_EPI_LD + lc);
            const float4 vb = *(const float4*)(ep + lr * GT_EPI_LD + lc + 4);
            float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
            if (vec_ok) {
                float ax[8];
                if (epi_reads_aux(g.epi)) {
                    const TO* ap = aux + (int64_t)min(row, g.M - 1) * g.ldaux + (col < g.N ? col : 0);
                    if constexpr (sizeof(TO) == 4) {
                        const float4 a0 = *(const float4*)ap, a1 = *(const float4*)(ap + 4);
                        ax[0] = a0.x; ax[1] = a0.y; ax[2] = a0.z; ax[3] = a0.w; ax[4] = a1.x; ax[5] = a1.y; ax[6] = a1.z; ax[7] = a1.w;
                    } else {
                        const bf16x8 a = *(const bf16x8*)ap;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ax[e] = (float)a.v[e];
                    }
                }
                float pre[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = g.alpha * v[e] + bias8[e];
                    pre[e] = x;
                    switch (g.epi) {
                        case EPI_GELU: x = gelu_t<TO>(x); break;
                        case EPI_GELU_DG: gelu_dgelu_t<TO>(x, x, pre[e]); break;      // aux <- gelu'(pre-activation)
                        case EPI_MUL_AUX: x *= ax[e]; break;
                        case EPI_ELU: x = elu1(x); break;
                        case EPI_MUL_DGELU: x *= dgelu_t<TO>(ax[e]); break;
                        case EPI_MUL_DELU: x *= (ax[e] > 0.f ? 1.0f : ax[e] + 1.0f); break;
                        case EPI_ADD_AUX: x += ax[e]; break;
                        default: break;
                    }
                    v[e] = x;
                }
                if (!valid) continue;
                TO* cp = C + (int64_t)row * g.ldc + col;
                if constexpr (sizeof(TO) == 4) {
                    *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    if (epi_writes_aux(g.epi) && aux) {
                        float* qp = (float*)(aux + (int64_t)row * g.ldaux + col);
                        *(float4*)qp = make_float4(pre[0], pre[1], pre[2], pre[3]);
                        *(float4*)(qp + 4) = make_float4(pre[4], pre[5], pre[6], pre[7]);
                    }
                } else {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o.v[e] = (bf16)v[e];
                    *(bf16x8*)cp = o;
                    if (epi_writes_aux(g.epi) && aux) {
                        bf16x8 q;
#pragma unroll
                        for (int e = 0; e < 8; ++e) q.v[e] = (bf16)pre[e];
                        *(bf16x8*)(aux + (int64_t)row * g.ldaux + col) = q;
                    }
                }
            } else if (valid) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (col + e < g.N) epilogue_store<TO>(g, C, aux, row, col + e, v[e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ MFMA bf16, "TN": weight gradients
// C[N,K] (+)= sum_m A[m,n] * B[m,k]   with A = dY [M,N] and B = X [M,K] in their natural row-major (token-major) layout.
// Both operands need 8 consecutive m per (n|k) column for the matrix cores, i.e. a transposed fragment: the tiles are
// staged row-major ([64 m][128 cols], 288-byte rows) and read with ds_read_b64_tr_b16 (4x16 hardware transpose read), using
// the same k-slot permutation on both sides (slot (g,j<4) <-> m 4g+j, (g,j>=4) <-> m 16+4g+j-4 within a 32-deep step).
// The contraction (M = tokens, up to 401k) is split over blockIdx.y; partial tiles leave through fp32 atomics.
typedef bf16 __attribute__((ext_vector_type(4))) bf16x4_t;
#define TN_BM 64            // contraction depth per LDS tile
#define TN_LD 144           // elements per LDS row (128 + 16 pad: conflict-free transposed reads)
#define TN_TILE_BYTES (TN_BM * TN_LD * 2)

__device__ __forceinline__ bf16x8_t tn_read_tr(const bf16* tile, int m0, int col0, int lane) {
    typedef __attribute__((address_space(3))) bf16x4_t* lds_p;
    const int fg = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const bf16* a0 = tile + (m0 + 4 * fg + q) * TN_LD + col0 + 4 * pp;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)a0);
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0 + 16 * TN_LD));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256, 2) void gemm_tn_mfma_bf16(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B, int64_t ldb,
                                                            float* __restrict__ C, int64_t ldc, int M, int N, int K, int splitk,
                                                            float* __restrict__ colsum, int tiles_n, int tiles_k,
                                                            float* __restrict__ slabs, unsigned* __restrict__ tickets) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    // XCD-aware order: workgroups b, b+8, ... share an XCD and its L2; give each XCD a contiguous run of (token slab, tile)
    // work items, slab-major, so a slab of dY / X is pulled from HBM by one XCD instead of all eight (fetch 3.5x -> ~1x).
    const int nt = tiles_n * tiles_k, total = nt * splitk;
    int bid = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int ks = bid / nt, tl = bid % nt;
    const int tn = tl / tiles_k, tk = tl % tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int mtiles = (M + TN_BM - 1) / TN_BM;
    const int per = (mtiles + splitk - 1) / splitk;
    const int mt0 = ks * per, mt1 = min(mtiles, mt0 + per);
    if (mt0 >= mt1) return;

    // staging: 64 rows x 16 chunks (16 B) per operand tile = 1024 chunks, 4 per thread
    int s_row[4], s_ch[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int c = tid + 256 * i; s_row[i] = c >> 4; s_ch[i] = c & 15; }
    u32x4_t ra[4], rb[4];
    int m_loaded = 0;
    // unconditional loads on clamped addresses; the zeroing select waits in stage_write behind an asm pin (see gemm_nt_mfma_bf16)
    auto stage_load = [&](int mt) {
        const int m0 = mt * TN_BM;
        m_loaded = m0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ca = n0 + s_ch[i] * 8, cb = k0 + s_ch[i] * 8;
            const int64_t mc = min(m0 + s_row[i], M - 1);
            ra[i] = *(const u32x4_t*)(A + mc * lda + (ca < N ? ca : 0));
            rb[i] = *(const u32x4_t*)(B + mc * ldb + (cb < K ? cb : 0));
        }
    };
    auto stage_write = [&](int buf) {
        bf16* sa = (bf16*)(smem + buf * 2 * TN_TILE_BYTES);
        bf16* sb = (bf16*)(smem + buf * 2 * TN_TILE_BYTES + TN_TILE_BYTES);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4_t ta = ra[i], tb = rb[i];
            asm volatile("" : "+v"(ta), "+v"(tb));
            const bool mok = (m_loaded + s_row[i]) < M;
            const bool oka = mok && (n0 + s_ch[i] * 8) < N, okb = mok && (k0 + s_ch[i] * 8) < K;
            *(u32x4_t*)(sa + s_row[i] * TN_LD + s_ch[i] * 8) = oka ? ta : (u32x4_t){0u, 0u, 0u, 0u};
            *(u32x4_t*)(sb + s_row[i] * TN_LD + s_ch[i] * 8) = okb ? tb : (u32x4_t){0u, 0u, 0u, 0u};
        }
    };
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;                 // bias gradient: column sums of A (only blocks with tk == 0)

    stage_load(mt0);
    stage_write(0);
    __syncthreads();
    int cur = 0;
    for (int mt = mt0; mt < mt1; ++mt) {
        const bool more = (mt + 1) < mt1;
        if (more) stage_load(mt + 1);
        const bf16* sa = (const bf16*)(smem + cur * 2 * TN_TILE_BYTES);
        const bf16* sb = (const bf16*)(smem + cur * 2 * TN_TILE_BYTES + TN_TILE_BYTES);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = tn_read_tr(sa, kk * 32, wr * 64 + i * 16, lane);
                fb[i] = tn_read_tr(sb, kk * 32, wc * 64 + i * 16, lane);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (colsum && tk == 0 && tid < 128) {
            float s = 0.f;
#pragma unroll 8
            for (int m = 0; m < TN_BM; ++m) s += (float)sa[m * TN_LD + tid];
            csum += s;
        }
        if (more) stage_write(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    if (colsum && tk == 0 && tid < 128 && n0 + tid < N) atomicAdd(colsum + n0 + tid, csum);
    // ---- combine the split contraction.  With a workspace: every split writes its 128 x 128 fp32 partial as a 64 KiB slab in
    // REGISTER order (one float4 per (fragment, lane): 1 KiB per store instruction, write-through `sc1` stores, so no release
    // fence), draws a ticket, and the split that draws the last one adds the other slabs to its registers and is the only one
    // to touch dW.  fp32 atomics from every split instead (no workspace) move splitk x the bytes at a fifth of the store rate:
    // 35 MB / launch on this model's shapes, about a third of the kernel's time.
    const int nsplit = (mtiles + per - 1) / per;             // splits with a non-empty slab range (the others returned above)
    if (slabs && nsplit > 1) {
        float* mine = slabs + ((size_t)tl * nsplit + ks) * (128 * 128);
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 128 * 128 * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), rsrc, ((wave * 16 + i * 4 + j) * 64 + lane) * 16, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores ...
        __syncthreads();                                      // ... before the one ticket that publishes them all
        unsigned* flag = (unsigned*)smem;                     // the staging tiles are dead: every wave passed the loop's last barrier
        if (tid == 0) *flag = __hip_atomic_fetch_add(tickets + tl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != (unsigned)(nsplit - 1)) return;
        // every wave takes the agent-scope acquire itself (buffer_inv): its slab loads below must not be served from lines its own
        // XCD's L2 cached before the other splits' write-through stores landed
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (tid == 0) __hip_atomic_store(tickets + tl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
        for (int sidx = 0; sidx < nsplit; ++sidx) {
            if (sidx == ks) continue;
            const float* other = slabs + ((size_t)tl * nsplit + sidx) * (128 * 128);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += *(const f32x4_t*)(other + ((wave * 16 + i * 4 + j) * 64 + lane) * 4);
        }
    }
    // epilogue: acc[i][j][r] = C[n = n0 + wr*64 + i*16 + 4*fg + r][k = k0 + wc*64 + j*16 + fr]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wr * 64 + i * 16 + 4 * fg + r, k = k0 + wc * 64 + j * 16 + fr;
                if (n < N && k < K) atomicAdd(C + (int64_t)n * ldc + k, acc[i][j][r]);
            }
}

// dW[N,K] += dY[M,N]^T . X[M,K]  (fp32 atomic accumulate);  optional db[N] += column sums of dY
// workspace of mvuld_gemm_tn_wgrad for an N x K weight: one 4-byte ticket per 128 x 128 output tile (kept at 0 between
// launches by the kernel itself: the caller zeroes the buffer ONCE, when it allocates it) followed by splitk 64 KiB slabs per tile
#define TN_TICKET_BYTES 4096            // (gemm_tn256.hip: TN_GROUP_WS_HEAD must equal it)
static int64_t tn128_workspace_bytes(int N, int K, int splitk) {
    const int64_t tiles = cdiv(N, 128) * cdiv(K, 128);
    if (splitk < 2 || tiles * 4 > TN_TICKET_BYTES) return 0;
    return TN_TICKET_BYTES + tiles * splitk * (int64_t)(128 * 128 * 4);
}
extern "C" int64_t mvuld_gemm_tn_wgrad_workspace_bytes(int M, int N, int K, int splitk) {
    const int64_t a = tn128_workspace_bytes(N, K, splitk);
    const int64_t b = mvuld_gemm_tn256_workspace_bytes(M, N, K);
    const int64_t c = b > 0 ? TN_TICKET_BYTES + b : 0;
    return a > c ? a : c;
}

// 1 (default): weights that fill 256 x 256 tiles go to the LDS-DMA kernel of gemm_tn256.hip when a workspace is given; 0: never
static std::atomic<int> g_tn256{-1};
extern "C" int mvuld_set_gemm_tn256(int on) {
    g_tn256.store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}

extern "C" int mvuld_gemm_tn_wgrad(const void* dY, int64_t ldy, const void* X, int64_t ldx, float* dW, int64_t ldw, int M, int N, int K,
                                   float* dbias, int splitk, void* ws, int64_t ws_bytes, hipStream_t stream) {
    MV_CHECK_ARG(dY && X && dW && M > 0 && N > 0 && K > 0, "gemm_tn_wgrad: bad args");
    MV_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && (((uintptr_t)dY | (uintptr_t)X) & 15) == 0,
                 "gemm_tn_wgrad: operands must be 16-byte aligned with N, K multiples of 8");
    const int tiles_n = (int)cdiv(N, 128), tiles_k = (int)cdiv(K, 128);
    const int mtiles = (int)cdiv(M, TN_BM);
    if (splitk < 1) splitk = 1;
    if (splitk > mtiles) splitk = mtiles;
    if (ws && ws_bytes > TN_TICKET_BYTES && (((uintptr_t)ws) & 15) == 0) {
        int on = g_tn256.load(std::memory_order_relaxed);
        if (on < 0) { const char* e = getenv("MVULD_GEMM_TN256"); on = e ? (atoi(e) != 0) : 1; g_tn256.store(on, std::memory_order_relaxed); }
        if (on && mvuld_gemm_tn256_try(dY, ldy, X, ldx, dW, ldw, M, N, K, dbias, (char*)ws + TN_TICKET_BYTES, ws_bytes - TN_TICKET_BYTES, stream) == 0) {
            MV_LAUNCH_CHECK("gemm_tn256");
            return 0;
        }
    }
    float* slabs = nullptr;
    unsigned* tickets = nullptr;
    if (ws && splitk >= 2 && splitk <= 8 && tn128_workspace_bytes(N, K, splitk) > 0) {
        const int64_t need = tn128_workspace_bytes(N, K, splitk);
        MV_CHECK_ARG(need > 0 && ws_bytes >= need && (((uintptr_t)ws) & 15) == 0, "gemm_tn_wgrad: workspace too small (%lld < %lld bytes) or misaligned",
                     (long long)ws_bytes, (long long)need);
        tickets = (unsigned*)ws;
        slabs = (float*)((char*)ws + TN_TICKET_BYTES);
    }
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)gemm_tn_mfma_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TN_TILE_BYTES);
        return true;
    }();
    (void)attr;
    const int lds = 4 * TN_TILE_BYTES;
    dim3 grid(tiles_n * tiles_k * splitk);
    hipLaunchKernelGGL(gemm_tn_mfma_bf16, grid, dim3(256), lds, stream, (const bf16*)dY, ldy, (const bf16*)X, ldx, dW, ldw, M, N, K, splitk,
                       dbias, tiles_n, tiles_k, slabs, tickets);
    MV_LAUNCH_CHECK("gemm_tn_wgrad");
    return 0;
}

// C = epi(scale_a * scale_b * A8 . B8^T + bias): OCP e4m3 operands (mvuld_quant_e4m3), fp32 accumulate, bf16 out
extern "C" int mvuld_gemm_nt_fp8(const void* A8, int64_t lda, const void* B8, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                                 const float* bias, int epilogue, void* aux, int64_t ldaux, const float* scale_a, const float* scale_b,
                                 void* q_out, int64_t ldq, float* q_state, hipStream_t stream) {
    MV_CHECK_ARG(A8 && B8 && (C || q_out) && M > 0 && N > 0 && K > 0, "gemm_nt_fp8: bad args");
    MV_CHECK_ARG(!q_out || (q_state && (epilogue == EPI_GELU || epilogue == EPI_GELU_DG)), "gemm_nt_fp8: the e4m3 side output belongs to the GELU epilogues and needs its {scale, amax} pair");
    MV_CHECK_ARG(epilogue == EPI_NONE || epilogue == EPI_BIAS || epilogue == EPI_GELU || epilogue == EPI_GELU_DG, "gemm_nt_fp8: epilogue %d (NONE / BIAS / GELU / GELU_DG only)", epilogue);
    GemmArgs g;
    g.A = A8; g.B = B8; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = g.sB = g.sC = 0;
    g.M = M; g.N = N; g.K = K; g.batch = 1; g.splitk = 1; g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.sAux = 0;
    g.alpha = 1.0f; g.epi = epilogue; g.out_mode = OUT_STORE; g.scale_a = scale_a; g.scale_b = scale_b;
    g.q_out = q_out; g.ldq = ldq; g.q_scale = q_state; g.q_amax = q_state ? (unsigned*)(q_state + 1) : nullptr;
    if (mvuld_gemm_nt_p256_fp8(g, stream) != 0) {
        mvuld_set_error("gemm_nt_fp8: shape M=%d N=%d K=%d not eligible (K %% 64 == 0, K >= 256, N %% 8 == 0, 16-byte aligned rows)", M, N, K);
        return 1;
    }
    MV_LAUNCH_CHECK("gemm_nt_fp8");
    return 0;
}

// ------------------------------------------------------------------------------------ fp32 operands, split in registers
// The head's fp32 tail (GAT / Rs_GCN / classifier products, M ~ 3200 nodes or 32 x [100 x 100]) runs on the bf16 matrix cores as
//   a . b ~= a_hi b_hi + a_lo b_hi + a_hi b_lo,   x_hi = bf16(x), x_lo = bf16(x - x_hi)   (error ~2^-16, fp32 accumulation).
// Rounds 1-2 materialised [hi | lo | hi] / [hi | hi | lo] copies of both operands (two split3_k launches per product, 224 per step)
// and ran a 3K-deep bf16 product; here the split happens on the way from the global loads to LDS: one launch per product, no copies.
// 64 x 64 tile, 4 waves (2 x 2) x (32 x 32), 64-deep steps, register-staged double buffering; every epilogue of mvuld_gemm_nt through
// epilogue_store (these products are small: the element-wise store is not what bounds them); batched; splitk == 1.
#define F3_KS 64                                    // contraction elements per step (two 32-wide chunks: these products are latency-bound, not MFMA-bound)
#define F3_LD (F3_KS + 8)                           // bf16 elements per LDS row (144-byte rows)
#define F3_LDS_BYTES (2 * 4 * 64 * F3_LD * 2)       // two stages x {A_hi, A_lo, B_hi, B_lo}
// one operand's 64 x 32 piece of a k-step: 8 floats per thread.  Row-major operand ([rows, K], ld): thread -> row t >> 2, k (t & 3) * 8 .. + 7.
// Transposed operand ([K, rows], ld): thread -> k t >> 3, rows (t & 7) * 8 .. + 7 (contiguous in memory).  Out-of-range elements are zeros.
template <bool trans>
__device__ __forceinline__ void f3_fetch(const float* __restrict__ P, int64_t ld, bool vec, int r0, int rows, int k0, int K, int tid, float (&x)[8]) {
    if constexpr (!trans) {
        const int lr = tid >> 2, lk = (tid & 3) * 8;
        const float* p = P + (int64_t)min(r0 + lr, rows - 1) * ld;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + lk + 4 * h;
            if (vec && k + 3 < K) {
                const float4 v = *(const float4*)(p + k);
                x[4 * h] = v.x; x[4 * h + 1] = v.y; x[4 * h + 2] = v.z; x[4 * h + 3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float v = p[min(k + e, K - 1)]; x[4 * h + e] = k + e < K ? v : 0.f; }
            }
        }
    } else {
        const int kk = k0 + (tid >> 3), seg = r0 + (tid & 7) * 8;
        const float* p = P + (int64_t)min(kk, K - 1) * ld;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = seg + 4 * h;
            if (vec && kk < K && r + 3 < rows) {
                const float4 v = *(const float4*)(p + r);
                x[4 * h] = v.x; x[4 * h + 1] = v.y; x[4 * h + 2] = v.z; x[4 * h + 3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float v = p[min(r + e, rows - 1)]; x[4 * h + e] = (kk < K && r + e < rows) ? v : 0.f; }
            }
        }
    }
}
template <bool trans>
__device__ __forceinline__ void f3_stage(bf16* __restrict__ hi, bf16* __restrict__ lo, int tid, const float (&x)[8]) {      // (hi / lo already offset to the chunk's column)
    bf16x8 h8, l8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { h8.v[e] = (bf16)x[e]; l8.v[e] = (bf16)(x[e] - (float)h8.v[e]); }
    if constexpr (!trans) {
        const int o = (tid >> 2) * F3_LD + (tid & 3) * 8;
        *(bf16x8*)(hi + o) = h8; *(bf16x8*)(lo + o) = l8;
    } else {                                             // kept as it is stored, [k][row]: the fragments come out through transposed reads
        const int o = (tid >> 3) * F3_LD + (tid & 7) * 8;
        *(bf16x8*)(hi + o) = h8; *(bf16x8*)(lo + o) = l8;
    }
}
// MFMA fragment of 16 rows r0 .. r0+15 x 32 k (chunk kc) of one plane: lane (fr, fg) holds k = 8 fg .. 8 fg + 7 of row r0 + fr.
// Row-major plane [row][F3_LD]: one 16-byte read.  Transposed plane [k][F3_LD] (chunk kc = rows 32 kc .. of it): two ds_read_b64_tr_b16 --
// lane 4q + p of a 16-lane group addresses k-row 8 fg (+ 4) + q, tile rows 4p .. 4p + 3, and receives tile row (lane & 15) of the four k-rows.
template <bool trans>
__device__ __forceinline__ bf16x8_t f3_frag(const bf16* __restrict__ plane, int r0, int kc, int lane) {
    const int fr = lane & 15, fg = lane >> 4;
    if constexpr (!trans) {
        return *(const bf16x8_t*)(plane + (r0 + fr) * F3_LD + 32 * kc + fg * 8);
    } else {
        typedef bf16 __attribute__((ext_vector_type(4))) f3_bf16x4_t;
        typedef __attribute__((address_space(3))) f3_bf16x4_t* lds_p4;
        const int q = fr >> 2, pp = fr & 3;
        const bf16* a0 = plane + (32 * kc + 8 * fg + q) * F3_LD + r0 + 4 * pp;
        const f3_bf16x4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p4)a0);
        const f3_bf16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p4)(a0 + 4 * F3_LD));
        return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}
template <bool ta, bool tb>
__global__ __launch_bounds__(256) void gemm_nt_f32x3_k(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char f3_smem[];
    bf16 (*sm)[4][64 * F3_LD] = (bf16 (*)[4][64 * F3_LD])f3_smem;            // [stage][A_hi, A_lo, B_hi, B_lo][64 rows][F3_LD]
    constexpr int NC = F3_KS / 32;
    const int b = blockIdx.z / g.splitk, ks = blockIdx.z % g.splitk;
    const float* A = (const float*)g.A + (int64_t)b * g.sA;
    const float* B = (const float*)g.B + (int64_t)b * g.sB;
    float* C = (float*)g.C + (int64_t)b * g.sC;
    float* aux = g.aux ? (float*)g.aux + (int64_t)b * g.sAux : nullptr;
    if (ks > 0) g.bias = nullptr;                        // (contraction splits add into C with atomics: the bias once)
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fg = lane >> 4;
    // 16-byte loads: along k for a row-major operand (K % 4), along the rows for a transposed one (M or N % 4)
    const bool veca = (g.lda % 4 == 0) && (g.sA % 4 == 0) && ((((uintptr_t)g.A) & 15) == 0) && ((ta ? g.M : g.K) % 4 == 0);
    const bool vecb = (g.ldb % 4 == 0) && (g.sB % 4 == 0) && ((((uintptr_t)g.B) & 15) == 0) && ((tb ? g.N : g.K) % 4 == 0);
    (void)g.ta; (void)g.tb;
    float xa[NC][8], xb[NC][8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            f3_fetch<ta>(A, g.lda, veca, m0, g.M, k0 + 32 * j, g.K, tid, xa[j]);
            f3_fetch<tb>(B, g.ldb, vecb, n0, g.N, k0 + 32 * j, g.K, tid, xb[j]);
        }
    };
    auto stage = [&](int st) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {               // chunk j: columns 32 j .. of a row-major plane, rows 32 j .. of a transposed one
            f3_stage<ta>(sm[st][0] + (ta ? 32 * j * F3_LD : 32 * j), sm[st][1] + (ta ? 32 * j * F3_LD : 32 * j), tid, xa[j]);
            f3_stage<tb>(sm[st][2] + (tb ? 32 * j * F3_LD : 32 * j), sm[st][3] + (tb ? 32 * j * F3_LD : 32 * j), tid, xb[j]);
        }
    };
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int nk_all = (g.K + F3_KS - 1) / F3_KS, per = (nk_all + g.splitk - 1) / g.splitk;
    const int c0 = ks * per, c1 = min(nk_all, c0 + per);
    if (c0 < c1) {
        fetch(c0 * F3_KS);
        stage(0);
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
        const int st = (c - c0) & 1;
        if (c + 1 < c1) fetch((c + 1) * F3_KS);          // next step's operands fly under this step's MFMAs
#pragma unroll
        for (int kc = 0; kc < NC; ++kc) {
            bf16x8_t fah[2], fal[2], fbh[2], fbl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fah[i] = f3_frag<ta>(sm[st][0], wr * 32 + i * 16, kc, lane); fal[i] = f3_frag<ta>(sm[st][1], wr * 32 + i * 16, kc, lane);
                fbh[i] = f3_frag<tb>(sm[st][2], wc * 32 + i * 16, kc, lane); fbl[i] = f3_frag<tb>(sm[st][3], wc * 32 + i * 16, kc, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
                }
        }
        if (c + 1 < c1) stage(st ^ 1);                   // the other stage was last read two barriers ago
        __syncthreads();
    }
    if (c0 >= c1) return;                                // (a split past the end of the contraction)
    // acc[i][j][r] = C[m0 + wr*32 + i*16 + 4*fg + r][n0 + wc*32 + j*16 + fr]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * 32 + i * 16 + 4 * fg + r, col = n0 + wc * 32 + j * 16 + fr;
                if (row < g.M && col < g.N) epilogue_store<float>(g, C, aux, row, col, acc[i][j][r]);
            }
}

// C[b] (fp32) = epilogue(alpha * op(A[b]) . op(B[b])^T + bias) for fp32 operands at near-fp32 accuracy on the bf16 matrix cores.
// trans_a / trans_b: the operand is stored [K, M] / [K, N] (the "TN" and "NN" forms of a product without a transpose pass: Rs_GCN's R^T dY,
// dR ph, and every weight gradient dY^T X of the head); splitk > 1: contraction splits adding into C with atomics (out_mode ATOMIC).
extern "C" int mvuld_gemm_nt_f32x3(const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb, int64_t strideB, float* C, int64_t ldc,
                                   int64_t strideC, int M, int N, int K, int batch, const float* bias, int epilogue, float* aux, int64_t ldaux,
                                   int64_t strideAux, float alpha, int out_mode, int trans_a, int trans_b, int splitk, hipStream_t stream) {
    MV_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && splitk >= 1, "gemm_nt_f32x3: bad args");
    MV_CHECK_ARG(epilogue >= EPI_NONE && epilogue <= EPI_MUL_AUX && !(epi_reads_aux(epilogue) && !aux), "gemm_nt_f32x3: epilogue %d", epilogue);
    MV_CHECK_ARG(out_mode == OUT_STORE || out_mode == OUT_ACCUM || (out_mode == OUT_ATOMIC && epilogue <= EPI_BIAS), "gemm_nt_f32x3: out_mode %d", out_mode);
    MV_CHECK_ARG(splitk == 1 || out_mode == OUT_ATOMIC, "gemm_nt_f32x3: a split contraction adds into C with atomics (out_mode ATOMIC)");
    MV_CHECK_ARG((int64_t)cdiv(M, 64) < 65536 && (int64_t)batch * splitk < 65536, "gemm_nt_f32x3: grid");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = strideA; g.sB = strideB; g.sC = strideC;
    g.M = M; g.N = N; g.K = K; g.batch = batch; g.splitk = splitk; g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.sAux = strideAux;
    g.alpha = alpha; g.epi = epilogue; g.out_mode = out_mode; g.scale_a = nullptr; g.scale_b = nullptr;
    g.ta = trans_a ? 1 : 0; g.tb = trans_b ? 1 : 0;
    const dim3 grid((unsigned)cdiv(N, 64), (unsigned)cdiv(M, 64), (unsigned)(batch * splitk));
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)gemm_nt_f32x3_k<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_f32x3_k<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_f32x3_k<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_f32x3_k<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS_BYTES);
        return true;
    }();
    (void)attr;
    if (g.ta && g.tb) hipLaunchKernelGGL((gemm_nt_f32x3_k<true, true>), grid, dim3(256), F3_LDS_BYTES, stream, g);
    else if (g.ta) hipLaunchKernelGGL((gemm_nt_f32x3_k<true, false>), grid, dim3(256), F3_LDS_BYTES, stream, g);
    else if (g.tb) hipLaunchKernelGGL((gemm_nt_f32x3_k<false, true>), grid, dim3(256), F3_LDS_BYTES, stream, g);
    else hipLaunchKernelGGL((gemm_nt_f32x3_k<false, false>), grid, dim3(256), F3_LDS_BYTES, stream, g);
    MV_LAUNCH_CHECK("gemm_nt_f32x3");
    return 0;
}

// ------------------------------------------------------------------------------------ C ABI
extern "C" int mvuld_gemm_nt(const void* A, int64_t lda, int64_t strideA, const void* B, int64_t ldb, int64_t strideB,
                             void* C, int64_t ldc, int64_t strideC, int M, int N, int K, int batch,
                             const float* bias, int epilogue, void* aux, int64_t ldaux, int64_t strideAux,
                             float alpha, int out_mode, int splitk, int dtype_in, int dtype_out, int force_simple,
                             hipStream_t stream) {
    MV_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "gemm_nt: empty problem M=%d N=%d K=%d batch=%d", M, N, K, batch);
    MV_CHECK_ARG(A && B && C, "gemm_nt: null operand");
    MV_CHECK_ARG(epilogue >= EPI_NONE && epilogue <= EPI_MUL_AUX, "gemm_nt: bad epilogue %d", epilogue);
    MV_CHECK_ARG(!(epi_reads_aux(epilogue) && !aux), "gemm_nt: epilogue %d needs aux", epilogue);
    MV_CHECK_ARG(out_mode >= OUT_STORE && out_mode <= OUT_ATOMIC, "gemm_nt: bad out_mode %d", out_mode);
    MV_CHECK_ARG(!(out_mode == OUT_ATOMIC && dtype_out != MVULD_F32), "gemm_nt: atomic output must be f32");
    MV_CHECK_ARG(!(out_mode == OUT_ATOMIC && epilogue > EPI_BIAS), "gemm_nt: atomic output with nonlinear epilogue");
    if (splitk < 1) splitk = 1;
    MV_CHECK_ARG(splitk == 1 || out_mode == OUT_ATOMIC, "gemm_nt: split-K needs atomic output");
    MV_CHECK_ARG(lda >= K && ldb >= K && ldc >= N, "gemm_nt: leading dim too small");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = strideA; g.sB = strideB; g.sC = strideC;
    g.M = M; g.N = N; g.K = K; g.batch = batch; g.splitk = splitk; g.bias = bias; g.aux = aux; g.ldaux = ldaux;
    g.sAux = strideAux; g.alpha = alpha; g.epi = epilogue; g.out_mode = out_mode; g.scale_a = nullptr; g.scale_b = nullptr;
    const bool aligned = (K % 8 == 0) && (lda % 8 == 0) && (ldb % 8 == 0) && (strideA % 8 == 0) && (strideB % 8 == 0) &&
                         (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0);
    const bool use_mfma = (dtype_in == MVULD_BF16) && aligned && !force_simple && (M >= 32) && (N >= 32);
    if (use_mfma) {
        // persistent 256 x 256 kernel (gemm_p256.hip): the tall bf16 -> bf16 products of the two encoders
        if (mvuld_gemm_nt_p256_try(g, dtype_out, stream) == 0) { MV_LAUNCH_CHECK("gemm_nt_bf16_p256"); return 0; }
        const int tiles_m = (int)cdiv(M, GT_BM), tiles_n = (int)cdiv(N, GT_BN);
        dim3 grid(tiles_m * tiles_n, batch * splitk);
        // 128 x 128 LDS-DMA ring for contractions up to MVULD_GEMM_RING_MAXK (default 2304; 0 = never): +8..15 % on the short-K
        // shapes (two tiles always in flight instead of one burst of loads per k-step); the register-staged loop below keeps the
        // long-K ones (K = 3072 / 4096: 10-15 % better there, its 64-deep steps halve the barriers per FLOP).
        // 256 x 128 LDS-DMA ring (two 8-wave workgroups / CU, 85 FLOP per L2 byte) whenever the taller tiles still fill the chip:
        // 3-15 % over the 128 x 128 ring on every shape of the step, -1.8 ms per step.  MVULD_GEMM_RING256X128 = minimum K (0 = off).
        static const int ring4 = [] { const char* e = getenv("MVULD_GEMM_RING256X128"); return e ? atoi(e) : 32; }();
        static const int ring4_tiles = [] { const char* e = getenv("MVULD_GEMM_RING256X128_TILES"); return e ? atoi(e) : 256; }();
        if (ring4 > 0 && splitk == 1 && out_mode == OUT_STORE && K % G2_BK == 0 && K >= ring4 && (int64_t)cdiv(M, G4_BM) * tiles_n * batch >= ring4_tiles) {
            const int tm4 = (int)cdiv(M, G4_BM);
            static const bool attr4 = [] {
                (void)hipFuncSetAttribute((const void*)gemm_nt_mfma_bf16_ring256x128<float>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)gemm_nt_mfma_bf16_ring256x128<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES);
                return true;
            }();
            (void)attr4;
            if (dtype_out == MVULD_F32) hipLaunchKernelGGL((gemm_nt_mfma_bf16_ring256x128<float>), dim3(tm4 * tiles_n, batch), dim3(512), G4_LDS_BYTES, stream, g, tm4, tiles_n);
            else hipLaunchKernelGGL((gemm_nt_mfma_bf16_ring256x128<bf16>), dim3(tm4 * tiles_n, batch), dim3(512), G4_LDS_BYTES, stream, g, tm4, tiles_n);
            MV_LAUNCH_CHECK("gemm_nt_mfma_bf16_ring256x128");
            return 0;
        }
        static const int ring = [] { const char* e = getenv("MVULD_GEMM_RING_MAXK"); return e ? atoi(e) : 2304; }();
        if (splitk == 1 && out_mode == OUT_STORE && K % G2_BK == 0 && K <= ring) {
            dim3 grid3(tiles_m * tiles_n, batch);
            static const bool attr3 = [] {
                (void)hipFuncSetAttribute((const void*)gemm_nt_mfma_bf16_ring128<float>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)gemm_nt_mfma_bf16_ring128<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES);
                return true;
            }();
            (void)attr3;
            if (dtype_out == MVULD_F32) hipLaunchKernelGGL((gemm_nt_mfma_bf16_ring128<float>), grid3, dim3(256), G3_LDS_BYTES, stream, g, tiles_m, tiles_n);
            else hipLaunchKernelGGL((gemm_nt_mfma_bf16_ring128<bf16>), grid3, dim3(256), G3_LDS_BYTES, stream, g, tiles_m, tiles_n);
            MV_LAUNCH_CHECK("gemm_nt_mfma_bf16_ring128");
            return 0;
        }
        // Variant: a long contraction (K >= 1024) on a grid that fits the chip at 2 workgroups / CU anyway runs the double-buffered
        // LDS-DMA loop (one barrier per k-step; the near-fp32 head GEMMs: -6 %); everything else the single-buffer loop at
        // 3 workgroups / CU, whose extra resident tile hides prologue / epilogue better than the deeper pipeline does (A/B in the
        // full step: equal or better).  MVULD_GEMM_NBUF=1|2|3 forces single | double | double + LDS-DMA.
        static const int forced = [] { const char* e = getenv("MVULD_GEMM_NBUF"); return (e && e[0] >= '1' && e[0] <= '3') ? e[0] - '0' : 0; }();
        const bool small_grid = (int64_t)tiles_m * tiles_n * batch * splitk <= 512;
        const int variant = forced ? forced : (K >= 1024 && K % GT_BK == 0 && small_grid ? 3 : 1);
        static const bool attr = [] {
            (void)hipFuncSetAttribute((const void*)gemm_nt_mfma_bf16<float, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES_N(2));
            (void)hipFuncSetAttribute((const void*)gemm_nt_mfma_bf16<bf16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES_N(2));
            (void)hipFuncSetAttribute((const void*)gemm_nt_mfma_bf16<float, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES_N(2));
            (void)hipFuncSetAttribute((const void*)gemm_nt_mfma_bf16<bf16, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GT_LDS_BYTES_N(2));
            return true;
        }();
        (void)attr;
        if (variant == 3 && K % GT_BK == 0) {
            if (dtype_out == MVULD_F32) hipLaunchKernelGGL((gemm_nt_mfma_bf16<float, 2, true>), grid, dim3(256), GT_LDS_BYTES_N(2), stream, g, tiles_m, tiles_n);
            else hipLaunchKernelGGL((gemm_nt_mfma_bf16<bf16, 2, true>), grid, dim3(256), GT_LDS_BYTES_N(2), stream, g, tiles_m, tiles_n);
        } else if (variant >= 2) {
            if (dtype_out == MVULD_F32) hipLaunchKernelGGL((gemm_nt_mfma_bf16<float, 2>), grid, dim3(256), GT_LDS_BYTES_N(2), stream, g, tiles_m, tiles_n);
            else hipLaunchKernelGGL((gemm_nt_mfma_bf16<bf16, 2>), grid, dim3(256), GT_LDS_BYTES_N(2), stream, g, tiles_m, tiles_n);
        } else {
            if (dtype_out == MVULD_F32) hipLaunchKernelGGL((gemm_nt_mfma_bf16<float, 1>), grid, dim3(256), GT_LDS_BYTES_N(1), stream, g, tiles_m, tiles_n);
            else hipLaunchKernelGGL((gemm_nt_mfma_bf16<bf16, 1>), grid, dim3(256), GT_LDS_BYTES_N(1), stream, g, tiles_m, tiles_n);
        }
        MV_LAUNCH_CHECK("gemm_nt_mfma_bf16");
        return 0;
    }
    dim3 grid((unsigned)cdiv(N, 64), (unsigned)cdiv(M, 64), batch * splitk);
    MV_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "gemm_nt: grid too large for the simple kernel (M=%d)", M);
    if (dtype_in == MVULD_F32 && dtype_out == MVULD_F32) hipLaunchKernelGGL((gemm_nt_simple<float, float>), grid, dim3(256), 0, stream, g);
    else if (dtype_in == MVULD_BF16 && dtype_out == MVULD_BF16) hipLaunchKernelGGL((gemm_nt_simple<bf16, bf16>), grid, dim3(256), 0, stream, g);
    else if (dtype_in == MVULD_BF16 && dtype_out == MVULD_F32) hipLaunchKernelGGL((gemm_nt_simple<bf16, float>), grid, dim3(256), 0, stream, g);
    else if (dtype_in == MVULD_F32 && dtype_out == MVULD_BF16) hipLaunchKernelGGL((gemm_nt_simple<float, bf16>), grid, dim3(256), 0, stream, g);
    else { mvuld_set_error("gemm_nt: bad dtypes %d %d", dtype_in, dtype_out); return 1; }
    MV_LAUNCH_CHECK("gemm_nt_simple");
    return 0;
}
