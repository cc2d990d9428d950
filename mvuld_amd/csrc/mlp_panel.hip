// Fused MLP of the narrow Swin stages (C = 128 / 256: stages 0 and 1 at 448 x 448; reference Mlp.forward, swin_transformer_v2.py:26-32,
// and its autograd).  At these widths the four MLP products of a block are HBM streams, not matrix problems: 401 408 tokens x 4C hidden
// is 411 MB per tensor and the unfused step writes the pre-activation and the activation, reads the activation back for fc2, reads the
// pre-activation for dGELU, writes d(pre-activation) and reads it back for the fc1^T product -- 3.0 GB per block of traffic that never
// needed to leave the chip (profiles/r02_gemm_shapes.csv: these products run at 3.3-4.9 TB/s, i.e. AT the HBM rate).
//
//   forward  : y = gelu(x W1^T + b1) W2^T + b2       writes y and the activation h (the fc2 weight gradient needs it); NO pre-activation
//   backward : dh = (dy W2) o gelu'(x W1^T + b1)     the pre-activation is RECOMPUTED (K = C is 128 / 256: cheap), dh is written once
//              dx = dh W1 + g                        (the fc1 weight gradient needs it) and consumed from registers for dx
//
// Decomposition: a wave owns TT x 16 tokens of a panel and ALL hidden / output columns, so both contractions are wave-local and the
// first product's accumulators ARE the second product's activation operand (the MFMA's activation port keeps the token on the lane:
// two 16-column accumulator tiles give a lane the 8 k-slots of a 32-deep step, and the k-slot permutation that implies -- slot (g, e<4)
// <-> hidden 4g+e, (g, e>=4) <-> hidden 16+4g+e-4 -- is applied to the WEIGHT operand's LDS reads, two 8-byte reads instead of one
// 16-byte one).  Nothing but weights ever sits in LDS: the hidden dimension is walked in chunks of HC columns whose weight slices
// (W1 rows; W2 / W1^T column slices) are staged through registers into a double-buffered, XOR-swizzled LDS image while the previous
// chunk computes; token fragments are read from HBM straight into registers, once per panel.  Persistent grid over panels.
// Outputs leave in the 16-byte row segments of gemm_p256.hip's epilogue (v_permlane16_swap of packed pairs); rows past M are clamped
// (loads) and stored onto row M - 1 with the identical values they duplicate, so every store is unconditional.
#include "gemm_common.h"

typedef bf16 __attribute__((ext_vector_type(4))) m_bf16x4_t;
#ifndef MLP_X
#define MLP_X 0      // timing experiments (results WRONG): bit 0 = no GELU / dGELU math, bit 1 = no activation stores, bit 2 = no second product
#endif

__device__ __forceinline__ unsigned m_pk(float a, float b) {
    typedef bf16 __attribute__((ext_vector_type(2))) bf16x2_t;
    bf16x2_t v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float m_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float m_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// byte offset of 16-byte slot `slot` of row `row` in a weight tile with a pitch of P bytes (P = 64, 128, 256, 512): the slot is XORed
// with a row term so that the 16 rows of a fragment read fall on different bank groups
template <int P>
__device__ __forceinline__ int m_off(int row, int slot) {
    const int swz = P == 64 ? ((row >> 2) & 3) : (P == 128 ? ((row >> 1) & 7) : (row & 15));
    return row * P + ((slot ^ swz) << 4);
}

// global -> registers -> LDS staging of one weight tile [R rows][P bytes] (row r at src + r * ld elements); NTH threads
template <int R, int P, int NTH>
struct MTile {
    static constexpr int SLOTS = R * P / 16, PER = (SLOTS + NTH - 1) / NTH;
    uint4 v[PER];
    __device__ __forceinline__ void load(const bf16* __restrict__ src, int64_t ld, int tid) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int s = tid + i * NTH;
            const int row = s / (P / 16), slot = s % (P / 16);
            v[i] = (SLOTS % NTH == 0 || s < SLOTS) ? *(const uint4*)(src + (int64_t)row * ld + slot * 8) : make_uint4(0, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void store(char* dst, int tid) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int s = tid + i * NTH;
            const int row = s / (P / 16), slot = s % (P / 16);
            if (SLOTS % NTH == 0 || s < SLOTS) *(uint4*)(dst + m_off<P>(row, slot)) = v[i];
        }
    }
};

struct MlpArgs {
    const bf16* X;  const bf16* D;  const bf16* G;      // [M, C] token-major: x (both passes), dy (backward), residual gradient added to dx (backward, may be null)
    const bf16* W1; const float* b1;                    // fc1.weight [4C, C], fc1.bias [4C]
    const bf16* W2; const float* b2;                    // forward: fc2.weight [C, 4C], fc2.bias [C];  backward: W2 = fc2.weight^T [4C, C]
    const bf16* W1T;                                    // backward: fc1.weight^T [C, 4C]
    bf16* H; bf16* Y;                                   // forward: activation [M, 4C], output [M, C];  backward: H = d(pre-activation) [M, 4C], Y = dx [M, C]
    int M;
};

// weight fragment (MFMA A port) of 16 rows `r0 ..` at 32-deep k-step `ks` of a tile with pitch P: one 16-byte read
template <int P>
__device__ __forceinline__ bf16x8_t m_frag(const char* tile, int r0, int ks, int fr, int fg) {
    return *(const bf16x8_t*)(tile + m_off<P>(r0 + fr, ks * 4 + fg));
}
// the same under the accumulator-to-operand k-slot permutation: columns 32 ks + 4 fg .. + 3 and 32 ks + 16 + 4 fg .. + 3 (two 8-byte reads)
template <int P>
__device__ __forceinline__ bf16x8_t m_frag_perm(const char* tile, int r0, int ks, int fr, int fg) {
    const int row = r0 + fr, half = (fg & 1) * 8;
    const m_bf16x4_t lo = *(const m_bf16x4_t*)(tile + m_off<P>(row, ks * 4 + (fg >> 1)) + half);
    const m_bf16x4_t hi = *(const m_bf16x4_t*)(tile + m_off<P>(row, ks * 4 + 2 + (fg >> 1)) + half);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// 16-byte row-segment store of two adjacent 16-column accumulator tiles (v0 = tile j0, v1 = tile j0 + 1) of token row `row`:
// after the swaps the lane owns columns c0 + (fg & 1) * 16 + (fg >> 1) * 8 .. + 7 (gemm_p256.hip's epilogue)
__device__ __forceinline__ void m_store_pair(bf16* base, int64_t ld, int64_t row, int c0, int fg, const float (&v0)[4], const float (&v1)[4]) {
    const auto s0 = __builtin_amdgcn_permlane16_swap(m_pk(v0[0], v0[1]), m_pk(v1[0], v1[1]), false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(m_pk(v0[2], v0[3]), m_pk(v1[2], v1[3]), false, false);
    typedef unsigned __attribute__((ext_vector_type(4))) u4;
    *(u4*)(base + row * ld + c0 + (fg & 1) * 16 + (fg >> 1) * 8) = (u4){s0[0], s1[0], s0[1], s1[1]};
}
// the matching 16-byte load of an operand laid out like the output (residual gradient): returns the values of tiles j0 / j0 + 1 in accumulator order
__device__ __forceinline__ void m_load_pair(const bf16* base, int64_t ld, int64_t row, int c0, int fg, float (&x0)[4], float (&x1)[4]) {
    const uint4 z = *(const uint4*)(base + row * ld + c0 + (fg & 1) * 16 + (fg >> 1) * 8);
    const auto s0 = __builtin_amdgcn_permlane16_swap(z.x, z.z, false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(z.y, z.w, false, false);
    x0[0] = m_lo(s0[0]); x0[1] = m_hi(s0[0]); x0[2] = m_lo(s1[0]); x0[3] = m_hi(s1[0]);
    x1[0] = m_lo(s0[1]); x1[1] = m_hi(s0[1]); x1[2] = m_lo(s1[1]); x1[3] = m_hi(s1[1]);
}

template <int C, int HC, int TT, bool BWD, int WPC, bool WH = true, int NTH = 512>
__global__ __launch_bounds__(NTH, WPC * NTH / 256) void mlp_panel_k(MlpArgs a, int npanel) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HID = 4 * C, NCH = HID / HC;
    constexpr int P1 = 2 * C, P2 = 2 * HC;                 // pitches: [HC rows][C] tiles (W1 / W2^T chunk), [C rows][HC] tile (W2 / W1^T column slice)
    constexpr int T1 = HC * P1, T2 = C * P2;               // tile bytes (equal: 2 C HC)
    constexpr int BUF = (BWD ? 2 * T1 : T1) + T2;          // one chunk's weights
    constexpr int KS1 = C / 32, KS2 = HC / 32, J1 = HC / 16, O2 = C / 16;
    constexpr int BIAS_OFF = 2 * BUF;                      // fp32 biases behind the two weight buffers: b1 [4C] (+ b2 [C], forward)
    static_assert(HC % 32 == 0 && C % 32 == 0 && WPC * (2 * BUF + 4 * (HID + C)) <= 163840, "tile geometry");
    // (a bias load from global memory inside the chunk loop would make hipcc wait for EVERY outstanding load -- the weight prefetch
    //  included -- in front of its first use: the biases sit in LDS)
    float* bias1 = (float*)(smem + BIAS_OFF);
    float* bias2 = bias1 + HID;
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = a.M;

    MTile<HC, P1, NTH> s1a;     // W1 rows of the chunk
    MTile<HC, P1, NTH> s1b;     // backward: W2^T rows of the chunk
    MTile<C, P2, NTH> s2;            // W2 (forward) / W1^T (backward) column slice of the chunk
    auto fetch = [&](int c) {   // global -> registers
        s1a.load(a.W1 + (int64_t)c * HC * C, C, tid);
        if (BWD) s1b.load(a.W2 + (int64_t)c * HC * C, C, tid);
        s2.load((BWD ? a.W1T : a.W2) + (int64_t)c * HC, HID, tid);
    };
    auto commit = [&](int buf) {   // registers -> LDS
        char* b = smem + buf * BUF;
        s1a.store(b, tid);
        if (BWD) s1b.store(b + T1, tid);
        s2.store(b + (BWD ? 2 * T1 : T1), tid);
    };
    if ((int)blockIdx.x >= npanel) return;
    for (int i = threadIdx.x; i < HID; i += NTH) bias1[i] = a.b1[i];
    if (!BWD)
        for (int i = threadIdx.x; i < C; i += NTH) bias2[i] = a.b2[i];
    fetch(0);
    commit(0);
    __syncthreads();
    int cbuf = 0;
    for (int panel = blockIdx.x; panel < npanel; panel += gridDim.x) {
        const int64_t row0 = (int64_t)panel * ((NTH / 64) * TT * 16) + wave * (TT * 16);
        // token fragments (MFMA B port): row = token fr, 16 bytes at k = 32 ks + 8 fg; rows past M read row M - 1
        bf16x8_t xf[TT][KS1], df[BWD ? TT : 1][BWD ? KS1 : 1];
        int64_t rowc[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int64_t r = row0 + t * 16 + fr;
            rowc[t] = r < M ? r : M - 1;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                xf[t][ks] = *(const bf16x8_t*)(a.X + rowc[t] * C + ks * 32 + fg * 8);
                if constexpr (BWD) df[t][ks] = *(const bf16x8_t*)(a.D + rowc[t] * C + ks * 32 + fg * 8);
            }
        }
        f32x4_t acc2[TT][O2];
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int o = 0; o < O2; ++o) acc2[t][o] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < NCH; ++c) {
            const bool last = c + 1 == NCH && panel + (int)gridDim.x >= npanel;      // nothing follows this chunk
            if (!last) fetch(c + 1 == NCH ? 0 : c + 1);      // next chunk's weights (the next panel starts at chunk 0 again): in flight under this chunk
            const char* w = smem + cbuf * BUF;
            // ---- first product(s): [TT x 16 tokens] x [HC hidden], K = C
            f32x4_t acc1[TT][J1], accd[BWD ? TT : 1][BWD ? J1 : 1];
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int j = 0; j < J1; ++j) {
                    acc1[t][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                    if constexpr (BWD) accd[t][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                }
            // weight fragments are read a group ahead of the MFMAs that use them (left to itself hipcc issues read -> wait -> 2 MFMAs -> read
            // ..., one exposed LDS latency per pair of MFMAs with only two waves per SIMD to cover it)
            {
                constexpr int NF = BWD ? 2 * J1 : J1;        // fragments per k-step: W1 (and W2^T) rows of every 16-column tile
                bf16x8_t wf[2][NF];
                auto rd = [&](int ks, bf16x8_t (&f)[NF]) {
#pragma unroll
                    for (int j = 0; j < J1; ++j) {
                        f[j] = m_frag<P1>(w, j * 16, ks, fr, fg);
                        if constexpr (BWD) f[J1 + j] = m_frag<P1>(w + T1, j * 16, ks, fr, fg);
                    }
                };
                rd(0, wf[0]);
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    if (ks + 1 < KS1) rd(ks + 1, wf[(ks + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < J1; ++j) {
#pragma unroll
                        for (int t = 0; t < TT; ++t) acc1[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][j], xf[t][ks], acc1[t][j], 0, 0, 0);
                        if constexpr (BWD) {
#pragma unroll
                            for (int t = 0; t < TT; ++t) accd[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][J1 + j], df[t][ks], accd[t][j], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- activation (forward) / d(pre-activation) (backward): written out, and kept as the next product's operand
            bf16x8_t hb[TT][KS2];
#pragma unroll
            for (int jp = 0; jp < J1 / 2; ++jp) {
                const f32x4_t bA = *(const f32x4_t*)(bias1 + c * HC + (2 * jp) * 16 + 4 * fg);
                const f32x4_t bB = *(const f32x4_t*)(bias1 + c * HC + (2 * jp + 1) * 16 + 4 * fg);
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    float v0[4], v1[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p0 = acc1[t][2 * jp][r] + bA[r], p1 = acc1[t][2 * jp + 1][r] + bB[r];
                        if constexpr (MLP_X & 1) {
                            v0[r] = BWD ? accd[t][2 * jp][r] * p0 : p0;
                            v1[r] = BWD ? accd[t][2 * jp + 1][r] * p1 : p1;
                        } else if constexpr (BWD) {
                            v0[r] = accd[t][2 * jp][r] * dgelu_fast(p0);
                            v1[r] = accd[t][2 * jp + 1][r] * dgelu_fast(p1);
                        } else {
                            v0[r] = gelu_fast(p0);
                            v1[r] = gelu_fast(p1);
                        }
                    }
                    typedef unsigned __attribute__((ext_vector_type(4))) u4;
                    hb[t][jp] = __builtin_bit_cast(bf16x8_t, (u4){m_pk(v0[0], v0[1]), m_pk(v0[2], v0[3]), m_pk(v1[0], v1[1]), m_pk(v1[2], v1[3])});
                    if constexpr (WH && !(MLP_X & 2)) m_store_pair(a.H, HID, rowc[t], c * HC + jp * 32, fg, v0, v1);      // (WH = false: inference, no activation copy)
                }
            }
            // ---- second product: [TT x 16 tokens] x [C outputs] += operand x (W2 / W1^T column slice), K = HC
            const char* w3 = w + (BWD ? 2 * T1 : T1);
            {
                constexpr int GO = 4, NG = KS2 * (O2 / GO);  // groups of four output tiles
                bf16x8_t wf[2][GO];
                auto rd = [&](int g, bf16x8_t (&f)[GO]) {
                    const int ks = g / (O2 / GO), o0 = (g % (O2 / GO)) * GO;
#pragma unroll
                    for (int i = 0; i < GO; ++i) f[i] = m_frag_perm<P2>(w3, (o0 + i) * 16, ks, fr, fg);
                };
                rd(0, wf[0]);
#pragma unroll
                for (int g = 0; g < ((MLP_X & 4) ? 1 : NG); ++g) {
                    if (g + 1 < NG) rd(g + 1, wf[(g + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    const int ks = g / (O2 / GO), o0 = (g % (O2 / GO)) * GO;
#pragma unroll
                    for (int i = 0; i < GO; ++i)
#pragma unroll
                        for (int t = 0; t < TT; ++t) acc2[t][o0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g & 1][i], hb[t][ks], acc2[t][o0 + i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!last) commit(cbuf ^ 1);         // the other buffer was last read in the previous chunk: every wave is past that chunk's barrier
            __syncthreads();
            cbuf ^= 1;
        }
        // ---- output: y = acc2 + b2 (forward) / dx = acc2 + g (backward)
#pragma unroll
        for (int op = 0; op < O2 / 2; ++op) {
            f32x4_t bA = {0.f, 0.f, 0.f, 0.f}, bB = {0.f, 0.f, 0.f, 0.f};
            if (!BWD) {
                bA = *(const f32x4_t*)(bias2 + (2 * op) * 16 + 4 * fg);
                bB = *(const f32x4_t*)(bias2 + (2 * op + 1) * 16 + 4 * fg);
            }
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                float v0[4], v1[4], g0[4] = {0.f, 0.f, 0.f, 0.f}, g1[4] = {0.f, 0.f, 0.f, 0.f};
                if (BWD && a.G) m_load_pair(a.G, C, rowc[t], op * 32, fg, g0, g1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0[r] = acc2[t][2 * op][r] + bA[r] + g0[r];
                    v1[r] = acc2[t][2 * op + 1][r] + bB[r] + g1[r];
                }
                m_store_pair(a.Y, C, rowc[t], op * 32, fg, v0, v1);
            }
        }
    }
}

static int mlp_cus() {
    static const int n = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        return v > 0 ? v : 256;
    }();
    return n;
}

template <int C, int HC, int TT, bool BWD, int WPC, bool WH = true, int NTH = 512>
static void mlp_launch(const MlpArgs& a, hipStream_t stream) {
    constexpr int BUF = ((BWD ? 2 : 1) * HC * 2 * C) + C * 2 * HC;
    constexpr int LDS = 2 * BUF + 4 * (4 * C + C);
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)mlp_panel_k<C, HC, TT, BWD, WPC, WH, NTH>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        return true;
    }();
    (void)attr;
    const int npanel = (int)cdiv(a.M, (NTH / 64) * TT * 16);
    const int grid = npanel < WPC * mlp_cus() ? npanel : WPC * mlp_cus();
    hipLaunchKernelGGL((mlp_panel_k<C, HC, TT, BWD, WPC, WH, NTH>), dim3(grid), dim3(NTH), LDS, stream, a, npanel);
}

static int mlp_check(const char* fn, int M, int C, const void* p0, const void* p1, const void* p2, const void* p3) {
    MV_CHECK_ARG(M > 0 && C == 128, "%s: the fused MLP covers C = 128 (got M = %d, C = %d)", fn, M, C);
    MV_CHECK_ARG(p0 && p1 && p2 && p3, "%s: null pointer", fn);
    MV_CHECK_ARG(((((uintptr_t)p0) | ((uintptr_t)p1) | ((uintptr_t)p2) | ((uintptr_t)p3)) & 15) == 0, "%s: operands must be 16-byte aligned", fn);
    return 0;
}

// (C = 256 -- Swin stage 1 -- was built and parity-tested in round 3 and lost to the unfused products on both passes (259 vs 215 us,
//  368 vs 215 us): no longer instantiated, tools/experiments/README.md; the kernel template still takes it.)
extern "C" int mvuld_mlp_fused_supported(int C) { return C == 128 ? 1 : 0; }

extern "C" int mvuld_mlp_fused_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* h, void* y, int M, int C,
                                   hipStream_t stream) {
    if (mlp_check("mlp_fused_fwd", M, C, x, w1, w2, y)) return 1;
    MV_CHECK_ARG(b1 && b2 && (((uintptr_t)h | (uintptr_t)b1 | (uintptr_t)b2) & 15) == 0, "mlp_fused_fwd: null / misaligned bias or activation buffer");
    MlpArgs a{(const bf16*)x, nullptr, nullptr, (const bf16*)w1, b1, (const bf16*)w2, b2, nullptr, (bf16*)h, (bf16*)y, M};
    // C = 128: two 256-thread workgroups per CU (their chunk barriers are independent, so one's GELU phase overlaps the other's MFMA phase:
    // 263 vs 290 us for one 512-thread workgroup; tools/bench_mlp.py); C = 256: one 512-thread workgroup (registers)
    if (!h) mlp_launch<128, 64, 2, false, 2, false, 256>(a, stream);
    else mlp_launch<128, 64, 2, false, 2, true, 256>(a, stream);
    MV_LAUNCH_CHECK("mlp_fused_fwd");
    return 0;
}

extern "C" int mvuld_mlp_fused_bwd(const void* x, const void* dy, const void* g, const void* w1, const float* b1, const void* w2t, const void* w1t,
                                   void* dh, void* dx, int M, int C, hipStream_t stream) {
    if (mlp_check("mlp_fused_bwd", M, C, x, dy, w1, w2t)) return 1;
    MV_CHECK_ARG(b1 && w1t && dh && dx && (((uintptr_t)w1t | (uintptr_t)dh | (uintptr_t)dx | (uintptr_t)b1 | (uintptr_t)g) & 15) == 0,
                 "mlp_fused_bwd: null / misaligned operand");
    MlpArgs a{(const bf16*)x, (const bf16*)dy, (const bf16*)g, (const bf16*)w1, b1, (const bf16*)w2t, nullptr, (const bf16*)w1t, (bf16*)dh, (bf16*)dx, M};
    mlp_launch<128, 32, 2, true, 2, true, 256>(a, stream);
    MV_LAUNCH_CHECK("mlp_fused_bwd");
    return 0;
}
