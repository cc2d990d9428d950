// Persistent 256 x 256-tile NT GEMM for the step's tall products (M = tokens >> N):  C = epi(alpha * A . B^T + bias), bf16 in / bf16 out.
//
// Why another kernel: the 256 x 128 ring of gemm.hip moves 85 FLOP per byte it pulls from L2 and tops out near the L2 -> LDS
// bandwidth (~12 TB/s observed = 1.0 PFLOP/s at 4096^3); a 256 x 256 tile is 128 FLOP per byte.  At one 8-wave workgroup per CU
// (128 KiB of LDS) the launch-per-tile form of that tile lost on this model's short contractions (K = 512 .. 768: 16-24 k-steps),
// because nothing overlapped a tile's prologue (first loads) and epilogue.  This version is persistent:
//   * grid = min(tiles, CUs); a workgroup walks tiles  b, b + G, b + 2G, ...  of an XCD-contiguous order (blocks b, b+8, ... share
//     an XCD, each XCD owns a contiguous run of tiles walked N-fastest, so an A row panel is pulled from HBM by one L2);
//   * ONE ring of four 32-deep LDS stages (A 256 x 32 + B 256 x 32 bf16 = 32 KiB each) filled by LDS-DMA
//     (global_load_lds_dwordx4, source-side XOR swizzle, no VGPR staging, no ds_write) runs ACROSS tiles: three k-steps are always in
//     flight behind counted `s_waitcnt vmcnt`, so the next tile's first stages land while this tile's epilogue runs;
//   * the epilogue uses no LDS (the ring is busy): the MFMA operands are swapped -- weights feed the A port, activations the B
//     port -- so each lane's four accumulator registers are four CONSECUTIVE output columns of one row; two v_permlane16_swap per
//     fragment pair turn them into 16-byte row segments (16 rows x 64 contiguous bytes per store instruction).  aux operands
//     (dGELU pre-activation, residual-gradient join) are fetched as the same 16-byte segments and un-swapped;
//   * GELU / dGELU use the 13-instruction erf of gemm_common.h (outputs are rounded to bf16 anyway).
// M and N tails: DMA rows are clamped, stores predicated (N % 8 == 0).  K % 32 == 0.
#include "gemm_common.h"
#include <stdlib.h>
#include <type_traits>
#include <mutex>

#define P_BN 256
#ifndef P256_X
#define P256_X 0      // timing experiments (tools/p256_variants.sh): bit 0 = epilogue without its stores, bit 1 = without GELU / dGELU math, bit 2 = non-temporal stores
#endif
#define P_BK 32
#ifndef P256_K64_DEFAULT
#define P256_K64_DEFAULT 1
#endif
#ifndef P256_TWO_PASS
#define P256_TWO_PASS 0      // 1: two-pass dGELU / residual-join epilogue (see the epilogue; measured: spills at 224 / 256-row tiles, so off)
#endif
#ifndef P256_EARLY_DEFAULT
#define P256_EARLY_DEFAULT 0
#endif
#ifndef P256_FP8_SCALED
#define P256_FP8_SCALED 1      // fp8 full-line ring on v_mfma_scale_f32_16x16x128_f8f6f4 (0: four v_mfma_f32_16x16x32_fp8_fp8 per stage and fragment pair)
#endif
#define P256_FP8_SCALED_MAX_NI 6
#ifndef P256_DYNAMIC_DEFAULT
#define P256_DYNAMIC_DEFAULT 0
#endif
#define P_STAGE_BYTES 32768
#define P_BIAS_OFF (4 * P_STAGE_BYTES)                 // two 1 KiB bias slices (256 fp32 columns), alternating per tile
#define P_LDS_BYTES (4 * P_STAGE_BYTES + 2048 + 64)

// byte offset of 16-byte chunk `ch` (0..3) of row `row` in a [256][32] bf16 tile (64-byte rows): slot XORed with the row group so
// that every 16-lane group of a ds_read_b128 fragment read covers all 64 banks (same image as gemm.hip's ring kernels)
__device__ __forceinline__ int p_off(int row, int ch) { return row * 64 + ((ch ^ ((4 - ((row >> 2) & 3)) & 3)) << 4); }

// the same for a [rows][64] bf16 tile (128-byte rows, chunks 0..7): chunk ^ ((row >> 1) & 7).  A ds_read_b128 is served in groups of
// 16 lanes = rows {0-3, 12-15} of chunk c and rows {4-11} of chunk c + 1 (or the complement): even rows take slots 0..7 of the 256-byte
// bank line, odd rows 8..15, and (c ^ {0, 1, 6, 7}) with ((c + 1) ^ {2, 3, 4, 5}) are eight different slots for every c.
__device__ __forceinline__ int p_off128(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef bf16 __attribute__((ext_vector_type(2))) bf16x2_t;
    bf16x2_t v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// 16-byte output store; P256_X bit 2 (timing experiment): non-temporal, so that the output stream does not displace the weights in L2
typedef unsigned __attribute__((ext_vector_type(4))) p_u32x4_t;
__device__ __forceinline__ void p_st16(void* p, unsigned a, unsigned b, unsigned c, unsigned d) {
    const p_u32x4_t v = {a, b, c, d};
    if (P256_X & 4) __builtin_nontemporal_store(v, (p_u32x4_t*)p);
    else *(p_u32x4_t*)p = v;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NI = 16-row accumulator fragments per wave along M: the tile is (32 * NI) x 256, NI = 8 -> 256 x 256.  The host picks NI per
// launch so that the tiles fill whole rounds of the persistent grid (a 25088 x 2048 product is 784 tiles of 256 rows = 3.06 rounds of
// 256 workgroups, but 1256 tiles of 160 rows = 4.9 rounds of 0.67 the length).
// FP8: the operands are OCP e4m3 bytes.  A 64-byte LDS row then holds 64 k instead of 32, the DMA / swizzle / fragment-read byte
// geometry is unchanged; a lane's 16-byte chunk feeds TWO v_mfma_f32_16x16x32_fp8_fp8 (its low and high 8 bytes: the same k subset
// on both operands, so any split is a valid contraction order).  Half the L2 -> LDS bytes and LDS reads per FLOP.
typedef long __attribute__((ext_vector_type(2))) i64x2_t;
// NS = ring stages: 4 (32 KiB stages at any NI) or 5 (NI <= 7: stages of (2 NI + 16) KiB, 152 KiB at NI = 7) -- one more k-step between an
// epilogue's burst of stores and the first operand load that has to wait for them (vmcnt retires in issue order).
// PP = ping-pong schedule of the two wave rows.  Waves w and w + 4 (wr = 0 / 1, same wc) share a SIMD.  With one barrier per k-step both
// reach the 12 fragment reads together and then the 32 MFMAs together: the matrix pipe idles through every read phase (the compiled loop
// is barrier, DMA issue, 12 ds_read_b128, lgkmcnt(0), 32 MFMAs: ~0.75 us per step against 0.43 us of MFMA time).  PP splits a k-step
// into two barrier epochs and runs row 1 one epoch behind row 0, so one wave of every SIMD reads its fragments (and issues its share of
// the DMA) while the other one owns the matrix pipe.  Same registers, same ring, same vmcnt accounting: every wave still issues one
// k-step of DMA per k-step, right after the barrier that opens its read epoch, and step c has been awaited by EVERY wave before the
// barrier that opens row 0's read epoch of c (row 0 waits in front of it as before, row 1 waits for c at the end of its read epoch of
// c - 1, which closes with that same barrier).  A tile's epilogue is not staggered: row 1 takes one barrier alone in front of the k-loop,
// row 0 one behind it, so both rows meet again at the epilogue (bias slices, store accounting and the tile walk are unchanged).
// K64 = 64-deep ring stages fetched as FULL 128-byte lines (bf16, PP only).  A 32-deep stage takes 64 bytes of every operand row per k-step:
// half a cache line per request, every line requested twice from L2 -- the L2 -> LDS path then tops out near 11 TB/s chip-wide (0.75 us per
// 32-deep step of a 256 x 256 tile, whatever the schedule of the waves: the ping-pong loop gained 3 %).  Here a DMA instruction fills
// 8 rows x 128 bytes (lane l -> row l >> 3, slot l & 7, source chunk = slot ^ ((row >> 1) & 7): every 16-lane group of a ds_read_b128 fragment
// read still covers all 64 banks), a stage is (32 NI + 256) rows x 128 bytes and is consumed as two 32-deep sub-steps (the second one's
// chunk index is the first one's ^ 4: byte offset ^ 64), so the contraction order -- and every output bit -- is that of the 32-deep ring.
// Stages: 2 at NI >= 6 (2 x 64 KiB at NI = 8), 3 at NI <= 5 (3 x 52 KiB at NI = 5).
// EI = early issue (K64 only).  vmcnt retires in issue order, so an operand load issued after an epilogue's stores is only counted as landed
// once those stores are acknowledged.  The two-stage full-line ring issues the next tile's second step after the stores (at that tile's
// first barrier) and waits for it one 64-deep step later: the stores get one step (~1 us) to drain instead of the 32-deep ring's three.
// With EI every wave issues that step BEFORE its epilogue -- right after the last hand-over barrier of the tile, when the stage the
// last step occupied has been read by every wave -- and skips the issue at the next tile's first barrier.  The stores are then younger
// than every load in flight: the first NS steps of a tile may leave them pending (NS - 1 before), a K = 128 tile never waits for its
// predecessor's stores at all.  The DMA stream then runs up to two tiles ahead, hence three bias slices instead of two.
// NF = "N full" (host: N % 64 == 0; used for the aux-reading epilogues): every wave's 64 columns are either all inside the matrix or all
// outside, so a wave outside skips its epilogue and a wave inside stores UNCONDITIONALLY -- rows past M go to row M - 1, whose values they
// duplicate exactly (the DMA clamps the operand rows the same way).  hipcc cannot count stores behind the `row < M && col < N` branch and
// then waits for (nearly) all of a tile's stores in front of the last rows' aux values (section 9b item 15 of DESIGN.md); without the
// branch the waits are counted.
template <int EPI, int NI, bool FP8 = false, int NS = 4, bool PP = false, bool K64 = false, bool EI = false, bool NF = false>
__global__ __launch_bounds__(512, 1) void gemm_nt_bf16_p256(GemmArgs g, int tiles_m, int tiles_n, unsigned* dynp = nullptr, int dyn_all = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* glb_vp;
    constexpr int BM = 32 * NI, WM = 16 * NI;           // tile rows, rows per wave
    constexpr int RB = K64 ? 128 : 64;                  // bytes of every operand row in a stage
    constexpr int A_BYTES = K64 ? NI * 4096 : (NS == 4 ? 16384 : 2 * NI * 1024);      // A region of a stage (B follows: 256 rows)
    constexpr int STAGE = A_BYTES + 256 * RB;
    constexpr int BIAS_OFF = NS * STAGE;
    static_assert(K64 || NS == 4 || (NS == 5 && NI <= 7), "five stages only fit below 256 rows");
    static_assert(!K64 || (PP && (NS == 2 || (NS == 3 && NI <= 5))), "128-byte stages: ping-pong, 2 stages (3 at <= 160 rows)");
    static_assert(!EI || K64, "early issue belongs to the full-line ring");
    constexpr int NBS = EI ? 3 : 2;                     // bias slices
    static_assert(NS * STAGE + NBS * 1024 + 64 <= 163840, "ring does not fit the LDS");
    constexpr int CLAIM_OFF = BIAS_OFF + NBS * 1024;    // two claimed tile positions (dynamic tile walk), alternating per tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;            // 2 x 4 waves, WM (m) x 64 (n) each
    const int fr = lane & 15, fg = lane >> 4;
    const int nt = tiles_m * tiles_n;
    const int G = gridDim.x, bx = blockIdx.x;
    constexpr int ES = FP8 ? 1 : 2;                      // bytes per operand element; a k-step is always 64 bytes of every row
    const int nk = g.K * ES / RB;
    // DYNAMIC TILE WALK (dynp != nullptr; host: K64 ring, nt > G, G % 8 == 0, nk >= NS + 1).  The static walk gives workgroup b the tiles
    // b, b + G, ...: a workgroup that starts late -- its CU was held by another kernel: a collective's channels, another stream's
    // persistent grid -- still owns its whole share and the launch lasts two shares.  Here only the first tile is fixed (position b: no
    // round trip in front of the first loads); every further one is claimed from a counter of the workgroup's XCD (positions of residue
    // b & 7 are that XCD's contiguous run of tiles, so an A row panel is still pulled from HBM by one L2): a late workgroup claims what is
    // left, usually nothing.  (dyn_all: the first tile is claimed too -- one exposed round trip per launch, ~2 us, and no tile is owned by
    // a workgroup that has not started: what a launch beside a resident collective wants, section 7 of DESIGN.md.)  The claim is a SCALAR atomic (s_atomic_add ... glc: tools/microbench/scalar_atomic.hip shows gfx950 executes
    // them, coherently across XCDs): its ticket returns into an SGPR under lgkmcnt, so it neither enters the vector-memory queue -- a
    // returning vector atomic would sit in vmcnt behind a tile's stores and in front of the next loads -- nor needs a register the
    // epilogue could spill.  Wave 0 issues the claim for tile k + 1 when tile k - 1's epilogue is done, publishes the position through LDS
    // in front of the first hand-over barrier of tile k (its lgkmcnt(0) after the fragment reads has retired the ticket), and every wave
    // reads it when its DMA stream leaves tile k, at k-step nk - NS >= 1.  One failing claim per workgroup ends its walk; the G-th workgroup
    // to finish zeroes the counters for the next launch on the stream.  Same tiles, same arithmetic per tile: bit-identical output.
    const bool dyn = K64 && dynp != nullptr;
    int my_tiles = bx < nt ? (dyn ? 1 : (nt - bx + G - 1) / G) : 0;
    int total = my_tiles * nk;
    int p_iss = bx, p_cmp = bx;                          // tile positions of the DMA stream / of the compute stream
    // The ticket is in flight across compiled code, so it must not be an asm OUTPUT: hipcc copies an output operand into the variable's
    // register right behind the statement, i.e. before the value has landed (first build's ISA, with "=s" and with "+s" alike).  It lives in
    // s101, which hipcc never allocates on this target (it reports s100 / s101 as reserved; naming it as a clobber makes the kernel
    // descriptor cover it) and which only these statements touch; it becomes a C++ value behind the lgkmcnt(0) that retires it.
    bool claim_fly = false;
    auto claim_issue = [&]() {                           // wave 0 only
        asm volatile("s_mov_b32 s101, 1\n\ts_atomic_add s101, %0, 0x0 glc" ::"s"(dynp + (bx & 7)) : "memory", "s101");
        claim_fly = true;
    };
    const unsigned claim_a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + CLAIM_OFF);
    const char* A = (const char*)g.A;
    const char* B = (const char*)g.B;
    bf16* C = (bf16*)g.C;
    bf16* aux = (bf16*)g.aux;

    auto tile_of = [&](int p, int& m0, int& n0) {        // tile at position p of the XCD-contiguous order
        const int q = nt >> 3, r = nt & 7, x = p & 7, i = p >> 3;
        const int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
        m0 = (t / tiles_n) * BM;
        n0 = (t % tiles_n) * P_BN;
    };

    // ---- DMA issue stream (runs three k-steps ahead of the compute stream, across tile boundaries)
    // one instruction fills 1 KiB = 16 tile rows in lane order (lane l -> row l >> 2, slot l & 3): the lane fetches the chunk the
    // swizzle keeps in its slot.  The A tile is 2 * NI pieces: wave w issues piece w and, if it exists, piece w + 8; the B tile is
    // 16 pieces: wave w issues pieces 2w, 2w + 1.  So a wave has 3 or 4 operations per k-step in flight (the counted waits below).
    // K64: the A tile is 4 * NI pieces of 8 rows: wave w issues pieces w, w + 8, ... (ALO or ALO + 1 of them); the B tile is 32 pieces:
    // wave w issues pieces 4w .. 4w + 3.
    constexpr int ALO = (4 * NI) / 8, AHI = (4 * NI + 7) / 8;
    const bool a2 = K64 ? (wave + 8 * ALO < 4 * NI) : (wave + 8 < 2 * NI);        // wave-uniform: this wave issues the extra A piece
    constexpr int OPS_LO = K64 ? ALO + 4 : 3, OPS_HI = K64 ? ALO + 5 : 4;          // DMA operations per k-step of a wave without / with it
    constexpr int NPA = K64 ? AHI : 2, NPB = K64 ? 4 : 2;
    const char* ga[K64 ? 1 : 2];
    const char* gb[K64 ? 1 : 2];
    unsigned fa_off[NPA], fb_off[NPB];                   // K64: 32-bit byte offsets from A / B (the host checks that they fit)
    int iss_ord = 0, iss_kt = 0, issued = 0;
    auto setup_ptrs = [&](int ord) {
        int m0, n0;
        tile_of(p_iss, m0, n0);
        if constexpr (K64) {
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const int rowa = (wave + 8 * i) * 8 + (lane >> 3);
                fa_off[i] = (unsigned)min(m0 + rowa, g.M - 1) * (unsigned)(g.lda * ES) + ((lane & 7) ^ ((rowa >> 1) & 7)) * 16;
            }
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                const int rowb = (wave * 4 + i) * 8 + (lane >> 3);
                fb_off[i] = (unsigned)min(n0 + rowb, g.N - 1) * (unsigned)(g.ldb * ES) + ((lane & 7) ^ ((rowb >> 1) & 7)) * 16;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rowa = (wave + 8 * i) * 16 + (lane >> 2);
                const int rowb = (wave * 2 + i) * 16 + (lane >> 2);
                ga[i] = A + (int64_t)min(m0 + rowa, g.M - 1) * g.lda * ES + ((lane & 3) ^ ((4 - ((rowa >> 2) & 3)) & 3)) * 16;
                gb[i] = B + (int64_t)min(n0 + rowb, g.N - 1) * g.ldb * ES + ((lane & 3) ^ ((4 - ((rowb >> 2) & 3)) & 3)) * 16;
            }
        }
        // the tile's 256 bias columns ride the same DMA queue (a VGPR load in the epilogue would make hipcc drain it with
        // vmcnt(0)); older than the tile's first operand stage, so the wait that retires that stage retires it too
        if (g.bias && wave == 0) {
            const int c = n0 + 4 * lane;
            __builtin_amdgcn_global_load_lds((glb_vp)(g.bias + (c < g.N ? c : 0)), (lds_vp)(smem + BIAS_OFF + (ord % NBS) * 1024), 16, 0, 0);
        }
    };
    auto issue_one = [&]() {
        if (issued < total) {
            char* st = smem + (issued % NS) * STAGE;
            const int k0 = iss_kt * RB;                  // bytes
            if constexpr (K64) {
                const char* Ak = A + k0;                 // scalar bases: the loads take the  saddr + 32-bit voffset  form
                const char* Bk = B + k0;
#pragma unroll
                for (int i = 0; i < NPA; ++i)
                    if (i < ALO || a2) __builtin_amdgcn_global_load_lds((glb_vp)(Ak + fa_off[i]), (lds_vp)(st + (wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
                for (int i = 0; i < NPB; ++i)
                    __builtin_amdgcn_global_load_lds((glb_vp)(Bk + fb_off[i]), (lds_vp)(st + A_BYTES + (wave * 4 + i) * 1024), 16, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds((glb_vp)(ga[0] + k0), (lds_vp)(st + wave * 1024), 16, 0, 0);
                if (a2) __builtin_amdgcn_global_load_lds((glb_vp)(ga[1] + k0), (lds_vp)(st + (wave + 8) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_vp)(gb[0] + k0), (lds_vp)(st + A_BYTES + (wave * 2) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_vp)(gb[1] + k0), (lds_vp)(st + A_BYTES + (wave * 2 + 1) * 1024), 16, 0, 0);
            }
            ++issued;
            if (++iss_kt == nk) {
                iss_kt = 0;
                ++iss_ord;
                if (dyn) {
                    int p;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(p) : "v"(claim_a + (iss_ord & 1) * 4) : "memory");
                    p = __builtin_amdgcn_readfirstlane(p);
                    if (p >= 0) {
                        p_iss = p;
                        ++my_tiles;
                        total += nk;
                        setup_ptrs(iss_ord);
                    }
                } else if (iss_ord < my_tiles) {
                    p_iss = bx + iss_ord * G;
                    setup_ptrs(iss_ord);
                }
            }
        }
    };
    const int dyn_base = dyn_all ? 0 : G;                // position of ticket t: (b & 7) + dyn_base + 8 t
    if (dyn && dyn_all) {
        if (wave == 0) {
            claim_issue();
            unsigned ticket;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, s101" : "=s"(ticket)::"s101", "memory");
            const int pn = (bx & 7) + 8 * (int)ticket;
            asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(claim_a), "v"(pn < nt ? pn : -1) : "memory");
            claim_fly = false;
        }
        __builtin_amdgcn_s_barrier();
        int p;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(p) : "v"(claim_a) : "memory");
        p = __builtin_amdgcn_readfirstlane(p);
        p_iss = p_cmp = p < 0 ? 0 : p;
        if (p < 0) my_tiles = total = 0;
    }
    if (dyn && wave == 0 && my_tiles > 0) claim_issue();     // the second tile's claim: in flight under the first tile's prologue
    if (my_tiles > 0) setup_ptrs(0);
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) issue_one();

    // fragment byte offsets inside a stage: weights (MFMA A port) 4 x 16 columns, activations (B port) NI x 16 rows
    int oa[NI], ob[4];
#pragma unroll
    for (int i = 0; i < NI; ++i) oa[i] = K64 ? p_off128(wr * WM + i * 16 + fr, fg) : p_off(wr * WM + i * 16 + fr, fg);
#pragma unroll
    for (int j = 0; j < 4; ++j) ob[j] = A_BYTES + (K64 ? p_off128(wc * 64 + j * 16 + fr, fg) : p_off(wc * 64 + j * 16 + fr, fg));

    // stores one epilogue leaves in flight (known exactly only for a wave whose sub-tile is interior: every store executes)
    constexpr int ST1 = 2 * NI, ST2 = 4 * NI, ST3 = 6 * NI;
    int pend = 0;
    const bool QE = FP8 && (EPI == EPI_GELU || EPI == EPI_GELU_DG) && g.q_out != nullptr;      // workgroup-uniform
    float amax_l = 0.f;
    int cs = 0;                                          // compute step, counted across tiles (ring stage = cs & 3)
    // Step xcs (k-step xkt of its tile) has landed once at most the operations issued after it are still in flight: normally the two
    // younger k-steps (3 or 4 loads each); in the first three steps after an epilogue also that epilogue's stores, which were issued
    // between step xcs's loads and now -- counting them lets the stores drain under the MFMAs instead of in front of them.  vmcnt
    // retires in issue order, so from the fourth step on the stores are older than the awaited loads and must be complete.
    // Precondition (both schedules): this wave has issued the DMA up to step xcs + NS - 2 and nothing younger.
    // EI: `prev` = the tile follows another one of this workgroup, whose end issued this tile's step NS - 1 ahead of its stores
    auto wait_step = [&](int xkt, int xcs, bool prev) {
        const int younger = total - 1 - xcs;
        constexpr int AH = NS - 2;                       // k-steps allowed to stay in flight behind the awaited one
        if constexpr (EI) {
            int steps = AH + ((prev && xkt == 0) ? 1 : 0);
            if (steps > younger) steps = younger;
            const int st = (prev && xkt <= NS - 1) ? pend : 0;
#define P_W(S, P)                                                                                               \
    do {                                                                                                        \
        constexpr int HI = (S) * OPS_HI + (P) > 63 ? 63 : (S) * OPS_HI + (P);                                    \
        constexpr int LO = (S) * OPS_LO + (P) > 63 ? 63 : (S) * OPS_LO + (P);                                    \
        if (a2) wait_vm<HI>(); else wait_vm<LO>();                                                              \
    } while (0)
#define P_WS(S)                                                                                                 \
    do {                                                                                                        \
        if (st == ST1) P_W(S, ST1);                                                                             \
        else if (st == ST2) P_W(S, ST2);                                                                        \
        else if (FP8 && st == ST3) P_W(S, ST3);                                                                 \
        else P_W(S, 0);                                                                                         \
    } while (0)
            if (steps <= 0) P_WS(0);
            else if (steps == 1) P_WS(1);
            else P_WS(2);
#undef P_WS
#undef P_W
            return;
        }
        if (younger >= AH) {
            if (xkt < NS - 1 && pend == ST1) { if (a2) wait_vm<OPS_HI * AH + ST1>(); else wait_vm<OPS_LO * AH + ST1>(); }
            else if (xkt < NS - 1 && pend == ST2) { if (a2) wait_vm<OPS_HI * AH + ST2>(); else wait_vm<OPS_LO * AH + ST2>(); }
            else if (FP8 && xkt < NS - 1 && pend == ST3) { if (a2) wait_vm<OPS_HI * AH + ST3>(); else wait_vm<OPS_LO * AH + ST3>(); }
            else if (a2) wait_vm<OPS_HI * AH>();
            else wait_vm<OPS_LO * AH>();
        } else if (younger == 2) {                       // NS == 5 only
            if (a2) wait_vm<2 * OPS_HI>(); else wait_vm<2 * OPS_LO>();
        } else if (younger == 1) {
            if (a2) wait_vm<OPS_HI>(); else wait_vm<OPS_LO>();
        } else {
            wait_vm<0>();
        }
    };
    for (int ord = 0; ord < my_tiles; ++ord) {
        f32x4_t acc[NI][4];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if constexpr (K64) {
            // opaque zeros: folded into the first MFMAs (C = 0) the peeled first k-step accumulates out of place, and its two MFMA
            // blocks then hold two accumulator sets: 256 registers and spills whose reloads drain the DMA queue (vmcnt(0))
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
        }
        const bool prev = EI && ord > 0;
        if (PP && wr == 1) {                             // row 1 falls one barrier epoch behind row 0
            wait_step(0, cs, prev);
            __builtin_amdgcn_s_barrier();
        }
        for (int kt = 0; kt < nk; ++kt, ++cs) {
            if (!PP || wr == 0) wait_step(kt, cs, prev);
            __builtin_amdgcn_s_barrier();                // step cs visible to every wave; every wave is done reading step cs - 1
            if (!(prev && kt == 0)) issue_one();         // step cs + NS - 1 refills the stage step cs - 1 occupied (EI: issued at the last tile's end)
            const char* st = smem + (cs % NS) * STAGE;
            if constexpr (K64 && FP8 && (P256_FP8_SCALED != 0) && NI <= P256_FP8_SCALED_MAX_NI) {
                // e4m3 on the block-scaled instruction: v_mfma_scale_f32_16x16x128_f8f6f4 takes 32 bytes of every operand row per lane and runs at
                // twice the bf16 rate (the non-scaled 16x16x32_fp8_fp8 form runs AT the bf16 rate: four of them, 64 cycles, for what this one does
                // in 32).  A 128-byte stage row is exactly one instruction's K: the lane takes the 16-byte chunk `fg` of BOTH 64-byte halves -- the
                // two reads the sub-steps below make -- as one 8-register operand; weights and activations split K the same way, so it is a valid
                // contraction (only the fp32 summation order differs from the non-scaled form).  Unit block scales (E8M0 127 in every byte); the
                // per-tensor scales stay folded into alpha.  ONE read epoch and ONE matrix epoch per stage: two barriers instead of four, row 1 still
                // one epoch behind row 0 (it reads stage c while row 0 multiplies it, and waits for its share of c + 1 before the barrier that
                // opens row 0's read epoch of c + 1).
                typedef int __attribute__((ext_vector_type(8))) i32x8_t;
                typedef int __attribute__((ext_vector_type(4))) i32x4_t;
                // Registers: 8 per fragment.  Up to 192 rows all NI + 4 fragments are read in the read epoch (80 registers beside 96 accumulators);
                // at 224 / 256 rows that spills (first build: 47 / 69 VGPRs), and reading half of the activation fragments behind the first half's
                // MFMAs is a RACE (the stage is handed back to the DMA by the barrier that ends the read epoch: row 0 refills it while row 1 is
                // still in its matrix epoch -- caught by test_gemm_nt_fp8_vs_dequantised_product).  So: this path up to 192 rows, which is what the
                // host picks for e4m3 products; taller tiles (forced by mvuld_set_gemm_p256_rows) keep the non-scaled sub-steps below.
                constexpr int HA = NI;
                i32x8_t fa[HA], fb[4];
                auto rd = [&](int off) {
                    const i32x4_t lo = *(const i32x4_t*)(st + off), hi = *(const i32x4_t*)(st + (off ^ 64));
                    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                };
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = rd(ob[j]);
#pragma unroll
                for (int i = 0; i < HA; ++i) fa[i] = rd(oa[i]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (wr == 1 && kt + 1 < nk) wait_step(kt + 1, cs + 1, prev);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < HA; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                if constexpr (HA < NI) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = HA; i < NI; ++i) fa[i - HA] = rd(oa[i]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = HA; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[i - HA], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            if constexpr (K64) {
                // two 64-byte sub-steps (32 bf16 / 64 e4m3 deep), each a read epoch and a matrix epoch; row 1 runs one epoch behind row 0
                typedef typename std::conditional<FP8, i64x2_t, bf16x8_t>::type frag_t;
                frag_t fa[NI], fb[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1) __builtin_amdgcn_s_barrier();
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[j] = *(const frag_t*)(st + (ob[j] ^ (h * 64)));
#pragma unroll
                    for (int i = 0; i < NI; ++i) fa[i] = *(const frag_t*)(st + (oa[i] ^ (h * 64)));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (dyn && h == 0 && kt == 0 && wave == 0 && claim_fly) {      // the ticket has landed (lgkmcnt(0) above): publish tile ord + 1
                        unsigned ticket;
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, s101" : "=s"(ticket)::"s101", "memory");      // (the wait is in the statement: nothing can come between)
                        const int pn = (bx & 7) + dyn_base + 8 * (int)ticket;
                        asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(claim_a + ((ord + 1) & 1) * 4), "v"(pn < nt ? pn : -1) : "memory");
                        claim_fly = false;
                    }
                    if (h == 1 && wr == 1 && kt + 1 < nk) wait_step(kt + 1, cs + 1, prev);
                    __builtin_amdgcn_s_barrier();
                    // EI: the tile's last stage has now been read by every wave: the next tile's step NS - 1 goes out ahead of the stores
                    if (EI && h == 1 && wr == 1 && kt == nk - 1) issue_one();
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if constexpr (FP8) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fb[j][0], fa[i][0], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fb[j][1], fa[i][1], acc[i][j], 0, 0, 0);
                            } else {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                            }
                        }
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                continue;
            }
            constexpr int H0 = (NI + 1) / 2;
            // PP: the fragments are in registers before the barrier that hands the matrix pipe over (and the stage back to the DMA)
            auto pp_mid = [&]() {
                if constexpr (PP) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (wr == 1 && kt + 1 < nk) wait_step(kt + 1, cs + 1, prev);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if constexpr (!FP8) {
                bf16x8_t fa[NI], fb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8_t*)(st + ob[j]);
#pragma unroll
                for (int i = 0; i < H0; ++i) fa[i] = *(const bf16x8_t*)(st + oa[i]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = H0; i < NI; ++i) fa[i] = *(const bf16x8_t*)(st + oa[i]);
                pp_mid();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < H0; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = H0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            } else {
                i64x2_t fa[NI], fb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *(const i64x2_t*)(st + ob[j]);
#pragma unroll
                for (int i = 0; i < H0; ++i) fa[i] = *(const i64x2_t*)(st + oa[i]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = H0; i < NI; ++i) fa[i] = *(const i64x2_t*)(st + oa[i]);
                pp_mid();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < H0; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fb[j][0], fa[i][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fb[j][1], fa[i][1], acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = H0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fb[j][0], fa[i][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fb[j][1], fa[i][1], acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_s_setprio(0);
            }
        }
        if (PP && wr == 0) {
            __builtin_amdgcn_s_barrier();               // pairs with row 1's last hand-over: the rows meet again at the epilogue
            if (EI) issue_one();
        }

        // ---- epilogue, straight from the accumulators:  acc[i][j][r] = C[m0 + wr*WM + i*16 + fr][n0 + wc*64 + j*16 + 4*fg + r]
        int m0, n0;
        tile_of(dyn ? p_cmp : bx + ord * G, m0, n0);
        p_cmp = p_iss;                                   // dynamic walk: the DMA stream is in the next tile by now, and not beyond it (nk > NS)
        const int nw = n0 + wc * 64, mw = m0 + wr * WM;
        if constexpr (NF) {
            if (nw >= g.N) { pend = 0; continue; }       // wave-uniform: nothing of this wave's 64 columns exists (never wave 0: wc = 0)
        }
        // tile ord + 2's claim, published during tile ord + 1: issued here so that the whole epilogue covers its round trip (behind the last
        // store the ticket was still in flight at wave 0's first lgkmcnt(0) of the next tile: +3..5 % on the K = 512 products)
        if (dyn && wave == 0 && ord + 1 < my_tiles) claim_issue();
        float alpha = g.alpha;
        if (FP8) alpha *= (g.scale_a ? g.scale_a[0] : 1.0f) * (g.scale_b ? g.scale_b[0] : 1.0f);      // per-tensor dequantisation
        const float qinv = QE ? 1.0f / g.q_scale[0] : 0.f;
        float b4[4][4];
        if (g.bias) {
            // inline asm: a ds_read hipcc can see makes it drain the LDS-DMA queue first (s_waitcnt vmcnt(0) in front of every read)
            const unsigned ba = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + BIAS_OFF + (ord % NBS) * 1024 + (wc * 64 + 4 * fg) * 4);
            f32x4_t t0, t1, t2, t3;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
                         "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(ba) : "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) { b4[0][r] = t0[r]; b4[1][r] = t1[r]; b4[2][r] = t2[r]; b4[3][r] = t3[r]; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) b4[j][0] = b4[j][1] = b4[j][2] = b4[j][3] = 0.f;
        }
        // aux operands (dGELU pre-activation / residual-gradient join) as the same 16-byte row segments the stores use, fetched one
        // fragment row ahead of their use: a load issued right before its use would wait for every store in front of it
        constexpr bool AUX_IN = epi_reads_aux(EPI);
        constexpr bool IS_GELU = EPI == EPI_GELU || EPI == EPI_GELU_DG;
        uint4 z[NI][2];
        auto load_aux = [&](int i) {
            const int row = mw + i * 16 + fr;
            const int rowc = row < g.M ? row : g.M - 1;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const int col = nw + (2 * jp + (fg & 1)) * 16 + (fg >> 1) * 8;
                z[i][jp] = *(const uint4*)(aux + (int64_t)rowc * g.ldaux + (col < g.N ? col : 0));
            }
        };
        // AUX_IN epilogues in TWO passes (P256_TWO_PASS): first the aux operand is folded into the accumulators, in place, while nothing but
        // loads is in flight; then everything is packed and stored.  In one pass every aux fetch sat behind the stores of the rows before it,
        // and hipcc cannot count stores that sit behind the `row < M && col < N` branch: it assumed none had been issued and emitted
        // `s_waitcnt vmcnt(1)`, `vmcnt(0)` in front of the last rows' aux values -- the wave waited for (nearly) all of its own stores to be
        // acknowledged before it could finish the tile (ISA of the dGELU / residual-join variants).
        // Built with -DP256_TWO_PASS=1 the waits come out as intended (counted, loads only), but the second copy of the row loop costs ~12
        // registers: 256 VGPRs + 9 (224 rows) / 68 (256 rows) spilled, and the spill reloads are scratch loads, i.e. vmcnt(0) again.  Off
        // until the epilogue is cut differently (round-3 item: an interior-tile epilogue with unconditional stores lets hipcc count them).
        constexpr bool TWO_PASS = AUX_IN && (P256_TWO_PASS != 0);
        if (AUX_IN) { load_aux(0); load_aux(1); }
        if constexpr (TWO_PASS) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (i + 2 < NI) load_aux(i + 2);
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const int j0 = 2 * jp;
                    const uint4 zz = z[i][jp];
                    const auto s0 = __builtin_amdgcn_permlane16_swap(zz.x, zz.z, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(zz.y, zz.w, false, false);
                    float x[2][4];
                    x[0][0] = bf_lo(s0[0]); x[0][1] = bf_hi(s0[0]); x[0][2] = bf_lo(s1[0]); x[0][3] = bf_hi(s1[0]);
                    x[1][0] = bf_lo(s0[1]); x[1][1] = bf_hi(s0[1]); x[1][2] = bf_lo(s1[1]); x[1][3] = bf_hi(s1[1]);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float t = alpha * acc[i][j0 + h][r] + b4[j0 + h][r];
                            if (P256_X & 2) { if (EPI == EPI_MUL_DGELU) t *= x[h][r]; }
                            else if (EPI == EPI_MUL_DGELU) t *= dgelu_fast(x[h][r]);
                            else if (EPI == EPI_MUL_AUX) t *= x[h][r];
                            else if (EPI == EPI_ADD_AUX) t += x[h][r];
                            acc[i][j0 + h][r] = t;
                        }
                }
                __builtin_amdgcn_sched_barrier(0);       // keep the fetches two rows ahead: hoisted to the front they are 56 live registers more
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (AUX_IN && !TWO_PASS && i + 2 < NI) load_aux(i + 2);
            const int row = mw + i * 16 + fr;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const int j0 = 2 * jp;
                // after the swaps this lane owns columns  nw + (j0 + (fg & 1)) * 16 + (fg >> 1) * 8 .. + 7  of `row`
                const int col = nw + (j0 + (fg & 1)) * 16 + (fg >> 1) * 8;
                const bool ok = (P256_X & 1) ? false : (NF ? true : (row < g.M && col < g.N));
                const int srow = NF ? min(row, g.M - 1) : row;      // NF: rows past M duplicate row M - 1 (identical values)
                float x[2][4];
                if (AUX_IN && !TWO_PASS) {
                    const uint4 zz = z[i][jp];
                    const auto s0 = __builtin_amdgcn_permlane16_swap(zz.x, zz.z, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(zz.y, zz.w, false, false);
                    x[0][0] = bf_lo(s0[0]); x[0][1] = bf_hi(s0[0]); x[0][2] = bf_lo(s1[0]); x[0][3] = bf_hi(s1[0]);
                    x[1][0] = bf_lo(s0[1]); x[1][1] = bf_hi(s0[1]); x[1][2] = bf_lo(s1[1]); x[1][3] = bf_hi(s1[1]);
                }
                float v[2][4], pre[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = TWO_PASS ? acc[i][j0 + h][r] : alpha * acc[i][j0 + h][r] + b4[j0 + h][r];
                        pre[h][r] = t;
                        if (TWO_PASS) {}
                        else if (P256_X & 2) { if (EPI == EPI_MUL_DGELU) t *= x[h][r]; }
                        else if (EPI == EPI_GELU) t = gelu_fast(t);
                        else if (EPI == EPI_GELU_DG) gelu_dgelu_fast(t, t, pre[h][r]);      // aux <- gelu'(pre-activation)
                        else if (EPI == EPI_MUL_DGELU) t *= dgelu_fast(x[h][r]);
                        else if (EPI == EPI_MUL_AUX) t *= x[h][r];
                        else if (EPI == EPI_ADD_AUX) t += x[h][r];
                        v[h][r] = t;
                    }
                if (!FP8 || C) {                         // (only the fp8 form may run without a bf16 output: no branch in the bf16 kernels)
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pk_bf16(v[0][0], v[0][1]), pk_bf16(v[1][0], v[1][1]), false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pk_bf16(v[0][2], v[0][3]), pk_bf16(v[1][2], v[1][3]), false, false);
                    if (ok) p_st16(C + (int64_t)srow * g.ldc + col, s0[0], s1[0], s0[1], s1[1]);
                    if (P256_X & 1) asm volatile("" ::"v"(s0[0]), "v"(s1[0]), "v"(s0[1]), "v"(s1[1]));
                }
                if constexpr (FP8 && IS_GELU) {
                    if (QE) {
                        // the bf16-rounded activation, re-quantised for the next product: 4 + 4 columns per lane before the swap
                        unsigned d[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const unsigned p0 = pk_bf16(v[h][0], v[h][1]), p1 = pk_bf16(v[h][2], v[h][3]);
                            const float r0 = bf_lo(p0), r1 = bf_hi(p0), r2 = bf_lo(p1), r3 = bf_hi(p1);
                            if (row < g.M && nw + (j0 + h) * 16 + 4 * fg < g.N)
                                amax_l = fmaxf(amax_l, fmaxf(fmaxf(fabsf(r0), fabsf(r1)), fmaxf(fabsf(r2), fabsf(r3))));
                            unsigned w = 0;
                            w = __builtin_amdgcn_cvt_pk_fp8_f32(q_clamp(r0 * qinv), q_clamp(r1 * qinv), w, false);
                            w = __builtin_amdgcn_cvt_pk_fp8_f32(q_clamp(r2 * qinv), q_clamp(r3 * qinv), w, true);
                            d[h] = w;
                        }
                        const auto t = __builtin_amdgcn_permlane16_swap(d[0], d[1], false, false);
                        if (ok) *(uint2*)((char*)g.q_out + (int64_t)row * g.ldq + col) = make_uint2(t[0], t[1]);
                    }
                }
                if (IS_GELU && aux) {
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pk_bf16(pre[0][0], pre[0][1]), pk_bf16(pre[1][0], pre[1][1]), false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pk_bf16(pre[0][2], pre[0][3]), pk_bf16(pre[1][2], pre[1][3]), false, false);
                    if (ok) p_st16(aux + (int64_t)srow * g.ldaux + col, s0[0], s1[0], s0[1], s1[1]);
                    if (P256_X & 1) asm volatile("" ::"v"(s0[0]), "v"(s1[0]), "v"(s0[1]), "v"(s1[1]));
                }
            }
        }
        const bool interior = NF || ((mw + WM <= g.M) && (nw + 64 <= g.N));      // wave-uniform: every store above was issued
        pend = interior ? ST1 * ((C ? 1 : 0) + ((IS_GELU && aux) ? 1 : 0) + (QE ? 1 : 0)) : 0;
    }
    if (dyn && wave == 0) {
        unsigned done;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(done) : "s"(dynp + 8), "0"(1u) : "memory");
        if (done == (unsigned)G - 1 && lane < 9) __hip_atomic_store(dynp + lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (FP8 && (EPI == EPI_GELU || EPI == EPI_GELU_DG)) {
        if (QE) {
            const float m = wave_max(amax_l);
            if (lane == 0 && m > __uint_as_float(__hip_atomic_load(g.q_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
                atomicMax(g.q_amax, __float_as_uint(m));
        }
    }
}

static int p256_num_cus() {
    static const int n = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        return v > 0 ? v : 256;
    }();
    return n;
}

// Ping-pong schedule of the two wave rows (template parameter PP): 1 = on (default), 0 = the lockstep loop.  Initialised from
// MVULD_P256_PINGPONG; mvuld_set_gemm_p256_pingpong() overrides it (tests, A/B timing).  Results are bit-identical either way.
#include <atomic>
static std::atomic<int> g_p256_pp{-1};
static bool p256_pingpong() {
    int v = g_p256_pp.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("MVULD_P256_PINGPONG");
        v = e ? (atoi(e) != 0) : 1;
        g_p256_pp.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
extern "C" int mvuld_set_gemm_p256_pingpong(int on) {
    g_p256_pp.store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}

// 64-deep full-line stages (template parameter K64; bf16, K % 64 == 0): 1 = on (default), 0 = the 32-deep ring.  Initialised from MVULD_P256_K64;
// mvuld_set_gemm_p256_k64() overrides it (tests, A/B timing).  Results are bit-identical either way.
static std::atomic<int> g_p256_k64{-1};
static bool p256_k64() {
    int v = g_p256_k64.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("MVULD_P256_K64");
        v = e ? (atoi(e) != 0) : P256_K64_DEFAULT;
        g_p256_k64.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
extern "C" int mvuld_set_gemm_p256_k64(int on) {
    g_p256_k64.store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}

// ---- dynamic tile walk: knob and per-stream counter blocks
// MVULD_GEMM_DYNAMIC_TILES / mvuld_set_gemm_dynamic_tiles: 0 = static walk, 1 = dynamic walk wherever a launch has more tiles than workgroups,
// 2 = the same with the first tile claimed too.
// A block is 16 words (8 per-XCD counters, the finished-workgroup count), all zero between launches (the kernel's last workgroup restores
// them), one block per stream: launches of a stream are ordered, launches of different streams must not share tickets.  The pool is
// allocated once, outside any capture; a stream that is being captured gets the static walk (a replay could run beside the stream whose
// block the captured pointer names), and so does the 65th stream.
static std::atomic<int> g_p256_dyn{-1};
static bool p256_dynamic() {
    int v = g_p256_dyn.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("MVULD_GEMM_DYNAMIC_TILES");
        v = e ? atoi(e) : P256_DYNAMIC_DEFAULT;
        if (v < 0 || v > 2) v = P256_DYNAMIC_DEFAULT;
        g_p256_dyn.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
extern "C" int mvuld_set_gemm_dynamic_tiles(int mode) {
    MV_CHECK_ARG(mode >= 0 && mode <= 2, "set_gemm_dynamic_tiles: mode must be 0, 1 or 2");
    g_p256_dyn.store(mode, std::memory_order_relaxed);
    return 0;
}
unsigned* mvuld_dyn_counter_block(hipStream_t stream) {
    constexpr int NBLK = 64;
    static std::mutex mu;
    static unsigned* pool = nullptr;
    static bool failed = false;
    static hipStream_t owner[NBLK];
    static int used = 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (failed) return nullptr;
    if (!pool) {
        if (hipMalloc((void**)&pool, NBLK * 64) != hipSuccess || hipMemset(pool, 0, NBLK * 64) != hipSuccess) {
            (void)hipGetLastError();
            pool = nullptr;
            failed = true;
            return nullptr;
        }
    }
    for (int i = 0; i < used; ++i)
        if (owner[i] == stream) return pool + i * 16;
    if (used == NBLK) return nullptr;
    owner[used] = stream;
    return pool + 16 * used++;
}
// the walk a launch of `nt` tiles of `nk` ring steps on `grid` workgroups takes: the counter block, or null = static
static unsigned* p256_dyn_for(hipStream_t stream, int nt, int grid, int nk, int ns) {
    if (!p256_dynamic() || nt <= grid || (grid & 7) != 0 || nk < ns + 1) return nullptr;
    return mvuld_dyn_counter_block(stream);
}
static int p256_dyn_all() { return g_p256_dyn.load(std::memory_order_relaxed) == 2; }

// (Round 4: the early-issue variant -- template parameter EI -- and the five-stage ring -- NS = 5 -- measured neutral / negative in rounds 2-3
// and are no longer instantiated: tools/experiments/README.md.  The kernel keeps the template parameters.)
template <int EPI, int NI>
static void p256_launch(const GemmArgs& g, int tiles_n, hipStream_t stream) {
    const int tiles_m = (int)cdiv(g.M, 32 * NI);
    const int nt = tiles_m * tiles_n;
    const int grid = nt < p256_num_cus() ? nt : p256_num_cus();
    if (p256_k64() && g.K % 64 == 0 && (int64_t)g.M * g.lda * 2 < ((int64_t)1 << 32) && (int64_t)g.N * g.ldb * 2 < ((int64_t)1 << 32)) {
        constexpr int NS6 = NI <= 5 ? 3 : 2;
        if constexpr (NI <= 5) {
            // MVULD_P256_K64_NS2=1: two stages below 192 rows too (A/B of the ring depth on one tile shape)
            // A three-stage ring runs its DMA stream two steps ahead; on a two-step contraction (K = 128) that is a whole tile ahead, and the
            // stream's NEXT wrap would refill the bias slice of the tile whose epilogue has not read it yet (two slices, alternating): the
            // ring must not be deeper than the tile is long.  (Round 4: found through the e4m3 path, whose picks changed; the bf16 products of
            // the step at K = 128 have N <= 512, where the tiles two apart share their bias columns, so it never showed.)
            static const bool two = [] { const char* e = getenv("MVULD_P256_K64_NS2"); return e && atoi(e) != 0; }();
            if (two || g.K / 64 < 3) {
                constexpr int LDS2 = 2 * (NI * 4096 + 32768) + 2048 + 64;
                static const bool attr2 = [] {
                    (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p256<EPI, NI, false, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
                    return true;
                }();
                (void)attr2;
                hipLaunchKernelGGL((gemm_nt_bf16_p256<EPI, NI, false, 2, true, true>), dim3(grid), dim3(512), LDS2, stream, g, tiles_m, tiles_n,
                                   p256_dyn_for(stream, nt, grid, g.K / 64, 2), p256_dyn_all());
                return;
            }
        }
        constexpr int LDS6 = NS6 * (NI * 4096 + 32768) + 2048 + 64;
        unsigned* dynp = p256_dyn_for(stream, nt, grid, g.K / 64, NS6);
        if constexpr (epi_reads_aux(EPI)) {
            // unconditional-store epilogue (template parameter NF) wherever every wave's 64 columns are all inside or all outside the
            // matrix and the output does not alias the aux operand: bit-identical, -2..8 % on these products
            if (g.N % 64 == 0 && g.aux != g.C) {
                static const bool attrn = [] {
                    (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p256<EPI, NI, false, NS6, true, true, false, true>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS6);
                    return true;
                }();
                (void)attrn;
                hipLaunchKernelGGL((gemm_nt_bf16_p256<EPI, NI, false, NS6, true, true, false, true>), dim3(grid), dim3(512), LDS6, stream, g, tiles_m,
                                   tiles_n, dynp, p256_dyn_all());
                return;
            }
        }
        static const bool attr6 = [] {
            (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p256<EPI, NI, false, NS6, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS6);
            return true;
        }();
        (void)attr6;
        hipLaunchKernelGGL((gemm_nt_bf16_p256<EPI, NI, false, NS6, true, true>), dim3(grid), dim3(512), LDS6, stream, g, tiles_m, tiles_n, dynp, p256_dyn_all());
        return;
    }
    if (p256_pingpong()) {
        static const bool attrp = [] {
            (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p256<EPI, NI, false, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES);
            return true;
        }();
        (void)attrp;
        hipLaunchKernelGGL((gemm_nt_bf16_p256<EPI, NI, false, 4, true>), dim3(grid), dim3(512), P_LDS_BYTES, stream, g, tiles_m, tiles_n, (unsigned*)nullptr, 0);
        return;
    }
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p256<EPI, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES);
        return true;
    }();
    (void)attr;
    hipLaunchKernelGGL((gemm_nt_bf16_p256<EPI, NI>), dim3(grid), dim3(512), P_LDS_BYTES, stream, g, tiles_m, tiles_n, (unsigned*)nullptr, 0);
}

template <int EPI>
static void p256_launch_ni(const GemmArgs& g, int ni, int tiles_n, hipStream_t stream) {
    switch (ni) {
        case 4: p256_launch<EPI, 4>(g, tiles_n, stream); break;
        case 5: p256_launch<EPI, 5>(g, tiles_n, stream); break;
        case 6: p256_launch<EPI, 6>(g, tiles_n, stream); break;
        case 7: p256_launch<EPI, 7>(g, tiles_n, stream); break;
        default: p256_launch<EPI, 8>(g, tiles_n, stream); break;
    }
}

// Routing of mvuld_gemm_nt to this kernel: 0 = never, 1 = default rule, 2 = whenever the shape is legal.  Initialised from
// MVULD_GEMM_P256 (A/B runs of tools/gemm_shapes.py); mvuld_set_gemm_p256_mode() overrides it (tests).
static std::atomic<int> g_p256_mode{-1};
static int p256_mode() {
    int m = g_p256_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("MVULD_GEMM_P256");
        m = e ? atoi(e) : 1;
        if (m < 0 || m > 2) m = 1;
        g_p256_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
extern "C" int mvuld_set_gemm_p256_mode(int mode) {
    MV_CHECK_ARG(mode >= 0 && mode <= 2, "set_gemm_p256_mode: mode must be 0, 1 or 2");
    g_p256_mode.store(mode, std::memory_order_relaxed);
    return 0;
}

template <int EPI, int NI>
static void p256_launch_fp8_ni(const GemmArgs& g, int tiles_n, hipStream_t stream) {
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p256<EPI, NI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES);
        return true;
    }();
    (void)attr;
    const int tiles_m = (int)cdiv(g.M, 32 * NI);
    const int nt = tiles_m * tiles_n;
    const int grid = nt < p256_num_cus() ? nt : p256_num_cus();
    if (p256_k64() && g.K % 128 == 0 && (int64_t)g.M * g.lda < ((int64_t)1 << 32) && (int64_t)g.N * g.ldb < ((int64_t)1 << 32)) {
        constexpr int NS6 = NI <= 5 ? 3 : 2;
        constexpr int LDS6 = NS6 * (NI * 4096 + 32768) + 2048 + 64;
        static const bool attr6 = [] {
            (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p256<EPI, NI, true, NS6, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS6);
            return true;
        }();
        (void)attr6;
        hipLaunchKernelGGL((gemm_nt_bf16_p256<EPI, NI, true, NS6, true, true>), dim3(grid), dim3(512), LDS6, stream, g, tiles_m, tiles_n, (unsigned*)nullptr, 0);
        return;
    }
    if (p256_pingpong()) {
        static const bool attrp = [] {
            (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_p256<EPI, NI, true, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES);
            return true;
        }();
        (void)attrp;
        hipLaunchKernelGGL((gemm_nt_bf16_p256<EPI, NI, true, 4, true>), dim3(grid), dim3(512), P_LDS_BYTES, stream, g, tiles_m, tiles_n, (unsigned*)nullptr, 0);
        return;
    }
    hipLaunchKernelGGL((gemm_nt_bf16_p256<EPI, NI, true>), dim3(grid), dim3(512), P_LDS_BYTES, stream, g, tiles_m, tiles_n, (unsigned*)nullptr, 0);
}
static int p256_pick_ni(int M, int tiles_n, int cus, int max_ni = 8);
static std::atomic<int> g_p256_ni{0};      // 0 = pick per shape; 4..8 = force (A/B timing)
static int g_p256_ni_forced() { return g_p256_ni.load(std::memory_order_relaxed); }
template <int EPI>
static void p256_launch_fp8(const GemmArgs& g, int tiles_n, hipStream_t stream) {
    // The block-scaled path exists up to 192-row tiles (register budget, see the kernel).  Pick among those where the contraction is long
    // enough for the matrix pipe to be what the tile waits for (tools/fp8_shapes.py, batch-256 forward shapes: K >= 1024 -7..-24 % against
    // the non-scaled 224 / 256-row tiles, K <= 768 +5..+19 %: those products are bound by their epilogue and output stream, and the
    // taller tile re-reads less).
    const bool scaled = P256_FP8_SCALED != 0 && p256_k64() && g.K % 128 == 0 && g.K >= 1024;
    int ni = g_p256_ni_forced();
    if (ni == 0) ni = p256_pick_ni(g.M, tiles_n, p256_num_cus(), scaled ? P256_FP8_SCALED_MAX_NI : 8);
    if (p256_k64() && g.K % 128 == 0 && g.K / 128 < 3 && ni < 6) ni = 6;      // two-step contraction: the two-stage ring (see p256_launch)
    switch (ni) {
        case 4: p256_launch_fp8_ni<EPI, 4>(g, tiles_n, stream); break;
        case 5: p256_launch_fp8_ni<EPI, 5>(g, tiles_n, stream); break;
        case 6: p256_launch_fp8_ni<EPI, 6>(g, tiles_n, stream); break;
        case 7: p256_launch_fp8_ni<EPI, 7>(g, tiles_n, stream); break;
        default: p256_launch_fp8_ni<EPI, 8>(g, tiles_n, stream); break;
    }
}

int mvuld_gemm_nt_p256_fp8(const GemmArgs& g, hipStream_t stream) {
    if (g.batch != 1 || g.splitk != 1 || g.out_mode != OUT_STORE) return -1;
    if (g.K % 64 != 0 || g.K < 256 || g.N % 8 != 0 || g.ldc % 8 != 0 || (((uintptr_t)g.C) & 15) != 0) return -1;
    if (g.lda % 16 != 0 || g.ldb % 16 != 0 || ((((uintptr_t)g.A) | ((uintptr_t)g.B)) & 15) != 0) return -1;
    if (g.aux && (g.ldaux % 8 != 0 || (((uintptr_t)g.aux) & 15) != 0)) return -1;
    if (g.bias && (((uintptr_t)g.bias) & 15) != 0) return -1;
    if (g.q_out && ((g.epi != EPI_GELU && g.epi != EPI_GELU_DG) || !g.q_scale || !g.q_amax || g.ldq % 8 != 0 || (((uintptr_t)g.q_out) & 7) != 0)) return -1;
    if (!g.C && !g.q_out) return -1;
    const int tiles_n = (int)cdiv(g.N, P_BN);
    switch (g.epi) {
        case EPI_NONE: case EPI_BIAS: p256_launch_fp8<EPI_BIAS>(g, tiles_n, stream); break;
        case EPI_GELU: p256_launch_fp8<EPI_GELU>(g, tiles_n, stream); break;
        case EPI_GELU_DG: p256_launch_fp8<EPI_GELU_DG>(g, tiles_n, stream); break;
        default: return -1;
    }
    return 0;
}

extern "C" int mvuld_set_gemm_p256_rows(int rows) {
    MV_CHECK_ARG(rows == 0 || (rows % 32 == 0 && rows >= 128 && rows <= 256), "set_gemm_p256_rows: 0 (auto) or 128, 160, 192, 224, 256");
    g_p256_ni.store(rows / 32, std::memory_order_relaxed);
    return 0;
}

// Tile height for an M x N product on a persistent grid of `cus` workgroups: the NI in 4..8 (rows = 32 * NI) that minimises
// rounds(NI) x cost(NI), cost = a fixed share per tile (weights stage, prologue / epilogue latencies) + a share per 16-row fragment.
static int p256_pick_ni(int M, int tiles_n, int cus, int max_ni) {
    int best = max_ni;
    double best_t = 1e30;
    for (int ni = max_ni; ni >= 4; --ni) {
        const int64_t nt = cdiv(M, 32 * ni) * tiles_n;
        const double rounds = (double)cdiv(nt, cus);
        const double t = rounds * (0.28 + 0.09 * ni);
        if (t < best_t * 0.97) { best_t = t; best = ni; }      // prefer the taller tile unless the shorter one wins by > 3 %
    }
    return best;
}

int mvuld_gemm_nt_p256_try(const GemmArgs& g, int dtype_out, hipStream_t stream) {
    const int mode = p256_mode();
    if (mode == 0 || dtype_out != MVULD_BF16 || g.batch != 1 || g.splitk != 1 || g.out_mode != OUT_STORE) return -1;
    if (g.K % P_BK != 0 || g.K < 4 * P_BK || g.N % 8 != 0 || g.ldc % 8 != 0 || (((uintptr_t)g.C) & 15) != 0) return -1;
    if (g.aux && (g.ldaux % 8 != 0 || (((uintptr_t)g.aux) & 15) != 0)) return -1;
    if (g.bias && (((uintptr_t)g.bias) & 15) != 0) return -1;
    if (g.epi == EPI_ELU || g.epi == EPI_MUL_DELU) return -1;      // head-only epilogues: small products, not this kernel's
    const int tiles_n = (int)cdiv(g.N, P_BN);
    if (mode == 1) {
        // default rule: enough whole tiles to occupy a good part of the chip, and not too much of the last column tile wasted.
        // Measured with the full-line ring (tools/gemm_shapes.py --p256-mode 2 against the rings of gemm.hip, same run): the 100-tile
        // products of the last Swin stage (6272 x 1024) are 13-20 % faster here, and the stage-0 products with N = 384 / 128 (the
        // second / only column tile half empty: wasted MFMAs, but these are streaming-bound) 30 % / 15-22 % faster.
        const int64_t tiles = cdiv(g.M, 256) * tiles_n;
        if (tiles < 96 || g.N % 128 != 0) return -1;
        if ((int64_t)tiles_n * P_BN * 2 > (int64_t)g.N * (tiles >= 1024 ? 4 : 3)) return -1;
    }
    int ni = g_p256_ni.load(std::memory_order_relaxed);
    if (ni == 0) ni = p256_pick_ni(g.M, tiles_n, p256_num_cus());
    switch (g.epi) {
        case EPI_NONE: case EPI_BIAS: p256_launch_ni<EPI_BIAS>(g, ni, tiles_n, stream); break;
        case EPI_GELU: p256_launch_ni<EPI_GELU>(g, ni, tiles_n, stream); break;
        case EPI_GELU_DG: p256_launch_ni<EPI_GELU_DG>(g, ni, tiles_n, stream); break;
        case EPI_MUL_AUX: p256_launch_ni<EPI_MUL_AUX>(g, ni, tiles_n, stream); break;      // (EPI_MUL_DGELU: the rings of gemm.hip)
        case EPI_ADD_AUX: p256_launch_ni<EPI_ADD_AUX>(g, ni, tiles_n, stream); break;
        default: return -1;
    }
    return 0;
}
