// Weight gradient  dW[N,K] += dY[M,N]^T . X[M,K]  on 256 x 256 output tiles (bf16 operands, token-major as they sit in HBM).
//
// The 128 x 128 kernel of gemm.hip stages its operands through VGPRs: 64 KiB of ds_write_b128 per 64-token step and CU is
// the LDS store path's whole budget (~79 B/clk), and at 64 FLOP per byte pulled from L2 it is bandwidth-bound besides.  This
// one mirrors gemm_p256.hip:
//   * 8 waves (2 x 4), each 128 (n) x 64 (k) of the tile: 128 FLOP per L2 byte;
//   * token slabs of 32 rows arrive by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write) into a ring of four
//     32 KiB stages, three slabs in flight behind counted `s_waitcnt vmcnt`, one barrier per slab;
//   * both MFMA operands are the hardware-transposed read (ds_read_b64_tr_b16) of the row-major [32 tokens][256 columns] images;
//     512-byte rows put every row on the same banks, so the DMA source lanes are permuted (16-byte chunk c of row r lands in
//     slot c ^ 2(r & 7)): the 8 rows of a 32-lane half then cover all 64 banks;
//   * the bias gradient (column sums of dY) is one more MFMA per fragment against a register of ones -- no LDS reads, no VALU;
//   * the contraction is split over workgroups so that tiles x splits fills the chip once; each split writes its 256 KiB fp32
//     partial in register order (1 KiB per store instruction) and a second, fully parallel kernel adds the partials into dW.
//     (fp32 atomics straight from the accumulators would be 256 KiB per workgroup at ~50 ns per 256 bytes and CU: as long as
//     the main loop itself.)
// N % 8 == 0, K % 8 == 0; column tails are computed on clamped addresses and dropped.  Any M: the LDS-DMA goes through buffer
// descriptors sized to the M valid rows, so the rows of the last 32-token slab beyond M read as zeros (out-of-range buffer loads
// return 0) and add nothing -- the pad-free text encoder hands over token counts that are multiples of nothing.
#include "gemm_common.h"
#include <atomic>

typedef bf16 __attribute__((ext_vector_type(4))) bf16x4_t;

#ifndef TN256_X
#define TN256_X 0      // timing experiments (tools/tn256_variants.sh; results are WRONG): bit 0 = fragments read once (no LDS reads in the loop),
#endif                 // bit 1 = no DMA after the first three slabs, bit 2 = no partial-slab / output stores
#define T_STAGE_BYTES 32768            // dY [32][256] bf16 (16 KiB) + X [32][256] bf16 (16 KiB)
#define T_LDS_BYTES (4 * T_STAGE_BYTES)
#define T_SLAB_FLOATS 65536            // one 256 x 256 fp32 partial

// byte offset of element (row r, column c) of a [32][256] bf16 image whose 16-byte chunks are XOR-swizzled by the row
__device__ __forceinline__ int t_off(int r, int c) { return r * 512 + ((((c >> 3) ^ (2 * (r & 7)))) << 4) + (c & 7) * 2; }

// PP = ping-pong schedule (as in gemm_p256.hip): waves w and w + 4 (wr = 0 / 1) share a SIMD; with one barrier per slab both reach the 24
// transposed reads together and then the 32 (+ 8) MFMAs together, and the matrix pipe idles through every read phase.  PP splits a slab
// step into a read epoch and a matrix epoch and runs row 1 one epoch behind row 0.  Every wave still issues one slab of DMA per step
// right after the barrier that opens its read epoch; slab c has been awaited by every wave before the barrier that opens row 0's read
// epoch of c (row 1 waits for it at the end of its read epoch of c - 1, which closes with that barrier).  Same contraction order.
// `bid` = index of the workgroup within its product (tile-major, splits of a tile adjacent); `force_slab`: write the partial slab even when
// the contraction is not split (grouped launches: the shared reduction kernel adds every product's tiles into dW)
template <bool PP>
__device__ __forceinline__ void tn256_body(char* smem, int bid, const bf16* __restrict__ dY, int64_t ldy, const bf16* __restrict__ X, int64_t ldx,
                                           int M, int N, int K, int tiles_k, int nsplit, int per, float* __restrict__ slabs,
                                           float* __restrict__ dW, int64_t ldw, float* __restrict__ dbias, bool force_slab) {
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef __attribute__((address_space(1))) const void* glb_vp;
    typedef __attribute__((address_space(3))) bf16x4_t* lds_p4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    // split-major order: the splits of one tile sit next to each other, workgroups b, b+8, ... share an XCD
    const int tl = bid / nsplit, ks = bid % nsplit;
    const int tn = tl / tiles_k, tk = tl % tiles_k;
    const int n0 = tn * 256, k0 = tk * 256;
    const int mslabs = (M + 31) / 32;
    const int s0 = ks * per, s1 = min(mslabs, s0 + per);
    const int nk = s1 - s0;                              // >= 1 by construction of nsplit / per on the host

    // LDS-DMA map: one instruction = 1 KiB = 2 image rows; lane l -> row l >> 5, slot l & 31, fetching chunk slot ^ 2(row & 7).
    // Wave w issues pieces 2w, 2w+1 (rows 4w .. 4w+3) of the dY image and of the X image.
    // byte offsets into the two buffers (< 4 GiB each, checked on the host); rows >= M fall outside the descriptors and load zeros
    const auto rsa = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)min((int64_t)M * ldy * 2, (int64_t)0x7fffffff), 0x00020000);
    const auto rsb = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)min((int64_t)M * ldx * 2, (int64_t)0x7fffffff), 0x00020000);
    unsigned va[2], vb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 2 + (lane >> 5);
        const int ch = (lane & 31) ^ (2 * (r & 7));
        const int ca = n0 + ch * 8, cb = k0 + ch * 8;
        va[i] = (unsigned)(((int64_t)(s0 * 32 + r) * ldy + (ca < N ? ca : 0)) * 2);
        vb[i] = (unsigned)(((int64_t)(s0 * 32 + r) * ldx + (cb < K ? cb : 0)) * 2);
    }
    const unsigned stepa = (unsigned)(64 * ldy), stepb = (unsigned)(64 * ldx);      // bytes per 32-row slab
    int issued = 0;
    auto issue_one = [&]() {
        if ((TN256_X & 2) && issued >= 3) return;
        if (issued < nk) {
            char* st = smem + (issued & 3) * T_STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_vp)(st + (wave * 2 + i) * 1024), 16, va[i], issued * stepa, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_vp)(st + 16384 + (wave * 2 + i) * 1024), 16, vb[i], issued * stepb, 0, 0);
            }
            ++issued;
        }
    };
    issue_one();
    issue_one();
    issue_one();

    // transposed-read addresses (constant over the loop).  Fragment of 16 columns col0.. over the 32 rows of a stage:
    // k-slot (g, j<4) <-> row 4g+j, (g, j>=4) <-> row 16+4g+j-4 on BOTH operands; lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3.
    const int q = (lane & 15) >> 2, pp = lane & 3;
    int oa[8], ob[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) oa[i] = t_off(4 * fg + q, wr * 128 + i * 16 + 4 * pp);
#pragma unroll
    for (int j = 0; j < 4; ++j) ob[j] = 16384 + t_off(4 * fg + q, wc * 64 + j * 16 + 4 * pp);

    f32x4_t acc[8][4], accb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        accb[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = dbias && tk == 0 && wc == 0;   // wave-uniform
    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    // slab x has landed once at most the slabs issued after it are in flight (precondition: this wave has issued up to slab x + 2)
    auto wait_slab = [&](int x) {
        const int younger = nk - 1 - x;
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (PP && wr == 1) {                                 // row 1 falls one barrier epoch behind row 0
        wait_slab(0);
        __builtin_amdgcn_s_barrier();
    }
    bf16x8_t fa[8], fb[4];
    for (int kt = 0; kt < nk; ++kt) {
        if (!PP || wr == 0) wait_slab(kt);
        __builtin_amdgcn_s_barrier();                    // slab kt visible to every wave; every wave is done with slab kt - 1
        issue_one();                                     // slab kt + 3 refills the stage slab kt - 1 occupied
        const char* st = smem + (kt & 3) * T_STAGE_BYTES;
        if (!(TN256_X & 1) || kt == 0) {
        // The transposed reads are inline asm: through the builtin hipcc sees LDS reads next to outstanding LDS-DMA (buffer_load ... lds) and
        // puts `s_waitcnt vmcnt(0)` in front of the first one -- every step then waited for ALL three slabs in flight, i.e. the four-stage
        // ring ran as a synchronous load per step (the compiled loop of rounds 1-2; found by reading the ISA).  Two blocks, each ending in
        // its own lgkmcnt(0), so the outputs are valid registers when the statement retires.
        {
            const unsigned sb = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)st;
            long bl[4], bh[4], al[8], ah[8];
            asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:8192\n\t"
                         "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:8192\n\t"
                         "ds_read_b64_tr_b16 %4, %10\n\tds_read_b64_tr_b16 %5, %10 offset:8192\n\t"
                         "ds_read_b64_tr_b16 %6, %11\n\tds_read_b64_tr_b16 %7, %11 offset:8192\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(bl[0]), "=&v"(bh[0]), "=&v"(bl[1]), "=&v"(bh[1]), "=&v"(bl[2]), "=&v"(bh[2]), "=&v"(bl[3]), "=&v"(bh[3])
                         : "v"(sb + ob[0]), "v"(sb + ob[1]), "v"(sb + ob[2]), "v"(sb + ob[3])
                         : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:8192\n\t"
                         "ds_read_b64_tr_b16 %2, %17\n\tds_read_b64_tr_b16 %3, %17 offset:8192\n\t"
                         "ds_read_b64_tr_b16 %4, %18\n\tds_read_b64_tr_b16 %5, %18 offset:8192\n\t"
                         "ds_read_b64_tr_b16 %6, %19\n\tds_read_b64_tr_b16 %7, %19 offset:8192\n\t"
                         "ds_read_b64_tr_b16 %8, %20\n\tds_read_b64_tr_b16 %9, %20 offset:8192\n\t"
                         "ds_read_b64_tr_b16 %10, %21\n\tds_read_b64_tr_b16 %11, %21 offset:8192\n\t"
                         "ds_read_b64_tr_b16 %12, %22\n\tds_read_b64_tr_b16 %13, %22 offset:8192\n\t"
                         "ds_read_b64_tr_b16 %14, %23\n\tds_read_b64_tr_b16 %15, %23 offset:8192\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(al[0]), "=&v"(ah[0]), "=&v"(al[1]), "=&v"(ah[1]), "=&v"(al[2]), "=&v"(ah[2]), "=&v"(al[3]), "=&v"(ah[3]),
                           "=&v"(al[4]), "=&v"(ah[4]), "=&v"(al[5]), "=&v"(ah[5]), "=&v"(al[6]), "=&v"(ah[6]), "=&v"(al[7]), "=&v"(ah[7])
                         : "v"(sb + oa[0]), "v"(sb + oa[1]), "v"(sb + oa[2]), "v"(sb + oa[3]), "v"(sb + oa[4]), "v"(sb + oa[5]), "v"(sb + oa[6]),
                           "v"(sb + oa[7])
                         : "memory");
            typedef long __attribute__((ext_vector_type(2))) t_i64x2;
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = __builtin_bit_cast(bf16x8_t, (t_i64x2){bl[j], bh[j]});
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = __builtin_bit_cast(bf16x8_t, (t_i64x2){al[i], ah[i]});
        }
        }
        if constexpr (PP) {
            // the fragments are in registers before the barrier that hands the matrix pipe over (and the stage back to the DMA)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (wr == 1 && kt + 1 < nk) wait_slab(kt + 1);
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PP) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 8; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
        }
        if constexpr (PP) __builtin_amdgcn_s_setprio(0);
    }
    if (PP && wr == 0) __builtin_amdgcn_s_barrier();     // pairs with row 1's last hand-over

    // bias gradient: every column of accb[i] holds sum_m dY[m][n]; lanes of column 0 add the split's share
    if (do_bias && fr == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wr * 128 + i * 16 + 4 * fg + r;
                if (n < N) atomicAdd(dbias + n, accb[i][r]);
            }
    }
    // acc[i][j][r] = dW[n0 + wr*128 + i*16 + 4*fg + r][k0 + wc*64 + j*16 + fr]
    if (nsplit > 1 || force_slab) {
        float* mine = slabs + ((size_t)tl * nsplit + ks) * T_SLAB_FLOATS;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (TN256_X & 4) asm volatile("" ::"v"(acc[i][j]));
                else *(f32x4_t*)(mine + ((wave * 32 + i * 4 + j) * 64 + lane) * 4) = acc[i][j];
            }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wr * 128 + i * 16 + 4 * fg + r, k = k0 + wc * 64 + j * 16 + fr;
                    if (n < N && k < K) atomicAdd(dW + (int64_t)n * ldw + k, acc[i][j][r]);
                }
    }
}

template <bool PP>
__global__ __launch_bounds__(512, 1) void gemm_tn256_k(const bf16* __restrict__ dY, int64_t ldy, const bf16* __restrict__ X, int64_t ldx,
                                                       int M, int N, int K, int tiles_k, int nsplit, int per, float* __restrict__ slabs,
                                                       float* __restrict__ dW, int64_t ldw, float* __restrict__ dbias) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tn256_body<PP>(smem, blockIdx.x, dY, ldy, X, ldx, M, N, K, tiles_k, nsplit, per, slabs, dW, ldw, dbias, false);
}

// ---- grouped launch (round 3): the weight gradients of ONE transformer block in one launch.  A single product has 4-16 tiles for 256 CUs,
// so its contraction was split 16-21 ways and every split wrote (and the reduction read back) a 256 KiB fp32 partial: 53 GB per step
// against 18.5 GB of operands and gradients (profiles/r02_hbm_traffic.csv).  A block's four products together have ~48 tiles: one launch
// over the job table splits each ~5 ways for the same one-round fill of the chip -- a third of the partial slabs, and one reduction
// launch per block instead of one per product.  The table travels as a kernel argument (no device copy to keep alive).
#define TN_MAX_JOBS 8
struct TnJob {
    const bf16* dY; const bf16* X; float* dW; float* dbias;
    int64_t ldy, ldx, ldw;
    int M, N, K, tiles_k, nsplit, per, wg0, slab0;       // wg0: first workgroup of the product; slab0: its first slab in the workspace
};
// map[g] = product << 12 | workgroup index inside the product, for launch-order block g.  Blocks g, g + 8, ... share an XCD and its L2 (observed
// placement, used for speed only): the host deals the blocks so that all tiles of one (product, token split) sit on ONE XCD -- they walk the same 32-token
// slabs of dY and X at the same time, so each operand row is fetched from HBM once per XCD instead of once per tile (tiles_n + tiles_k times:
// profiles/r03_hbm_traffic.csv had the grouped launch reading 1.0 GB for 0.41 GB of operands, at 6 TB/s -- bandwidth-bound on its own re-reads).
#define TN_MAP_MAX 512
struct TnJobs { int n, wg_total, tiles_total, mapped; int tile0[TN_MAX_JOBS + 1]; TnJob j[TN_MAX_JOBS]; unsigned short map[TN_MAP_MAX]; };

template <bool PP>
__global__ __launch_bounds__(512, 1) void gemm_tn256_group_k(const TnJobs jobs, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int ji = 0, local;
    if (jobs.mapped) {
        const unsigned m = jobs.map[blockIdx.x];
        ji = (int)(m >> 12);
        local = (int)(m & 0xfff);
    } else {
#pragma unroll
        for (int i = 1; i < TN_MAX_JOBS; ++i) ji += (i < jobs.n && (int)blockIdx.x >= jobs.j[i].wg0) ? 1 : 0;      // products sit in wg0 order
        local = blockIdx.x - jobs.j[ji].wg0;
    }
    const TnJob& J = jobs.j[ji];
    tn256_body<PP>(smem, local, J.dY, J.ldy, J.X, J.ldx, J.M, J.N, J.K, J.tiles_k, J.nsplit, J.per,
                   slabs + (size_t)J.slab0 * T_SLAB_FLOATS, J.dW, J.ldw, J.dbias, true);
}

// dW += sum over splits of the partial tiles (register-order slabs): one thread per (tile, wave, fragment, lane) float4
__global__ __launch_bounds__(256) void gemm_tn256_reduce_k(const float* __restrict__ slabs, int nsplit, int tiles_k, int N, int K,
                                                           float* __restrict__ dW, int64_t ldw) {
    const int tl = blockIdx.x >> 6;                      // 64 blocks of 256 threads per tile
    const int pos = ((blockIdx.x & 63) << 8) + threadIdx.x;          // 0 .. 16383
    const int lane = pos & 63, frag = (pos >> 6) & 31, wave = pos >> 11;
    const int i = frag >> 2, j = frag & 3, wr = wave >> 2, wc = wave & 3, fr = lane & 15, fg = lane >> 4;
    const float* p = slabs + (size_t)tl * nsplit * T_SLAB_FLOATS + (size_t)pos * 4;
    f32x4_t s = *(const f32x4_t*)p;
    for (int k = 1; k < nsplit; ++k) s += *(const f32x4_t*)(p + (size_t)k * T_SLAB_FLOATS);
    const int n0 = (tl / tiles_k) * 256, k0 = (tl % tiles_k) * 256;
    const int kk = k0 + wc * 64 + j * 16 + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + wr * 128 + i * 16 + 4 * fg + r;
        if (n < N && kk < K) atomicAdd(dW + (int64_t)n * ldw + kk, s[r]);
    }
}

// the same for a job table: blockIdx = (tile over all products) * 64 + part
__global__ __launch_bounds__(256) void gemm_tn256_group_reduce_k(const TnJobs jobs, const float* __restrict__ slabs) {
    const int gt = blockIdx.x >> 6;
    int ji = 0;
#pragma unroll
    for (int i = 1; i < TN_MAX_JOBS; ++i) ji += (i < jobs.n && gt >= jobs.tile0[i]) ? 1 : 0;
    const TnJob& J = jobs.j[ji];
    const int tl = gt - jobs.tile0[ji];
    const int pos = ((blockIdx.x & 63) << 8) + threadIdx.x;
    const int lane = pos & 63, frag = (pos >> 6) & 31, wave = pos >> 11;
    const int i = frag >> 2, j = frag & 3, wr = wave >> 2, wc = wave & 3, fr = lane & 15, fg = lane >> 4;
    const float* p = slabs + ((size_t)J.slab0 + (size_t)tl * J.nsplit) * T_SLAB_FLOATS + (size_t)pos * 4;
    f32x4_t s = *(const f32x4_t*)p;
    for (int k = 1; k < J.nsplit; ++k) s += *(const f32x4_t*)(p + (size_t)k * T_SLAB_FLOATS);
    const int n0 = (tl / J.tiles_k) * 256, k0 = (tl % J.tiles_k) * 256;
    const int kk = k0 + wc * 64 + j * 16 + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + wr * 128 + i * 16 + 4 * fg + r;
        if (n < J.N && kk < J.K) atomicAdd(J.dW + (int64_t)n * J.ldw + kk, s[r]);
    }
}

static int tn256_all_cus() {
    static const int n = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        return v > 0 ? v : 256;
    }();
    return n;
}
// CUs the contraction splits are planned for.  Alone on the chip: all of them.  In the multi-stream training step this kernel always
// runs beside the data-gradient chain of another stream (one LDS-heavy workgroup per CU either way); there a footprint of HALF the
// chip, with half the slab traffic, is ~1.8x slower for the kernel and 1 % faster for the step (the host tells which case it is:
// mvuld_set_gemm_tn256_budget).  MVULD_TN256_CUS pins the number (tuning runs).
static std::atomic<int> g_tn256_budget{0};
extern "C" int mvuld_set_gemm_tn256_budget(int cus) {
    MV_CHECK_ARG(cus >= 0, "set_gemm_tn256_budget: cus >= 0 (0 = every CU)");
    g_tn256_budget.store(cus, std::memory_order_relaxed);
    return 0;
}
static int tn256_num_cus() {
    static const int pinned = [] { const char* e = getenv("MVULD_TN256_CUS"); return e ? atoi(e) : 0; }();
    if (pinned > 0) return pinned;
    const int b = g_tn256_budget.load(std::memory_order_relaxed);
    return b > 0 && b < tn256_all_cus() ? b : tn256_all_cus();
}

// schedule of the main loop: 1 (default) = ping-pong, 0 = lockstep; MVULD_TN256_PINGPONG, mvuld_set_gemm_tn256_pingpong (tests, A/B timing).
// Measured (tools/gemm_shapes.py --only tn, same box): while hipcc's `s_waitcnt vmcnt(0)` in front of the builtin transposed reads made every
// step a synchronous load, ping-pong was 1 % slower (12.25 vs 12.13 ms per step's worth); with the reads in inline asm the ring really runs
// three slabs ahead (10.77 ms) and the read / MFMA alternation is what is left to hide: 10.40 ms with ping-pong.
static std::atomic<int> g_tn256_pp{-1};
static bool tn256_pingpong() {
    int v = g_tn256_pp.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("MVULD_TN256_PINGPONG");
        v = e ? (atoi(e) != 0) : 1;
        g_tn256_pp.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
extern "C" int mvuld_set_gemm_tn256_pingpong(int on) {
    g_tn256_pp.store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}

// split plan of the 256 x 256-tile kernel: 0 splits = shape not eligible
static void tn256_plan(int M, int N, int K, int& nsplit, int& per) {
    nsplit = 0;
    per = 0;
    if (M < 256) return;
    const int64_t tiles = cdiv(N, 256) * cdiv(K, 256);
    // fewer than 8 tiles would mean > 32 splits: the partial slabs (256 KiB each) then cost more than the 128 x 128 kernel's atomics
    // (measured, tools/gemm_shapes.py: 512 x 512 and 256 x 256 weights lose, 768 x 768 breaks even, everything wider wins 15-25 %)
    static const int min_tiles = [] { const char* e = getenv("MVULD_TN256_MIN_TILES"); const int v = e ? atoi(e) : 8; return v > 0 ? v : 8; }();
    static const int max_pad4 = [] { const char* e = getenv("MVULD_TN256_MAX_PAD4"); const int v = e ? atoi(e) : 5; return v >= 4 ? v : 5; }();
    // ... except where the contraction is very long: with the ring running asynchronously (reads in inline asm) the 3- and 4-tile weights of
    // Swin stage 1 (100 352 tokens; 1024 x 256: 114 -> 80 us, 256 x 1024: 114 -> 89 us, 768 x 256: 91 -> 83 us) win too; 2-tile weights and
    // the 4-tile 512 x 512 at 25 088 tokens still lose (tools/gemm_shapes.py --only tn with MVULD_TN256_MIN_TILES = 1 / 2 / 4)
    if (tiles < min_tiles && !(min_tiles == 8 && tiles >= 3 && M >= 65536)) return;
    // most of every tile must be real weight: (padded area) <= 1.25 x (N x K)   (MVULD_TN256_MAX_PAD4 / 4: A/B runs)
    if (cdiv(N, 256) * 256 * cdiv(K, 256) * 256 * 4 > (int64_t)N * K * max_pad4) return;
    const int mslabs = (M + 31) / 32;
    int want = (int)(tn256_num_cus() / tiles);
    if (want < 1) want = 1;
    if (want > mslabs / 8) want = mslabs / 8 > 0 ? mslabs / 8 : 1;      // at least 8 ring steps per split
    per = (int)cdiv(mslabs, want);
    nsplit = (int)cdiv(mslabs, per);                     // every split non-empty
}

int64_t mvuld_gemm_tn256_workspace_bytes(int M, int N, int K) {
    int nsplit, per;
    tn256_plan(M, N, K, nsplit, per);
    if (nsplit < 1) return 0;
    return nsplit > 1 ? cdiv(N, 256) * cdiv(K, 256) * (int64_t)nsplit * T_SLAB_FLOATS * 4 : 16;
}

// returns 0 when it took the launch, -1 when the shape (or the workspace) is not its to take
int mvuld_gemm_tn256_try(const void* dY, int64_t ldy, const void* X, int64_t ldx, float* dW, int64_t ldw, int M, int N, int K, float* dbias,
                         void* ws, int64_t ws_bytes, hipStream_t stream) {
    int nsplit, per;
    tn256_plan(M, N, K, nsplit, per);
    if (nsplit < 1) return -1;
    if ((int64_t)M * ldy * 2 >= (int64_t)0x7fffffff || (int64_t)M * ldx * 2 >= (int64_t)0x7fffffff) return -1;     // 32-bit buffer offsets
    const int64_t need = mvuld_gemm_tn256_workspace_bytes(M, N, K);
    if (nsplit > 1 && (!ws || ws_bytes < need || (((uintptr_t)ws) & 15) != 0)) return -1;
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)gemm_tn256_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_tn256_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
        return true;
    }();
    (void)attr;
    const int tiles_k = (int)cdiv(K, 256);
    const int tiles = (int)(cdiv(N, 256) * tiles_k);
    if (tn256_pingpong())
        hipLaunchKernelGGL(gemm_tn256_k<true>, dim3(tiles * nsplit), dim3(512), T_LDS_BYTES, stream, (const bf16*)dY, ldy, (const bf16*)X, ldx, M, N,
                           K, tiles_k, nsplit, per, (float*)ws, dW, ldw, dbias);
    else
        hipLaunchKernelGGL(gemm_tn256_k<false>, dim3(tiles * nsplit), dim3(512), T_LDS_BYTES, stream, (const bf16*)dY, ldy, (const bf16*)X, ldx, M, N,
                           K, tiles_k, nsplit, per, (float*)ws, dW, ldw, dbias);
    if (nsplit > 1)
        hipLaunchKernelGGL(gemm_tn256_reduce_k, dim3(tiles * 64), dim3(256), 0, stream, (const float*)ws, nsplit, tiles_k, N, K, dW, ldw);
    return 0;
}

// ---- grouped launch, host side.  desc: njobs x 10 int64 {dY, ldy, X, ldx, dW, ldw, M, N, K, dbias}.
static bool tn256_group_shape_ok(int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t ldx) {
    static const int max_pad4 = [] { const char* e = getenv("MVULD_TN256_MAX_PAD4"); const int v = e ? atoi(e) : 5; return v >= 4 ? v : 5; }();
    if (M < 256 || N % 8 || K % 8 || ldy % 8 || ldx % 8) return false;
    if (cdiv(N, 256) * 256 * cdiv(K, 256) * 256 * 4 > N * K * max_pad4) return false;
    return M * ldy * 2 < (int64_t)0x7fffffff && M * ldx * 2 < (int64_t)0x7fffffff;
}
extern "C" int mvuld_gemm_tn_wgrad_group_ok(int M, int N, int K, int64_t ldy, int64_t ldx) { return tn256_group_shape_ok(M, N, K, ldy, ldx) ? 1 : 0; }

// XCD-aware deal of the launch-order blocks (see TnJobs::map): (product, split) groups, largest first, each onto the XCD with the least
// room that still takes it whole (else spread over the emptiest ones).  MVULD_TN256_XCD_MAP=0: launch order = product order.
static void tn256_group_map(TnJobs& T) {
    static const bool on = [] { const char* e = getenv("MVULD_TN256_XCD_MAP"); return !e || atoi(e) != 0; }();
    T.mapped = 0;
    if (!on || T.wg_total > TN_MAP_MAX) return;
    for (int i = 0; i < T.n; ++i)
        if ((int64_t)cdiv(T.j[i].N, 256) * T.j[i].tiles_k * T.j[i].nsplit > 4096) return;
    int freec[8], nexti[8];
    for (int x = 0; x < 8; ++x) { freec[x] = (T.wg_total - x + 7) / 8; nexti[x] = 0; }
    struct Grp { int job, ks, size; };
    Grp g[TN_MAX_JOBS * 64];
    int ng = 0;
    for (int i = 0; i < T.n; ++i) {
        const int tiles = (int)cdiv(T.j[i].N, 256) * T.j[i].tiles_k;
        for (int ks = 0; ks < T.j[i].nsplit; ++ks) {
            if (ng == TN_MAX_JOBS * 64) return;
            g[ng++] = Grp{i, ks, tiles};
        }
    }
    for (int a = 1; a < ng; ++a) {          // insertion sort, largest first
        const Grp v = g[a];
        int b = a - 1;
        while (b >= 0 && g[b].size < v.size) { g[b + 1] = g[b]; --b; }
        g[b + 1] = v;
    }
    for (int a = 0; a < ng; ++a) {
        int tile = 0;
        while (tile < g[a].size) {
            const int need = g[a].size - tile;
            int best = -1;
            for (int x = 0; x < 8; ++x)      // smallest room that takes the rest whole
                if (freec[x] >= need && (best < 0 || freec[x] < freec[best])) best = x;
            if (best < 0)
                for (int x = 0; x < 8; ++x)  // else the emptiest XCD
                    if (freec[x] > 0 && (best < 0 || freec[x] > freec[best])) best = x;
            if (best < 0) return;           // (cannot happen: the slots add up to wg_total)
            const int take = freec[best] < need ? freec[best] : need;
            for (int k = 0; k < take; ++k, ++tile) {
                const int slot = best + 8 * nexti[best]++;
                T.map[slot] = (unsigned short)((g[a].job << 12) | (tile * T.j[g[a].job].nsplit + g[a].ks));
            }
            freec[best] -= take;
        }
    }
    T.mapped = 1;
}

static int tn256_group_plan(const int64_t* desc, int njobs, TnJobs& T) {
    if (njobs < 1 || njobs > TN_MAX_JOBS) return -1;
    int64_t steps = 0;
    for (int i = 0; i < njobs; ++i) {
        const int64_t* d = desc + 10 * i;
        if (!tn256_group_shape_ok(d[6], d[7], d[8], d[1], d[3])) return -1;
        steps += cdiv(d[7], 256) * cdiv(d[8], 256) * ((d[6] + 31) / 32);
    }
    const int cus = tn256_num_cus();
    int64_t target = cdiv(steps, cus);                   // 32-token slabs per workgroup for one full round of the chip
    if (target < 8) target = 8;
    for (int attempt = 0; attempt < 64; ++attempt) {
        int wg = 0, slab = 0, tile = 0;
        for (int i = 0; i < njobs; ++i) {
            const int64_t* d = desc + 10 * i;
            TnJob& J = T.j[i];
            J.dY = (const bf16*)d[0]; J.ldy = d[1]; J.X = (const bf16*)d[2]; J.ldx = d[3]; J.dW = (float*)d[4]; J.ldw = d[5];
            J.M = (int)d[6]; J.N = (int)d[7]; J.K = (int)d[8]; J.dbias = (float*)d[9];
            J.tiles_k = (int)cdiv(J.K, 256);
            const int tiles = (int)cdiv(J.N, 256) * J.tiles_k, mslabs = (J.M + 31) / 32;
            int ns = (int)((mslabs + target / 2) / target);
            if (ns < 1) ns = 1;
            J.per = (int)cdiv(mslabs, ns);
            J.nsplit = (int)cdiv(mslabs, J.per);
            J.wg0 = wg; J.slab0 = slab;
            T.tile0[i] = tile;
            wg += tiles * J.nsplit; slab += tiles * J.nsplit; tile += tiles;
        }
        T.tile0[njobs] = tile;
        T.n = njobs; T.wg_total = wg; T.tiles_total = tile; T.mapped = 0;
        if (wg <= cus || target >= steps) { tn256_group_map(T); return 0; }
        target += (target + 15) / 16;                    // a few workgroups over one round: longer splits
    }
    tn256_group_map(T);
    return 0;
}
// The workspace has the layout of mvuld_gemm_tn_wgrad's: its first TN_GROUP_WS_HEAD bytes are the ticket block of the 128 x 128 kernel's
// last-arriver reduction (zero between launches: the caller may hand the SAME per-stream buffer to both entry points) and are not touched.
#define TN_GROUP_WS_HEAD 4096
extern "C" int64_t mvuld_gemm_tn_wgrad_group_workspace_bytes(const int64_t* desc, int njobs) {
    TnJobs T;
    if (tn256_group_plan(desc, njobs, T)) return -1;
    return TN_GROUP_WS_HEAD + (int64_t)T.wg_total * T_SLAB_FLOATS * 4;
}
extern "C" int mvuld_gemm_tn_wgrad_group(const int64_t* desc, int njobs, void* ws, int64_t ws_bytes, hipStream_t stream) {
    TnJobs T;
    MV_CHECK_ARG(desc && njobs >= 1 && njobs <= TN_MAX_JOBS, "gemm_tn_wgrad_group: 1..%d products per launch", TN_MAX_JOBS);
    MV_CHECK_ARG(tn256_group_plan(desc, njobs, T) == 0, "gemm_tn_wgrad_group: a product is not eligible (mvuld_gemm_tn_wgrad_group_ok)");
    for (int i = 0; i < njobs; ++i)
        MV_CHECK_ARG(T.j[i].dY && T.j[i].X && T.j[i].dW && ((((uintptr_t)T.j[i].dY) | ((uintptr_t)T.j[i].X)) & 15) == 0, "gemm_tn_wgrad_group: product %d: null / misaligned operand", i);
    const int64_t need = TN_GROUP_WS_HEAD + (int64_t)T.wg_total * T_SLAB_FLOATS * 4;
    MV_CHECK_ARG(ws && ws_bytes >= need && (((uintptr_t)ws) & 15) == 0, "gemm_tn_wgrad_group: workspace too small (%lld < %lld bytes) or misaligned",
                 (long long)ws_bytes, (long long)need);
    ws = (char*)ws + TN_GROUP_WS_HEAD;
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)gemm_tn256_group_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_tn256_group_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
        return true;
    }();
    (void)attr;
    if (tn256_pingpong()) hipLaunchKernelGGL(gemm_tn256_group_k<true>, dim3(T.wg_total), dim3(512), T_LDS_BYTES, stream, T, (float*)ws);
    else hipLaunchKernelGGL(gemm_tn256_group_k<false>, dim3(T.wg_total), dim3(512), T_LDS_BYTES, stream, T, (float*)ws);
    hipLaunchKernelGGL(gemm_tn256_group_reduce_k, dim3(T.tiles_total * 64), dim3(256), 0, stream, T, (const float*)ws);
    MV_LAUNCH_CHECK("gemm_tn_wgrad_group");
    return 0;
}
