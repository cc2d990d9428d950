/* libmvuld_hip.so -- C ABI of the MI355X (gfx950) kernels behind MVulD's fused multimodal hot path.
 *
 * The reference (jacknichao/MVulD) has no native code and no FFI: every GPU kernel it runs comes
 * from torch / cuDNN / DGL / transformers.  Each entry below therefore cites the reference *call
 * site* (file:line under /root/reference/mvuld) whose implicit third-party kernel(s) it replaces.
 *
 * Conventions
 *   - plain device pointers + explicit sizes; no torch types.  `dtype`: 0 = f32, 1 = bf16 storage
 *     (math is always fp32; bf16 GEMMs accumulate in fp32 on the MFMA units).
 *   - the caller owns every buffer, including workspaces and saved statistics.
 *   - functions never allocate, never synchronise, never throw; they enqueue on `stream`
 *     (pass torch.cuda.current_stream().cuda_stream) and return 0, or non-zero with a message
 *     retrievable through mvuld_last_error() (thread-local).
 *   - "atomic accumulate" outputs (parameter gradients) are fp32 and must be zeroed by the caller
 *     once per optimisation step.
 *   - the only process-wide state is a handful of ROUTING / TUNING settings (mvuld_set_*: which of several
 *     equivalent kernels or schedules a call uses).  Each is one atomic integer, read once from its MVULD_*
 *     environment variable on first use unless a setter ran first (function-local statics / atomics: safe to
 *     call from several threads); none of them changes what a call computes beyond the rounding order noted
 *     at its declaration, and kernel launches themselves keep no state.
 */
#ifndef MVULD_HIP_H
#define MVULD_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mvuld_stream_t;

enum { MVULD_F32 = 0, MVULD_BF16 = 1 };
enum { MVULD_EPI_NONE = 0, MVULD_EPI_BIAS = 1, MVULD_EPI_GELU = 2, MVULD_EPI_ELU = 3, MVULD_EPI_MUL_DGELU = 4, MVULD_EPI_MUL_DELU = 5, MVULD_EPI_ADD_AUX = 6,
       MVULD_EPI_GELU_DG = 7, MVULD_EPI_MUL_AUX = 8 };
enum { MVULD_OUT_STORE = 0, MVULD_OUT_ACCUM = 1, MVULD_OUT_ATOMIC = 2 };

int mvuld_version(void);
const char* mvuld_last_error(void);

/* C[b] = epilogue(alpha * A[b] . B[b]^T (+ bias[N]))   A [M,K] (lda), B [N,K] (ldb), C [M,N] (ldc), batch strides in elements.
 * Replaces every nn.Linear / F.linear / Conv1d(k=1) / Conv2d(4x4,s4) / torch.matmul on the path:
 *   swin_transformer_v2.py:150 (qkv), :177 (proj), :27-30 (Mlp), :361 (reduction), :490 (patch conv);
 *   HF RobertaModel dense layers (unixcoder.py:36); GATConv.fc, GraphModel.py:153-209 Linear layers;
 *   Rs_GCN.py:57-70 (g/theta/phi/W convs, theta^T.phi, R.g).
 * epilogue GELU writes the pre-activation to `aux` (if non-null); MUL_DGELU / MUL_DELU multiply by the
 * activation derivative taken from `aux` (pre-activation / ELU output); ADD_AUX adds `aux` (residual-gradient join).
 * GELU_DG / MUL_AUX are the FFN's training pair: GELU_DG is GELU whose `aux` receives gelu'(pre-activation) instead (same erf and
 * exponential: two more instructions per element), MUL_AUX multiplies by `aux` -- the backward product of Mlp.fc2 / RobertaOutput.dense
 * then has a one-instruction epilogue instead of recomputing erf + exp (swin_transformer_v2.py:27-30, nn.GELU backward).  out_mode ATOMIC (fp32 C only)
 * with splitk > 1 is the weight-gradient form  dW += dY^T . X . */
int mvuld_gemm_nt(const void* A, int64_t lda, int64_t strideA, const void* B, int64_t ldb, int64_t strideB,
                  void* C, int64_t ldc, int64_t strideC, int M, int N, int K, int batch,
                  const float* bias, int epilogue, void* aux, int64_t ldaux, int64_t strideAux,
                  float alpha, int out_mode, int splitk, int dtype_in, int dtype_out, int force_simple,
                  mvuld_stream_t stream);
/* The same product for fp32 operands and an fp32 result at near-fp32 accuracy on the bf16 matrix cores: every operand element is split
 * into hi = bf16(x) and lo = bf16(x - hi) ON THE WAY from memory to LDS and the kernel accumulates a_hi b_hi + a_lo b_hi + a_hi b_lo in
 * fp32 (error ~2^-16).  One launch, no operand copies (rounds 1-2: mvuld_split_bf16x3 on both operands + a 3K-deep bf16 product).
 * Every epilogue and out_mode of mvuld_gemm_nt (atomic: NONE / BIAS only), batched.  trans_a / trans_b: that operand is stored [K, M] /
 * [K, N] (lda / ldb = its row stride) -- products like R^T dY or the weight gradient dY^T X without a transpose pass; splitk > 1 splits the
 * contraction over workgroups that add into C with atomics (out_mode ATOMIC only).  The head's fp32 tail:
 * GATConv.fc, GraphModel.py:153-209, Rs_GCN.py:57-70 in the bf16 activation mode. */
int mvuld_gemm_nt_f32x3(const float* A, int64_t lda, int64_t strideA, const float* B, int64_t ldb, int64_t strideB, float* C, int64_t ldc,
                        int64_t strideC, int M, int N, int K, int batch, const float* bias, int epilogue, float* aux, int64_t ldaux,
                        int64_t strideAux, float alpha, int out_mode, int trans_a, int trans_b, int splitk, mvuld_stream_t stream);

/* fp8 forward GEMMs (BASELINE configs[4]: "fp8 (CDNA4 MFMA) QKV/FFN GEMMs in SwinV2 + UniXcoder"; same call sites as mvuld_gemm_nt:
 * swin_transformer_v2.py:146-152,177,26-32 and the RobertaModel dense layers behind unixcoder.py:36).
 * mvuld_quant_e4m3: per-tensor quantisation to OCP e4m3 (gfx950's native fp8): scale_out[0] = max|x| / 448, out = e4m3(x / scale);
 *   n % 8 == 0; `partials` = 1024 floats of scratch.
 * mvuld_gemm_nt_fp8: C[M,N] (bf16) = epi(scale_a[0] * scale_b[0] * A8[M,K] . B8[N,K]^T + bias) on the e4m3 matrix-core forms (v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales where K >= 1024, v_mfma_f32_16x16x32_fp8_fp8 elsewhere) with fp32
 *   accumulation (the persistent 256 x 256 kernel; K % 64 == 0, K >= 256, N % 8 == 0, lda / ldb multiples of 16); epilogue NONE / BIAS /
 *   GELU (+ pre-activation to `aux`) / GELU_DG (+ gelu' to `aux`).  With a GELU epilogue the activation can leave as e4m3 for the next product without a pass of
 *   its own: q_out[M,N] (row stride ldq bytes) = e4m3(bf16(gelu) / q_state[0]); max|gelu| is folded into q_state[1] (atomic max on the
 *   float's bits) for the NEXT step's scale ("delayed scaling": mvuld_fp8_roll_scales); C may then be null (inference: fp8 only).
 * mvuld_layernorm_fwd_q8: mvuld_layernorm_fwd that also emits y as e4m3 under q_state[0] and folds max|y| into q_state[1].
 * mvuld_fp8_roll_scales: for each of n {scale, amax} pairs: amax > 0 ? (scale = amax / 448, amax = 0) : unchanged. */
int mvuld_quant_e4m3(const void* x, int64_t n, int dtype, void* out, float* scale_out, float* partials, mvuld_stream_t stream);
/* the same for many fp32 tensors in three launches (all QKV / FFN weights after an optimizer step).  jobs: device array of
 * {const float* src; uint8_t* dst; float* scale; int64_t n (% 8 == 0); int64_t blk0 (first of the tensor's ceil(n / 8192) blocks)};
 * partials: total_blocks floats of scratch. */
int mvuld_quant_e4m3_batched(const void* jobs, int njobs, int64_t total_blocks, float* partials, mvuld_stream_t stream);
int mvuld_gemm_nt_fp8(const void* A8, int64_t lda, const void* B8, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                      const float* bias, int epilogue, void* aux, int64_t ldaux, const float* scale_a, const float* scale_b,
                      void* q_out, int64_t ldq, float* q_state, mvuld_stream_t stream);
int mvuld_layernorm_fwd_q8(const void* x, const void* pre, void* xsum, const float* gamma, const float* beta, const void* residual,
                           const float* rowscale, int rows_per_sample, void* y, float* mean, float* rstd, int64_t rows, int C, float eps,
                           void* q_out, float* q_state, mvuld_stream_t stream);
int mvuld_fp8_roll_scales(float* state, int n, mvuld_stream_t stream);

/* Routing of mvuld_gemm_nt's bf16 -> bf16 plain-store products to the persistent 256 x 256-tile kernel (csrc/gemm_p256.hip):
 * 0 = never, 1 = default rule (>= 96 tiles, N % 128 == 0, at most half of the last column tile empty: the tall Linear layers of the two encoders, same call sites as
 * mvuld_gemm_nt), 2 = every legal shape (K % 32 == 0, K >= 128, N % 8 == 0, no ELU epilogues; tests and A/B timing).  Host-side setting, no stream. */
/* CUs the 256x256 weight-gradient kernel plans its contraction splits for: 0 = all (best alone); the fused training step sets half
 * the chip while its streams run concurrently (a smaller footprint beside the data-gradient chain: step -1 %) */
int mvuld_set_gemm_tn256_budget(int cus);
/* Schedule of the 256x256 weight-gradient kernel's main loop: 1 (default) = ping-pong (the two waves of a SIMD half a slab step apart:
 * one reads its transposed fragments while the other owns the matrix pipe), 0 = lockstep.  Bit-identical weight gradients;
 * MVULD_TN256_PINGPONG. */
int mvuld_set_gemm_tn256_pingpong(int on);
int mvuld_set_gemm_p256_mode(int mode);
/* Tile height of that kernel: 0 = chosen per shape so the tiles fill whole rounds of the persistent grid (default),
 * or 128 / 160 / 192 / 224 / 256 rows for every launch (A/B timing, tests). */
int mvuld_set_gemm_p256_rows(int rows);
/* Tile walk of that kernel on launches with more tiles than workgroups: 0 = static (workgroup b owns tiles b, b + G, ...),
 * 1 = dynamic: only a workgroup's first tile is fixed, the others are claimed from per-XCD counters, so a workgroup whose CU was held by
 * another kernel when the launch began (a collective's channels, another stream's persistent grid) claims what is left instead of owning
 * a full share; 2 = the first tile is claimed too (one exposed ~2 us round trip per launch; no tile waits for a workgroup that has not
 * started: the mode for runs with a resident collective, mvuld_amd/distributed.py selects it).  Bit-identical results (same tiles, same
 * arithmetic per tile); the library keeps one 64-byte counter block per stream.  Launches on a stream under capture keep the static
 * walk.  MVULD_GEMM_DYNAMIC_TILES. */
int mvuld_set_gemm_dynamic_tiles(int mode);
/* Schedule of that kernel's main loop: 1 (default) = ping-pong -- the two waves of every SIMD run half a k-step apart, one reading its
 * fragments from LDS while the other owns the matrix pipe; 0 = both in lockstep (one barrier per k-step).  Bit-identical results;
 * initialised from MVULD_P256_PINGPONG (A/B timing, tests). */
int mvuld_set_gemm_p256_pingpong(int on);
/* Ring geometry of that kernel for bf16 products with K % 64 == 0: 1 (default) = 64-deep stages fetched as full 128-byte lines (8 rows x 128 bytes
 * per LDS-DMA instruction; 2 stages, 3 at <= 160-row tiles), 0 = 32-deep stages (16 rows x 64 bytes per instruction, 4 stages).
 * Bit-identical results; initialised from MVULD_P256_K64. */
int mvuld_set_gemm_p256_k64(int on);
/* Fused MLP of the narrow Swin stages (Mlp.forward, swin_transformer_v2.py:26-32, and its autograd), C = 128 / 256, bf16, hidden = 4C:
 * at these widths the MLP's products are HBM streams; these two kernels keep the hidden dimension on the chip (csrc/mlp_panel.hip).
 *   fwd: y [M, C] = gelu(x W1^T + b1) W2^T + b2 and the activation h [M, 4C] (the fc2 weight gradient reads it; null = not
 *        written: inference); no pre-activation.
 *   bwd: dh [M, 4C] = (dy W2) o gelu'(x W1^T + b1) (pre-activation recomputed; the fc1 weight gradient reads dh) and
 *        dx [M, C] = dh W1 + g (g: residual gradient, may be null).  w2t = fc2.weight^T [4C, C], w1t = fc1.weight^T [C, 4C].
 * Same values as mvuld_gemm_nt with the GELU / dGELU / residual-join epilogues up to bf16 rounding of the stored intermediates
 * (the recomputed pre-activation is fp32 here, bf16 there).  All operands 16-byte aligned, row-major, leading dimension = width. */
int mvuld_mlp_fused_supported(int C);
int mvuld_mlp_fused_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* h, void* y, int M, int C,
                        mvuld_stream_t stream);
int mvuld_mlp_fused_bwd(const void* x, const void* dy, const void* g, const void* w1, const float* b1, const void* w2t, const void* w1t,
                        void* dh, void* dx, int M, int C, mvuld_stream_t stream);
/* Weight gradient on the matrix cores without transposes: dW[N,K] += dY[M,N]^T . X[M,K] (bf16 operands in their token-major
 * layout, fp32 accumulate); dbias[N] += column sums of dY when non-null.  The token contraction is split over workgroups.
 * `ws` (optional, caller-owned, ws_bytes >= mvuld_gemm_tn_wgrad_workspace_bytes(M, N, K, splitk), 16-byte aligned, ZEROED ONCE by
 * the caller when allocated, then private to one stream) lets the splits exchange fp32 partial tiles instead of adding
 * every partial into dW with fp32 atomics: weights that fill 256 x 256 tiles (M % 32 == 0) take the LDS-DMA kernel of
 * csrc/gemm_tn256.hip (its own split plan + a reduction launch; `splitk` is ignored), 2..8-way splits of the 128 x 128
 * kernel a last-arriver reduction; everything else, and every call without `ws`, the atomic form.
 * The autograd of every nn.Linear weight/bias on the path (same call sites as mvuld_gemm_nt). */
int64_t mvuld_gemm_tn_wgrad_workspace_bytes(int M, int N, int K, int splitk);   /* 0 = no workspace form for this shape */
int mvuld_gemm_tn_wgrad(const void* dY, int64_t ldy, const void* X, int64_t ldx, float* dW, int64_t ldw, int M, int N, int K,
                        float* dbias, int splitk, void* ws, int64_t ws_bytes, mvuld_stream_t stream);
/* 0: never route mvuld_gemm_tn_wgrad to the 256 x 256-tile kernel (A/B timing, tests); 1 (default): as described above. */
int mvuld_set_gemm_tn256(int on);
/* The weight gradients of ONE transformer block (up to 8 products) in one launch of the 256 x 256-tile kernel plus one reduction
 * launch: together the products have enough tiles to fill the chip with ~5-way instead of 16-21-way contraction splits, a third of
 * the fp32 partial-slab traffic.  desc: njobs x 10 int64 {dY, ldy, X, ldx, dW, ldw, M, N, K, dbias (0 = none)}, read on the host
 * during the call (the table travels as a kernel argument).  Every product must satisfy mvuld_gemm_tn_wgrad_group_ok (M >= 256,
 * N, K, ldy, ldx multiples of 8, N x K at least 80 % of its 256 x 256 tiles, operands < 2 GiB); ws: caller-owned, 16-byte aligned,
 * >= mvuld_gemm_tn_wgrad_group_workspace_bytes(desc, njobs) (-1 = not eligible), private to one stream; it may be the buffer handed to
 * mvuld_gemm_tn_wgrad on the same stream (same layout: the first 4096 bytes, that entry point's zeroed ticket block, are left alone).
 * Same arithmetic as mvuld_gemm_tn_wgrad on the same kernel (partials summed in split order, then one fp32 atomic add per element).
 * The autograd of the nn.Linear weights of a SwinTransformerBlock (swin_transformer_v2.py:270-306) / RobertaLayer. */
int mvuld_gemm_tn_wgrad_group_ok(int M, int N, int K, int64_t ldy, int64_t ldx);
int64_t mvuld_gemm_tn_wgrad_group_workspace_bytes(const int64_t* desc, int njobs);
int mvuld_gemm_tn_wgrad_group(const int64_t* desc, int njobs, void* ws, int64_t ws_bytes, mvuld_stream_t stream);

/* dst[b][c][r] = src[b][r][c]  (activation / weight transposes feeding the NT GEMM in backward) */
int mvuld_transpose(const void* src, void* dst, int R, int C, int batch, int dtype, mvuld_stream_t stream);

/* njobs independent 2-D transposes in one launch: jobs = device array of {const void* src; void* dst; int64 R, C, tile0}
 * (40 bytes each; tile0 = running sum of ceil(R/64)*ceil(C/64) over the preceding jobs; total_tiles = the full sum).
 * Refreshes every transposed weight copy (the W^T operands of the dgrad GEMMs, autograd of Linear) after an optimizer step. */
int mvuld_transpose_batched(const void* jobs, int njobs, int64_t total_tiles, int dtype, mvuld_stream_t stream);

/* out[c] += sum_r x[r*ld + c]  -- bias gradients (autograd of the `+ bias` in every Linear) */
int mvuld_colsum(const void* x, int64_t ld, float* out, int64_t M, int N, int dtype, mvuld_stream_t stream);

/* y = residual + rowscale[row / rows_per_sample] * (LayerNorm(x + pre) * gamma + beta).
 * Swin res-post-norm + DropPath: swin_transformer_v2.py:301,304; plain LN :492,:362,:632; RoBERTa post-LN
 * (LayerNorm(dense + input)) uses `pre` and keeps the sum in `xsum` for backward. */
int mvuld_layernorm_fwd(const void* x, const void* pre, void* xsum, const float* gamma, const float* beta,
                        const void* residual, const float* rowscale, int rows_per_sample, void* y, float* mean,
                        float* rstd, int64_t rows, int C, float eps, int dtype, mvuld_stream_t stream);
/* y = LayerNorm(dropout(x, p) + pre) * gamma + beta with mvuld_dropout's mask (same seed / counter: bit-identical to mvuld_dropout followed by
 * mvuld_layernorm_fwd), bf16 only: the hidden-state dropouts of RobertaSelfOutput / RobertaOutput (HF modeling_roberta) in one pass. */
int mvuld_layernorm_fwd_drop(const void* x, const void* pre, void* xsum, const float* gamma, const float* beta, void* y, float* mean,
                             float* rstd, int64_t rows, int C, float eps, float drop_p, uint64_t drop_seed, const uint64_t* seed_offset,
                             mvuld_stream_t stream);
/* mvuld_layernorm_bwd (no row scale) that also writes dxd = mvuld_dropout(dx, p, seed), bit-identical to the two launches (bf16, C % 8 == 0,
 * 16-byte aligned): the backward of the same two RoBERTa blocks -- d(dense output) = mask o d(dropout(dense) + input) / (1 - p), while the
 * residual branch keeps dx.  ws as for mvuld_layernorm_bwd. */
int mvuld_layernorm_bwd_drop(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, void* dxd,
                             float* dgamma, float* dbeta, int64_t rows, int C, float* ws, int64_t ws_bytes, float drop_p, uint64_t drop_seed,
                             const uint64_t* seed_offset, mvuld_stream_t stream);
/* Deferred LayerNorm parameter gradients: mvuld_layernorm_bwd (or _drop) with dgamma = dbeta = NULL and a workspace writes dx and
 * nparts = mvuld_layernorm_bwd_nparts(rows, C, ws_bytes, dtype) rows of column partials into ws ([nparts][2C] floats; 0 = this call
 * would not take the partial form: pass the gradients instead), and mvuld_layernorm_bwd_reduce adds them into dgamma / dbeta later --
 * on any stream ordered after the first call (the caller keeps ws alive and unshared until then).  Nothing in backward reads these
 * gradients; launched on the backward chain the 5 us reduction costs the step ~25 us beside other streams' kernels. */
int mvuld_layernorm_bwd_nparts(int64_t rows, int C, int64_t ws_bytes, int dtype);
int mvuld_layernorm_bwd_reduce(const float* ws, int nparts, int C, float* dgamma, float* dbeta, mvuld_stream_t stream);
/* ... any number of them in one launch per 64: desc = njobs x 5 int64 {ws, nparts, C, dgamma, dbeta}, read on the host during the call */
int mvuld_layernorm_bwd_reduce_batch(const int64_t* desc, int njobs, mvuld_stream_t stream);
/* dx for the normalised input (x, or xsum when `pre` was used); dgamma/dbeta accumulate (+=).  `ws` (optional, fp32,
 * ws_bytes >= 512*C) lends room for per-block column partials summed by a second kernel; without it the column sums
 * are device atomics (slower: ~1.5M contended atomics per launch at C=768). */
int64_t mvuld_layernorm_bwd_workspace_bytes(int C);   /* size of `ws` for the two-pass column sums */
int mvuld_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                        const float* rowscale, int rows_per_sample, void* dx, float* dgamma, float* dbeta,
                        int64_t rows, int C, float* ws, int64_t ws_bytes, int dtype, mvuld_stream_t stream);

/* BatchNorm1d over a strided view: element (o,c,i) at o*so + c*sc + i*si, statistics over (o,i).
 * GraphModel.py:153,158,186,187,208 (swinbn, bn_text, bn_gat, bn_bbox, final_fc_bn) and Rs_GCN.py:27-34 (W[1]). */
int mvuld_batchnorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* run_mean,
                        float* run_var, float* save_mean, float* save_rstd, int O, int C, int I, int64_t so,
                        int64_t sc, int64_t si, float eps, float momentum, int training, int dtype, mvuld_stream_t stream);
int mvuld_batchnorm_bwd(const void* dy, const void* x, const float* gamma, const float* save_mean,
                        const float* save_rstd, void* dx, float* dgamma, float* dbeta, int O, int C, int I,
                        int64_t so, int64_t sc, int64_t si, int training, float* dxsum, int dtype, mvuld_stream_t stream);
/* dxsum (optional, [C] fp32, accumulated): per-channel sum of dx taken before dx is rounded to its storage type -- the bias
 * gradient of a Linear / Conv1d(k=1) directly in front of the BatchNorm (Rs_GCN.py:27-34: W = Sequential(Conv1d, BatchNorm1d)). */

/* Fused attention (reference-quality VALU kernels; f32 | bf16 storage).
 * mode 0: SwinV2 shifted-window cosine attention + continuous position bias + shift mask, with the roll /
 *         window_partition / window_reverse of swin_transformer_v2.py:279-299 folded into the token index map
 *         (WindowAttention.forward :140-179).  table16 = 16*sigmoid(cpb_mlp(coords)) [(2ws-1)^2, H].
 * mode 1: pad-masked attention of the UniXcoder encoder (unixcoder.py:35-36; additive -10000 mask).
 * qkv [tokens, 3*H*hd] rows = [3][H][hd]; out [tokens, H*hd]; lse [B*nW, H, N] fp32 saved for backward. */
int mvuld_attn_fwd_simple(int mode, int B, int H, int hd, int N, int nW, int res, int ws, int shift, float scale,
                          const void* qkv, const float* table16, const float* logit_scale, const int* valid,
                          void* out, float* lse, int dtype, mvuld_stream_t stream);
int mvuld_attn_bwd_simple(int mode, int B, int H, int hd, int N, int nW, int res, int ws, int shift, float scale,
                          const void* qkv, const float* table16, const float* logit_scale, const int* valid,
                          const void* out, const void* dout, const float* lse, void* dqkv, float* dtable16,
                          float* dlogit_scale, int dtype, mvuld_stream_t stream);

/* attn_drop_p / drop_seed (modes 1, 2): dropout on the attention probabilities (HF attention_probs_dropout_prob, active when the
 * text encoder trains: RobertaConfig built at unixcoder.py:107-110): kept probabilities are scaled by 1/(1-p) before they multiply
 * V, the softmax normalisation is untouched; the mask is a counter-based hash of (seed, batch, head, query, key), regenerated by
 * the backward passes from the same seed.  p = 0 switches it off. */
/* mode 2 (matrix-core kernels only): mode 1 over PACKED sequences: `valid` carries cu [B+1], N = the longest sequence allowed
 * (sizes LDS and the lse rows: lse is [B, H, N]), `res` = total packed tokens; every packed token is a valid key. */
/* mode 0, head_dim 32, window side a multiple of 4 (the SwinV2 stages with 28 x 28 windows): the FORWARD pass runs on the window fast
 * path (three shifted copies of the bias table read with aligned 8-byte LDS reads, two query tiles per wave).  1 (default) = with the
 * deferred softmax maximum (the running maximum of a query moves only when a block exceeds it by 2^6; out / lse equal the general
 * kernel's to bf16 rounding), 2 = on the general kernel's exact schedule (bit-identical to it), 0 = the general kernel
 * (MVULD_ATTN_WIN). */
int mvuld_set_attn_win(int mode);
/* 1 (default): in shifted blocks, the forward, dQ and dK/dV passes skip the tiles whose pairs all lie across the vertical
 * mask split of a last-row window (each carries the -100 of swin_transformer_v2.py:245-268 and is < 2^-57 of its row's largest
 * probability while tau <= 22; heads with a larger tau compute every pair); 0: nothing is skipped.  Same output bits.  Env: MVULD_ATTN_YSKIP. */
int mvuld_set_attn_yskip(int on);
/* The same attention on the matrix cores (bf16 storage only; v_mfma_f32_16x16x32_bf16, K / V^T (forward), K / K^T / V (dQ pass)
 * and Q~ / dO and their transposes (dK,dV pass) staged in LDS).  ws_delta: caller-owned fp32 [tokens*H] workspace;
 * ws_qt: caller-owned bf16 [tokens, H*hd] workspace (mode 0: normalised queries shared between the dQ and bias-gradient passes).
 * ws_part (optional, mode 0): fp32 room for one (2ws-1)^2 partial bias-table gradient per workgroup of the bias pass
 * (<= B*nW*H*2 of them), summed by a second kernel; without it every workgroup adds its table with device atomics.
 * sample_scale (optional, mode 0): [B] fp32, the per-sample DropPath factor of the residual branch this attention feeds
 * (swin_transformer_v2.py:301: x = shortcut + drop_path(norm1(attn))).  A sample whose factor is exactly 0 contributes nothing to
 * the block's output or to any gradient, so its workgroups write zeros (out, lse; dqkv) and add nothing to the table gradient
 * instead of computing them -- stochastic depth as a saving, the same values downstream.  Pass the SAME vector to both calls. */
int mvuld_attn_fwd_mfma(int mode, int B, int H, int hd, int N, int nW, int res, int ws, int shift, float scale,
                        const void* qkv, const float* table16, const float* logit_scale, const int* valid,
                        void* out, float* lse, float attn_drop_p, uint64_t drop_seed, const uint64_t* seed_offset,
                        const float* sample_scale, int dtype, mvuld_stream_t stream);
int64_t mvuld_attn_bwd_mfma_workspace_bytes(int mode, int B, int H, int nW, int ws);   /* size of `ws_part` */
int mvuld_attn_bwd_mfma(int mode, int B, int H, int hd, int N, int nW, int res, int ws, int shift, float scale,
                        const void* qkv, const float* table16, const float* logit_scale, const int* valid,
                        const void* out, const void* dout, const float* lse, void* dqkv, float* dtable16,
                        float* dlogit_scale, float* ws_delta, void* ws_qt, float* ws_part, int64_t ws_part_bytes,
                        int passes, float attn_drop_p, uint64_t drop_seed, const uint64_t* seed_offset,
                        const float* sample_scale, int dtype, mvuld_stream_t stream);
/* passes: 1 = delta + dQ + dK/dV, 2 = bias-table gradient (mode 0; reads ws_delta / ws_qt written by pass 1), 3 = both.
 * The bias-table gradient feeds nothing else in backward, so a caller may issue pass 2 later, on another stream. */
/* Round 4: the FUSED window backward.  For mode 0, head_dim 32, ws % 4 == 0 and ws <= 28 (SwinV2-base stages 0-2) one kernel forms dQ, dK, dV
 * and the bias-table gradient from a single recomputation of the scores (swin_transformer_v2.py:140-179 backward; the three passes above
 * recompute S, exp and dP once each).  mvuld_attn_bwd_fused_active(mode, hd, ws) = 1 says that mvuld_attn_bwd_mfma will take the geometry that
 * way: the caller then issues ONE call with passes = 3 and a ws_part of mvuld_attn_bwd_mfma_workspace_bytes (required, not optional);
 * a call with passes = 2 alone returns without work.  mvuld_set_attn_bwd_fused(0) / MVULD_ATTN_BWD_FUSED=0 keep the three passes
 * (results agree to the rounding of the bf16 operands and of the order of fp32 sums; tests compare both with the fp32 restatement). */
int mvuld_attn_bwd_fused_active(int mode, int hd, int ws);
int mvuld_set_attn_bwd_fused(int on);

/* Continuous position bias table and its backward: swin_transformer_v2.py:159-163 (cpb_mlp over relative_coords_table) */
int mvuld_cpb_table_fwd(const float* coords, const float* W1, const float* b1, const float* W2, float* hidden,
                        float* table16, int T2, int H, mvuld_stream_t stream);
/* ... for every block at once (the tables depend on parameters only: one launch after each optimizer step).  jobs: device array of
 * {coords, W1, b1, W2, hidden (out), table16 (out), int64 T2, int64 H, int64 row0 = prefix sum of T2}. */
int mvuld_cpb_table_fwd_batched(const void* jobs, int njobs, int64_t total_rows, mvuld_stream_t stream);
int64_t mvuld_cpb_table_bwd_workspace_bytes(int T2, int H);   /* size of `ws` */
int mvuld_cpb_table_bwd(const float* coords, const float* W2, const float* hidden, const float* table16,
                        const float* dtable16, float* dW1, float* db1, float* dW2, int T2, int H,
                        float* ws, int64_t ws_bytes, mvuld_stream_t stream);     /* ws (optional): >= ceil(T2/16)*(32*512+1536)*4 bytes of
                        fp32 scratch for per-block partial sums (else ~1 M device atomics per call) */

/* activation backward: mode 0 GELU(erf) with ref = pre-activation (Mlp, swin_transformer_v2.py:28; RoBERTa
 * intermediate); mode 1 ELU with ref = output (F.elu, GraphModel.py:154-187) */
int mvuld_act_bwd(const void* dy, const void* ref, void* dx, int64_t n, int mode, int dtype, mvuld_stream_t stream);
int mvuld_elu_fwd(const void* x, void* y, int64_t n, int dtype, mvuld_stream_t stream);
int mvuld_cast(const void* x, int dtype_in, void* y, int dtype_out, int64_t n, mvuld_stream_t stream);
int mvuld_add(const void* a, const void* b, void* y, int64_t n, int dtype, mvuld_stream_t stream);
/* y = a * b elementwise: the text x graph feature product of Multi_DefectModel_noGlobalImage (new_model.py:196) */
int mvuld_mul(const void* a, const void* b, void* y, int64_t n, int dtype, mvuld_stream_t stream);
/* y = x * keep / (1-p), keep = hash(seed, index) >= p: nn.Dropout of GraphModel.py:171-177, GATConv feat_drop,
 * RoBERTa hidden dropout; the same call with the same seed is the backward */
int mvuld_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, const uint64_t* seed_offset, int dtype,
                  mvuld_stream_t stream);

/* fp32 -> [hi | lo | hi] (mode 0) or [hi | hi | lo] (mode 1) bf16 rows of width 3*Kp: feeding both to mvuld_gemm_nt with K = 3*Kp
 * gives a near-fp32 product (error ~2^-16) at matrix-core speed; used for the head's fp32 tail (Rs_GCN.py:57-70, GraphModel.py:201-209) */
int mvuld_split_bf16x3(const float* src, int64_t ld, void* dst, int64_t rows, int K, int Kp, int mode, mvuld_stream_t stream);

/* PatchEmbed im2col (swin_transformer_v2.py:490) and PatchMerging's 2x2 gather-concat (:352-359) / its inverse */
int mvuld_im2col_patch4(const float* img, void* cols, int B, int S, int dtype, mvuld_stream_t stream);
int mvuld_patch_merge_gather(const void* src, void* dst, int B, int res, int C, int inverse, int dtype, mvuld_stream_t stream);

/* RoBERTa embeddings (HF RobertaEmbeddings as called from unixcoder.py:36) */
int mvuld_position_ids(const int64_t* ids, int* pos, int* valid, int B, int L, int pad, mvuld_stream_t stream);
int mvuld_embed_fwd(const int64_t* ids, const int* pos, const float* word, const float* posw, const float* type0,
                    void* out, int64_t ntok, int H, int vocab, int maxpos, int dtype, mvuld_stream_t stream);
int mvuld_embed_bwd(const int64_t* ids, const int* pos, const void* dy, float* dword, float* dposw, int64_t ntok, int H,
                    int vocab, int maxpos, int dtype, mvuld_stream_t stream);

/* DropPath factors of a whole forward in one launch: out[k*B + b] = keep(seed, k, b) / (1 - rates[k]) (timm DropPath semantics,
 * swin_transformer_v2.py:301,304); rates [nblk] fp32 on the device (< 1), counter-based hash, no host RNG or copy. */
int mvuld_droppath_scales(const float* rates, float* out, int nblk, int B, uint64_t seed, const uint64_t* seed_offset,
                          mvuld_stream_t stream);
/* `seed_offset` (optional, here and in mvuld_dropout / mvuld_attn_*_mfma): one uint64 in DEVICE memory mixed into the host seed --
 * the step counter of a hipGraph-captured training step (host seeds are frozen into the captured kernel arguments; the counter is
 * advanced by mvuld_counter_add inside the graph, so every replay draws fresh masks).  NULL = host seed only. */
int mvuld_counter_add(uint64_t* counter, uint64_t inc, mvuld_stream_t stream);

/* y[i] = x[i] * s[0], s one fp32 on the device: upstream gradient of the scalar loss applied to dlogits (autograd of
 * CrossEntropyLoss when the loss is scaled or summed with other terms, main_bigvul.py:331-333) */
int mvuld_scale_by_dev(const float* x, const float* s, float* y, int64_t n, mvuld_stream_t stream);

/* Packed (pad-free) token sequences.  The reference pads every function / source line to 512 tokens and masks the pad keys
 * (unixcoder.py:33-38,56-68; data_list.py:293-299); pad rows never reach a result (masked mean, :37), so the text encoder may
 * run on the non-pad tokens only.  cu [B+1] (int32, device): cu[b] .. cu[b+1]-1 are the packed rows of sequence b (exclusive scan
 * of the per-sequence non-pad counts).  pack_tokens writes the packed ids, their HF position ids and rowmap[t] = b*L + l (the
 * padded row of packed row t).  segment_mean_* = the masked mean over a packed sequence and its gradient.
 * rows_map: scatter = 0: dst[t] = src[map[t]]; scatter = 1: dst[map[t]] = src[t]  (t < rows): unpacking to [B*L, C] and back. */
int mvuld_pack_tokens(const int64_t* ids, const int* cu, int64_t* ids_packed, int* pos_packed, int* rowmap, int B, int L, int pad,
                      mvuld_stream_t stream);
int mvuld_segment_mean_fwd(const void* x, const int* cu, void* out, int B, int C, int dtype, mvuld_stream_t stream);
int mvuld_segment_mean_bwd(const void* dout, const int* cu, void* dx, int B, int C, int dtype, mvuld_stream_t stream);
int mvuld_rows_map(const void* src, const int* map, void* dst, int64_t rows, int C, int scatter, int dtype, mvuld_stream_t stream);

/* (masked) mean over tokens: AdaptiveAvgPool1d (swin_transformer_v2.py:633) and the sentence vector (unixcoder.py:37) */
int mvuld_mean_pool_fwd(const void* x, const int* valid, void* out, int B, int L, int C, int dtype, mvuld_stream_t stream);
int mvuld_mean_pool_bwd(const void* dout, const int* valid, void* dx, int B, int L, int C, int dtype, mvuld_stream_t stream);

/* l2norm over the node axis (no eps) then mean over nodes: GraphModel.py:74-79,201-204 */
int mvuld_l2norm_mean_fwd(const void* g, void* hf, float* ssum, float* snrm, int B, int Nn, int C, int dtype, mvuld_stream_t stream);
int mvuld_l2norm_mean_bwd(const void* g, const void* dhf, const float* ssum, const float* snrm, void* dg, int B, int Nn,
                          int C, int dtype, mvuld_stream_t stream);

/* CrossEntropyLoss + softmax: main_bigvul.py:298,330-333 */
int mvuld_cross_entropy(const float* logits, const int64_t* target, float* loss, float* probs, float* dlogits, int B,
                        int K, float loss_scale, mvuld_stream_t stream);
/* The Swin fine-tune job's criteria (main.py:136-140): SoftTargetCrossEntropy on mixed targets (target [B, K] fp32, target_i null) or
 * LabelSmoothingCrossEntropy (target_i [B] int64 + smoothing, target null): loss += mean_b sum_k -t log softmax * loss_scale,
 * dlogits = (probs * sum_k t - t) * loss_scale / B.  timm (third party) is absent from the reference tree: restated from its algorithm. */
int mvuld_cross_entropy_soft(const float* logits, const float* target, const int64_t* target_i, float smoothing, float* loss,
                             float* probs, float* dlogits, int B, int K, float loss_scale, mvuld_stream_t stream);
/* timm.data.Mixup in "batch" mode (main.py:268-269, data/build.py:86-95), out of place on [B, C, H, W]: y[b] = lam x[b] + (1 - lam) x[B-1-b],
 * or with cutmix != 0 the box rows [yl, yh) x columns [xl, xh) of x[B-1-b] pasted into x[b]; with target / soft_target the mixed,
 * label-smoothed one-hot targets [B, K] (mixup_target: on = 1 - smoothing + smoothing / K, off = smoothing / K).  lam and the box are
 * drawn on the host (numpy, as timm draws them). */
int mvuld_mixup_batch(const void* x, void* y, const int64_t* target, float* soft_target, int B, int C, int H, int W, int K, float lam,
                      int cutmix, int yl, int yh, int xl, int xh, float smoothing, int dtype, mvuld_stream_t stream);
/* The "elem" / "pair" modes of the same class (AUG.MIXUP_MODE, config.py:219): one parameter row per sample, params [B, 6] fp32 on the
 * device = {lam, cutmix flag, yl, yh, xl, xh}; the partner of sample b is B-1-b as in batch mode ("pair": the host writes the same row
 * for b and B-1-b).  Soft targets use each sample's own lam. */
int mvuld_mixup_rows(const void* x, void* y, const int64_t* target, float* soft_target, int B, int C, int H, int W, int K,
                     const float* params, float smoothing, int dtype, mvuld_stream_t stream);

/* GATConv sparse part (dgl 0.8.1 u_add_v / edge_softmax / u_mul_e_sum; GraphModel.py:167-170) over CSR by destination */
int mvuld_gat_scores_fwd(const void* ft, const float* al, const float* ar, float* el, float* er, int N, int H, int O,
                         int dtype, mvuld_stream_t stream);
int mvuld_gat_aggregate_fwd(const void* ft, const float* el, const float* er, const int* indptr_dst, const int* src_by_dst,
                            const float* bias, void* out, float* alpha, int N, int E, int H, int O, float slope, int dtype,
                            mvuld_stream_t stream);
int mvuld_gat_aggregate_bwd(const void* dout, const void* ft, const float* el, const float* er, const float* alpha,
                            const float* al, const float* ar, const int* indptr_dst, const int* src_by_dst,
                            const int* indptr_src, const int* dst_by_src, const int* slot_by_src, void* dft, float* dal,
                            float* dar, float* ws_dlogit, float* ws_der, float* ws_del, int N, int E, int H, int O,
                            float slope, int dtype, mvuld_stream_t stream);

/* unbatch_features: pad with zero rows / truncate to max_node (GraphModel.py:30-54) */
int mvuld_segment_pad_fwd(const void* h, const int* node_offsets, void* out, int B, int maxn, int F, int dtype, mvuld_stream_t stream);
int mvuld_segment_pad_bwd(const void* dout, const int* node_offsets, void* dh, int B, int maxn, int F, int64_t total_nodes,
                          int dtype, mvuld_stream_t stream);

/* clip_grad_norm_(5.0) + AdamW: utils_multi.py:229-232, optimizer.py:27-31 */
/* out[0] += sum x^2, bit-reproducible (no float atomics: every data-parallel rank must get the same clip coefficient);
 * partials = caller-owned scratch of >= 2048 floats */
int mvuld_sumsq(const float* x, int64_t n, float* partials, float* out, mvuld_stream_t stream);
int mvuld_clip_coef(const float* sumsq, float max_norm, float grad_scale, float* norm_out, mvuld_stream_t stream);
int mvuld_adamw(float* p, float* g, float* m, float* v, void* p16, int64_t n, float lr, float beta1, float beta2,
                float eps, float weight_decay, int step, const float* coef, int zero_grad, const float* dev_hyper,
                mvuld_stream_t stream);
/* zero_grad != 0: the kernel also clears g after consuming it (replaces optimizer.zero_grad()'s separate pass).
 * dev_hyper (optional, 3 floats on the device): {lr, 1 - beta1^step, sqrt(1 - beta2^step)} read by the kernel INSTEAD of lr / step --
 * the per-iteration cosine learning rate and the bias corrections of a hipGraph-captured step, refreshed by a small copy before
 * each replay. */

/* SwinV2's qkv bias (q_bias, zeros, v_bias) (swin_transformer_v2.py:147-150) for every block in one launch.
 * jobs: device array of {const float* q_bias; const float* v_bias; float* dst [3C]; int64_t C}. */
int mvuld_qkv_bias_pack_batched(const void* jobs, int njobs, mvuld_stream_t stream);

/* CSR index of a batched graph on the device (what DGL builds lazily on the host for GATConv, GraphModel.py:171-176): edges grouped by
 * destination and by source in edge-id order (stable), and for every edge in by-source order its slot in the by-destination order.
 * src / dst: int64 [E] device arrays of node ids < N; outputs int32 (indptr_* [N + 1], the others [E]); ws: scratch of
 * mvuld_graph_csr_workspace_bytes(E) bytes.  Identical to the host builder mvuld_amd/graph.py: BatchedGraph.index. */
int64_t mvuld_graph_csr_workspace_bytes(int E);
int mvuld_graph_csr_build(const int64_t* src, const int64_t* dst, int E, int N, int* indptr_dst, int* src_by_dst, int* indptr_src,
                          int* dst_by_src, int* slot_by_src, void* ws, int64_t ws_bytes, mvuld_stream_t stream);

/* Image ingestion on the device -- the reference's evaluation transform, data/build.py:146-168:
 *   transforms.Resize((S, S), bicubic) [= PIL.Image.resize, Pillow Resample.c] -> ToTensor -> Normalize(mean, std).
 * images [B][H][W][3] uint8 RGB -> out [B][3][Ho][Wo] fp32 / bf16, bit-exact with Pillow's two-pass 8-bit resampler (uint8 rounding
 * after each pass).  bounds_h / kk_h (horizontal) and bounds_v / kk_v (vertical): Pillow's precompute_coeffs + normalize_coeffs_8bpc
 * tables for the axis (int32 [n_out][2] = first tap, tap count; int32 [n_out][ksize] 22-bit fixed point), built on the host
 * (mvuld_amd/data/image_ingest.py).  bounds_h == NULL skips the horizontal pass (W == Wo, as Pillow does).  tmp: B*H*Wo*3 bytes.
 * u8_out (optional, [B][Ho][Wo][3]): the resized 8-bit image itself. */
int mvuld_image_resize_bicubic_normalize(const void* images, int B, int H, int W, const int* bounds_h, const int* kk_h, int ksize_h,
                                         const int* bounds_v, const int* kk_v, int ksize_v, int Ho, int Wo, void* tmp, void* out,
                                         int out_dtype, void* u8_out, float mean_r, float mean_g, float mean_b, float std_r, float std_g, float std_b,
                                         mvuld_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
