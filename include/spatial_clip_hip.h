/* spatial_clip_hip.h -- C ABI of libspatialclip_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * Spatial-CLIP contrastive training step (Biogod2020/Spatial-Clip, SpatialClipLitModule.training_step).
 *
 * The reference has no FFI for this path: its hot path is stock PyTorch ops reached from Python
 * (SURVEY.md section 8b).  Each entry point below therefore replaces an ATen op *site* of the reference and
 * cites it (paths relative to the reference repository).  A reference maintainer binds this library with
 * ctypes; the build's own Python host layer does exactly that (spatial-clip_amd/_lib.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; the callee borrows it for the enqueue only;
 *   - all workspace / outputs are caller-allocated (no hipMalloc inside: safe under a caching allocator);
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue, they never synchronise;
 *   - return 0 on success, negative on error; sc_last_error() gives a thread-local message;
 *   - bf16 tensors are passed as void*, fp32 as float*, tile ids as int64 (long long);
 *   - "ld*" = row stride in ELEMENTS; matrices are row-major.
 */
#ifndef SPATIAL_CLIP_HIP_H
#define SPATIAL_CLIP_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* sc_last_error(void);
int sc_abi_version(void);

/* ------------------------------------------------------------------------------------------------ GEMM
 * bf16 MFMA GEMM, fp32 accumulate.  Replaces nn.Linear / nn.MultiheadAttention in_proj,out_proj / conv1
 * (patch embed as GEMM) forward, dgrad and wgrad:  src/open_clip/transformer.py:253,260-264,624-630.
 *   mode NT: C[M,N] = A[M,K] . B[N,K]^T   (K % 8 == 0; K % 64 == 0 takes the fast path)
 *   mode TN: C[M,N] = At[K,M]^T . Bt[K,N] (weight gradients straight from row-major activations; any K)
 * epilogues:
 *   SC_EPI_BF16          C(bf16) = acc
 *   SC_EPI_BF16_BIAS     C(bf16) = acc + bias[N]
 *   SC_EPI_F32_BIAS_RES  C(f32)  = acc + bias[N] + res[M,N](f32)        (residual stream update)
 *   SC_EPI_GELU_PAIR     C(bf16) = u = acc + bias ; C2(bf16) = gelu_erf(u)  (mlp.c_fc + nn.GELU)
 *   SC_EPI_BF16_DGELU    C(bf16) = acc * gelu'(aux[M,N](bf16))          (c_proj dgrad fused with GELU bwd)
 *   SC_EPI_F32           C(f32)  = acc ; with splitk > 1 partial slabs go to `slabs` and are reduced into C
 *   SC_EPI_BF16_BIAS_RES C(bf16) = acc + bias[N] + res[M,N](bf16)       (residual stream kept in bf16, as the reference's
 *                        autocast keeps it; `res` then points at bf16 data and ldres counts bf16 elements; one rounding)
 *   SC_EPI_GELU_GRAD_PAIR  u = bf16(acc + bias) ; C(bf16) = gelu_erf'(u) ; C2(bf16) = gelu_erf(u)   (round 4: what the
 *                        backward of nn.GELU needs is the factor gelu'(u), not u -- the forward already holds the erf
 *                        pieces in registers, so the default path stores the factor and never u; the u-storing
 *                        SC_EPI_GELU_PAIR / SC_EPI_BF16_DGELU pair stays for activation recomputation, which rebuilds
 *                        h = gelu(u) from the saved u)
 *   SC_EPI_BF16_MUL_AUX  C(bf16) = acc * aux[M,N](bf16)                 (c_proj dgrad x the stored gelu'(u): dU = dH . g;
 *                        SC_EPI_BF16_DGELU rounds its recomputed factor to bf16 first, so both give the same bits)
 *   SC_EPI_QGELU_PAIR / SC_EPI_BF16_DQGELU / SC_EPI_QGELU_GRAD_PAIR  (round 5) the three GELU epilogues with QuickGELU
 *                        x * sigmoid(1.702 x) and its derivative in place of the erf GELU: the activation of the
 *                        OpenAI-pretrained towers (`quick_gelu: true`, src/open_clip/transformer.py:32-35,
 *                        src/open_clip/model.py:142-145,228, model_configs/ViT-B-16-quickgelu.json).  Same kernels.
 * N % 8 == 0, lda/ldb % 8 == 0, ldc % 4 == 0, 16-byte aligned bases.  Outer-dimension edges are handled. */
enum { SC_GEMM_NT = 0, SC_GEMM_TN = 1 };
enum { SC_EPI_BF16 = 0, SC_EPI_BF16_BIAS = 1, SC_EPI_F32_BIAS_RES = 2, SC_EPI_GELU_PAIR = 3,
       SC_EPI_BF16_DGELU = 4, SC_EPI_F32 = 5, SC_EPI_BF16_BIAS_RES = 6, SC_EPI_GELU_GRAD_PAIR = 7,
       SC_EPI_BF16_MUL_AUX = 8, SC_EPI_QGELU_PAIR = 9, SC_EPI_BF16_DQGELU = 10, SC_EPI_QGELU_GRAD_PAIR = 11 };
int sc_gemm_bf16(int mode, int epi, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                 void* C, int ldc, void* C2, int ldc2, const float* bias, const void* res, int ldres,
                 const void* aux, int ldaux, int splitk, float* slabs, void* stream);
long long sc_gemm_slab_floats(int M, int N, int K, int splitk);
/* Weight AND bias gradient of one Linear in one pass: dW[M,N](f32) = dY[K,M]^T . X[K,N], dbias[M] = column sums of dY
 * (fused into the TN kernel: the sums are taken from the MFMA fragments already in registers).
 * ws: sc_gemm_wgrad_ws_floats() floats. */
long long sc_gemm_wgrad_ws_floats(int M, int N, int K, int splitk);
int sc_gemm_wgrad_bias(const void* dY, int lddy, const void* X, int ldx, int M, int N, int K, float* dW, int ldw,
                       float* dbias, int splitk, float* ws, void* stream);

/* Several weight (+ bias) gradients that share their reduction length K (the token axis of one transformer block) in ONE
 * launch and one slab reduction: problem p is dW_p[M_p, N_p](f32, dense: ldw == N_p) = dY_p[K, M_p]^T . X_p[K, N_p] and,
 * when dbias_p != NULL, dbias_p[M_p] = column sums of dY_p -- sc_gemm_wgrad_bias for up to SC_WGRAD_GROUP_MAX Linears
 * (src/open_clip/transformer.py:253,260-264: in_proj, out_proj, c_fc, c_proj).  All problems run at the same split-K.
 * ws: sc_gemm_wgrad_group_ws_floats() floats. */
enum { SC_WGRAD_GROUP_MAX = 4 };
typedef struct {
    const void* dY; long long lddy;
    const void* X; long long ldx;
    float* dW; float* dbias;
    int M; int N;
} sc_wgrad_desc;
long long sc_gemm_wgrad_group_ws_floats(const sc_wgrad_desc* descs, int n, int K, int splitk);
int sc_gemm_wgrad_group(const sc_wgrad_desc* descs, int n, int K, int splitk, float* ws, void* stream);

/* FP8 (OCP e4m3fn) forward GEMM of BASELINE configs[4] ("fp8 MFMA"; the reference has no fp8 path).  Recipe: every
 * operand ROW (token activations; weight output channel) is scaled by s = 2^floor(log2(448 / amax(row))) when it is
 * quantised (sc_quantize_rows_fp8 writes 1/s per row; fixed_scale > 0 skips the amax pass), the products accumulate in
 * fp32 on the matrix cores (v_mfma_scale_f32_16x16x128_f8f6f4, block scale 1.0) and accumulator (m, n) is multiplied
 * by a_scale_inv[m] * b_scale_inv[n] before the bf16 kernel's epilogue (SC_EPI_BF16 / _BIAS / F32_BIAS_RES / GELU_PAIR /
 * BF16_DGELU / F32).  NT only: C[M,N] = A8[M,K] . B8[N,K]^T, K % 128 == 0, lda / ldb in bytes (= elements), multiples of 16. */
int sc_quantize_rows_fp8(const void* src, int src_is_f32, long long ld_src, int rows, int cols, void* dst_fp8,
                         long long ld_dst, float* scale_inv, float fixed_scale, void* stream);
/* The same for n matrices in ONE launch (the e4m3 copies of all Linear weights after an optimiser step; device tables):
 * desc[8 i ..] = {src pointer, src_is_f32, ld_src, rows, cols, dst pointer, ld_dst, scale_inv pointer} as int64,
 * block_prefix[i] = first 4-row block of matrix i, block_prefix[n] = total_blocks.  Same shape / alignment rules per matrix. */
int sc_quantize_rows_fp8_batched(const long long* desc, const int* block_prefix, int n, int total_blocks, void* stream);
int sc_gemm_fp8(int epi, const void* A8, int lda, const float* a_scale_inv, const void* B8, int ldb,
                const float* b_scale_inv, int M, int N, int K, void* C, int ldc, void* C2, int ldc2, const float* bias,
                const void* res, int ldres, const void* aux, int ldaux, void* stream);
/* sc_gemm_fp8 with (a) a_scale_scalar != 0: a_scale_inv points at ONE factor for all rows of A8 (an operand quantised
 * with a per-tensor scale) and (b) an optional e4m3 copy of the epilogue's bf16 output (SC_EPI_GELU_PAIR: h;
 * SC_EPI_BF16_DGELU: dU) for the next GEMM: q8_out[M][ldq8] = e4m3(value * *q8_scale), and max|value| of the launch is
 * max-reduced into q8_amax[64].  sc_fp8_scale_update turns the maxima of n tensors (amax_slots[n][64]) into next step's
 * power-of-two scales (2^(floor(log2(448 / amax)) - margin_bits), left alone when nothing was recorded) and clears the
 * slots: "delayed scaling" -- a tile cannot know its rows' maxima, the step before can. */
int sc_gemm_fp8_q(int epi, const void* A8, int lda, const float* a_scale_inv, int a_scale_scalar, const void* B8, int ldb,
                  const float* b_scale_inv, int M, int N, int K, void* C, int ldc, void* C2, int ldc2, const float* bias,
                  const void* res, int ldres, const void* aux, int ldaux, void* q8_out, long long ldq8,
                  const float* q8_scale, float* q8_amax, void* stream);
int sc_fp8_scale_update(float* amax_slots, float* scale, float* scale_inv, int n, int margin_bits, void* stream);
/* The same with an amax history: this step's maxima are stored in hist[slot][n] (slot = step % hist_len, owned by the
 * caller) and the scales follow the maximum over all hist_len rows -- a quiet step does not expose the next one to the
 * saturating clamp. */
int sc_fp8_scale_update_hist(float* amax_slots, float* hist, int hist_len, int slot, float* scale, float* scale_inv, int n,
                             int margin_bits, void* stream);
/* e4m3 weight (+ bias) gradient (round 4): dW[M,N](f32, dense) = dy_scale_inv * x_scale_inv * dY8[K,M]^T . X8[K,N], both operands
 * token-major e4m3 bytes quantised with ONE scale per tensor -- the reduction runs over the tokens, so the per-token-row scales of
 * sc_layernorm_*_q8 cannot be used here; the delayed per-tensor copies of sc_gemm_fp8_q / sc_layernorm_*_t8 can.  dbias[M] (may be
 * NULL) = dy_scale_inv * column sums of dY8.  M >= 256, N >= 192, M % 16 == N % 16 == 0, K % 128 == 0; lddy / ldx in bytes (% 16);
 * ws: sc_gemm_wgrad_ws_floats(M, N, K, splitk) floats. */
int sc_gemm_wgrad_fp8(const void* dY8, long long lddy, const float* dy_scale_inv, const void* X8, long long ldx,
                      const float* x_scale_inv, int M, int N, int K, float* dW, int ldw, float* dbias, int splitk, float* ws,
                      void* stream);
/* The quantiser fused into the kernels that hold a complete row (round 3): LayerNorm forward also emits the e4m3 copy
 * of its output (A operand of the qkv / c_fc forward GEMMs), LayerNorm backward the e4m3 copy of the new residual
 * gradient (A operand of the c_proj / out_proj data-gradient GEMMs; SC_EPI_BF16_DGELU takes aux = the pre-GELU tensor),
 * each with its per-row 1/scale.  Same arguments as sc_layernorm_fwd / sc_layernorm_bwd plus the fp8 pointer, its row
 * stride in bytes (multiple of 4) and scale_inv[rows]. */
int sc_layernorm_fwd_q8(const float* x, long long ldx, const float* gamma, const float* beta, void* y, long long ldy,
                        void* y_fp8, long long ldy8, float* scale_inv, float* mean, float* rstd, int rows, int d,
                        float eps, void* stream);
int sc_layernorm_bwd_q8(const void* dy, long long lddy, const float* x, long long ldx, const float* mean,
                        const float* rstd, const float* gamma, float* dres, long long lddres, void* dres_bf16,
                        long long lddbf, void* dres_fp8, long long ldd8, float* scale_inv, int accumulate,
                        float* dgamma, float* dbeta, float* colsum, float* ws, int rows, int d, void* stream);
/* LayerNorm backward with the residual gradient carried in bf16 -- the precision the reference carries it in: under its
 * bf16 autocast the residual stream x + attn(ln_1(x)) (src/open_clip/transformer.py:253,260-264) and hence its gradient
 * are bf16 tensors.  gout = gin + LNbwd(dy) is read and written in bf16 only (10 instead of 16 bytes per element cross
 * HBM); the fp32 buffer is read for the sparse form (accumulate = -P: the incoming gradient of rows r % P == 0 sits
 * there) and written only when write_f32 != 0 (the last hop, in front of the embedding backward).  gout_fp8 / scale_inv
 * as in sc_layernorm_bwd_q8, or null.  dgamma / dbeta / colsum and the deferred reduction as in sc_layernorm_bwd. */
int sc_layernorm_bwd_g16(const void* dy, long long lddy, const float* x, long long ldx, const float* mean,
                         const float* rstd, const float* gamma, const void* gin_bf16, long long ldgin, float* dres,
                         long long lddres, int write_f32, void* gout_bf16, long long ldgout, void* gout_fp8,
                         long long ldd8, float* scale_inv, int accumulate, float* dgamma, float* dbeta, float* colsum,
                         float* ws, int rows, int d, void* stream);
/* The same two passes on a residual stream that is kept in bf16 (SC_EPI_BF16_BIAS_RES writes it): x points at bf16 rows,
 * ldx counts bf16 elements.  y_fp8 / gout_fp8 + scale_inv as in the _q8 forms or null; gin_bf16 null = the incoming
 * gradient (if accumulate != 0) is read from the fp32 buffer as in sc_layernorm_bwd. */
int sc_layernorm_fwd_x16(const void* x_bf16, long long ldx, const float* gamma, const float* beta, void* y, long long ldy,
                         void* y_fp8, long long ldy8, float* scale_inv, float* mean, float* rstd, int rows, int d,
                         float eps, void* stream);
int sc_layernorm_bwd_x16(const void* dy, long long lddy, const void* x_bf16, long long ldx, const float* mean,
                         const float* rstd, const float* gamma, const void* gin_bf16, long long ldgin, float* dres,
                         long long lddres, int write_f32, void* gout_bf16, long long ldgout, void* gout_fp8,
                         long long ldd8, float* scale_inv, int accumulate, float* dgamma, float* dbeta, float* colsum,
                         float* ws, int rows, int d, void* stream);

/* ------------------------------------------------------------------------------------------------ attention
 * Fused multi-head self-attention on the packed in_proj output qkv[B*L, 3*H*dh] (q | k | v, head h at
 * columns h*dh): softmax(q k^T / sqrt(dh)) v, fp32 softmax, optional causal mask.  Replaces
 * nn.MultiheadAttention's SDPA core (src/open_clip/transformer.py:272-287; mask :1080-1086).
 * out[B*L, H*dh] bf16, lse[B,H,L] fp32 (log-sum-exp of the scaled scores, kept for backward).
 * sc_attn_bwd writes dqkv[B*L, 3*H*dh] (bf16) and uses delta[B,H,L] as scratch.  L <= 320, dh in {32, 64}.
 * q_rows > 0 restricts the work to the first q_rows query positions of every sequence (the last ViT block only
 * feeds the CLS token downstream): outputs of the other rows are not written, dk / dv receive only those queries'
 * contributions; dout / out of the unused rows may hold any FINITE values (they meet exact zeros only).  dq of the
 * other rows: q_rows == 1 writes exact zeros (the caller needs no memset of dqkv); 1 < q_rows < L leaves them untouched. */
int sc_attn_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int dh, int causal, int q_rows,
                void* stream);
int sc_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                int B, int L, int H, int dh, int causal, int q_rows, void* stream);

/* ------------------------------------------------------------------------------------------------ LayerNorm
 * LayerNorm over the last dim of the fp32 residual stream (eps 1e-5; src/open_clip/transformer.py:23-29),
 * bf16 output feeding the next GEMM; mean/rstd [rows] saved for backward (may be NULL).
 * Backward: dres = (accumulate ? dres : 0) + LN'(dy) -- accumulate = 1: every row of dres carries an incoming residual
 * gradient; accumulate = -P: only the rows r with r % P == 0 do (class-token rows behind a class-token-only block), the
 * others are neither read nor pre-zeroed; also writes the bf16 copy of the new dres (may be NULL),
 * dgamma, dbeta and colsum = column sums of the new dres (= bias gradient of the Linear that produced the
 * residual branch; may be NULL).  ws: sc_layernorm_bwd_ws_floats() floats.  d % 4 == 0, d <= 2048.
 * dgamma == NULL defers the column reductions: the per-block partial sums stay in ws and
 * sc_layernorm_bwd_reduce(ws, ...) finishes dgamma / dbeta / colsum later (any stream ordered after the first call;
 * ws must stay untouched until then) -- these only feed the optimiser, not the data-gradient chain. */
int sc_layernorm_fwd(const float* x, long long ldx, const float* gamma, const float* beta, void* y, long long ldy,
                     float* mean, float* rstd, int rows, int d, float eps, void* stream);
long long sc_layernorm_bwd_ws_floats(int rows, int d);
int sc_layernorm_bwd(const void* dy, long long lddy, const float* x, long long ldx, const float* mean,
                     const float* rstd, const float* gamma, float* dres, long long lddres, void* dres_bf16,
                     long long lddbf, int accumulate, float* dgamma, float* dbeta, float* colsum, float* ws,
                     int rows, int d, void* stream);
/* sc_layernorm_fwd_x16 / sc_layernorm_bwd_x16 with a SECOND e4m3 output quantised with one scale for the whole tensor (round 4):
 * y_t8 = e4m3(value * *t_scale), max |value| of the launch max-reduced into t_amax[64] (delayed scaling, sc_fp8_scale_update*).
 * These copies are the X / dY operands of sc_gemm_wgrad_fp8, whose reduction runs over the token rows. */
int sc_layernorm_fwd_x16_t8(const void* x_bf16, long long ldx, const float* gamma, const float* beta, void* y, long long ldy,
                            void* y_fp8, long long ldy8, float* scale_inv, void* y_t8, long long ldt8, const float* t_scale,
                            float* t_amax, float* mean, float* rstd, int rows, int d, float eps, void* stream);
int sc_layernorm_bwd_x16_t8(const void* dy, long long lddy, const void* x_bf16, long long ldx, const float* mean,
                            const float* rstd, const float* gamma, const void* gin_bf16, long long ldgin, float* dres,
                            long long lddres, int write_f32, void* gout_bf16, long long ldgout, void* gout_fp8, long long ldd8,
                            float* scale_inv, void* gout_t8, long long ldt8, const float* t_scale, float* t_amax, int accumulate,
                            float* dgamma, float* dbeta, float* colsum, float* ws, int rows, int d, void* stream);
int sc_layernorm_bwd_reduce(const float* ws, int rows, int d, float* dgamma, float* dbeta, float* colsum,
                            void* stream);

/* u = bf16(x + bias[n]), h = bf16(gelu_erf(u)) on a dense fp32 [rows, n]: epilogue of a split-K forward Linear
 * (gene-MLP fc1, K = 20k genes, M = batch: the K loop is split over the chip and reduced in fp32 first). */
int sc_bias_gelu_pair(const float* x, const float* bias, void* u, void* h, int rows, int n, void* stream);

/* column sums of a bf16 matrix -> fp32 (bias gradients).  ws: sc_colsum_ws_floats() floats. */
long long sc_colsum_ws_floats(int rows, int n);
int sc_colsum_bf16(const void* x, long long ld, int rows, int n, float* out, float* ws, void* stream);

/* F.normalize(dim=-1, eps 1e-12) of the embedding heads (src/open_clip/model.py:328,345) and its backward
 * dx = (dy - y <y,dy>) / max(|x|, eps), emitted in bf16 for the projection dgrad / wgrad GEMMs. */
int sc_l2norm_fwd(const float* x, float* y, void* y_bf16, float* inv_norm, int rows, int d, void* stream);
int sc_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, void* dx_bf16, int rows, int d,
                  void* stream);

/* fp32 -> bf16 casts of master weights / inputs: row-padded copy and transposed copy (dst[c][r] = src[r][c]). */
int sc_cast_pad_bf16(const float* src, long long ld_src, void* dst, long long ld_dst, int rows, int cols,
                     int cols_pad, void* stream);
int sc_cast_transpose_bf16(const float* src, void* dst, int rows, int cols, long long ld_dst, void* stream);
/* All transposed copies in ONE launch: desc[n][5] (device int64) = {src offset in elements from `master` / the
 * mirror, dst pointer, rows, cols, ld_dst}; tile_prefix[n+1] (device int32) = running count of 64x64 tiles.  With
 * mirror_bf16 != NULL (the flat bf16 mirror of the masters, same element offsets, e.g. just written by sc_adamw_step)
 * the copies are made from it -- half the bytes read, bit-identical results; otherwise from the fp32 master. */
int sc_cast_transpose_batched(const float* master, const void* mirror_bf16, const long long* desc,
                              const int* tile_prefix, int n, int total_tiles, void* stream);

/* ------------------------------------------------------------------------------------------------ patch embedding
 * VisionTransformer._embeds (src/open_clip/transformer.py:783-798): conv1 with kernel = stride = patch is a GEMM
 * over the im2col'd patches (inner order c,py,px = conv1.weight.view(width,-1)); then class token, positional
 * embedding and ln_pre, written as the fp32 residual stream x[B*L, d]. */
int sc_im2col(const float* images, void* patches, int B, int C, int H, int W, int P, long long ld_out, void* stream);
int sc_embed_ln_fwd(const float* patch_out, const float* cls, const float* pos, const float* gamma,
                    const float* beta, float* x, float* mean, float* rstd, int B, int L, int d, float eps,
                    void* stream);
/* The same with the residual stream starting in bf16 (model.net.residual_stream: bf16 -- under bf16-mixed the reference's ln_pre
 * output is a bf16 tensor, src/open_clip/transformer.py:26-29,789-791): x_bf16[B*L, d], no fp32 copy and no cast pass. */
int sc_embed_ln_fwd_x16(const float* patch_out, const float* cls, const float* pos, const float* gamma,
                        const float* beta, void* x_bf16, float* mean, float* rstd, int B, int L, int d, float eps,
                        void* stream);
long long sc_embed_ln_bwd_ws_floats(int B, int L, int d);
int sc_embed_ln_bwd(float* dres, const float* patch_out, const float* cls, const float* pos, const float* mean,
                    const float* rstd, const float* gamma, void* dpatch_bf16, float* dgamma, float* dbeta,
                    float* dpos, float* dcls, float* ws, int B, int L, int d, void* stream);

/* ------------------------------------------------------------------------------------------------ text tower glue
 * CLIP.encode_text (src/open_clip/model.py:330-345): x = token_embedding[text] + positional_embedding (fp32 residual
 * stream [B*L, d]); pooling = row at text.argmax(-1) (EOT has the largest id; src/open_clip/transformer.py:931-934).
 * Backward: dtable is zeroed then scatter-added with float atomics (duplicate tokens), dpos[t] = sum_b dres[b,t].
 * gather/scatter move the pooled rows between the [B*L, d] stream and a compact [B, d] buffer. */
int sc_token_embed_fwd(const long long* tokens, const float* table, const float* pos, float* x, int B, int L, int d,
                       int V, void* stream);
int sc_token_embed_bwd(const long long* tokens, const float* dres, float* dtable, float* dpos, int B, int L, int d,
                       int V, void* stream);
/* The same gradients, bit-reproducible (round 6; the default of the text tower): table row t is summed by ONE wave per
 * 256-column slab that walks the token list in order -- no float atomics, so repeated tokens (<start_of_text> in every caption)
 * are added in increasing position.  eot[B] (int32, nullable) = pooled position per caption: positions behind it carry an exactly
 * zero gradient in the causal tower and are skipped. */
int sc_token_embed_bwd_det(const long long* tokens, const int* eot, const float* dres, float* dtable, float* dpos, int B, int L,
                           int d, int V, void* stream);
int sc_argmax_rows_i64(const long long* tokens, int* out_idx, int B, int L, void* stream);
int sc_gather_rows_f32(const float* src, const int* idx, int L, float* dst, int B, int d, void* stream);
int sc_scatter_rows_f32(const float* src, const int* idx, int L, float* dst, void* dst_bf16, int B, int d,
                        void* stream);

/* ------------------------------------------------------------------------------------------------ contrastive head
 * ClipLoss (src/open_clip/loss.py:91-155, local_loss layout) and SpatialLoss
 * (src/models/components/losses.py:44-124) on the device.  z[2][B][G] holds the cosine similarities
 * (z[0] = image_features . all_text^T, z[1] = text_features . all_image^T), produced with sc_sgemm_f32(_grouped).
 * Soft labels are sparse: lab_col/lab_w[2][B][nlab] ((column, weight) pairs, column -1 = unused); they come from
 * sc_onehot_labels (ClipLoss: arange(B)+B*rank, loss.py:94-96) or sc_neighbor_join (SpatialLoss label loop,
 * losses.py:91-111: id -> LAST index, alpha*scale <= 0 skipped, L1-normalised).
 * loss_out[4] = {loss, gap, CE_image, CE_text}; cap_logit_scale <= 0 disables the STE cap; temp_reg_weight 0
 * disables the gap^2 term.  Backward overwrites z with dL/dz and returns d logit_scale (of the *exponentiated*
 * scale) and d logit_bias; feature grads are then two sc_sgemm_f32 calls per direction. */
int sc_sgemm_f32(const float* A, long long sam, long long sak, const float* B, long long sbn, long long sbk,
                 float* C, long long ldc, int M, int N, int K, int accumulate, void* stream);
/* The same GEMM for up to SC_SGEMM_MAX_GROUP independent problems in ONE launch (flat tile list): the two similarity
 * matrices of the forward, the four gradient products of the backward.  C[m][n] (+)= sum_k A[m*sam + k*sak] *
 * B[n*sbn + k*sbk]; exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32).  Each operand needs one unit stride;
 * `descs` is a HOST array. */
#define SC_SGEMM_MAX_GROUP 6
typedef struct {
    const float* A; long long sam; long long sak;
    const float* B; long long sbn; long long sbk;
    float* C; long long ldc;
    int M; int N; int K; int accumulate;
} sc_sgemm_desc;
int sc_sgemm_f32_grouped(const sc_sgemm_desc* descs, int n, void* stream);
/* Send buffer of the feature all-gather (gather_features, src/open_clip/loss.py:21-65 + the tile-id gathers of
 * src/models/components/losses.py:63-68, as ONE payload): out[r] = feat[r][0..D) | ids_a[r] | ids_b[r] (each int64 as
 * two float slots; ids may both be NULL).  D and ldo even when ids are given. */
int sc_pack_rows(const float* feat, long long ldf, const long long* ids_a, const long long* ids_b, float* out,
                 long long ldo, int B, int D, void* stream);
int sc_neighbor_join(const long long* all_image_tile_ids, const long long* all_text_tile_ids,
                     const long long* neighbor_tile_ids, const float* neighbor_alphas, int B, int G, int K, int rank,
                     float neighbor_alpha_scale, int* lab_col, float* lab_w, void* stream);
int sc_onehot_labels(int B, int rank, int* lab_col, float* lab_w, void* stream);
int sc_contrastive_loss_fwd(const float* z, int B, int G, const float* logit_scale, float cap_logit_scale,
                            const float* logit_bias, const int* lab_col, const float* lab_w, int nlab,
                            float temp_reg_weight, float* rowstats, float* loss_out, void* stream);
int sc_contrastive_loss_bwd(float* z_inout, int B, int G, const float* logit_scale, float cap_logit_scale,
                            const float* logit_bias, const int* lab_col, const float* lab_w, int nlab,
                            float temp_reg_weight, const float* rowstats, const float* loss_out,
                            const float* grad_out, float* rowgrad, float* dscale, float* dbias, void* stream);
/* RecallAtK (src/models/components/metrics.py:22-36) on the local [B,B] block of z[0]: hits3 += {R@1,R@5,R@10}. */
int sc_recall_hits(const float* z_image_rows, int G, int B, int col0, int* hits3, void* stream);
/* Validation-only zero-shot gene-expression metric (src/metrics/zero_shot.py:62-88): sample-wise
 * Pearson correlation of pred[rows, cols] (image_features @ gene_bank^T) against the rank-weighted targets, rows with
 * sqrt(sum pc^2) * sqrt(sum tc^2) <= 1e-6 score 0.  pcc[rows] (may be NULL); sum_count[0] += sum(pcc),
 * sum_count[1] += rows (the metric's two states; may be NULL). */
int sc_pcc_rows(const float* pred, long long ldp, const float* target, long long ldt, int rows, int cols, float* pcc,
                float* sum_count, void* stream);
/* logit_scale.exp() (src/models/components/spatial_clip_net.py:51) and its backward dx = dy * y * mult. */
int sc_exp_scalar(const float* x, float* y, void* stream);
int sc_exp_scalar_bwd(const float* y, const float* dy, float* dx, float mult, void* stream);
/* out[n] = x[n] * *s (device scalar): the upstream gradient of the loss applied to the gradients the fused head formed for
 * an upstream gradient of 1 (the `grad_output` of autograd for the loss node, src/models/spatial_clip_module.py:103-108). */
int sc_scale_by_scalar(const float* x, const float* s, float* out, long long n, void* stream);

/* ------------------------------------------------------------------------------------------------ input pipeline
 *  the data-dependent dataloader steps, on the device.
 * sc_knn_alpha: K nearest tiles of every tile of ONE slide from its (x, y) centroids (xy[N][2]), self excluded, ties by
 * index; nbr_index[N][K] holds slide-local indices (-1 = fewer than K other tiles), alpha[N][K] the loss weights,
 * normalised per row: mode 0 weight = 1 / (distance + 1e-6) (docs/spatial_clip_data_pipeline.html, Step 1), mode 1
 * weight = exp(-d^2 / (2 sigma^2)) (notebooks/d1_dataset_construct_cw.ipynb).
 * sc_augment_tiles: decoded uint8 tiles [B][H][W][3] -> fp32 [B][3][S][S]: integer crop box, PIL's antialiased BICUBIC
 * resize (two 8-bit passes, 22-bit fixed-point coefficients: what the reference's resized-crop transform does to a PIL tile,
 * src/open_clip/transform.py:153-154,186-204 with use_timm), optional horizontal flip, ColorJitter with PIL's 8-bit
 * ImageEnhance semantics (brightness / contrast / saturation factors applied in the per-sample order code 0..5),
 * ToTensor, Normalize: byte-identical to the PIL pipeline before ToTensor.
 * params12[B][12] = {x0, y0, crop_w, crop_h, brightness, contrast, saturation, order, flip, 0, 0, 0}; mean3 / std3 are
 * HOST pointers (configs/model/spatial_clip.yaml:12-17, src/open_clip/constants.py:1-2). */
int sc_knn_alpha(const float* xy, int N, int K, int mode, float sigma, int* nbr_index, float* alpha, void* stream);
/* sc_png_decode: B PNG files as they sit in the shards (8-bit RGB / RGBA, not interlaced; concatenated in `files`, file b =
 * bytes [offsets[b], offsets[b+1]), all DEVICE memory) -> uint8 tiles out_rgb[B][H][W][3], one wave per tile: zlib inflate
 * (dynamic / fixed / stored blocks, IDAT chunks joined) and the five scanline filters on the device -- what
 * PIL.Image.open(...).convert("RGB") does on the reference's dataloader workers.  status[b] = 0, or an error code (wrong size,
 * unsupported colour type, damaged stream, ...): the caller decodes those tiles on the host.  scratch:
 * sc_png_decode_scratch_bytes(B, H, W) bytes. */
long long sc_png_decode_scratch_bytes(int B, int H, int W);
int sc_png_decode(const void* files, const long long* offsets, int B, void* out_rgb, int H, int W, void* scratch, int* status,
                  void* stream);
int sc_augment_tiles(const void* src_u8_hwc, int B, int H, int W, const float* params12, float* out_nchw, int S,
                     const float* mean3_host, const float* std3_host, void* stream);

/* h = gelu(u) on contiguous bf16 (exact-erf GELU, the formula and input of the fused GEMM epilogue: bit-identical to the
 * epilogue's second output).  Used by the activation-recomputation mode (open_clip's CLIP.set_grad_checkpointing,
 * src/open_clip/model.py:313-315): the block's GELU output is not kept for the backward.  n % 8 == 0. */
int sc_gelu_bf16(const void* u, void* h, long long n, void* stream);
/* The same pass for towers built with `quick_gelu: true`: h = u * sigmoid(1.702 u) (src/open_clip/transformer.py:32-35),
 * bit-identical to the second output of SC_EPI_QGELU_PAIR. */
int sc_quick_gelu_bf16(const void* u, void* h, long long n, void* stream);

/* ------------------------------------------------------------------------------------------------ RCCL
 * The collectives of the data-parallel step on a caller-supplied HIP stream (enqueue only, never a host wait); RCCL is
 * resolved at run time, reusing the instance the process already loaded.  One process per GPU.
 * sc_comm_unique_id: rank 0 fills 128 bytes that the caller distributes (any side channel); sc_comm_init: collective
 * over `world` processes, binds to the current HIP device, returns an opaque handle (0 on error); sc_comm_destroy.
 * sc_allgather_feats_async: recv[world * bytes_per_rank] = rank-major concatenation of every rank's send block -- the
 * packed features (+ tile ids) of gather_features (src/open_clip/loss.py:21-65, src/models/components/losses.py:58-68).
 * sc_reduce_scatter_grads_async: recv[floats_per_rank] = sum over ranks of block `rank` of send[world * floats_per_rank]
 * (backward of the gather with gather_with_grad=True).  sc_allreduce_sum_async: in-place sum of a gradient bucket
 * (DDP's gradient reduction, configs/trainer/ddp.yaml; the 1/world factor is folded into sc_adamw_step). */
int sc_comm_unique_id(void* id_out_128);
long long sc_comm_init(const void* id_128, int rank, int world);
int sc_comm_destroy(void* comm);
int sc_allgather_feats_async(void* comm, const void* send, void* recv, long long bytes_per_rank, void* stream);
int sc_reduce_scatter_grads_async(void* comm, const float* send, float* recv, long long floats_per_rank, void* stream);
int sc_allreduce_sum_async(void* comm, float* buf, long long n, void* stream);

/* ------------------------------------------------------------------------------------------------ optimiser
 * clip_grad_norm_(max_norm) + AdamW over flat fp32 buffers (src/models/spatial_clip_module.py:138-158,
 * configs/optimizer/adamw.yaml, configs/trainer/default.yaml:19).  grad_scale = 1/world_size folds DDP's
 * gradient mean.  sc_grad_norm writes norm_clip_out[2] = {|g|*grad_scale, min(1, max_norm/(norm+1e-6))};
 * ws: 1024 doubles.  sc_adamw_step reads the clip coefficient from norm_clip[1] (NULL = no clipping) and, if
 * params_bf16 is given, also writes the bf16 mirror of the updated parameters (same flat layout). */
int sc_grad_norm(const float* grads, long long n, float grad_scale, float max_norm, double* ws, float* norm_clip_out,
                 void* stream);
int sc_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                  const float* norm_clip, void* params_bf16, void* stream);
/* sc_adamw_step with its step-dependent scalars in DEVICE memory, hyper[3] = {lr, 1 - beta1^step, sqrt(1 - beta2^step)}:
 * the launch is identical from step to step, so a whole training step (forward, backward, clip, AdamW) can be captured
 * into ONE hipGraph and replayed while the LambdaLR schedule (src/models/spatial_clip_module.py:146-158) and Adam's bias
 * correction advance on the host -- the reference's optimizer.step() + scheduler.step() pair, minus ~700 launches of host
 * work per step on the small-batch configurations (configs/experiment/medium_*.yaml: ViT-B-32, batch 32). */
int sc_adamw_hyper_host(float lr, float beta1, float beta2, int step, float* out3_host);   /* fills a HOST triple for the copy */
int sc_adamw_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                      const float* hyper, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                      const float* norm_clip, void* params_bf16, void* stream);
/* The two halves of sc_grad_norm for an optimiser that owns 1/W of every gradient bucket (reduce-scatter ->
 * AdamW on the rank's shard -> all-gather; Lightning's DDP mean, configs/trainer/ddp.yaml:4, replaced): 1024 fp64 partial
 * sums of squares per contiguous piece of the shard, then -- after the caller has all-reduced (SUM) the concatenated
 * partial arrays over the ranks -- the norm of the whole averaged gradient and the clip coefficient, as sc_grad_norm. */
int sc_grad_sumsq_partial(const float* grads, long long n, double* partial1024, void* stream);
int sc_grad_norm_final(const double* partial, int n_partial, float grad_scale, float max_norm, float* norm_clip_out,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPATIAL_CLIP_HIP_H */
