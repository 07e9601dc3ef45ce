/*
 * ecgbyte_decoder.h -- C ABI of the causal-LM step kernels in libecgbyte_hip.so (MI355X, gfx950).
 *
 * The reference runs this part of the hot path through HuggingFace transformers on ATen kernels
 * (vendored transformers 4.46.0.dev0; no hand-written CUDA of its own on this path).  Each entry
 * point below replaces the ATen calls behind one module; paths relative to the reference root:
 *   ecgb_embed_fwd/bwd     nn.Embedding in LlamaModel.forward, transformers/src/transformers/models/llama/modeling_llama.py:889
 *   ecgb_rmsnorm_fwd/bwd   LlamaRMSNorm.forward, modeling_llama.py:67-72 (gemma=1: GemmaRMSNorm, models/gemma/modeling_gemma.py:51-68)
 *   ecgb_rope              apply_rotary_pos_emb / rotate_half, modeling_llama.py:193-224
 *   ecgb_gemm_nt_bf16      every nn.Linear of the block (q/k/v/o_proj, gate/up/down_proj, lm_head), modeling_llama.py:227-395,1209
 *   ecgb_attn_fwd/bwd      LlamaSdpaAttention.forward incl. the causal + left-padding mask, modeling_llama.py:526-614,981-1100
 *   ecgb_softmax_causal_fwd/bwd   the same through materialised scores (QK^T and PV as ecgb_gemm_nt_bf16 batches);
 *                          kept as the second implementation the fused kernels are tested against
 *   ecgb_glu_fwd/bwd       LlamaMLP act_fn(gate) * up, modeling_llama.py:238-258 (gelu_tanh=1: GemmaMLP)
 *   ecgb_ce_fwd_bwd        ForCausalLMLoss, transformers/src/transformers/loss/loss_utils.py:24-47
 *   ecgb_sumsq, ecgb_adam_step   clip_grad_norm_(1.0) + Adam(weight_decay = L2), ecg_byte/runners/train.py:26, ecg_byte/main.py:262-264
 * All tensors are device pointers; bf16 = IEEE bfloat16 bit patterns; `stream` as in ecgbyte.h.
 * Same status codes as ecgbyte.h.
 */
#ifndef ECGBYTE_DECODER_H
#define ECGBYTE_DECODER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* out[t,:] = table[ids[t],:] * scale (bf16);  grad_table(fp32)[ids[t],:] += dout[t,:] * scale */
int ecgb_embed_fwd(const int64_t *ids_dev, const void *table_dev, void *out_dev, size_t tokens, int hidden,
                   float scale, void *stream);
int ecgb_embed_bwd(const int64_t *ids_dev, const void *dout_dev, float *grad_table_dev, size_t tokens, int hidden,
                   float scale, void *stream);
/* The same scatter-add into a BF16 table without atomics: tokens sorted by id (ids_sorted_dev, order_dev = the stable argsort of ids); every
 * run of equal ids is added in sorted order, in fp32, on top of what the row holds (the tied lm_head's gradient) -- the same bits every
 * launch, no fp32 copy of the table.  skip_id: the padding_idx row (no gradient from the lookup, modeling_llama.py:889), or -1. */
int ecgb_embed_bwd_sorted(const int64_t *ids_sorted_dev, const int64_t *order_dev, const void *dout_dev, void *grad_table_dev,
                          size_t tokens, int hidden, float scale, int64_t skip_id, void *stream);

/* y = w * bf16(x * rsqrt(mean(x^2)+eps)).  If residual_dev != NULL: x := x + residual first, written to
 * sum_out_dev.  rstd_dev (fp32 per row) is saved for the backward. */
int ecgb_rmsnorm_fwd(const void *x_dev, const void *residual_dev, const void *w_dev, void *y_dev, void *sum_out_dev,
                     float *rstd_dev, size_t rows, int hidden, float eps, int gemma, void *stream);
/* dx = rstd * (dy*w - xhat * mean(dy*w*xhat)) [+ dres];  dw_dev (fp32) += sum over rows of dy * xhat; dw_dev NULL: frozen norm weights, no weight gradient */
/* scratch_dev (ecgb_rmsnorm_bwd_scratch_floats floats, or null): with it the weight gradient is summed in a fixed order (per-workgroup partial
 * rows added in workgroup order: the same bits every launch; hidden 2048 / 4096), without it by LDS + global float atomics. */
size_t ecgb_rmsnorm_bwd_scratch_floats(size_t rows, int hidden);
int ecgb_rmsnorm_bwd(const void *x_dev, const void *w_dev, const float *rstd_dev, const void *dy_dev,
                     const void *dres_dev, void *dx_dev, float *dw_dev, size_t rows, int hidden, int gemma, float *scratch_dev, void *stream);

/* In-place rotary embedding of x [tokens, n_heads, head_dim] (row stride in elements), half-split
 * layout; cos/sin fp32 [tokens, head_dim/2].  inverse=1 applies the transpose rotation (backward). */
int ecgb_rope(void *x_dev, const float *cos_dev, const float *sin_dev, size_t tokens, int n_heads, int head_dim,
              size_t row_stride, int inverse, void *stream);

/* h = act(gate) * up with gate|up stored side by side: gate_up [tokens, 2*inter], h [tokens, inter] */
int ecgb_glu_fwd(const void *gate_up_dev, void *h_dev, size_t tokens, int inter, int gelu_tanh, void *stream);
int ecgb_glu_bwd(const void *gate_up_dev, const void *dh_dev, void *dgate_up_dev, size_t tokens, int inter,
                 int gelu_tanh, void *stream);

/* Inverted dropout with a counter-based generator: out[i] = keep(seed, i) ? x[i] / (1 - p) : 0; the same (seed, p)
 * reproduces the mask, so the backward pass is the same call on the gradient.  In place allowed. */
int ecgb_dropout_bf16(const void *x_dev, void *out_dev, size_t n, float p, uint64_t seed, void *stream);
int ecgb_add_bf16(const void *a_dev, const void *b_dev, void *out_dev, size_t n, void *stream);
int ecgb_transpose_bf16(const void *in_dev, void *out_dev, int rows, int cols, void *stream);
/* Batch of strided matrices: entry z = (z / inner, z % inner) starts at base + zo*outer + zi*inner_stride. */
int ecgb_transpose_bf16_strided(const void *in_dev, void *out_dev, int rows, int cols, long long ld_in, long long ld_out,
                                int batch, int inner, long long outer_in, long long inner_in, long long outer_out,
                                long long inner_out, void *stream);
/* n row-major matrices transposed in ONE launch (dst[t] = src[t]^T, src[t] is rows[t] x cols[t]); tile_off_dev[t] = number of 64x64 tiles of the
 * matrices before t, total_tiles = their sum over all n.  The LoRA adapters' transposed copies after an optimizer step. */
int ecgb_transpose_multi_bf16(const void *const *src_dev, void *const *dst_dev, const int *rows_dev, const int *cols_dev,
                              const int *tile_off_dev, int n, int total_tiles, void *stream);
int ecgb_f32_to_bf16(const float *in_dev, void *out_dev, size_t n, void *stream);
/* out[i] = bf16(sum_s slabs[s * slab_stride + i]) for i < n, summed in slab order; accumulate != 0: added to the bf16 already in out.
 * The K-slices of ecgb_gemm_tn_bf16 (splits > 1) meet here.  n, slab_stride multiples of 4. */
int ecgb_sum_slabs_bf16(const float *slabs_dev, long long slab_stride, int n_slabs, void *out_dev, size_t n, int accumulate, void *stream);

/* C[M,N] = alpha * A[M,K] . B[N,K]^T, bf16 operands, fp32 accumulation on the matrix cores.
 * accumulate_f32 = 0: C is bf16; 1: C is fp32 and C += result; 2: C is bf16 and C += result (LoRA branch added to
 * the base projection).  batch > 1: blockIdx.z strides. */
int ecgb_gemm_nt_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc,
                      int M, int N, int K, float alpha, int accumulate_f32, int batch, long long batch_a,
                      long long batch_b, long long batch_c, void *stream);
/* K-concatenated product: C[M,N] = alpha * (A[M,K] . B[N,K]^T + A2[M,K2] . B2[N,K2]^T) in ONE launch (K, K2 multiples of 64;
 * same accumulate modes).  A LoRA branch rides in its projection's launch: A2 = t = scale * dropout(x) lora_A^T [M, 64],
 * B2 = lora_B [N, 64] (peft LoraLayer.forward, result = base(x) + lora_B(lora_A(dropout(x))) * scaling; ecg_byte/main.py:131-155). */
int ecgb_gemm_nt_bf16_cat(const void *a_dev, long long lda, const void *b_dev, long long ldb, const void *a2_dev, long long lda2,
                          const void *b2_dev, long long ldb2, int K2, void *c_dev, long long ldc, int M, int N, int K, float alpha,
                          int accumulate_f32, void *stream);
/* A decode step's adapter site (o, down: peft LoraLayer.forward at one or two rows, M <= 2) in ONE launch instead of two: the launch's first workgroups form
 * t[M, K2] = t_scale * x . lora_A^T, the others y[M, N] = x . W^T + t . lora_B^T.  t travels through t64_dev ([M, K2] 64-bit words owned by the site, K2 <= 64): an element's
 * bf16 bits under the value of *epoch_dev in bits 16 .. 47, ONE store that other XCDs can see; a wave of the projection asks for the row's words with its first weight pieces
 * and uses them behind its weight row once every word carries the epoch.  *epoch_dev must differ from one call on the site to the next (ecgb_decode_advance_e adds one a
 * step; a fresh t64_dev must hold no current epoch: zeros, the counter starting at 1).  The bits of ecgb_gemm_nt_bf16(x, lora_A, alpha = t_scale) followed by
 * ecgb_gemm_nt_bf16_cat: every sum in the same order (tests/test_gpu_decode_fused.py).  ECGB_ERR_UNSUPPORTED where the two products would take different column-per-wave kernels
 * (N > 8192 with K >= 8192), M > 2 or K2 > 64: use the two calls. */
int ecgb_gemm_nt_bf16_lora_decode(const void *x_dev, long long ldx, const void *w_dev, long long ldw, const void *a_lora_dev, long long lda_lora, float t_scale,
                                  const void *b_lora_dev, long long ldb_lora, int K2, unsigned long long *t64_dev, void *y_dev, long long ldy, int M, int N, int K,
                                  const int *epoch_dev, void *stream);

/* LoRA adapter branch (peft LoraLayer, ecg_byte/main.py:131-155), stacked adapters of a fused projection.  The stacked down-projection
 * has n_sub 16-row sub-blocks (a_dev = [16 * n_sub, in]) split evenly over n_fields modules ("blocks": q | k | v share one x but each
 * draws its OWN dropout mask); element (row, col) of x is kept for block f iff 16-bit field f of a counter-based hash of
 * (seed, row * in + col) is >= p * 65536.  p = 0: no dropout.
 * t and dt are [T, 64] (columns past 16 * n_sub zero: one K-step of ecgb_gemm_nt_bf16_cat), A^T is [in, 64].
 *   ecgb_lora_down   t = scale / (1 - p) * (mask_f . x) A_f^T  (one pass over x);  xd_dev, if not NULL, receives the
 *                    n_fields masked copies of x, [n_fields, T, in] (tests compare them with the masks the backward kernels replay)
 *   ecgb_lora_dx     dx[T, in] += scale / (1 - p) * sum_f mask_f . (dt_f A_f), at_dev = A^T  (one read-modify-write of dx) */
int ecgb_lora_down(const void *x_dev, const void *a_dev, void *t_dev, void *xd_dev, int T, int in, int n_sub, int n_fields,
                   float scale, float p, uint64_t seed, void *stream);
int ecgb_lora_dx(const void *dt_dev, const void *at_dev, void *dx_dev, int T, int in, int n_sub, int n_fields,
                 float scale, float p, uint64_t seed, void *stream);
/* The adapters' own weight gradient from x itself: da_dev [16 * n_sub, in] (bf16: the first rows of the stacked A's gradient) (+)=
 * scale / (1 - p) * dt[:, :16 * n_sub]^T . (mask_f . x), the masks evaluated again from (seed, row * in + col) -- the forward keeps no masked
 * copy of x (peft's autograd saves dropout(x) per module; the model here saves x once per site).  One pass over x for all modules of the
 * site; chunks of rows meet in fp32 slabs (scratch_dev, ecgb_lora_da_scratch_bytes) summed in chunk order: no atomics, the same bits every time. */
size_t ecgb_lora_da_scratch_bytes(int T, int in, int n_sub);
int ecgb_lora_da(const void *x_dev, const void *dt_dev, void *da_dev, int T, int in, int n_sub, int n_fields, float scale, float p,
                 uint64_t seed, int accumulate, void *scratch_dev, size_t scratch_bytes, void *stream);
/* ecgb_lora_dx with the GLU backward behind it (the down-projection site, one adapter block): dx [T, inter] = d(act(gate) * up) of the
 * frozen base is READ, the adapters' contribution added, and d(gate|up) [T, 2*inter] = ecgb_glu_bwd(gate|up, that sum) written -- the same
 * bits as ecgb_lora_dx followed by ecgb_glu_bwd, one read-modify-write pass over [T, inter] less. */
int ecgb_lora_dx_glu(const void *dt_dev, const void *at_dev, const void *dx_dev, const void *gate_up_dev, void *d_gate_up_dev, int T,
                     int inter, int n_sub, int n_fields, float scale, float p, uint64_t seed, int gelu_tanh, void *stream);

/* GPT-2 block pieces (BASELINE config C1's model; transformers/src/transformers/models/gpt2/modeling_gpt2.py:565-661, pytorch_utils.py:87-113):
 *   ecgb_layernorm_fwd   y = (x - mean) * rstd * w + b (nn.LayerNorm); residual_dev != NULL: x := x + residual first, written to sum_out_dev;
 *                        mean_dev / rstd_dev (fp32 per row) are saved for the backward
 *   ecgb_layernorm_bwd   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ dres], g = dy * w;  dw_dev / db_dev (fp32) += sum over rows of dy * xhat / dy
 *   ecgb_bias_act        u += bias in place (Conv1D's bias);  gelu_new != 0: h = NewGELUActivation(u) as well (u keeps the pre-activation)
 *   ecgb_gelu_new_bwd    dpre = dh * gelu_new'(pre)
 *   ecgb_colsum          out (fp32) += sum over rows of dy: the gradient of a bias */
int ecgb_layernorm_fwd(const void *x_dev, const void *residual_dev, const void *w_dev, const void *b_dev, void *y_dev, void *sum_out_dev,
                       float *mean_dev, float *rstd_dev, size_t rows, int hidden, float eps, void *stream);
/* scratch_dev of ecgb_layernorm_bwd / ecgb_colsum (…_scratch_floats floats, or null): per-workgroup partial rows added in workgroup order -- the same
 * bits every launch; null: float atomics. */
size_t ecgb_layernorm_bwd_scratch_floats(size_t rows, int hidden);
size_t ecgb_colsum_scratch_floats(size_t rows, int n);
int ecgb_partial_rows_sum_f32(const float *partials_dev, int n_rows, int n, long long ld, float *dst_dev, void *stream);
int ecgb_layernorm_bwd(const void *x_dev, const void *w_dev, const float *mean_dev, const float *rstd_dev, const void *dy_dev,
                       const void *dres_dev, void *dx_dev, float *dw_dev, float *db_dev, size_t rows, int hidden, float *scratch_dev, void *stream);
int ecgb_bias_act(void *u_dev, const void *bias_dev, void *h_dev, size_t rows, int n, int gelu_new, void *stream);
int ecgb_gelu_new_bwd(const void *pre_dev, const void *dh_dev, void *dpre_dev, size_t n, void *stream);
int ecgb_colsum(const void *dy_dev, float *out_dev, size_t rows, int n, float *scratch_dev, void *stream);

/* C[M,N] = alpha * A[M,K] . B[K,N] with B ROW-major: the input gradient dX = dY . W against the weight as nn.Linear stores it
 * ([out, in], modeling_llama.py:227-258,273-395) -- what ecgb_gemm_nt_bf16 computes on a transposed copy of B, without the copy (the
 * same bits).  accumulate_f32 as in ecgb_gemm_nt_bf16.  K % 64 == 0, N % 8 == 0. */
int ecgb_gemm_nn_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc, int M, int N,
                      int K, float alpha, int accumulate_f32, void *stream);
/* ecgb_gemm_nn_bf16 with the GLU backward in its epilogue (full fine-tune, the MLP's down projection): d_gate_up [M, 2 * inter] =
 * ecgb_glu_bwd(gate_up, dY . W), W = [K, inter] row-major as nn.Linear stores the down projection; the product [M, inter] is never written.
 * The same bits as the two calls.  Whole 256x256 tiles only: M % 256 == 0 and inter % 256 == 0, else ECGB_ERR_UNSUPPORTED (callers take the two calls). */
int ecgb_gemm_nn_glu_bwd_bf16(const void *dy_dev, long long lddy, const void *w_dev, long long ldw, const void *gate_up_dev, long long ldgu,
                              void *d_gate_up_dev, long long ldd, int M, int inter, int K, int gelu_tanh, void *stream);
/* The same on the down-projection site of a LoRA fine-tune (peft LoraLayer on down_proj; ecg_byte/main.py:131-155): the adapter's share of the input gradient
 * joins the product before the GLU backward,
 *     d(gate|up) = ecgb_glu_bwd(gate_up, bf16(dY . W) + scale / (1 - p') * mask . (dt A)),   dt [M, 64] = dY . B_lora,   at_dev = A_lora^T [inter, 64]
 * (rank 16 in columns 0..15, the rest zero), the forward's dropout mask replayed from (seed, p) as ecgb_lora_down drew it -- ecgb_gemm_nn_bf16 followed by
 * ecgb_lora_dx_glu in one launch, the same bits, d(act(gate) * up) never written.  Four-wave kernel only: ECGB_ERR_UNSUPPORTED where it does not take the
 * shape (whole 256x256 tiles, its share of K-tiles per CU) or the backward runs non-persistent; callers then take the two calls. */
int ecgb_gemm_nn_glu_bwd_lora_bf16(const void *dy_dev, long long lddy, const void *w_dev, long long ldw, const void *gate_up_dev, long long ldgu,
                                   const void *dt_dev, const void *at_dev, void *d_gate_up_dev, long long ldd, int M, int inter, int K, int gelu_tanh,
                                   float scale, float p, uint64_t seed, void *stream);
/* The input gradient of a frozen projection that carries ONE LoRA module (o_proj): dx [M, in] = bf16(dY . W) + scale / (1 - p') * mask . (dt A) in one launch --
 * ecgb_gemm_nn_bf16 followed by ecgb_lora_dx (n_sub = n_fields = 1), the same bits.  Four-wave kernel only: ECGB_ERR_UNSUPPORTED otherwise (callers take the two calls). */
int ecgb_gemm_nn_lora_bf16(const void *dy_dev, long long lddy, const void *w_dev, long long ldw, const void *dt_dev, const void *at_dev,
                           void *dx_dev, long long lddx, int M, int in, int K, float scale, float p, uint64_t seed, void *stream);
/* ecgb_gemm_nn_bf16 with the contraction cut into n_splits K-slices (few output tiles, a long contraction: the loss head's input gradient dlogits . E over the
 * vocabulary): slice s writes its fp32 partial product to slab s of slabs_dev (n_splits x M x N floats, 16-byte aligned), the slabs are added in slice
 * order into c_dev (bf16, contiguous [M, N]).  No atomics: the same bits every launch. */
int ecgb_gemm_nn_splitk_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, float *slabs_dev, int M, int N, int K,
                             int n_splits, float alpha, void *stream);
/* The MLP's gate|up projection with the GLU in the GEMM's epilogue (modeling_llama.py:227-258
 * `down_proj(act_fn(gate_proj(x)) * up_proj(x))`; gelu_tanh != 0: Gemma's GeGLU):  gate|up [M, 2*inter] = alpha * (A . B^T
 * [+ A2 . B2^T]) with B = [2*inter, K] (gate rows, then up rows), H [M, inter] = act(gate) * up on the bf16-rounded gate and up
 * (bit for bit ecgb_glu_fwd of the stored projection).  c_dev may be null (inference: gate|up is never written).
 * a2_dev / b2_dev ([M, K2], [2*inter, K2]) as in ecgb_gemm_nt_bf16_cat, or null with K2 = 0.  inter % 128 == 0, K, K2 % 64 == 0. */
int ecgb_gemm_nt_glu_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, const void *a2_dev, long long lda2,
                          const void *b2_dev, long long ldb2, int K2, void *c_dev, long long ldc, void *h_dev, long long ldh,
                          int M, int inter, int K, float alpha, int gelu_tanh, void *stream);
/* Weight-gradient product without transposed copies: C[N,K] = alpha * A^T . B with A = [M,N] and B = [M,K]
 * row-major (dW = dY^T . X, the contraction index is the row index of both operands).  M % 64 == 0.
 * splits == 1: C is bf16.  splits > 1: the contraction is cut into `splits` slices run by separate workgroups
 * (small outputs would otherwise leave most CUs idle) and C is fp32 [splits][N, ldc]: slice s stores its partial product
 * in slab s (no zeroing needed, no atomics; ecgb_sum_slabs_bf16 adds the slabs in slice order). */
int ecgb_gemm_tn_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc,
                      int M, int N, int K, float alpha, int splits, void *stream);
/* Output tile choice: 0 = automatic (256x256 tiles / 8 waves on v_mfma_f32_16x16x32_bf16 when they fill the chip,
 * else 128x128 / 4 waves on 32x32x16); forced: 128, 256, 257 = 256x256 on 32x32x16, 258 = 256x256 on 16x16x32
 * without the four-phase schedule (tests, tuning). */
int ecgb_set_gemm_tile(int tile);
/* The input-gradient GEMMs (ecgb_gemm_nn_bf16, ecgb_gemm_nn_glu_bwd_bf16) with a persistent tile loop (default, 1) or one tile per workgroup (0).
 * Persistent workgroups hold a static share of the tiles on every CU; beside a collective that occupies CUs (the gradient exchange of a
 * data-parallel backward) the one-tile kernels degrade gracefully where a static share would not.  parallel.GradAllReduce sets 0 for world > 1. */
int ecgb_set_gemm_backward_persistent(int on);
int ecgb_get_gemm_backward_persistent(void);   /* the switch as it stands (1 / 0): a scope that changes it restores what it found */
/* Order in which an XCD's workgroups walk its range of output tiles: 0 or 1 = row by row (default); g > 1 = blocks of g tile rows, column by column (the ~32
 * tiles an XCD runs at a time are then g x 32 / g, both operand panels shared in its L2: measured 1-9 % slower at the C3 shapes, kept for A/B).  Same results. */
int ecgb_set_gemm_group_m(int group_m);
/* Same, with two-level batch addressing for attention heads: batch entry z = (zo, zi), zo = z / inner,
 * zi = z % inner; operand X starts at X + zo*outer_x + (zi / div_x)*inner_x  (div_b > 1 shares one KV head
 * among div_b query heads). */
int ecgb_gemm_nt_bf16_heads(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc,
                            int M, int N, int K, float alpha, int accumulate_f32, int batch, int inner, long long outer_a, long long inner_a,
                            int div_a, long long outer_b, long long inner_b, int div_b, long long outer_c,
                            long long inner_c, void *stream);

/* inv_count = 1 / max(#labels in [0, vocab), 1) */
int ecgb_count_labels(const int64_t *labels_dev, size_t n, int vocab, float *inv_count_dev, void *stream);
/* Cross-entropy of `rows` rows of logits [rows, ld] (first `vocab` columns valid) against labels (-100 =
 * ignored); sum_loss_dev += sum(row losses) * inv_count; logits are overwritten with d(loss)/d(logits). */
int ecgb_ce_fwd_bwd(void *logits_dev, const int64_t *labels_dev, float *row_loss_dev, float *sum_loss_dev,
                    const float *inv_count_dev, size_t rows, int vocab, size_t ld, void *stream);
/* 1 (default): rows with ld <= 163 840 are held in registers by 1 024-thread workgroups -- one read of the logits, one write of the gradient;
 * 0: the three-sweep kernel (maximum, sum, gradient), which wider vocabularies always take.  Same loss and gradients up to the order of the fp32 sums. */
int ecgb_set_ce_in_registers(int on);
/* ecgb_rmsnorm_fwd at hidden 2048: 1 (default) the row kept in registers between the sum of squares and the scaling, 0 the generic kernel (A/B, tests).  Same bits. */
int ecgb_set_rmsnorm_fwd_rows(int on);
/* (tuning) rows per workgroup of ecgb_rmsnorm_bwd at hidden 2048 / 4096 (default 64; measured at [32768, 2048]: 112 us, 32: 121, 16: 143 -- the partial rows of dw grow): changes what ecgb_rmsnorm_bwd_scratch_floats reports and the order of the dw sum. */
int ecgb_set_rmsnorm_bwd_rows_per_wg(int n);

/* acc_dev += sum(g^2) */
int ecgb_sumsq(const void *g_dev, size_t n, int is_fp32, float *acc_dev, void *stream);
/* The same over a list of bf16 tensors in ONE launch: ptrs_dev[t] / counts_dev[t] describe tensor t, and block b of the launch sums
 * the (at most 2^20) elements of tensor chunk_tensor_dev[b] that start at chunk_off_dev[b].  *acc_dev += the total. */
/* partials_dev (n_chunks floats, or null): per-chunk sums stored and added in chunk order -- the same gradient norm bits every step; null: atomics. */
int ecgb_sumsq_multi_bf16(const void *const *ptrs_dev, const unsigned long long *counts_dev, const int *chunk_tensor_dev,
                          const unsigned long long *chunk_off_dev, int n_chunks, float *acc_dev, float *partials_dev, void *stream);
/* One Adam step (moments fp32, params bf16) with the gradient first scaled by min(1, max_norm/(sqrt(*sumsq)+1e-6))
 * and weight decay applied as L2 (grad += wd * param), as torch.optim.Adam does. */
int ecgb_adam_step(void *param_dev, const void *grad_dev, int grad_is_fp32, float *m_dev, float *v_dev, size_t n,
                   const float *sumsq_dev, float max_norm, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, void *stream);
/* The same step over a LIST of bf16 parameter tensors in ONE launch (pointer tables of the parameters, their bf16 gradients and their fp32 moments; the chunk
 * table of ecgb_sumsq_multi_bf16): element by element ecgb_adam_step's arithmetic, the same bits; 16-byte accesses where the four pointers of a tensor allow. */
int ecgb_adam_multi_bf16(void *const *params_dev, const void *const *grads_dev, float *const *m_dev, float *const *v_dev, const unsigned long long *counts_dev,
                         const int *chunk_tensor_dev, const unsigned long long *chunk_off_dev, int n_chunks, const float *sumsq_dev, float max_norm,
                         float lr, float beta1, float beta2, float eps, float weight_decay, int step, void *stream);

/* Attention probabilities, round 1: scores are materialised ([batch*heads, S, S] bf16) by batched
 * ecgb_gemm_nt_bf16 calls and normalised here (a fused flash-style kernel is the next step).
 * In place: P[bh,i,j] = softmax_j(scale * scores[bh,i,j]) over the keys j <= i with attn_mask[b,j] != 0
 * (b = bh / n_heads; attn_mask float32 [batch, S]); fp32 softmax, bf16 result; rows with no visible key
 * become all zeros (pad rows: the reference's outputs there are don't-care, modeling_llama.py:1032-1042). */
int ecgb_softmax_causal_fwd(void *scores_dev, const float *attn_mask_dev, int batch_heads, int n_heads, int seq,
                            float scale, void *stream);
/* In place on dp_dev: dS = scale * P * (dP - rowsum(P * dP)). */
int ecgb_softmax_bwd(const void *p_dev, void *dp_dev, int batch_heads, int seq, float scale, void *stream);

/* Fused causal grouped-query attention (no S x S tensor in HBM): K/V tiles staged in LDS, softmax in
 * registers, MFMA for QK^T and PV.  q/k/v/o are [batch*seq] rows with the given row strides (elements);
 * query head hq starts at q + hq*head_dim, KV head g at k + g*head_dim (so slices of a fused qkv buffer work).
 * Key j is visible to query i iff j <= i and attn_mask[b, j] != 0; rows without a visible key give zeros.
 * lse_dev [batch, n_q_heads, seq] fp32 is saved for the backward.  head_dim 64 (Llama, GPT-2), 128 or 256 (Gemma). */
int ecgb_attn_fwd(const void *q_dev, long long ldq, const void *k_dev, long long ldk, const void *v_dev, long long ldv,
                  const float *attn_mask_dev, void *o_dev, long long ldo, float *lse_dev, int batch, int seq,
                  int n_q_heads, int n_kv_heads, int head_dim, float scale, void *stream);
/* dq/dk/dv from dO (same layout as o).  delta_dev [batch, n_q_heads, seq] fp32 is scratch (rowsum(dO*O)).
 * dK/dV sum over the query heads of each KV group inside the kernel (no atomics). */
/* head_dim 64 forward and backward: 2 = the lean kernels (default): tiles by LDS-DMA two tiles ahead into a ring of three buffers, the register operand
 * pre-scaled by scale * log2 e, the softmax row constants (running maximum / log-sum-exp / delta) as the initial accumulators of the score products,
 * forward running maximum deferred; 1 = the round-2 LDS-DMA kernels; 0 = the register-staged kernels (1 and 0 are bit-identical to each other and kept
 * for A/B and as cross-checks).  With 2, | 0x100 / 0x200 / 0x400 keeps the forward / dQ / dK-dV kernel alone on mode 1.  ECGB_ERR_INVALID otherwise. */
int ecgb_set_attn_fwd_staging(int mode);
/* waves per workgroup of the lean kernels: 4 (default) or 8 (a K / V tile serves gcd(G, 8) query heads x 256 / gcd rows; A/B). */
int ecgb_set_attn_lean_waves(int waves);
/* scratch of ecgb_attn_bwd: fp32 partial dK / dV slabs when the query heads of a KV group are split over workgroups (head_dim 256 with
 * few key blocks: Gemma); 0 for every other shape (scratch_dev may then be null).  With ecgb_set_attn_d256_pass_p(1) also, behind the slabs, room for the probabilities
 * (bf16, batch * n_q_heads * seq^2 rounded up to whole 128 x 64 tiles; transient) that the dK pass then hands to a dV kernel instead of both passes forming the scores;
 * a scratch that holds the slabs but not the probabilities is accepted (the pair kernel runs). */
size_t ecgb_attn_bwd_scratch_bytes(int batch, int seq, int n_q_heads, int n_kv_heads, int head_dim);
/* head_dim 256 backward: 0 (default) = the pair kernel (the dV pass forms the scores a second time), 1 = the probabilities travel from the dK pass to a dV kernel through the
 * scratch where it has room (the same bits; at Gemma-2B dims, batch 8, seq 2048: 0.66 against 0.68 ms a layer for 537 MB of transient scratch). */
int ecgb_set_attn_d256_pass_p(int on);
int ecgb_attn_bwd(const void *q_dev, long long ldq, const void *k_dev, long long ldk, const void *v_dev, long long ldv,
                  const float *attn_mask_dev, const void *o_dev, const void *do_dev, long long ldo, const float *lse_dev,
                  float *delta_dev, void *dq_dev, long long lddq, void *dk_dev, long long lddk, void *dv_dev,
                  long long lddv, int batch, int seq, int n_q_heads, int n_kv_heads, int head_dim, float scale, void *scratch_dev, size_t scratch_bytes, void *stream);
/* ecgb_attn_bwd followed by RoPE's backward on dQ and dK (ecgb_rope with inverse != 0: modeling_llama.py:151-176 apply_rotary_pos_emb, transposed), inside the
 * attention kernels' own stores: q and k are the ROTATED projections the forward saw, dq / dk come out as gradients of the unrotated ones -- the same bits as the two
 * calls, one read and one write of the q|k gradient less.  rope_cos_dev / rope_sin_dev: [batch * seq, 32] fp32, the tables ecgb_rope takes.  head_dim 64 on the lean
 * kernels with 16-byte aligned dq / dk rows only: ECGB_ERR_UNSUPPORTED otherwise (the caller then runs the two steps apart). */
int ecgb_attn_bwd_rope(const void *q_dev, long long ldq, const void *k_dev, long long ldk, const void *v_dev, long long ldv,
                       const float *attn_mask_dev, const void *o_dev, const void *do_dev, long long ldo,
                       const float *lse_dev, float *delta_dev, void *dq_dev, long long lddq, void *dk_dev, long long lddk,
                       void *dv_dev, long long lddv, const float *rope_cos_dev, const float *rope_sin_dev, int batch, int seq, int n_q_heads,
                       int n_kv_heads, int head_dim, float scale, void *scratch_dev, size_t scratch_bytes, void *stream);

/* One decode step of generate(): q [batch, n_q_heads*head_dim] against a KV cache whose rows (one per position) are
 * `ld` elements apart, `capacity` rows per batch entry, KV head g at + g*head_dim; the first kv_len rows are valid
 * (the new token included); attn_mask [batch, mask_ld] marks padded positions with 0.
 * Reference: LlamaSdpaAttention.forward with past_key_value (modeling_llama.py:563-575), DynamicCache.update
 * (cache_utils.py:408).  o [batch, n_q_heads*head_dim]. */
int ecgb_attn_decode(const void *q_dev, const void *k_cache_dev, const void *v_cache_dev, long long ld, long long capacity,
                     const float *attn_mask_dev, long long mask_ld, void *o_dev, int batch, int kv_len, int n_q_heads,
                     int n_kv_heads, int head_dim, float scale, void *stream);

/* The same step with the keys of every head split over `n_splits` workgroups (long caches, few heads: one CU cannot pull a
 * long cache fast enough, the chip has 256): scores + per-split statistics, then partial P.V with the global statistics,
 * then the sum of the partial outputs -- three launches; same arithmetic per key as ecgb_attn_decode.  `scratch_dev` holds
 * ecgb_attn_decode_split_scratch_bytes(...) bytes (fp32 scores, statistics and partial outputs). */
size_t ecgb_attn_decode_split_scratch_bytes(long long capacity, int batch, int n_q_heads, int head_dim, int n_splits);
int ecgb_attn_decode_split(const void *q_dev, const void *k_cache_dev, const void *v_cache_dev, long long ld, long long capacity,
                           const float *attn_mask_dev, long long mask_ld, void *o_dev, int batch, int kv_len, int n_q_heads,
                           int n_kv_heads, int head_dim, float scale, int n_splits, void *scratch_dev, size_t scratch_bytes,
                           void *stream);

/* The same with the number of valid cache rows read from device memory, and the append of the new token's K | V row at
 * index *kv_len_dev - 1: nothing in the launch depends on the step, so one decode step captured in a HIP graph can be
 * replayed for every token (the host only bumps the device counter). */
int ecgb_attn_decode_dyn(const void *q_dev, const void *k_cache_dev, const void *v_cache_dev, long long ld, long long capacity,
                         const float *attn_mask_dev, long long mask_ld, void *o_dev, int batch, const int *kv_len_dev,
                         int n_q_heads, int n_kv_heads, int head_dim, float scale, void *stream);
int ecgb_kv_append(const void *src_dev, long long src_ld, long long col_off, int width, void *cache_dev, long long capacity,
                   int batch, const int *kv_len_dev, void *stream);
/* ecgb_attn_decode_split with the number of valid cache rows in device memory: the same three launches, each split's key range computed on the
 * device as the host computes it (chunk = ceil(len / n_splits)) -- the same bits as ecgb_attn_decode_split with the same n_splits. */
int ecgb_attn_decode_split_dyn(const void *q_dev, const void *k_cache_dev, const void *v_cache_dev, long long ld, long long capacity,
                               const float *attn_mask_dev, long long mask_ld, void *o_dev, int batch, const int *kv_len_dev, int n_q_heads,
                               int n_kv_heads, int head_dim, float scale, int n_splits, void *scratch_dev, size_t scratch_bytes, void *stream);

/* Tuning / A-B: the most workgroups ecgb_glu_fwd / ecgb_glu_bwd launch (default 2^20: one trip of a workgroup; rounds 2-5: 4 096 walking the tensor in a grid-stride loop). */
int ecgb_set_stream_grid_cap(int n);
/* Tuning: the most workgroups of ecgb_rmsnorm_bwd at hidden 2048 / 4096 (with ecgb_set_rmsnorm_bwd_rows_per_wg; changes ecgb_rmsnorm_bwd_scratch_floats and the order of the dw sum). */
int ecgb_set_rmsnorm_bwd_grid_cap(int n);

/* Round 6: a decode step's RoPE + KV-cache append + attention in ONE launch, from the step's raw q|k|v projection [batch, (n_q + 2 n_kv) head_dim] (q and k are NOT
 * rotated in place: the cache row kv_len - 1 and o are what leaves) -- replaces ecgb_rope_append + ecgb_attn_decode_split[_dyn] (cache_utils.py:408-470 DynamicCache.update,
 * modeling_llama.py:526-614 at one query row), the same bits for the same n_splits.  The workgroups of a (sequence, head) meet through counters inside the launch, so all
 * n_splits * n_q_heads * batch of them must be resident: ECGB_ERR_UNSUPPORTED above two workgroups a CU (and outside head_dim 64 / 128 / 256, n_splits <= 64, 2048 keys a split); the
 * caller then runs the separate launches.  scratch: ecgb_attn_decode_one_scratch_floats() floats whose FIRST batch * n_q_heads * n_splits * 2 (the splits' statistics) hold
 * the bit pattern 0x7FC0DEAD ("not stored yet") and whose LAST batch * n_q_heads * 2 words (the counters) are zero before the first call; every launch leaves them so.  kv_len_dev: the number of valid cache rows AFTER the append in device memory (a replayed graph), else kv_len. */
size_t ecgb_attn_decode_one_scratch_floats(int batch, int n_q_heads, int head_dim, int n_splits);
int ecgb_attn_decode_one(const void *qkv_dev, long long ld_qkv, const float *cos_dev, const float *sin_dev, void *cache_dev, long long ld, long long capacity,
                         const float *attn_mask_dev, long long mask_ld, void *o_dev, int batch, int kv_len, const int *kv_len_dev, int n_q_heads, int n_kv_heads,
                         int head_dim, float scale, int n_splits, float *scratch_dev, size_t scratch_floats, void *stream);

/* C = alpha * A . B^T like ecgb_gemm_nt_bf16 (plain bf16 store, one problem), on four waves per workgroup with 128x128 wave tiles (csrc/gemm_w4.hip).
 * Whole 256x256 tiles only: M and N multiples of 256, K of 64, 16-byte aligned operands; ECGB_ERR_UNSUPPORTED otherwise.  ecgb_gemm_nt_bf16 dispatches here
 * when the problem also has at least two tiles per CU (batch 1, plain store); the same bits as the eight-wave kernels. */
int ecgb_gemm_nt_w4_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc,
                         int M, int N, int K, float alpha, void *stream);
/* The q|k|v projection with RoPE's forward in its epilogue (modeling_llama.py:151-176 apply_rotary_pos_emb on the projection's output): C = alpha * (A . B^T
 * [+ A2 . B2^T, K2 > 0: the LoRA pair of ecgb_gemm_nt_bf16_cat]), then every head of 64 columns below rope_cols rotated with row t of the [M, 32] fp32 tables --
 * ecgb_rope's arithmetic on the bf16-rounded projection: the same bits as the two calls, one write and one read of q|k less.  Whole 256x256 tiles, at least one per
 * CU, 16-byte aligned operands; ECGB_ERR_UNSUPPORTED otherwise (the caller runs ecgb_gemm_nt_bf16[_cat] and ecgb_rope). */
int ecgb_gemm_nt_bf16_rope(const void *a_dev, long long lda, const void *b_dev, long long ldb, const void *a2_dev, long long lda2,
                           const void *b2_dev, long long ldb2, int K2, void *c_dev, long long ldc, int M, int N, int K, float alpha,
                           const float *rope_cos_dev, const float *rope_sin_dev, int rope_cols, void *stream);
/* ecgb_gemm_nn_bf16's product (B [K, N] row-major) on the four-wave kernel, by name: whole 256x256 tiles, plain bf16 store; ECGB_ERR_UNSUPPORTED otherwise.  The same bits
 * as the eight-wave NN kernels; ecgb_gemm_nn_bf16 dispatches here from 256 K-tiles per CU on while persistent backward kernels are allowed. */
int ecgb_gemm_nn_w4_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc,
                         int M, int N, int K, float alpha, void *stream);
/* ecgb_gemm_tn_bf16's product with one K-slice (C [M, N] = A^T . B, A [K, M] and B [K, N] row-major: the weight gradient dY^T . X) on the four-wave kernel, by name;
 * whole 256x256 tiles, plain bf16 store, the same bits; ecgb_gemm_tn_bf16 dispatches here under the same conditions as ecgb_gemm_nn_bf16. */
int ecgb_gemm_tn_w4_bf16(const void *a_dev, long long lda, const void *b_dev, long long ldb, void *c_dev, long long ldc,
                         int M, int N, int K, float alpha, void *stream);
int ecgb_set_gemm_w4(int on);              /* 1 (default): ecgb_gemm_nt_bf16 / _cat / _glu send eligible problems to the four-wave kernel; 0: never; 2: every form it has (A/B, tests) */
int ecgb_set_gemm_w4_group_m(int group_m);  /* its tile order: blocks of group_m tile rows (default 8; 0 = row by row) */
int ecgb_set_gemm_w4_min_ktiles(int n);     /* K-tiles per workgroup from which the dispatch picks the four-wave kernel (default 128) */
int ecgb_gemm_w4_span_ok(int lay, long long lda, long long ldb, long long K);   /* 1 if the four-wave kernel's 32-bit DMA offsets cover these strides over a contraction of K (lay 0 NT, 1 NN, 2 TN); else the dispatch uses the eight-wave kernels.  Pure arithmetic, no device. */
int ecgb_set_gemm_w4_sched(int sched);      /* its K-tile schedule: 1 (default) four barriers per K-tile behind counted waits, 0 one rendezvous per K-tile (round 3; A/B).  Same bits. */

/* ecgb_rmsnorm_fwd for the few rows of a decode step with adapters, followed in the same launch by the site's stacked LoRA down-projection of the normalised row:
 * t_dev [rows, ldt] (bf16) = lora_scale * y . A^T for the n_a rows of lora_a_dev [n_a, lda] -- bit for bit ecgb_gemm_nt_bf16(y, A, alpha = lora_scale) on the few-row
 * kernel (the same pieces in the same order).  hidden % 512 == 0; one workgroup per row. */
int ecgb_rmsnorm_lora_fwd(const void *x_dev, const void *residual_dev, const void *w_dev, void *y_dev, void *sum_out_dev, float *rstd_dev, size_t rows,
                          int hidden, float eps, int gemma, const void *lora_a_dev, long long lda, int n_a, float lora_scale, void *t_dev, long long ldt,
                          void *stream);

/* A decode step's ecgb_rope (forward, on the new token's q and k heads) and its KV-cache append in one launch: qkv_dev [batch, (n_q + 2 n_kv) * head_dim] is rotated in
 * place, the rotated k and the v of every sequence go to row kv_len - 1 of cache_dev [batch, capacity, 2 * n_kv * head_dim] (k | v).  kv_len_dev (int32[1] in device
 * memory) replaces kv_len when given (a replayed graph).  The same bits as ecgb_rope followed by the copy / ecgb_kv_append. */
int ecgb_rope_append(void *qkv_dev, const float *cos_dev, const float *sin_dev, int batch, int n_q_heads, int n_kv_heads, int head_dim, size_t row_stride,
                     void *cache_dev, long long capacity, int kv_len, const int *kv_len_dev, void *stream);

/* Greedy token choice of generate() (GenerationMixin._sample, generation/utils.py:3205: `next_tokens = torch.argmax(next_token_scores, dim=-1)`):
 * out[r] = index of the first maximum of the bf16 row x[r, 0:n] (rows `ld` elements apart).  One launch, no workspace. */
int ecgb_argmax_bf16(const void *x_dev, long long ld, int rows, int n, int64_t *out_dev, void *stream);

/* The RoPE table rows of a decode step: cos_out / sin_out [n, half] (fp32) = cos / sin of float(pos[i]) * inv_freq[j], the fp32 product rounded once -- the bits of
 * `fr = pos.float()[:, None] * inv_freq[None, :]; fr.cos(), fr.sin()` (LlamaRotaryEmbedding.forward, modeling_llama.py:119-139, in fp32), one launch instead of four. */
int ecgb_rope_table(const int64_t *pos_dev, int n, const float *inv_freq_dev, int half, float *cos_out_dev, float *sin_out_dev, void *stream);

/* What generate()'s loop does with the chosen tokens, for a decode step replayed from a graph (GenerationMixin._sample, generation/utils.py:3208-3232: finished sequences
 * take pad_token_id, `input_ids = torch.cat([input_ids, next_tokens[:, None]], dim=-1)`, the attention mask grows by a column of ones, `unfinished_sequences` is and-ed with
 * "not an eos id"), in ONE launch instead of six to eleven element-wise ones (5 us each in the graph).  For every sequence b of `batch`:
 *     t = unfinished[b] ? next[b] : pad_id;   unfinished[b] &= t is none of eos[0..n_eos);   out[b][col[b]] = t;   mask[b][col[b]] = 1;   tok[b] = t;   pos[b] += 1;   col[b] += 1
 * and *n_dev += 1.  All int64 except mask (float32) and n_dev (int32[1]); eos_dev may be NULL (n_eos 0): `unfinished` is then neither read nor written. */
int ecgb_decode_advance(const int64_t *next_dev, int batch, int64_t *tok_dev, int64_t *pos_dev, int64_t *col_dev, int *n_dev, int64_t *out_dev, long long out_ld,
                        float *mask_dev, long long mask_ld, int64_t *unfinished_dev, long long pad_id, const int64_t *eos_dev, int n_eos, void *stream);
/* The same, and *epoch_dev += 1 (int32[1]): the step counter the one-launch adapter sites of a replayed decode step synchronise on (ecgb_gemm_nt_bf16_lora_decode). */
int ecgb_decode_advance_e(const int64_t *next_dev, int batch, int64_t *tok_dev, int64_t *pos_dev, int64_t *col_dev, int *n_dev, int64_t *out_dev, long long out_ld,
                          float *mask_dev, long long mask_ld, int64_t *unfinished_dev, long long pad_id, const int64_t *eos_dev, int n_eos, int *epoch_dev, void *stream);

/* ---- the decode step of generate() for one or two sequences, fused (csrc/decode.hip; round 5) -----------------------------------------------------------------
 * Replaces, per layer and token: ecgb_rmsnorm_lora_fwd + ecgb_gemm_nt_bf16_cat (q|k|v), ecgb_rope_append + ecgb_attn_decode_split (three kernels), the few-row GEMMs of
 * the adapters' down-projections, ecgb_rmsnorm_fwd + the GLU GEMV.  Every sum is formed in the order of the kernels replaced: the same bits (tests/test_gpu_decode_fused.py).
 * Reference: modeling_llama.py:635-701 (decoder layer), cache_utils.py:408-470 (DynamicCache.update), generation/utils.py:3131 (the per-token call). */

/* residual add (delta may be NULL) + RMSNorm (gemma != 0: (1 + w) form) + optional LoRA branch (A [64, H], n_a rows used; B [N or 2 N, 64]) + projection of M <= 2 rows.
 * glu 0: y[M, N] = h W^T (+ t B^T); glu 1 / 2: W = [gate rows; up rows] (N each), y = act(gate) * up (SiLU / tanh-GELU).  x_out = x + delta when delta is given. */
int ecgb_decode_norm_gemv(const void *x_dev, const void *delta_dev, const void *norm_w_dev, float eps, int gemma, int M, int H, void *x_out_dev,
                          const void *w_dev, long long ldw, int N, const void *lora_a_dev, long long lda, int n_a, float lora_scale, const void *lora_b_dev,
                          long long ldb, void *y_dev, long long ldy, int glu, void *stream);
/* y[M, N] = a W^T (+ t B^T), M <= 2: t = bf16(scale a A^T) formed inside (lora_a_dev, K <= 4096) or given (t_dev [M, 64]); both NULL: no adapter. */
int ecgb_decode_gemv(const void *a_dev, long long lda_act, int M, int K, const void *w_dev, long long ldw, int N, const void *lora_a_dev, long long lda, int n_a,
                     float lora_scale, const void *t_dev, const void *lora_b_dev, long long ldb, void *y_dev, long long ldy, void *stream);
/* t[M, 64] = bf16(scale a A^T) for the n_a used rows of A [64, K] (the rest zero); K a multiple of 2048 (the down-projection site). */
int ecgb_decode_lora_t(const void *a_dev, long long lda_act, int M, int K, const void *lora_a_dev, long long lda, int n_a, float lora_scale, void *t_dev, void *stream);
/* scratch floats of ecgb_decode_attn; the last batch * n_kv_heads words (tickets) must be zero before the first call and are left zero by every call */
size_t ecgb_decode_attn_scratch_floats(long long capacity, int batch, int n_q_heads, int n_kv_heads, int head_dim, int n_splits);
/* One decode step's attention from the raw q|k|v projection [batch, (Hq + 2 Hkv) D]: RoPE on q and on the new key (ecgb_rope's arithmetic), key / value append at cache row
 * len - 1 (cache [batch, capacity, 2 Hkv D]: keys | values), softmax(q K^T scale, keys with mask == 0 excluded) V -> o [batch, Hq D].  len = *kv_len_dev when given. */
int ecgb_decode_attn(void *qkv_dev, long long ld_qkv, const float *cos_dev, const float *sin_dev, void *cache_dev, long long capacity, const float *attn_mask_dev,
                     long long mask_ld, void *o_dev, int batch, int kv_len, const int *kv_len_dev, int n_q_heads, int n_kv_heads, int head_dim, float scale, int n_splits,
                     float *scratch_dev, size_t scratch_floats, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ECGBYTE_DECODER_H */
