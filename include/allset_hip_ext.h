/*
 * allset_hip_ext.h -- the EXTENSION surface of liballset_hip.so: everything the library exports beyond the core
 * aggregation ABI of allset_hip.h.  These entry points are the plumbing of the allset_amd Python package (its dense
 * tail, bf16 regime, exchange-layout helpers, loss / optimizer kernels); they follow the same conventions (extern "C",
 * int status, borrowed device pointers, caller's stream, no hidden sync, no environment reads) and are versioned
 * separately: ALLSET_ABI_VERSION below moves with them, ALLSET_CORE_ABI_VERSION (allset_hip.h) does not.
 *
 *   reference code replaced                                                        entry points here
 *   -----------------------------------------------------------------------------  --------------------------------------
 *   src/layers.py:571-579 MLP.forward (norm -> [Linear -> ReLU -> norm -> dropout]* -> Linear) and the relu -> dropout
 *   SetGNN puts behind every conv (src/models.py:475-481)                          allset_fused_linear_* (widths 64 / 128),
 *                                                                                  allset_gemm_x6* / allset_row_stats (256 / 512),
 *                                                                                  allset_ln_*, allset_relu_dropout_*, allset_wgrad*
 *   src/layers.py:153-157 PMA tail: + att_r, ln0, ln1(z + relu(rFF(z)))            allset_ln_res_*, allset_pma_fold_*
 *   src/layers.py:499-517 Normalization='bn' in training mode                       allset_col_moments*, allset_col_affine_add, *_nm
 *   src/train.py:169-199, 478-486 loss / metrics / Adam                            allset_nll_*, allset_split_metrics, allset_adam_*
 */
#ifndef ALLSET_HIP_EXT_H
#define ALLSET_HIP_EXT_H

#include "allset_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ALLSET_ABI_VERSION 15  /* 15: additions only (allset_gemm_wide_sgn(_supported): the relu backward of a wide Linear's input as the backward-data GEMM's epilogue).  14: additions only (allset_reduce_partials_batched_ex2: an output type per buffer; allset_reduce_partials_batchable / _batched* now also take the reductions the single entry runs as TWO launches, at most 512 partial rows, with the same association -- a [1M, 128] or [250k, 256] step reduces every parameter gradient of its backward pass in one launch; allset_sparse_linear_supported / _pitch / _wt / _fwd / _bwd: PMA's value projection + folded logits on sparse raw features).  13: additions only (bf16 regime: allset_linear_bf16_mask_pitch / _fwd_mask / _bwd_bits -- the relu mask as one bit per element --, allset_wgrad_bf16_ex2(_supported) -- that mask and PMA's four auxiliary logit rows inside the weight-gradient pass --, allset_pma_fold_fwd_bf16 / _bwd_bf16).  Earlier:  12: additions only (allset_fused_linear_bwd_pma_tail(_supported), allset_fused_linear_bwd_ln_pro(_supported)); the auxiliary-column forward at 128 x 128 also runs under ALLSET_ARITH_FP16X3 now;   2: dense-tail entries gained seed_base / mask / acc_in / aux parameters, new ln_res_* and pma_merge_pack; 3: additions only (bf16 ln / ln_res / wgrad, pma_*_ld, block_transpose); 4: additions only (fused_linear_bwd_all); 5: additions only (ln_res_bwd_pma, linear_bf16_*, wgrad_bf16_ex, reduce_partials_ex, nll_logsoftmax_*, split_metrics, adam_step*, pma_fold_*, wgrad_fused_ex); 6: addition only (fused_linear_bwd_all_slices_for); 7: additions only (fused_linear_bwd_all_aux, _aux_supported); 8: additions only (fused_linear_blocked_supported, fused_linear_fwd_blocked, fused_linear_bwd_all_blocked); 9: BREAKING -- the library reads no environment variable any more: which kernel an entry point launches, and the partial-slice count a caller sizes its buffers with, are pure functions of the call's arguments (the ALLSET_DENSE_MFMA=f32 comparison family and the ALLSET_BWD_ROLES / _BWD_STAGE / _BWD_PAIR / _BWD_ROLES3 / _FWD_ROLES / _LNRES_CAP / _WGRAD_BF16_TILED switches are gone with the kernels that lost their A/B); added in the same version: allset_fused_linear_fwd_nm / allset_fused_linear_bwd_all_nm (norm_mode: LayerNorm or per-column affine prologue), allset_col_moments(_slices, _supported), allset_col_moments2, allset_col_affine_add -- training-mode BatchNorm1d.  Note for ABI 4-5 callers (true since ABI 6, recorded here): allset_fused_linear_bwd_all at O = I = 128 takes the slice count of allset_fused_linear_bwd_all_slices_for, NOT that of the width-less allset_fused_linear_bwd_all_slices -- a behaviour break of ABI 6, which was wrongly listed as "addition only"; 10: additions only (dataset-scale step: allset_input_linear_*, allset_xhat_rows, allset_fold_ln_linear, allset_unfold_ln_linear, allset_reduce_partials_batch_max / _batch_max_counters / _batchable / _batched / _batched_ex, allset_linear_narrow_supported / _slices / _bwd, allset_nll_logsoftmax_fwd_total, allset_sparse_ln_linear_* / allset_fold_ln_linear_t / allset_unfold_ln_linear_ex); 11: additions only -- the header is split (the 15 aggregation entry points of SURVEY 8(b2) are allset_hip.h with their own frozen ALLSET_CORE_ABI_VERSION and allset_core_version(); this file is everything else); allset_fused_linear_fwd_ex / allset_fused_linear_bwd_all_ex / allset_fused_linear_arith_supported: the arithmetic of the fused Linear kernels (exact-split bf16x6 or fp16x3) becomes the caller's choice.  BEHAVIOUR CHANGE of ABI 11, recorded here because "additions only" undersells it: the unchanged legacy entries (allset_fused_linear_fwd / _blocked / _nm, allset_fused_linear_bwd_all / _blocked / _nm) now run ALLSET_ARITH_AUTO -- at K = N = 128 the row- / launch-scaled fp16x3 planes instead of the exact bf16x6 split (error per product <= 2^-21 relative + 2^-38 x the row's largest |gy| x |u|, relative to the ROW maximum; bf16x6 is exact to 2^-23 for any dynamic range), and the tiled 256 / 512-wide GEMMs and the weight gradient take fp16 planes under AUTO as well; a caller that needs the old numerics calls the _ex entries with ALLSET_ARITH_BF16X6 (python: dense.set_arithmetic("strict")) */

/* ---------------------------------------------------------------------------------------------
 * Dense tail (reference MLP.forward, layers.py:571-579: norm -> [Linear -> ReLU -> norm -> dropout]* -> Linear,
 * and the relu/dropout that HalfNLHconv.forward / SetGNN.forward wrap around it, layers.py:631-634,
 * models.py:473-481).  fp32, row-major.  The two well-shaped GEMMs (y = x W^T, gx = gy W) stay on hipBLASLt;
 * these entry points cover what the torch ops do badly at [1M,128]: LayerNorm fwd/bwd (fused with the
 * neighbouring ReLU / dropout) and the weight-gradient GEMM.
 * Dropout: keep iff hash(seed, element index) >= p, kept values scaled by 1/(1-p) (a counter hash: one 32-bit hash per four
 * consecutive elements at 8 bits each when p * 256 is an integer -- p = 0.5, 0.25, ... exact -- otherwise per pair at 16 bits
 * each; every entry point derives the resolution from p alone, so all sites agree); the backward regenerates the mask
 * from the same seed.  Every dropout-bearing entry point also takes `seed_base` (may be NULL): a DEVICE pointer to a
 * 64-bit counter; when given, the effective seed is counter * 0x9E3779B97F4A7C15 + seed, read when the kernel starts, so
 * a captured hipGraph draws fresh masks on every replay (the caller bumps the counter inside the graph).
 * ------------------------------------------------------------------------------------------- */

/* y = dropout_p( LayerNorm_{gamma,beta,eps}( relu_in ? relu(x) : x ) );  stats[row] = {mean, rstd} (f32[n*2]). */
int allset_ln_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int relu_in,
                  float p, uint64_t seed, float* y, int64_t ldy, float* stats, int64_t n, int64_t d,
                  const uint64_t* seed_base, void* stream);

/* Backward of allset_ln_fwd.  gx = d loss / d x (may be NULL when only the parameter partials are wanted);  partials: f32[n_partials*2*d], row k holds block k's
 * (dgamma[d], dbeta[d]) partial sums -- the caller sums over k.  n_partials from allset_ln_bwd_partials. */
int allset_ln_bwd_partials(int64_t n, int64_t d, int64_t* n_partials);
int allset_ln_bwd(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* stats, const float* gamma,
                  int relu_in, float p, uint64_t seed, float* gx, int64_t ldgx, float* partials, int64_t n_partials,
                  int64_t n, int64_t d, const uint64_t* seed_base, void* stream);

/* y = dropout_p(relu(x)) over numel contiguous elements, and its backward (gx = gy/(1-p) where y > 0, else 0). */
int allset_relu_dropout_fwd(const float* x, float p, uint64_t seed, float* y, int64_t numel,
                            const uint64_t* seed_base, void* stream);
int allset_relu_dropout_bwd(const float* gy, const float* y, float p, float* gx, int64_t numel, void* stream);

/* Weight gradient of y = u W^T + b:  gW[o][i] = sum_r ga[r][o] * u[r][i],  gb[o] = sum_r ga[r][o], as
 * n_slices split-K partials (part_w: f32[n_slices*O*I], part_b: f32[n_slices*O] or NULL) that the caller
 * sums -- deterministic, no atomics.  fp32-accurate arithmetic (bf16x6 on the bf16 matrix pipe, see
 * allset_fused_linear_fwd).  O, I, lda, ldu must be multiples of 4 and
 * the inputs 16-byte aligned, else ALLSET_ERR_UNSUPPORTED. */
int allset_wgrad_slices(int64_t n, int64_t O, int64_t I, int64_t* n_slices);
int allset_wgrad(const float* ga, int64_t lda, const float* u, int64_t ldu, float* part_w, float* part_b,
                 int64_t n_slices, int64_t n, int64_t O, int64_t I, void* stream);

/* The same weight gradient for bf16 activations (ga, u: bf16 row-major; fp32 partials as above; bf16 x bf16 products are
 * exact in the fp32 accumulator).  n_slices from allset_wgrad_bf16_slices (widths in {64,128,256} with 16-byte aligned rows take
 * a full-width kernel -- one workgroup owns all of gW for its rows, each operand is read once, the transposes are LDS
 * transpose-reads -- with its own slice count; other shapes fall back to the 128 x 128-tiled kernel and allset_wgrad_slices). */
int allset_wgrad_bf16_slices(int64_t n, int64_t O, int64_t I, int64_t* n_slices);
int allset_wgrad_bf16(const void* ga, int64_t lda, const void* u, int64_t ldu, float* part_w, float* part_b,
                      int64_t n_slices, int64_t n, int64_t O, int64_t I, void* stream);
/* The same with ONE partial buffer: part: f32[n_slices][part_stride], row k = slice k's gW (O*I floats) followed, when
 * want_bias, by its gb (O floats) -- so a single allset_reduce_partials_ex call sums every parameter gradient of the Linear
 * and can write them as bf16.  part_stride >= O*I (+O), a multiple of 4. */
int allset_wgrad_bf16_ex(const void* ga, int64_t lda, const void* u, int64_t ldu, float* part, int64_t part_stride,
                         int want_bias, int64_t n_slices, int64_t n, int64_t O, int64_t I, void* stream);
/* ABI 13.  The same with (a) `bits` (may be NULL): the relu BIT mask allset_linear_bf16_fwd_mask wrote for the Linear whose output
 * gradient `ga` is -- applied to ga as it is staged, so the masked gradient never exists in memory -- and (b) `g4` (fp32 [n, 4], may be
 * NULL): the gradient of four auxiliary output columns (PMA's folded logits, reference layers.py:126-131), rounded to bf16 and
 * multiplied in the same pass.  Partial row: [gW (O*I) | gb (O, if want_bias) | gWa (4*I) | gba (4)], the last two only with g4.
 * O, I in {128, 256} (g4: 256 x 256), see allset_wgrad_bf16_ex2_supported; n_slices from allset_wgrad_bf16_slices. */
int allset_wgrad_bf16_ex2_supported(int64_t O, int64_t I, int has_bits, int has_aux);
int allset_wgrad_bf16_ex2(const void* ga, int64_t lda, const void* bits, const float* g4, const void* u, int64_t ldu, float* part,
                          int64_t part_stride, int want_bias, int64_t n_slices, int64_t n, int64_t O, int64_t I, void* stream);

/* The Linear of the bf16 regime (BASELINE configs[4]; replaces MLP.forward's Linear + ReLU, reference layers.py:571-579, and
 * PMA's value projection + folded logits, layers.py:120-145, with their autograd) -- bf16 activations and parameters, fp32
 * accumulation, ONE pass per direction with the element-wise neighbours folded in.  in/out features in {128, 256}
 * (allset_linear_bf16_supported); rows 16-byte aligned, leading dimensions multiples of 8 elements.
 *   fwd:  y[n, N] = act(x[n, K] W[N, K]^T + bias[N])      act = relu if relu_out else identity; bias may be NULL
 *         aux_out[n, 4] (fp32) = x aux_w[4, K]^T + aux_b[4]   when aux_out != NULL (aux_w, aux_b: bf16; aux_b may be NULL)
 *   bwd:  ga = gy[n, O] where ymask[n, O] > 0 (ymask = the forward's relu output, or NULL: ga = gy);
 *         ga_out (optional, needs ymask) receives ga for the weight-gradient kernel (allset_wgrad_bf16);
 *         gx[n, I] = ga W[O, I]  [+ acc_in[n, I] (bf16: another gradient branch of x)]  [+ galpha[n, 4] (fp32) aux_w[4, I]]
 *         -- all terms summed in fp32 and rounded to bf16 once. */
int allset_linear_bf16_supported(int64_t in_features, int64_t out_features);
int allset_linear_bf16_fwd(const void* x, int64_t ldx, const void* W, const void* bias, int relu_out, const void* aux_w,
                           const void* aux_b, float* aux_out, void* y, int64_t ldy, int64_t n, int64_t K, int64_t N,
                           void* stream);
int allset_linear_bf16_bwd(const void* gy, int64_t ldg, const void* ymask, int64_t ldm, void* ga_out, int64_t lda,
                           const void* W, const float* galpha, const void* aux_w, const void* acc_in, int64_t ldacc,
                           void* gx, int64_t ldgx, int64_t n, int64_t O, int64_t I, void* stream);
/* ABI 13.  The relu mask as ONE BIT per element: allset_linear_bf16_fwd_mask is the forward with relu_out = 1 and a second output
 * mask_out -- n dense rows of allset_linear_bf16_mask_pitch(N) = N / 8 bytes, 16-byte aligned; bit (16 hb + j) of the row's word sq
 * (four words of N / 32 bytes per row) is set where y[row][64 hb + 16 sq + j] > 0 -- which allset_linear_bf16_bwd_bits
 * (gx = (gy where bit) W [+ acc_in]) and allset_wgrad_bf16_ex2 consume instead of the 2-byte-per-element activation: the masked
 * backward reads gy + n N / 8 bytes and writes gx only (no ga_out: the weight-gradient kernel masks on its own). */
int64_t allset_linear_bf16_mask_pitch(int64_t N);
int allset_linear_bf16_fwd_mask(const void* x, int64_t ldx, const void* W, const void* bias, void* y, int64_t ldy, void* mask_out,
                                int64_t n, int64_t K, int64_t N, void* stream);
int allset_linear_bf16_bwd_bits(const void* gy, int64_t ldg, const void* bits, const void* W, const void* acc_in, int64_t ldacc,
                                void* gx, int64_t ldgx, int64_t n, int64_t O, int64_t I, void* stream);

/* torch.optim.Adam's update (reference train.py:469; non-amsgrad, L2 weight decay) for up to allset_adam_max_tensors() fp32
 * tensors in ONE launch.  params / grads / exp_avg / exp_avg_sq / numel: HOST arrays of `count` device pointers / sizes (read
 * during the call); steps: one DEVICE float per tensor holding its t >= 1, which the caller increments before each call (so a
 * captured graph advances the bias corrections on replay).  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps). */
int allset_adam_max_tensors(void);
int allset_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                     const float* const* steps, const int64_t* numel, int64_t count, float lr, float beta1, float beta2, float eps,
                     float weight_decay, void* stream);
/* The same for tensors of one storage type (ALLSET_F32 or ALLSET_BF16: parameter, gradient and both moments in that type, as torch
 * keeps them; fp32 arithmetic, each stored value rounded once). */
int allset_adam_step_dtype(int dtype, void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                           const float* const* steps, const int64_t* numel, int64_t count, float lr, float beta1, float beta2,
                           float eps, float weight_decay, void* stream);

/* The training loop's loss (reference train.py:479-480: F.log_softmax over the classes, NLLLoss over the train split):
 *   loss = inv_count * sum_r w[r] * (logsumexp(logits[r, :]) - logits[r, y[r]])       w: 0/1 per row (NULL = all rows)
 * fwd writes allset_nll_partials(n) partial sums (the caller adds them); bwd writes d loss / d logits for EVERY row (zeros
 * where w is 0), scaled by gout[0] (a device scalar, NULL = 1).  y: int64 class per row (read only where w != 0). */
int allset_nll_partials(int64_t n, int64_t* n_partials);
int allset_nll_logsoftmax_fwd(const float* logits, int64_t ld, const int64_t* y, const float* w, float inv_count,
                              float* partials, int64_t n_partials, int64_t n, int64_t C, void* stream);
/* The same, with the sum of the partials finished inside the launch (ABI 10): the last workgroup to arrive -- counted on `ticket`, a
 * uint32 the caller zeroes ONCE and the launch re-arms (one per stream that may run this concurrently) -- adds partials[0 ..
 * n_partials) in index order into total[0].  Deterministic; one launch instead of kernel + reduction at dataset scale. */
int allset_nll_logsoftmax_fwd_total(const float* logits, int64_t ld, const int64_t* y, const float* w, float inv_count,
                                    float* partials, int64_t n_partials, uint32_t* ticket, float* total, int64_t n, int64_t C,
                                    void* stream);
/* PMA's folded attention logits (reference layers.py:126-131 forms K = lin_K(x) and contracts it with att_r; alpha is linear in
 * x, so the layer multiplies x by the folded weight instead):  w[h, k] = sum_c W_K[h C + c, k] att_r[h, c]  (f32 [H, K]),
 * b[h] = sum_c b_K[h C + c] att_r[h, c] (bk may be NULL: b = 0), and the backward of that fold. */
int allset_pma_fold_fwd(const float* Wk, const float* bk, const float* att, float* w, float* b, int64_t H, int64_t C, int64_t K,
                        void* stream);
int allset_pma_fold_bwd(const float* Wk, const float* bk, const float* att, const float* gw, const float* gb, float* gWk,
                        float* gbk, float* gatt, int64_t H, int64_t C, int64_t K, void* stream);
/* ABI 13.  The same for bf16 parameters (every pointer bf16; fp32 arithmetic, each output rounded once). */
int allset_pma_fold_fwd_bf16(const void* Wk, const void* bk, const void* att, void* w, void* b, int64_t H, int64_t C, int64_t K,
                             void* stream);
int allset_pma_fold_bwd_bf16(const void* Wk, const void* bk, const void* att, const void* gw, const void* gb, void* gWk, void* gbk,
                             void* gatt, int64_t H, int64_t C, int64_t K, void* stream);

/* Accuracy and loss of the reference's evaluate() (train.py:169-199) for three row sets in one pass: split[r] in {0, 1, 2} names the
 * row's set (anything else: none); partials: f32[allset_nll_partials(n)][6] = per-block sums {correct_0, correct_1, correct_2,
 * nll_0, nll_1, nll_2}; the caller adds the blocks and divides by the set sizes. */
int allset_split_metrics(const float* logits, int64_t ld, const int64_t* y, const int8_t* split, float* partials,
                         int64_t n_partials, int64_t n, int64_t C, void* stream);
int allset_nll_logsoftmax_bwd(const float* logits, int64_t ld, const int64_t* y, const float* w, float inv_count,
                              const float* gout, float* glogits, int64_t ldg, int64_t n, int64_t C, void* stream);


/* out[c] = sum_p part[p*M + c], p < P <= 4096: sums the partial buffers the kernels above hand back (M % 4 == 0).
 * scratch: f32[ceil(P/64) * M], required when P > 64 (two-level tree). */
int allset_reduce_partials(const float* part, int64_t P, int64_t M, float* out, float* scratch, void* stream);
/* The same with rows row_stride floats apart (only the first M of each row are summed) and a choice of output type
 * (out_dtype: ALLSET_F32 -> f32[M], ALLSET_BF16 -> bf16[M], rounded once from the fp32 sum). */
int allset_reduce_partials_ex(const float* part, int64_t P, int64_t row_stride, int64_t M, void* out, int out_dtype,
                              float* scratch, void* stream);
/* MANY small reductions in ONE launch (ABI 10; a dataset-scale training step is a chain of ~5 us kernels, a third of them
 * allset_reduce_partials): out[k][c] = sum_p parts[k][p * row_stride[k] + c], p < P[k], c < M[k], for k < count <=
 * allset_reduce_partials_batch_max(); every (P[k], M[k]) must satisfy allset_reduce_partials_batchable (P <= 512, P * M < 2^31:
 * the reductions the single entry finishes in one launch -- P <= 64, or P * M <= 2^21 -- and, since ABI 14, the ones it runs as a
 * two-launch tree, which one workgroup per column block then walks with the tree's association).  The pointer / size arrays are
 * HOST arrays (copied into the kernel's arguments).  Sums are bit-identical to count separate allset_reduce_partials calls. */
int allset_reduce_partials_batch_max(void);
int allset_reduce_partials_batchable(int64_t P, int64_t M);
int allset_reduce_partials_batched(const float* const* parts, const int64_t* P, const int64_t* row_stride, const int64_t* M,
                                   float* const* outs, int64_t count, void* stream);
/* ... and, by one more workgroup of the same launch, *inc_i64 += 1 (may be NULL) and *inc_f32[k] += 1.0f for k < n_inc_f32 <=
 * allset_reduce_partials_batch_max_counters(): the counters a training step advances once (its dropout-seed counter -- `seed_base`
 * of the dropout-bearing entries -- and the optimizer's step counters, allset_adam_step's `steps`).  count may be 0. */
int allset_reduce_partials_batch_max_counters(void);
int allset_reduce_partials_batched_ex(const float* const* parts, const int64_t* P, const int64_t* row_stride, const int64_t* M,
                                      float* const* outs, int64_t count, int64_t* inc_i64, float* const* inc_f32, int64_t n_inc_f32,
                                      void* stream);
/* ABI 14.  The same with an output type per buffer: out_dtypes[k] = ALLSET_F32 (outs[k]: f32[M[k]], 16-byte aligned) or ALLSET_BF16
 * (outs[k]: bf16[M[k]], 8-byte aligned, each sum rounded once -- bit-identical to allset_reduce_partials_ex); NULL = all fp32.
 * `| ALLSET_REDUCE_AS_TREE`: sum buffer k with the two-launch tree's association whatever its size -- for a COLUMN SLICE of a
 * wider buffer that must come out with the bits the reduction of the whole buffer gives (allset_reduce_partials_is_tree(P, M) tells
 * which association the single entry uses for a buffer of M columns). */
#define ALLSET_REDUCE_AS_TREE 0x100
int allset_reduce_partials_is_tree(int64_t P, int64_t M);
/* accs (may be NULL; entries may be NULL): accs[k] = f32[M[k]], 16-byte aligned, ADDED to the sum of buffer k before it is written (fp32
 * outputs only) -- the gradient a parameter already holds, so that accumulation costs no launch of its own.  parts[k] may point INTO a
 * wider partial buffer (a column range of it, row_stride[k] = the buffer's): one section of a kernel's partial row as its own entry. */
int allset_reduce_partials_batched_ex2(const float* const* parts, const int64_t* P, const int64_t* row_stride, const int64_t* M,
                                       void* const* outs, const float* const* accs, const int32_t* out_dtypes, int64_t count,
                                       int64_t* inc_i64, float* const* inc_f32, int64_t n_inc_f32, void* stream);

/* allset_wgrad with both operands recomputed on the fly from what allset_fused_linear_fwd keeps:
 *   ga = gy * (y > 0 ? 1/(1-p_out) : 0)  if y != NULL (relu/dropout epilogue), else gy;
 *   u  = dropout_{p_in,seed_in}( LayerNorm_{stats,gamma,beta}( relu_in ? relu(x) : x ) )  (LayerNorm iff stats != NULL). */
int allset_wgrad_fused(const float* gy, int64_t ldg, const float* y, int64_t ldy, float p_out,
                       const float* x, int64_t ldx, const float* stats, const float* gamma, const float* beta,
                       int relu_in, float p_in, uint64_t seed_in, float* part_w, float* part_b,
                       int64_t n_slices, int64_t n, int64_t O, int64_t I, const uint64_t* seed_base,
                       const uint32_t* mask, void* stream);
/* The same with ONE partial buffer (part: f32[n_slices][part_stride]: a slice's gW, then -- when want_bias -- its gb), so that a
 * single allset_reduce_partials_ex call sums both.  With y, mask, stats, gamma, beta NULL and relu_in = 0, p_in = p_out = 0 this is
 * allset_wgrad. */
int allset_wgrad_fused_ex(const float* gy, int64_t ldg, const float* y, int64_t ldy, float p_out, const float* x, int64_t ldx,
                          const float* stats, const float* gamma, const float* beta, int relu_in, float p_in, uint64_t seed_in,
                          float* part, int64_t part_stride, int want_bias, int64_t n_slices, int64_t n, int64_t O, int64_t I,
                          const uint64_t* seed_base, const uint32_t* mask, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Linear layers wider than 128 (reference layers.py:571-579 with MLP_hidden 256 / 512, src/run_AllSetTransformer.sh):
 * a tiled GEMM with fp32-accurate arithmetic on the bf16 matrix pipe (csrc/wide_mlp.hip).
 *   out[r, n] = epi( sum_k pro(A)[r, k] * B[n, k] + bias[n] ),   A: [rows, K] fp32,  B: [N, K]
 *   pro(a)[r,k]: a *= (mask_y[r,k] > 0 ? 1/(1-p_mask) : 0) if mask_y;  a = relu(a) if relu_in;
 *                a = (a - mean_r) * rstd_r * gamma[k] + beta[k] if stats ({mean, rstd} per row, allset_row_stats or
 *                allset_ln_fwd);  a *= dropout_{p_in, seed_in}(r*K + k)
 *   epi(v)[r,n]: v = relu(v) if relu_out;  v *= dropout_{p_out, seed_out}(r*N + n)
 * (the same dropout hash as every other dense-tail entry, so allset_ln_bwd / allset_wgrad_fused regenerate the masks).
 * B is passed as pre-split bf16 planes: allset_gemm_x6_planes(W, ldw, transpose, planes, N, K) with
 *   transpose == 0: B = W [N, K] (forward: N = out features, K = in features);
 *   transpose != 0: B[n, k] = W[k, n], W [K, N] (backward-data: N = in features, K = out features);
 * planes: allset_gemm_x6_plane_bytes(N, K) bytes, 16-byte aligned.  Supported: K % 32 == 0, N % 4 == 0, both <= 4096.
 * ------------------------------------------------------------------------------------------- */
int allset_gemm_x6_supported(int64_t N, int64_t K);
int64_t allset_gemm_x6_plane_bytes(int64_t N, int64_t K);
int allset_gemm_x6_planes(const float* W, int64_t ldw, int transpose, void* planes, int64_t N, int64_t K, void* stream);
int allset_row_stats(const float* x, int64_t ldx, int relu_in, float eps, float* stats, int64_t rows, int64_t d, void* stream);
/* The same GEMM with a LayerNorm-BACKWARD epilogue, for N <= 256 (a tile holds whole rows) -- the autograd of one
 * `norm -> Linear` stage of reference MLP.forward (layers.py:571-579) with respect to the stage's input: backward-data of a Linear whose
 * input was u = dropout_p(LN(relu_in ? relu(x) : x)) in one kernel --
 *   gu = (G * epilogue mask from mask_y / p_mask) @ B^T  stays on chip;  gx (stored) = LayerNorm backward of gu wrt x
 *   (dropout mask regenerated from (seed, r*N + n), relu mask from the sign of x, {mean, rstd} from `stats`);
 *   partials[w][0|1][N] = workgroup w's sums of dgamma = gu * xhat and dbeta = gu over its rows; the caller adds the
 *   allset_gemm_x6_lnb_partials(rows) rows up (allset_reduce_partials).  Replaces allset_gemm_x6 + allset_ln_bwd. */
int64_t allset_gemm_x6_lnb_partials(int64_t rows);
int allset_gemm_x6_lnb(const float* G, int64_t ldg, const float* mask_y, int64_t ldy, float p_mask, const void* planes,
                       const float* x, int64_t ldx, const float* stats, const float* gamma, int relu_in, float p, uint64_t seed,
                       float* gx, int64_t ldgx, float* partials, int64_t n_partials, int64_t rows, int64_t N, int64_t K,
                       const uint64_t* seed_base, void* stream);
int allset_gemm_x6(const float* A, int64_t lda, const float* mask_y, int64_t ldy, float p_mask, int relu_in,
                   const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                   const void* planes, const float* bias, int relu_out, float p_out, uint64_t seed_out,
                   float* out, int64_t ldo, int64_t rows, int64_t N, int64_t K, const uint64_t* seed_base, void* stream);
/* The same three entries in the fp16x3 arithmetic (ABI 11; ALLSET_ARITH_* above has the error model): the weight as TWO fp16 planes,
 * each output column scaled into fp16's window (inverse scales stored behind the planes: allset_gemm_f16x3_plane_bytes), A behind a
 * LayerNorm-apply prologue by one power of two for the launch (bound from gamma / beta), any other A per ROW from its largest element
 * -- read one tile ahead inside the kernel, no extra pass.  Three MFMAs per product instead of six.  Contracts as allset_gemm_x6 /
 * allset_gemm_x6_lnb / allset_gemm_x6_planes (`planes` from allset_gemm_f16x3_planes). */
int64_t allset_gemm_f16x3_plane_bytes(int64_t N, int64_t K);
int allset_gemm_f16x3_planes(const float* W, int64_t ldw, int transpose, void* planes, int64_t N, int64_t K, void* stream);
/* ABI 14.  The plane images of MANY weights in ONE launch (a dataset-scale training step at 256 / 512-wide layers rebuilds W and W^T of
 * every wide Linear, each a ~5-us launch on the step's dependent chain): planes[k] as allset_gemm_f16x3_planes would write it, k < count <=
 * allset_gemm_f16x3_planes_batch_max().  The arrays are HOST arrays. */
int allset_gemm_f16x3_planes_batch_max(void);
int allset_gemm_f16x3_planes_batched(const float* const* Ws, const int64_t* ldws, const int32_t* transposes, void* const* planes,
                                     const int64_t* Ns, const int64_t* Ks, int64_t count, void* stream);
int allset_gemm_f16x3_lnb(const float* G, int64_t ldg, const float* mask_y, int64_t ldy, float p_mask, const void* planes,
                          const float* x, int64_t ldx, const float* stats, const float* gamma, int relu_in, float p, uint64_t seed,
                          float* gx, int64_t ldgx, float* partials, int64_t n_partials, int64_t rows, int64_t N, int64_t K,
                          const uint64_t* seed_base, void* stream);
int allset_gemm_f16x3(const float* A, int64_t lda, const float* mask_y, int64_t ldy, float p_mask, int relu_in,
                      const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                      const void* planes, const float* bias, int relu_out, float p_out, uint64_t seed_out,
                      float* out, int64_t ldo, int64_t rows, int64_t N, int64_t K, const uint64_t* seed_base, void* stream);
/* Supersets of allset_gemm_x6 / _f16x3 and their _lnb forms (ABI 11): `arith` names the planes' format (ALLSET_ARITH_BF16X6 or
 * ALLSET_ARITH_FP16X3, declared below), and the forward's 1-bit activation mask travels instead of its fp32 output: mask_out (forward,
 * N % 64 == 0, may be NULL) receives "out > 0" after the epilogue in the "mask layout" of allset_fused_linear_fwd; mask_bits (backward,
 * K % 64 == 0, may be NULL; then mask_y must be NULL) replaces mask_y -- one dword per thread and K step where mask_y costs a second
 * [rows, K] fp32 read.  allset_wgrad_fused_ex takes the same buffer as `mask`.  stats_out (forward, N == 256, may be NULL): f32[rows][2]
 * {mean, rstd} of (stats_relu ? relu(out) : out) with eps = stats_eps -- what allset_row_stats would compute from `out` for the NEXT
 * Linear's LayerNorm prologue, written by the epilogue's row pass instead of by a second pass over the output. */
int allset_gemm_wide(int arith, const float* A, int64_t lda, const float* mask_y, int64_t ldy, const uint32_t* mask_bits, float p_mask,
                     int relu_in, const float* stats, const float* gamma, const float* beta, float p_in, uint64_t seed_in,
                     const void* planes, const float* bias, int relu_out, float p_out, uint64_t seed_out, uint32_t* mask_out,
                     float* stats_out, float stats_eps, int stats_relu, float* out, int64_t ldo, int64_t rows, int64_t N, int64_t K,
                     const uint64_t* seed_base, void* stream);
/* ABI 15.  Backward-data of a wide Linear whose INPUT was relu(x) with no LayerNorm in between (PMA's rFF, reference layers.py:128-130
 * with Normalization 'None'): gx = ((G masked as in allset_gemm_wide) @ B^T) where x > 0, zero elsewhere -- the relu's backward as the
 * GEMM's epilogue instead of an allset_relu_dropout_bwd pass over [rows, N] (bit-identical to that pair).  x: f32 [rows, N] with
 * 16-byte aligned rows (ldx % 4 == 0); planes as for allset_gemm_wide.  Built on the split-role kernel only:
 * allset_gemm_wide_sgn_supported(arith, N, K) -- fp16x3 planes, K % 128 == 0; otherwise ALLSET_ERR_UNSUPPORTED (the caller keeps the
 * elementwise pass). */
int allset_gemm_wide_sgn_supported(int arith, int64_t N, int64_t K);
int allset_gemm_wide_sgn(int arith, const float* G, int64_t ldg, const float* mask_y, int64_t ldy, const uint32_t* mask_bits, float p_mask,
                         const void* planes, const float* x, int64_t ldx, float* gx, int64_t ldgx, int64_t rows, int64_t N, int64_t K,
                         const uint64_t* seed_base, void* stream);
int allset_gemm_wide_lnb(int arith, const float* G, int64_t ldg, const float* mask_y, int64_t ldy, const uint32_t* mask_bits, float p_mask,
                         const void* planes, const float* x, int64_t ldx, const float* stats, const float* gamma, int relu_in, float p,
                         uint64_t seed, float* gx, int64_t ldgx, float* partials, int64_t n_partials, int64_t rows, int64_t N, int64_t K,
                         const uint64_t* seed_base, void* stream);
/* The weight / bias gradient of a wide Linear behind a LayerNorm prologue on TWO fp16 planes per operand (csrc/wgrad_f16.hip; the
 * ALLSET_ARITH_FP16X3 arithmetic): allset_wgrad_fused_ex's contract with `mask` the forward's 1-bit activation mask (NULL: the Linear has
 * no relu / dropout epilogue; a fp32 mask source y is not taken) and stats / gamma / beta REQUIRED.  Built for O % 256 == 0,
 * I % 128 == 0, I <= 512 (_supported); its own slice count (_slices).  gy is scaled per 32-row stage, u by one power of two for the
 * launch times the stage's distance to the largest stage met so far -- the contract stated under "arithmetic" below: a gradient COLUMN
 * more than 2^14 below the largest element of its stage loses low bits.  ALLSET_ARITH_BF16X6 callers use allset_wgrad_fused_ex. */
int allset_wgrad_f16x3_supported(int64_t O, int64_t I);
int allset_wgrad_f16x3_slices(int64_t n, int64_t O, int64_t I, int64_t* n_slices);
int allset_wgrad_f16x3(const float* gy, int64_t ldg, const uint32_t* mask, float p_out, const float* x, int64_t ldx, const float* stats,
                       const float* gamma, const float* beta, int relu_in, float p_in, uint64_t seed_in, float* part,
                       int64_t part_stride, int want_bias, int64_t n_slices, int64_t n, int64_t O, int64_t I,
                       const uint64_t* seed_base, void* stream);



/* allset_pma_fwd_ex / allset_pma_bwd_stats / allset_pma_bwd_src_ex with explicit leading dimensions for the small per-row
 * operands the kernels GATHER next to a feature row: the logits (`lda` floats between rows, >= H) and the backward
 * statistics (`lds` floats between rows, even, >= 2H; stats 8-byte aligned).  A caller whose feature rows are narrower
 * than a cache line (the column-sharded layer: d/P columns) interleaves [V row | logits] and [gout row | stats] in one
 * 128-byte-pitched buffer and passes pointers into it, so each incidence costs one cache-line request instead of two. */
int allset_pma_fwd_ld(int dtype, int variant, int64_t nnz, const int32_t* row_order, const int32_t* rowptr, const int32_t* col,
                      const float* alpha, int64_t lda, const void* V, int64_t ldv, float slope, void* out, int64_t ldo,
                      float* m, float* l, int64_t n_t, int64_t n_s, int64_t H, int64_t C, void* stream);
int allset_pma_bwd_stats_ld(int dtype, const void* out, int64_t ldo, const void* gout, int64_t ldg, const float* m,
                            const float* l, float* stats, int64_t lds, int64_t n_t, int64_t H, int64_t C, void* stream);
int allset_pma_bwd_src_ld(int dtype, int variant, int64_t nnz, const int32_t* row_order, const int32_t* rowptrT,
                          const int32_t* colT, const float* alpha, const void* V, int64_t ldv, const void* gout, int64_t ldg,
                          const float* stats, int64_t lds, float slope, void* gV, int64_t ldgv, float* galpha, int64_t n_s,
                          int64_t n_t, int64_t H, int64_t C, void* stream);

/* Layout change around the all-to-all of the column-sharded layer (allset_amd/dist.py; no reference counterpart -- the
 * reference is single-device).  A row-major matrix of `rows` rows whose row holds P column blocks of
 * `block_bytes` (a multiple of 16) each, leading dimension `ld_bytes`, and the block-major buffer [P][rows][block_bytes]
 * an all-to-all sends / receives:   to_blocks != 0: src row-major -> dst block-major (pack);  0: src block-major -> dst
 * row-major (unpack).  Any element type; all pointers 16-byte aligned. */
int allset_block_transpose(const void* src, void* dst, int64_t rows, int64_t P, int64_t block_bytes, int64_t ld_bytes,
                           int to_blocks, void* stream);

/* Multi-GPU E->V attention pooling: pack this rank's partial result for the cross-rank merge.
 *   packed[r] = [ out_loc[r,h,:] * w[r,h] for all h | w[r,0..H-1] ],  w = l_loc > 0 ? l_loc * exp(m_loc - m_glob) : 0
 * (out_loc, m_loc, l_loc from allset_pma_fwd on the local incidences; m_glob = max over ranks of m_loc).  A sum over ranks of
 * packed rows gives numerator and denominator of the global softmax pooling.  packed: f32[n*ldp], ldp >= H*C + H. */
int allset_pma_merge_pack(const float* out_loc, int64_t ldo, const float* m_loc, const float* l_loc, const float* m_glob,
                          float* packed, int64_t ldp, int64_t n, int64_t H, int64_t C, void* stream);

/* allset_ln_fwd / allset_ln_bwd for bf16 activations and bf16 gamma / beta (BASELINE configs[4] regime): fp32 statistics
 * and arithmetic, bf16 in and out, stats and parameter partials fp32.  Widths: allset_ln_bf16_supported(d) (d % 8 == 0,
 * d <= 512).  gx may be NULL (partials only). */
int allset_ln_bf16_supported(int64_t d);
int allset_ln_fwd_bf16(const void* x, int64_t ldx, const void* gamma, const void* beta, float eps, int relu_in, float p,
                       uint64_t seed, void* y, int64_t ldy, float* stats, int64_t n, int64_t d, const uint64_t* seed_base,
                       void* stream);
int allset_ln_bwd_bf16_partials(int64_t n, int64_t d, int64_t* n_partials);
int allset_ln_bwd_bf16(const void* gy, int64_t ldg, const void* x, int64_t ldx, const float* stats, const void* gamma,
                       int relu_in, float p, uint64_t seed, void* gx, int64_t ldgx, float* partials, int64_t n_partials,
                       int64_t n, int64_t d, const uint64_t* seed_base, void* stream);

/* allset_ln_res_fwd / _bwd (below) for bf16 activations and parameters: same semantics, fp32 arithmetic, fp32 stats and
 * partials (n_partials from allset_ln_bwd_bf16_partials, 3 rows of d per block).  Widths: allset_ln_bf16_supported(d). */
int allset_ln_res_fwd_bf16(const void* x, int64_t ldx, const void* colb, const void* res, int64_t ldr, const void* gamma,
                           const void* beta, float eps, int relu_out, float p, uint64_t seed, void* y, int64_t ldy,
                           float* stats, int64_t n, int64_t d, const uint64_t* seed_base, void* stream);
int allset_ln_res_bwd_bf16(const void* gy, int64_t ldg, const void* x, int64_t ldx, const void* colb, const void* res,
                           int64_t ldr, const float* stats, const void* gamma, const void* beta, int relu_out, float p,
                           uint64_t seed, void* gs, int64_t ldgs, float* partials, int64_t n_partials, int64_t n, int64_t d,
                           const uint64_t* seed_base, void* stream);

/* LayerNorm with a fused sum in front and a relu behind -- the PMA tail (reference layers.py:153-157) and the
 * relu -> dropout SetGNN puts behind every conv (models.py:475-481):
 *   y = dropout_{p,seed}( relu_out ? relu(.) : . )( LayerNorm_{gamma,beta,eps}( x + colb + res ) )
 * colb f32[d] (may be NULL: e.g. PMA's seed vector att_r) and res f32[n*ldr] (may be NULL: the residual branch) are added
 * in registers; stats f32[n*2] = {mean, rstd} of the sum.  Widths: allset_ln_res_supported(d) (d % 4 == 0, d <= 512; two 16-byte chunks per lane above 256, ABI 14).
 * Backward: gs = d loss / d (x + colb + res) -- the gradient of x AND of res; partials f32[n_partials*3*d], row k holds
 * block k's (dgamma[d], dbeta[d], dcolb[d]); the caller sums over k.  The relu mask is recomputed from the statistics. */
int allset_ln_res_supported(int64_t d);
int allset_ln_res_fwd(const float* x, int64_t ldx, const float* colb, const float* res, int64_t ldr, const float* gamma,
                      const float* beta, float eps, int relu_out, float p, uint64_t seed, float* y, int64_t ldy,
                      float* stats, int64_t n, int64_t d, const uint64_t* seed_base, void* stream);
int allset_ln_res_bwd_partials(int64_t n, int64_t d, int64_t* n_partials);
int allset_ln_res_bwd(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* colb, const float* res,
                      int64_t ldr, const float* stats, const float* gamma, const float* beta, int relu_out, float p,
                      uint64_t seed, float* gs, int64_t ldgs, float* partials, int64_t n_partials, int64_t n, int64_t d,
                      const uint64_t* seed_base, void* stream);
/* The PMA tail's first LayerNorm, backward, with the statistics pass folded in: y = LayerNorm(x + colb) where x is the pooled
 * output of allset_pma_fwd (reference layers.py:153-154).  Same results as allset_ln_res_bwd(res = NULL, relu_out = 0, p = 0)
 * plus pma_stats f32[n*heads*2] = {m + log(l + 1e-16), <x[t,h,:], gs[t,h,:]>} -- exactly what allset_pma_bwd_stats(x, gs, m, l)
 * would write, from the registers that hold x and gs (one pass over both saved).  allset_ln_res_bwd_pma_supported(d, heads):
 * d / heads must be 4 x a power of two. */
int allset_ln_res_bwd_pma_supported(int64_t d, int64_t heads);
int allset_ln_res_bwd_pma(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* colb, const float* stats,
                          const float* gamma, const float* beta, float* gs, int64_t ldgs, float* partials, int64_t n_partials,
                          int64_t n, int64_t d, const float* pma_m, const float* pma_l, float* pma_stats, int64_t heads,
                          void* stream);
/* The same for bf16 activations (channels per head = 8 x a power of two). */
int allset_ln_res_bwd_pma_bf16_supported(int64_t d, int64_t heads);
int allset_ln_res_bwd_pma_bf16(const void* gy, int64_t ldg, const void* x, int64_t ldx, const void* colb, const float* stats,
                               const void* gamma, const void* beta, void* gs, int64_t ldgs, float* partials,
                               int64_t n_partials, int64_t n, int64_t d, const float* pma_m, const float* pma_l,
                               float* pma_stats, int64_t heads, void* stream);

/* Fused tall-skinny Linear (K = in features, N = out features, both in {64, 128}; W row-major [N][K] contiguous):
 *   y = epi( pro(x) @ W^T + b ),  pro = [relu_in] -> [LayerNorm(gamma,beta,eps) if gamma != NULL] -> [dropout p_in],
 *                                 epi = [relu_out] -> [dropout p_out].
 * One read + one write of the activation matrix.  Arithmetic: fp32 on the bf16 matrix pipe -- every operand is split
 * exactly into three bf16 values and six of the nine partial products are accumulated in fp32 ("bf16x6", dropped terms
 * <= 2^-23 relative: as accurate as a native fp32 MFMA, 2.7x its rate on gfx950; the only
 * kernel family since ABI 9 -- except at K = N = 128 behind a LayerNorm prologue (ALLSET_NORM_LAYER), where the forward uses the
 * two-fp16-plane scheme of the one-pass backward, see allset_fused_linear_bwd_all: the LayerNorm output is bounded, one power of
 * two for the launch and one per 32-column slice of W bring the operands into fp16's window; error per product <= 2^-21 relative,
 * tests/test_gpu_dense.py test_fused_linear_forward_fp16x3_against_float64).  stats (f32[n*2] =
 * {mean, rstd}) is written when the LayerNorm prologue is on.  allset_fused_linear_supported(K, N) -> 1/0.
 *
 * Auxiliary output columns (optional, bf16x6 kernels): aux_out f32[n*4] = pro(x) @ aux_w^T + aux_b with aux_w f32[4*K],
 * aux_b f32[4] or NULL -- four extra output columns from the rows already in registers (PMA's folded attention logits
 * next to its value projection, reference layers.py:126-131).  Their gradient w.r.t. x is the rank-4 update
 * gx += aux_g[n,4] @ aux_w[4,I] of allset_fused_linear_bwd (aux_g, aux_w; NULL = none).
 *
 * Activation mask (optional, bf16x6 kernels): mask_out receives 1 bit per output element, "y > 0" after the epilogue,
 * so the backward kernels need not re-read y.  Layout ("mask layout"): blocks of 16 rows x 64 columns, 32 dwords each,
 * block index (row / 16) * (N / 64) + col / 64; inside a block, dword ((row % 16) / 4) * 8 + (row % 4) * 2 +
 * (col % 64) / 32, bit 8 * (col % 4) + (col % 32) / 4.  Size: allset_fused_linear_mask_words(n, N) dwords (0 when the
 * mask is not supported: N % 64 != 0).  Pass the same buffer as `mask` to
 * allset_fused_linear_bwd / allset_wgrad_fused instead of y. */
int allset_fused_linear_supported(int64_t K, int64_t N);
int64_t allset_fused_linear_mask_words(int64_t n, int64_t N);
int allset_fused_linear_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                            int relu_in, float p_in, uint64_t seed_in, const float* W, const float* bias,
                            int relu_out, float p_out, uint64_t seed_out, float* y, int64_t ldy, float* stats,
                            int64_t n, int64_t K, int64_t N, const uint64_t* seed_base, uint32_t* mask_out,
                            const float* aux_w, const float* aux_b, float* aux_out, void* stream);

/* Backward of allset_fused_linear_fwd w.r.t. x (O = out features, I = in features, both in {64,128}):
 *   ga = gy * (y > 0 ? 1/(1-p_out) : 0) if y != NULL else gy;   gu = ga @ W;   gz = gu * dropout_{p_in,seed_in} mask;
 *   gx = LayerNorm-backward(gz; x, stats, gamma) through relu_in   (stats != NULL), else gz through relu_in.
 * partials (stats != NULL): f32[n_partials*2*I], row w = wave w's (dgamma[I], dbeta[I]); the caller sums over rows.
 * n_partials from allset_fused_linear_bwd_partials(n).
 * gx may be NULL when only the LayerNorm parameter partials are wanted (the input needs no gradient: a model's first layer).
 * acc_in (may be NULL; f32[n*ldacc], may alias gx): added to the result, gx = acc_in + (this Linear's gradient) -- where
 * a tensor feeds two branches the second branch's backward kernel does the sum instead of a separate add pass. */
int allset_fused_linear_bwd_partials(int64_t n, int64_t* n_partials);
int allset_fused_linear_bwd(const float* gy, int64_t ldg, const float* y, int64_t ldy, float p_out, const float* W,
                            const float* x, int64_t ldx, const float* stats, const float* gamma, int relu_in,
                            float p_in, uint64_t seed_in, float* gx, int64_t ldgx, float* partials,
                            int64_t n_partials, int64_t n, int64_t O, int64_t I, const uint64_t* seed_base,
                            const uint32_t* mask, const float* acc_in, int64_t ldacc, const float* aux_g,
                            const float* aux_w, void* stream);

/* The WHOLE backward of allset_fused_linear_fwd in one pass over gy and x (bf16x6 kernel family): everything
 * allset_fused_linear_bwd returns (gx, LayerNorm partials) AND the weight / bias gradient of allset_wgrad_fused,
 *   gW[O][I] = ga^T @ u,   gb[O] = column sums of ga,    u = dropout_{p_in,seed_in}(LayerNorm(relu_in(x))) recomputed,
 * from one read of gy, the activation mask, x and the row statistics (1.6 GB per [1M,128]x[128,128] Linear instead of 2.7 GB
 * for the pair of kernels).  What torch autograd computes for reference MLP.forward, layers.py:571-579.
 *   mask     the 1-bit activation mask of the forward ("mask layout" above) or NULL for a Linear without relu/dropout epilogue
 *   x        always required (the weight gradient recomputes the Linear's input from it); stats/gamma/beta come together or NULL
 *   gx       required (a Linear whose input needs no gradient keeps the two-kernel pair); acc_in as in allset_fused_linear_bwd
 *   part_w   f32[n_slices*O*I], part_b f32[n_slices*O] or NULL, part_ln f32[n_slices*2*I] (stats != NULL): one partial per
 *            wave or workgroup, n_slices from allset_fused_linear_bwd_all_slices_for(n, O, I, acc_in != NULL); the caller sums over slices (allset_reduce_partials).
 *   part_stride  0: part_w / part_b / part_ln are three dense arrays as sized above; > 0 (>= O*I + O + 2*I in practice): they
 *            point into ONE f32[n_slices*part_stride] buffer, slice k's sections at k*part_stride from each pointer -- one
 *            allset_reduce_partials launch then sums all of a Linear's parameter gradients.
 * Arithmetic: bf16x6 as in the forward, except at O = I = 128, whatever the prologue, with or without acc_in (i.e. everything
 * but allset_fused_linear_bwd_all_aux's auxiliary columns):
 * "fp16x3" (csrc/fused_bwd6.hip) -- every operand is scaled by a power of two (gy per row, W per 32-column slice, the
 * recomputed input against the workgroup's running largest row product) and split into TWO fp16 values, three of the four partial
 * products are accumulated in fp32 on the f16 matrix pipe (half the matrix instructions of bf16x6).  Error per product <= 2^-21
 * relative + 2^-38 of (the row's largest |gy|) x |input|: on sums, a library fp32 GEMM's level (tests/test_gpu_dense.py
 * test_one_pass_backward_fp16x3_*_against_float64); a gradient column 2^17 below its rows' largest element loses low bits.
 * No atomics: bitwise reproducible run to run.  allset_fused_linear_bwd_all_supported(O, I, flags) -> 1/0: widths in
 * {64,128} and the prologue / epilogue combinations the module surface produces (dropout_in only behind relu_in, acc_in only
 * on the plain Linear).  Unsupported -> ALLSET_ERR_UNSUPPORTED; use the two-kernel pair. */
int allset_fused_linear_bwd_all_supported(int64_t O, int64_t I, int has_ln, int drop_in, int relu_in, int has_mask, int has_acc);
int allset_fused_linear_bwd_all_slices(int64_t n, int64_t* n_slices);      /* DEPRECATED: the one-wave-per-SIMD kernel's count (widths other than O = I = 128); size buffers with _slices_for */
/* The slice count allset_fused_linear_bwd_all expects for these widths (ABI 6): the O = I = 128 kernel keeps ONE weight-gradient
 * accumulator per workgroup (n_slices = number of workgroups), the other widths one per wave. */
int allset_fused_linear_bwd_all_slices_for(int64_t n, int64_t O, int64_t I, int has_acc, int64_t* n_slices);

/* ---- training-mode BatchNorm1d of the reference MLP (`Normalization='bn'`, the constructor default: layers.py:499-517, 571-579) ----
 * BatchNorm(f(x)) with batch statistics is a per-column affine map f(x) * a + b (a = gamma * rsqrt(var + eps), b = beta - mean * a):
 * the caller takes the two column moments with allset_col_moments (+ allset_reduce_partials), hands (a, b) to the fused Linear
 * as its prologue (norm_mode = ALLSET_NORM_COLUMN_AFFINE) and adds the statistics' own dependence on x to the input gradient
 * with allset_col_affine_add.  No normalised tensor is written.  csrc/batchnorm.hip; allset_amd/dense.py _BatchNormLinear. */
#define ALLSET_NORM_LAYER 0            /* (gamma, beta) = LayerNorm weight / bias; row statistics computed in the kernel */
#define ALLSET_NORM_COLUMN_AFFINE 1    /* (gamma, beta) = per-column scale / shift; `stats` is filled with {0, 1} per row */
/* allset_fused_linear_fwd without auxiliary columns + norm_mode (ALLSET_NORM_LAYER reproduces allset_fused_linear_fwd). */
int allset_fused_linear_fwd_nm(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int norm_mode,
                               int relu_in, float p_in, uint64_t seed_in, const float* W, const float* bias, int relu_out,
                               float p_out, uint64_t seed_out, float* y, int64_t ldy, float* stats, int64_t n, int64_t K,
                               int64_t N, const uint64_t* seed_base, uint32_t* mask_out, void* stream);
/* allset_fused_linear_bwd_all without acc_in + norm_mode.  ALLSET_NORM_COLUMN_AFFINE: stats = the {0, 1} rows the forward wrote;
 * gx = (gy W) * gamma through the relu / dropout masks (no row-mean terms); part_ln[slice] = {sum_r gu * f(x), sum_r gu} per column
 * = the gradients of the column scale and shift. */
int allset_fused_linear_bwd_all_nm(const float* gy, int64_t ldg, const uint32_t* mask, float p_out, const float* W, const float* x,
                                   int64_t ldx, const float* stats, const float* gamma, const float* beta, int norm_mode,
                                   int relu_in, float p_in, uint64_t seed_in, float* gx, int64_t ldgx, float* part_ln,
                                   float* part_w, float* part_b, int64_t n_slices, int64_t n, int64_t O, int64_t I,
                                   const uint64_t* seed_base, int64_t part_stride, void* stream);
/* part[slice][c] = sum over the slice's rows of f(x)[r,c] (center == NULL) or of (f(x)[r,c] - center[c])^2, f = relu if relu_in
 * else identity; x: f32 rows of d (4 <= d <= 1024, d % 4 == 0), 16-byte aligned; n_slices from allset_col_moments_slices(n); the
 * caller sums the slices (allset_reduce_partials) and divides by n.  Two calls give mean and the centred (biased) variance. */
int allset_col_moments_supported(int64_t d);
int allset_col_moments_slices(int64_t n, int64_t* n_slices);
int allset_col_moments(const float* x, int64_t ldx, int64_t n, int64_t d, int relu_in, const float* center, float* part,
                       int64_t n_slices, void* stream);
/* Both raw moments in one read: part f64 [n_slices][2][d] = per-slice sums of f(x) and of f(x)^2, accumulated in double precision
 * (the caller sums the slices and takes var = E[f^2] - mean^2 in fp64). */
int allset_col_moments2(const float* x, int64_t ldx, int64_t n, int64_t d, int relu_in, double* part, int64_t n_slices, void* stream);
/* gx[r,c] += m * (f(x)[r,c] * s[c] + t[c]) in place, m = (x[r,c] > 0) if relu_mask else 1: the gradient of the batch statistics
 * (s = 2 dvar / n, t = dmean / n - s * mean). */
int allset_col_affine_add(float* gx, int64_t ldgx, const float* x, int64_t ldx, const float* s, const float* t, int relu_mask,
                          int64_t n, int64_t d, void* stream);
/* ---- the same first Linear on SPARSE raw features (bag-of-words rows; the loaders hand over a dense matrix, reference models.py:473-476
 * multiplies all of it): the two GEMMs become sums over the non-zeros.  The caller holds the features' CSR (rowptr [n+1], col, val
 * [nnz], row-major order) and CSC (colptr [d+1], rowT [nnz] = row of each entry in column-major order, posT [nnz] = its CSR position).
 *   allset_fold_ln_linear_t   WT [d + 2, O]: rows j < d = W[:, j] * gamma_j, row d = sum_j W[:, j] gamma_j, row d + 1 = b + W beta
 *   allset_sparse_ln_linear_fwd   y[n, O] = Linear(LayerNorm(dropout_p(x))) from the non-zeros (statistics: exact two-pass sums + the
 *       zeros' closed form; dropout = the counter hash of (seed, r * d + j), the positions the dense kernels drop); keeps
 *       w_out[p] = rstd_r * v_p per non-zero (CSR order) and rm[r] = rstd_r * mean_r for the backward
 *   allset_sparse_ln_linear_bwd   M[k, j] = sum_{p in column j} w[posT[p]] gy[rowT[p], k]  ([O, ldm >= d]) and su_part
 *       [allset_sparse_ln_linear_slices()][2][O] = per-slice sums of gy[r, :] and of rm[r] gy[r, :]
 *   allset_unfold_ln_linear_ex    allset_unfold_ln_linear with M_true = M - u and the ones column = s taken from su_part
 * O in {64, 128, 256} (allset_sparse_ln_linear_supported).  csrc/sparse_input.hip; allset_amd/dense.py _SparseInputNormLinear. */
int allset_sparse_ln_linear_supported(int64_t O);
int allset_sparse_ln_linear_slices(void);
int allset_fold_ln_linear_t(const float* W, int64_t ldw, const float* gamma, const float* beta, const float* b, int64_t O, int64_t d,
                            float* WT, void* stream);
int allset_sparse_ln_linear_fwd(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n, int64_t d, const float* WT,
                                int64_t O, float eps, float p_pre, uint64_t seed, const uint64_t* seed_base, float* y, int64_t ldy,
                                float* w_out, float* rm, void* stream);
int allset_sparse_ln_linear_bwd(const int32_t* colptr, const int32_t* rowT, const int32_t* posT, const float* w, const float* rm,
                                const float* gy, int64_t ldg, int64_t n, int64_t d, int64_t O, float* M, int64_t ldm, float* su_part,
                                void* stream);
int allset_unfold_ln_linear_ex(const float* M, int64_t ldm, const float* W, int64_t ldw, const float* gamma, const float* beta,
                               int64_t O, int64_t d, float* gW, int64_t ldgw, float* gb, float* ggamma, float* gbeta,
                               const float* su_part, int64_t n_slices, void* stream);

/* ABI 14.  The PLAIN Linear on the same sparse rows, with up to four auxiliary output rows: PMA's value projection and its folded
 * logits on the FIRST conv of an AllSetTransformer (reference models.py:473 -- the input dropout -- and layers.py:126-131: lin_V and
 * lin_K applied to Citeseer's 3703-wide bag-of-words rows; as library GEMMs + a dropout pass they were 165 us of a 440-us step):
 *   allset_sparse_linear_wt    WT[d + 1, pitch]: row j < d = column j of the stacked weight [W1 (O1 rows); W2 (O2 <= 8 rows); 0 ...],
 *                              row d = the stacked bias (b1, b2 may be NULL); pitch = allset_sparse_linear_pitch(O1, O2) = O1 + 4 ceil(O2 / 4)
 *   allset_sparse_linear_fwd   y[n, O1] = dropout_p(x) W1^T + b1 and y2[n, pitch - O1] = dropout_p(x) W2^T + b2 (columns >= O2 zero) from the
 *                              non-zeros; the dropout is the library's hash of (seed [, *seed_base], r * d + j); w_out[nnz] keeps the
 *                              values after the dropout for the backward
 *   allset_sparse_linear_bwd   gW1[O1, ldw1 >= d] and gW2[O2, ldw2 >= d] from gy[n, O1] and g2[n, pitch - O1] over the CSC of x; sb_part
 *                              [allset_sparse_ln_linear_slices()][pitch] = per-slice column sums of [gy | g2] (the bias gradients);
 *                              with ticket (a zeroed uint32 the launch re-arms, one per stream) and sb_total[pitch] the last slice
 *                              workgroup to finish sums the slices in index order into sb_total -- no reduction launch; both NULL:
 *                              the caller sums the slices (allset_reduce_partials)
 * x needs no gradient (raw features).  O1 in {64, 128, 256, 512}, O2 <= 8 (allset_sparse_linear_supported: the widths and head
 * counts of the reference's tuned run_AllSetTransformer.sh configurations).  csrc/sparse_input.hip;
 * allset_amd/dense.py _SparsePmaProject. */
int allset_sparse_linear_supported(int64_t O1, int64_t O2);
int64_t allset_sparse_linear_pitch(int64_t O1, int64_t O2);
int allset_sparse_linear_wt(const float* W1, int64_t ld1, int64_t O1, const float* W2, int64_t ld2, int64_t O2, const float* b1,
                            const float* b2, int64_t d, float* WT, void* stream);
int allset_sparse_linear_fwd(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n, int64_t d, const float* WT,
                             int64_t O1, int64_t O2, float p_pre, uint64_t seed, const uint64_t* seed_base, float* y, int64_t ldy,
                             float* y2, float* w_out, void* stream);
int allset_sparse_linear_bwd(const int32_t* colptr, const int32_t* rowT, const int32_t* posT, const float* w, const float* gy,
                             int64_t ldg, const float* g2, int64_t n, int64_t d, int64_t O1, int64_t O2, float* gW1, int64_t ldw1,
                             float* gW2, int64_t ldw2, float* sb_part, uint32_t* ticket, float* sb_total, void* stream);

/* ---- backward of a Linear with a NARROW output: the classifier head Linear(hidden -> num_classes) (reference models.py:449-456) ----
 * ONE kernel instead of the library's three (input gradient, weight gradient on a single workgroup, bias gradient):
 *   gx[n, K] = gy[n, N] W[N, K] (gx may be NULL);  part[slice][k * K + j] = partial sums of gW = gy^T x;  part[slice][N * K + k] of gb.
 * N <= 16, K <= 256, K % 4 == 0 (allset_linear_narrow_supported); x / gx rows 16-byte aligned, W dense; n_slices from
 * allset_linear_narrow_slices(n); part_stride >= N * K + N; the caller sums the slices (allset_reduce_partials / _batched).
 * csrc/narrow_linear.hip; allset_amd/dense.py _Linear. */
int allset_linear_narrow_supported(int64_t N, int64_t K);
int allset_linear_narrow_slices(int64_t n, int64_t* n_slices);
int allset_linear_narrow_bwd(const float* gy, int64_t ldg, const float* x, int64_t ldx, const float* W, int64_t n, int64_t N, int64_t K,
                             float* gx, int64_t ldgx, float* part, int64_t part_stride, int64_t n_slices, void* stream);

/* ---- the FIRST Linear of a model: LayerNorm(raw features) -> Linear on an input that needs no gradient (reference models.py:473-476,
 * layers.py:571-579 with InputNorm; raw widths 1433 / 3703 fit none of the resident-weight kernels) ----
 * y = [x_hat | 1] [W * gamma | b + W beta]^T with x_hat the LayerNorm WITHOUT its affine part, and every parameter gradient from the
 * one product M = gy^T [x_hat | 1]: gW = gamma * M[:, :d] + beta (x) M[:, d], gb = M[:, d], ggamma_j = sum_k W[k,j] M[k,j],
 * gbeta_j = sum_k W[k,j] M[k,d] -- the [n, d] input gradient and the LayerNorm backward over it are never formed.  The two GEMMs
 * are the caller's (library).  csrc/input_linear.hip; allset_amd/dense.py _InputNormLinear. */
int64_t allset_input_linear_k(int64_t d);            /* leading dimension of [x_hat | 1 | 0...]: d + 1 rounded up to 16 elements */
int allset_input_linear_supported(int64_t d);         /* 1 <= d <= 4096 */
/* xh[r, :d] = LayerNorm_noaffine(dropout_{p_pre}(x[r, :])), xh[r, d] = 1, xh[r, d+1 : ldxh] = 0.  The dropout (counter hash of
 * (seed, r * d + c), as everywhere in this library) is applied BEFORE the statistics: models.py:473 `F.dropout(x, p=0.2)`. */
int allset_xhat_rows(const float* x, int64_t ldx, int64_t n, int64_t d, float eps, float p_pre, uint64_t seed,
                     const uint64_t* seed_base, float* xh, int64_t ldxh, void* stream);
/* Wp[k, :d] = W[k, :d] * gamma, Wp[k, d] = b[k] + sum_j W[k,j] beta[j] (b may be NULL), Wp[k, d+1 : ldwp] = 0. */
int allset_fold_ln_linear(const float* W, int64_t ldw, const float* gamma, const float* beta, const float* b, int64_t O, int64_t d,
                          float* Wp, int64_t ldwp, void* stream);
/* M [O, ldm >= d + 1] = gy^T [x_hat | 1]  ->  gW [O, d], gb [O] (may be NULL), ggamma [d], gbeta [d] (closed forms above). */
int allset_unfold_ln_linear(const float* M, int64_t ldm, const float* W, int64_t ldw, const float* gamma, const float* beta,
                            int64_t O, int64_t d, float* gW, int64_t ldgw, float* gb, float* ggamma, float* gbeta, void* stream);
int allset_fused_linear_bwd_all(const float* gy, int64_t ldg, const uint32_t* mask, float p_out, const float* W, const float* x,
                                int64_t ldx, const float* stats, const float* gamma, const float* beta, int relu_in, float p_in,
                                uint64_t seed_in, float* gx, int64_t ldgx, float* part_ln, float* part_w, float* part_b,
                                int64_t n_slices, int64_t n, int64_t O, int64_t I, const uint64_t* seed_base,
                                const float* acc_in, int64_t ldacc, int64_t part_stride, void* stream);
/* The same single pass for the plain Linear that carried four auxiliary output columns in the forward (allset_fused_linear_fwd's
 * aux_w / aux_out: PMA's folded attention logits, reference layers.py:126-131 `x_K = self.lin_K(x); alpha = (x_K * self.att_r).sum(-1)`
 * restated as x @ w_a^T):  gx = gy W + aux_g aux_w,  gW = gy^T x,  gb = colsum(gy),  gaux_w = aux_g^T x [4, I],  gaux_b = colsum(aux_g).
 * aux_g [n,4] and aux_w [4,I] dense fp32, 16-byte aligned.  part: [n_slices][part_stride] with sections gW [O*I] | gb [O] | gaux_w
 * [4*I] | gaux_b [4] (part_stride >= O*I + O + 4*I + 4; n_slices = allset_fused_linear_bwd_all_slices_for(n, O, I, 0)); the caller
 * sums over slices (allset_reduce_partials).  Built where _aux_supported(O, I) returns 1 (O = I = 128, default kernel family);
 * otherwise ALLSET_ERR_UNSUPPORTED and the caller keeps allset_fused_linear_bwd (aux_g / aux_w) + allset_wgrad. */
int allset_fused_linear_bwd_all_aux_supported(int64_t O, int64_t I);
int allset_fused_linear_bwd_all_aux(const float* gy, int64_t ldg, const float* W, const float* x, int64_t ldx, const float* aux_g,
                                    const float* aux_w, float* gx, int64_t ldgx, float* part, int64_t part_stride, int64_t n_slices,
                                    int64_t n, int64_t O, int64_t I, void* stream);

/* ---- the PMA tail's first rFF Linear, backward, with ln0's backward and the pooling's backward statistics in the same pass (ABI 12;
 * reference layers.py:153-157: out = ln0(pooled + att_r) feeds rFF's first Linear AND the residual add in front of ln1).  O = I = 128,
 * fp16x3 arithmetic, heads with 128 / heads = 4 x a power of two (allset_fused_linear_bwd_pma_tail_supported):
 *   gu = gy W + gres            (gres [n, 128]: the residual branch's gradient of out, i.e. ln1's input gradient)
 *   gx = the gradient of `pooled` through ln0's backward (x = pooled, colb = att_r flattened or NULL, stats / gamma / beta of ln0)
 *   gW = gy^T out, gb = colsum(gy), with out = ln0(pooled + colb) RECOMPUTED (the forward's saved copy is not read)
 *   part: [n_slices][part_stride] = gW [O*I] | gb [O] | dgamma [I] | dbeta [I] | dcolb [I]  (part_stride >= O*I + O + 3*I; n_slices =
 *         allset_fused_linear_bwd_all_slices_for(n, 128, 128, 0); the caller sums over slices)
 *   pma_stats[r, h] = {pma_m[r, h] + log(pma_l[r, h] + 1e-16)  (FLT_MAX for an empty target),  <pooled[r, h, :], gx[r, h, :]>}
 * -- what allset_fused_linear_bwd_all (acc_in = gres) followed by allset_ln_res_bwd_pma computed in two passes. */
int allset_fused_linear_bwd_pma_tail_supported(int64_t O, int64_t I, int64_t heads);
int allset_fused_linear_bwd_pma_tail(const float* gy, int64_t ldg, const float* W, const float* pooled, int64_t ldx, const float* colb,
                                     const float* stats, const float* gamma, const float* beta, const float* gres, int64_t ldgres,
                                     float* gx, int64_t ldgx, float* part, int64_t part_stride, int64_t n_slices, const float* pma_m,
                                     const float* pma_l, float* pma_stats, int64_t heads, int64_t n, int64_t O, int64_t I, void* stream);

/* ---- the PMA tail's SECOND rFF Linear, backward, with ln1's backward as its gy prologue (ABI 12).  y = dropout_p(relu_post?(
 * LayerNorm_{gamma2,beta2}(s))) with s = out + relu(z), z = relu_in?(x) W^T + b (allset_fused_linear_fwd_res_ln saved s, stats2 and the
 * 1-bit mask of z > 0).  Given gy = d y:  gs_out = d s (ln1's backward: the dropout's keep factors from (seed, seed_base) as in the
 * forward, the relu mask from the recomputed LayerNorm output);  gx = (gs under the mask) W through relu_in?;  gW, gb as
 * allset_fused_linear_bwd_all;  part: [n_slices][part_stride] = gW [O*I] | gb [O] | dgamma2 [O] | dbeta2 [O] (part_stride >= O*I + 3*O;
 * n_slices = allset_fused_linear_bwd_all_slices_for(n, 128, 128, 0)).  O = I = 128, fp16x3 arithmetic.  What allset_ln_res_bwd followed
 * by allset_fused_linear_bwd_all computed in two passes. */
int allset_fused_linear_bwd_ln_pro_supported(int64_t O, int64_t I);
int allset_fused_linear_bwd_ln_pro(const float* gy, int64_t ldg, const float* s, int64_t lds, const float* stats2, const float* gamma2,
                                   const float* beta2, int relu_post, float p, uint64_t seed, const uint64_t* seed_base,
                                   const uint32_t* mask, const float* W, const float* x, int64_t ldx, int relu_in, float* gs_out,
                                   int64_t ldgs, float* gx, int64_t ldgx, float* part, int64_t part_stride, int64_t n_slices, int64_t n,
                                   int64_t O, int64_t I, void* stream);

/* ---- the PMA tail folded into its two rFF Linears (ABI 11; reference layers.py:153-157: out = ln0(pooled + att_r);
 * out = ln1(out + relu(rFF(out)))), K = N = 128, fp16x3 arithmetic (allset_fused_linear_tail_supported):
 *   _ln_side : uo = LayerNorm(x + colb) (colb f32[K] or NULL; stats[r] = {mean, rstd}), y = relu_out?(uo W^T + bias) -- ln0 as the
 *              prologue of rFF's first Linear, its output written once as a side result (the residual branch and the backward use it);
 *   _res_ln  : z = relu_out?(relu_in?(x) W^T + bias), mask_out = 1 bit per element "z > 0" (mask layout above; may be NULL);
 *              s = res + z -> s_out; stats[r] = {mean, rstd} of s; y = dropout_{p_out}(relu_post?(LayerNorm(s))) -- the residual add
 *              and ln1 (with the relu -> dropout SetGNN puts behind the conv, models.py:475-481) as the epilogue of rFF's last Linear.
 * Backward: allset_ln_res_bwd on s_out (x = s_out, no colb / res) gives the gradient of s; the Linears' own backward is
 * allset_fused_linear_bwd_all as before.  The operand of _res_ln has no bound known in advance: its fp16 planes are scaled per ROW
 * (largest element to [2^13, 2^14)), undone per row in the epilogue; the plain forwards without a LayerNorm prologue use the same scheme
 * under ALLSET_ARITH_AUTO / _FP16X3 since this version. */
int allset_fused_linear_tail_supported(int64_t K, int64_t N);
int allset_fused_linear_fwd_ln_side(const float* x, int64_t ldx, const float* colb, const float* gamma, const float* beta, float eps,
                                    const float* W, const float* bias, int relu_out, float* y, int64_t ldy, float* uo, int64_t lduo,
                                    float* stats, int64_t n, int64_t K, int64_t N, void* stream);
int allset_fused_linear_fwd_res_ln(const float* x, int64_t ldx, int relu_in, const float* W, const float* bias, int relu_out,
                                   const float* res, int64_t ldres, const float* gamma, const float* beta, float eps, int relu_post,
                                   float p_out, uint64_t seed_out, const uint64_t* seed_base, float* y, int64_t ldy, float* s_out,
                                   int64_t lds, float* stats, uint32_t* mask_out, int64_t n, int64_t K, int64_t N, void* stream);

/* ---- choice of arithmetic for the fused Linear kernels (ABI 11) ------------------------------------------------------------------
 * The reference computes every Linear in IEEE fp32 (torch, layers.py:571-579).  gfx950's fp32 matrix rate is 1/16 of its bf16 / f16
 * rate, so these kernels EMULATE fp32 products on the 16-bit matrix pipe, in one of two ways:
 *   ALLSET_ARITH_BF16X6  "exact split": each fp32 operand = three bf16 planes (an exact decomposition for every finite fp32 value);
 *                        six of the nine plane products are accumulated in fp32, the dropped ones are <= 2^-23 relative per product.
 *                        NO dependence on the data's dynamic range: as accurate as a native fp32 MFMA for any input.  Built for every
 *                        shape the entries take.  The strict mode.
 *   ALLSET_ARITH_FP16X3  two fp16 planes per operand, three products: half the matrix work and 2/3 of the plane traffic.  fp16 has five
 *                        exponent bits, so operands are scaled by exact powers of two into its window: forward -- one scale per launch
 *                        (the LayerNorm output is bounded) and one per 32-column slice of W; backward -- gy per ROW, W per 32-column
 *                        slice, the recomputed input per row, with an online rescale of the weight-gradient accumulators.  Error per
 *                        product <= 2^-21 relative + 2^-38 x (largest |gy| of the ROW) x |u|: elements more than 2^17 below their row's
 *                        largest lose low bits.  Harmless for gx (a row sum dominated by the row's large elements) but visible in the
 *                        weight gradient when a whole COLUMN of gy sits > 2^17 below the other columns of the same rows: that column's
 *                        gW row then carries relative error 2^-(38 - k) at k binary orders below.  Built for K = N = 128 only; the
 *                        forward behind ALLSET_NORM_LAYER (one scale per launch) or without a norm (one scale per row), not in the
 *                        column-affine mode (allset_fused_linear_arith_supported).
 *   ALLSET_ARITH_AUTO    the library's choice -- a pure function of the SHAPE arguments, never of the data (no data pass, no sync):
 *                        FP16X3 wherever it is built, BF16X6 elsewhere.  AUTO therefore does NOT fall back on hostile dynamic range;
 *                        a caller whose gradient columns spread over more than 2^17 (not seen in any AllSet configuration: LayerNorm
 *                        and relu keep a layer's gradient columns within a few binary orders) asks for BF16X6.
 * The entries without an `arith` argument behave as AUTO.  Measured on MI355X at [1M,128] x [128,128] (profiles/r05_*): see DESIGN 6.1.
 * The two _ex entries are supersets of allset_fused_linear_fwd / _nm / _blocked and allset_fused_linear_bwd_all / _nm / _blocked:
 * block widths 0 = row-major, norm_mode as above, aux / acc_in NULL = none. */
#define ALLSET_ARITH_AUTO 0
#define ALLSET_ARITH_BF16X6 1
#define ALLSET_ARITH_FP16X3 2
int allset_fused_linear_arith_supported(int direction /* 0 forward, 1 one-pass backward */, int64_t K, int64_t N, int has_ln,
                                        int norm_mode, int arith);
int allset_fused_linear_fwd_ex(const float* x, int64_t ldx, int64_t x_block_cols, const float* gamma, const float* beta, float eps,
                               int norm_mode, int relu_in, float p_in, uint64_t seed_in, const float* W, const float* bias,
                               int relu_out, float p_out, uint64_t seed_out, float* y, int64_t ldy, int64_t y_block_cols,
                               float* stats, int64_t n, int64_t K, int64_t N, const uint64_t* seed_base, uint32_t* mask_out,
                               const float* aux_w, const float* aux_b, float* aux_out, int arith, void* stream);
int allset_fused_linear_bwd_all_ex(const float* gy, int64_t ldg, int64_t gy_block_cols, const uint32_t* mask, float p_out,
                                   const float* W, const float* x, int64_t ldx, int64_t x_block_cols, const float* stats,
                                   const float* gamma, const float* beta, int norm_mode, int relu_in, float p_in, uint64_t seed_in,
                                   float* gx, int64_t ldgx, int64_t gx_block_cols, float* part_ln, float* part_w, float* part_b,
                                   int64_t n_slices, int64_t n, int64_t O, int64_t I, const uint64_t* seed_base, const float* acc_in,
                                   int64_t ldacc, int64_t part_stride, int arith, void* stream);

/* Column-blocked operands (ABI 8).  An [n, C] operand with block width cb is stored [C / cb][n][cb] (its ld argument == cb):
 * exactly the send / receive buffer of an equal-split all-to-all that turns a row block of all C columns into all rows of a
 * C / P column slice (allset_amd/dist.py _rows_to_cols and back).  The fused Linear reading / writing that layout removes the
 * pack / unpack pass on either side of the exchange (no reference counterpart: the reference is single-device; the layer these
 * serve is reference layers.py:623-656 sharded by feature columns).  *_block_cols = 0: that operand is plain row-major.
 * cb must be a power of two with 4 <= cb <= C / 2.  Built where allset_fused_linear_blocked_supported(K, N) returns 1 (K = N = 128,
 * default kernel family); no auxiliary columns, no acc_in.  Everything else as in the un-blocked entry points. */
int allset_fused_linear_blocked_supported(int64_t K, int64_t N);
int allset_fused_linear_fwd_blocked(const float* x, int64_t ldx, int64_t x_block_cols, const float* gamma, const float* beta, float eps,
                                    int relu_in, float p_in, uint64_t seed_in, const float* W, const float* bias, int relu_out,
                                    float p_out, uint64_t seed_out, float* y, int64_t ldy, int64_t y_block_cols, float* stats,
                                    int64_t n, int64_t K, int64_t N, const uint64_t* seed_base, uint32_t* mask_out, void* stream);
int allset_fused_linear_bwd_all_blocked(const float* gy, int64_t ldg, int64_t gy_block_cols, const uint32_t* mask, float p_out,
                                        const float* W, const float* x, int64_t ldx, int64_t x_block_cols, const float* stats,
                                        const float* gamma, const float* beta, int relu_in, float p_in, uint64_t seed_in, float* gx,
                                        int64_t ldgx, int64_t gx_block_cols, float* part_ln, float* part_w, float* part_b,
                                        int64_t n_slices, int64_t n, int64_t O, int64_t I, const uint64_t* seed_base,
                                        int64_t part_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ALLSET_HIP_EXT_H */
