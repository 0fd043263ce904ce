/*
 * allset_hip.h -- the CORE C ABI of liballset_hip.so: AllSet's vertex<->hyperedge multiset aggregation
 * (HalfNLHconv / PMA propagate step) as hand-written HIP kernels for gfx950 (MI355X, CDNA4).
 *
 * What this boundary replaces.  The reference (jianhao2016/AllSet) is pure Python; its path reaches
 * native code only through the PyG / torch_scatter operator seam:
 *
 *   reference call site                         third-party op it lands in        entry point here
 *   ------------------------------------------  --------------------------------  -----------------------
 *   src/layers.py:633  self.propagate(x, norm)  index_select (PyG __lift__)   \
 *   src/layers.py:639  norm.view(-1,1) * x_j    mixed-dtype mul                 >  allset_segreduce_fwd
 *   src/layers.py:656  scatter(.., reduce=aggr) torch_scatter.scatter          /   (+ its autograd mirror:
 *                                                                                    same entry on the transposed
 *                                                                                    CSR, allset_segmax_bwd,
 *                                                                                    allset_sddmm_rowdot)
 *   src/layers.py:145  self.propagate(x_V, a)   index_select x2               \
 *   src/layers.py:173  F.leaky_relu             elementwise                     \
 *   src/layers.py:174  softmax(alpha, index)    torch_geometric.utils.softmax    >  allset_pma_fwd
 *   src/layers.py:177  x_j * alpha[..., None]   elementwise                     /   (+ allset_pma_bwd_stats,
 *   src/layers.py:194  scatter(.., 'add')       torch_scatter.scatter          /     allset_pma_bwd_src,
 *                                                                                    allset_pma_attention)
 *   src/models.py:453-456 edge_index re-basing + torch.stack of the reversed index,
 *   src/layers.py:174,656 index.max()+1 sizing (host syncs every forward)       ->  allset_csr_build (once)
 *
 *   The dense tail either side of the aggregation -- torch modules in the reference:
 *   src/layers.py:571-579 MLP.forward: norm -> (Linear -> ReLU -> norm -> dropout)* -> Linear, and the
 *   relu -> dropout SetGNN puts behind every conv (src/models.py:475-481)        ->  allset_fused_linear_fwd / _bwd,
 *                                                                                    allset_wgrad_fused (widths 64, 128);
 *                                                                                    allset_ln_*, allset_relu_dropout_*,
 *                                                                                    allset_wgrad (any width)
 *   src/layers.py:153-157 PMA tail: + att_r, ln0, ln1(z + relu(rFF(z)))         ->  allset_ln_res_fwd / _bwd
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.  Every function returns an int status:
 *    0 = ALLSET_OK, <0 = error (see enum); allset_last_error() gives a thread-local message.
 *    Nothing throws across the ABI.
 *  - All data pointers are DEVICE pointers borrowed for the duration of the call; the library
 *    allocates nothing and keeps no global mutable state (re-entrant, safe from the autograd thread).
 *    Scratch is passed in by the caller, sized by the matching *_workspace_bytes query.
 *  - Every launch goes to the caller's hipStream_t (passed as void*), is asynchronous and performs no
 *    hidden synchronisation.
 *  - Incidence is CSR: rowptr[n_rows+1] int32, col[nnz] int32; "row" = the TARGET of the aggregation
 *    (hyperedge for V->E, vertex for E->V), "col" = the SOURCE row gathered from.  Per-incidence
 *    arrays (w, p) are in CSR order; perm[] from allset_csr_build maps CSR position -> position in the
 *    caller's original [2,nnz] edge list.
 *  - Feature matrices are row-major with an explicit leading dimension in ELEMENTS (ld >= d).
 */
#ifndef ALLSET_HIP_H
#define ALLSET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALLSET_CORE_ABI_VERSION 1  /* The core surface: the 15 entry points of this file -- what a reference-side maintainer binds in place of the PyG / torch_scatter operator seam (table above).  FROZEN: version 1 == these signatures exactly as they stood in ALLSET_ABI_VERSION 10 of the former single header (tests/test_abi.py::test_core_surface_is_frozen keeps the text of every prototype).  It changes only by a new version number.  Everything else the library exports -- the dense tail, the bf16 regime, the exchange helpers, the optimizer / loss kernels: this package's own plumbing -- lives in allset_hip_ext.h under its own version, ALLSET_ABI_VERSION. */

enum allset_status {
  ALLSET_OK = 0,
  ALLSET_ERR_INVALID_ARGUMENT = -1, /* null pointer, negative size, bad enum, misaligned ld     */
  ALLSET_ERR_HIP = -2,              /* a HIP runtime call / kernel launch failed                  */
  ALLSET_ERR_UNSUPPORTED = -3,      /* valid request this build has no kernel for (e.g. a dtype)  */
  ALLSET_ERR_WORKSPACE = -4         /* workspace missing or smaller than *_workspace_bytes says   */
};

enum allset_reduce { ALLSET_SUM = 0, ALLSET_MEAN = 1, ALLSET_MAX = 2, ALLSET_MIN = 3 };
enum allset_dtype { ALLSET_F32 = 0, ALLSET_BF16 = 1 };

/* Version of the core surface of the loaded library (== ALLSET_CORE_ABI_VERSION it was built with). */
int allset_core_version(void);
/* Version of the WHOLE library (== ALLSET_ABI_VERSION of allset_hip_ext.h it was built with; >= 10). */
int allset_version(void);
/* Thread-local, NUL-terminated description of the last error returned on this thread ("" if none). */
const char* allset_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Incidence structure.  Replaces the per-forward index handling of reference models.py:453-456 and
 * the implicit `index.max()+1` host syncs of layers.py:174,656: sizes are fixed once, here.
 *
 * Stable counting sort of the nnz incidences by (row_ids[i] - row_base): equal rows keep the
 * caller's order, so a sequential reduction over a CSR row visits incidences in the same order
 * as the reference's CPU scatter_add_.
 *   row_ids, col_ids : int64[nnz] device (the two rows of the reference's edge_index)
 *   row_base/col_base: subtracted from the ids (models.py:453 `edge_index[1] -= cidx`, not in place)
 *   rowptr int32[n_rows+1], col int32[nnz], perm int32[nnz] : outputs
 * Ids outside [base, base+n) are a caller error the kernel cannot report without a sync; the host
 * wrapper validates ranges once.
 * ------------------------------------------------------------------------------------------- */
int allset_csr_build_workspace_bytes(int64_t nnz, int64_t n_rows, size_t* bytes);
int allset_csr_build(const int64_t* row_ids, const int64_t* col_ids, int64_t nnz,
                     int64_t row_base, int64_t col_base, int64_t n_rows,
                     int32_t* rowptr, int32_t* col, int32_t* perm,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Deep Sets aggregation (reference layers.py:633,638-656):
 *   out[t, :] = reduce_{j in [rowptr[t], rowptr[t+1])}  w[j] * x[col[j], :]
 * reduce = SUM | MEAN (sum / max(count,1)) | MAX | MIN; rows with no incidence produce 0 in every
 * mode (torch_scatter semantics).  w may be NULL (all ones; the reference default `norm`).
 * argext (int32[n_t*d], MAX/MIN only, may be NULL): CSR position of the winning incidence per
 * (row, column), -1 for empty rows; ties go to the smallest CSR position.
 * The backward of SUM/MEAN w.r.t. x is this same entry point on the transposed CSR.
 * ------------------------------------------------------------------------------------------- */
int allset_segreduce_fwd(int reduce, int dtype,
                         const int32_t* rowptr, const int32_t* col, const float* w,
                         const void* x, int64_t ldx, void* out, int64_t ldo, int32_t* argext,
                         int64_t n_t, int64_t n_s, int64_t d, void* stream);

/* Same, with an explicit kernel choice.  variant: 0 = auto (uses nnz, the number of incidences -- known to the caller,
 * unknowable to the library without a sync: mean degree nnz/n_t < 6 selects the short-row kernel), 1 = one wavefront
 * per row, 2 = short-row kernel (several consecutive rows per half-wave walked as one incidence stream; sum/mean only,
 * 16-byte aligned rows, d <= 256 f32 / 512 bf16; otherwise ALLSET_ERR_UNSUPPORTED).  Results are identical up to fp32
 * summation order (the short-row kernel sums a row strictly in CSR order).
 * row_order (int32[n_t], may be NULL; one-wavefront-per-row kernels only): a permutation giving the order in which rows
 * are PROCESSED -- for skewed degree distributions the long rows of each XCD's row range are listed first so they do not
 * set the tail of the launch (a 4096-incidence row occupies one wave for ~0.3 ms).  Results do not depend on it. */
int allset_segreduce_fwd_ex(int reduce, int dtype, int variant, int64_t nnz, const int32_t* row_order,
                            const int32_t* rowptr, const int32_t* col, const float* w,
                            const void* x, int64_t ldx, void* out, int64_t ldo, int32_t* argext,
                            int64_t n_t, int64_t n_s, int64_t d, void* stream);

/* Backward of MAX/MIN w.r.t. x, deterministic (no atomics), on the TRANSPOSED CSR (rows = sources):
 *   gx[s, c] = sum_{j in T-row s} [argext[colT[j], c] == posT[j]] * wT[j] * gout[colT[j], c]
 * posT[j] = forward-CSR position of T-incidence j.  wT may be NULL. */
int allset_segmax_bwd(const int32_t* rowptrT, const int32_t* colT, const int32_t* posT, const float* wT,
                      const int32_t* argext, const float* gout, int64_t ldg, float* gx, int64_t ldx,
                      int64_t n_s, int64_t n_t, int64_t d, void* stream);

/* Gradient w.r.t. the per-incidence weight (reference models.py:451-452 LearnMask path):
 *   gw[j] = scale(t) * sum_c m(j,c) * x[col[j], c] * gout[t, c]        for j in CSR row t
 * SUM: scale=1, m=1.  MEAN: scale=1/max(count,1).  MAX/MIN: m = [argext[t,c] == j]. */
int allset_sddmm_rowdot(int reduce, const int32_t* rowptr, const int32_t* col,
                        const float* x, int64_t ldx, const float* gout, int64_t ldg,
                        const int32_t* argext, float* gw,
                        int64_t n_t, int64_t n_s, int64_t d, void* stream);

/* ---------------------------------------------------------------------------------------------
 * PMA aggregation (reference layers.py:145,168-194), heads H, channels per head C, d = H*C:
 *   a_j      = leaky_relu(alpha[col[j], h], slope)
 *   p_j      = exp(a_j - m[t,h]) / (l[t,h] + 1e-16),  m = max_j a_j,  l = sum_j exp(a_j - m)
 *   out[t,h,:] = sum_j p_j * V[col[j], h, :]          (empty row -> 0; the caller adds att_r)
 * alpha: f32[n_s*H] (pre-activation logits), V: [n_s, ldv], out: [n_t, ldo]; m, l: f32[n_t*H]
 * outputs saved for the backward (empty row: m = 0, l = 0).
 * ------------------------------------------------------------------------------------------- */
int allset_pma_fwd(int dtype, const int32_t* rowptr, const int32_t* col,
                   const float* alpha, const void* V, int64_t ldv, float slope,
                   void* out, int64_t ldo, float* m, float* l,
                   int64_t n_t, int64_t n_s, int64_t H, int64_t C, void* stream);

/* Same with an explicit kernel choice (see allset_segreduce_fwd_ex): variant 0 auto by nnz / n_t, 1 one wavefront per
 * row, 2 short-row kernel (16-byte aligned rows, C a multiple of the packet width, H*C <= 256 f32 / 512 bf16). */
int allset_pma_fwd_ex(int dtype, int variant, int64_t nnz, const int32_t* row_order, const int32_t* rowptr, const int32_t* col,
                      const float* alpha, const void* V, int64_t ldv, float slope,
                      void* out, int64_t ldo, float* m, float* l,
                      int64_t n_t, int64_t n_s, int64_t H, int64_t C, void* stream);

/* Attention weights p[j,h] in CSR order (reference PMA.forward(..., return_attention_weights=True),
 * layers.py:159-162).  p: f32[nnz*H]. */
int allset_pma_attention(const int32_t* rowptr, const int32_t* col, const float* alpha,
                         const float* m, const float* l, float slope, float* p,
                         int64_t n_t, int64_t H, void* stream);

/* Backward, step 1 (dense, target-major): stats[t,h] = { M, delta } with M = m + log(l + 1e-16) (so that
 * p_j = exp(a_j - M)) and delta[t,h] = <out[t,h,:], gout[t,h,:]>.   stats: f32[n_t*H*2], 8-byte aligned. */
int allset_pma_bwd_stats(int dtype, const void* out, int64_t ldo, const void* gout, int64_t ldg,
                         const float* m, const float* l, float* stats,
                         int64_t n_t, int64_t H, int64_t C, void* stream);

/* Backward, step 2 (one gather pass, source-major, on the TRANSPOSED CSR; no [nnz,*] temporaries):
 *   p_j        = exp(lrelu(alpha[s,h]) - M[t_j,h])
 *   gV[s,h,:]  = sum_j p_j * gout[t_j,h,:]
 *   galpha[s,h]= lrelu'(alpha[s,h]) * sum_j p_j * (<V[s,h,:], gout[t_j,h,:]> - delta[t_j,h])
 * with lrelu'(x) = 1 if x > 0 else slope (PyTorch's convention at 0). */
int allset_pma_bwd_src(int dtype, const int32_t* rowptrT, const int32_t* colT,
                       const float* alpha, const void* V, int64_t ldv,
                       const void* gout, int64_t ldg, const float* stats, float slope,
                       void* gV, int64_t ldgv, float* galpha,
                       int64_t n_s, int64_t n_t, int64_t H, int64_t C, void* stream);

/* Same with an explicit kernel choice (variant as in allset_pma_fwd_ex; nnz / n_s decides in auto mode). */
int allset_pma_bwd_src_ex(int dtype, int variant, int64_t nnz, const int32_t* row_order, const int32_t* rowptrT, const int32_t* colT,
                          const float* alpha, const void* V, int64_t ldv,
                          const void* gout, int64_t ldg, const float* stats, float slope,
                          void* gV, int64_t ldgv, float* galpha,
                          int64_t n_s, int64_t n_t, int64_t H, int64_t C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ALLSET_HIP_H */
