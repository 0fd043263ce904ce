"""ctypes binding of ``liballset_hip.so`` (C ABI: ``include/allset_hip.h`` = the core aggregation surface, ``include/allset_hip_ext.h`` =
the dense tail and the rest of this package's plumbing).

There is deliberately no fallback: if the shared library is missing or a call fails, this raises.
``import torch`` must precede the ``CDLL`` so that the library's ``libamdhip64.so.7`` dependency
binds to the HIP runtime torch has already mapped (one runtime per process: device pointers and
streams are shared with torch).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p, POINTER

import torch  # noqa: F401  (must be imported before the CDLL below)

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liballset_hip.so")
ABI_VERSION = 15
CORE_ABI_VERSION = 1

SUM, MEAN, MAX, MIN = 0, 1, 2, 3
F32, BF16 = 0, 1
REDUCE_AS_TREE = 0x100
ARITH_AUTO, ARITH_BF16X6, ARITH_FP16X3 = 0, 1, 2          # include/allset_hip_ext.h ALLSET_ARITH_*
REDUCE_CODES = {"add": SUM, "sum": SUM, "mean": MEAN, "max": MAX, "min": MIN}

# name -> argtypes, in the order of include/allset_hip.h.  Every function returns int except
# allset_last_error.
_P = c_void_p
SIGNATURES = {
    "allset_version": [],
    "allset_core_version": [],
    "allset_csr_build_workspace_bytes": [c_int64, c_int64, POINTER(c_size_t)],
    "allset_csr_build": [_P, _P, c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P, c_size_t, _P],
    "allset_segreduce_fwd": [c_int, c_int, _P, _P, _P, _P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int64, _P],
    "allset_segreduce_fwd_ex": [c_int, c_int, c_int, c_int64, _P, _P, _P, _P, _P, c_int64, _P, c_int64, _P, c_int64, c_int64,
                                c_int64, _P],
    "allset_segmax_bwd": [_P, _P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int64, c_int64, _P],
    "allset_sddmm_rowdot": [c_int, _P, _P, _P, c_int64, _P, c_int64, _P, _P, c_int64, c_int64, c_int64, _P],
    "allset_pma_fwd": [c_int, _P, _P, _P, _P, c_int64, c_float, _P, c_int64, _P, _P, c_int64, c_int64, c_int64, c_int64, _P],
    "allset_pma_fwd_ex": [c_int, c_int, c_int64, _P, _P, _P, _P, _P, c_int64, c_float, _P, c_int64, _P, _P, c_int64, c_int64,
                          c_int64, c_int64, _P],
    "allset_pma_fwd_ld": [c_int, c_int, c_int64, _P, _P, _P, _P, c_int64, _P, c_int64, c_float, _P, c_int64, _P, _P, c_int64,
                          c_int64, c_int64, c_int64, _P],
    "allset_pma_bwd_stats_ld": [c_int, _P, c_int64, _P, c_int64, _P, _P, _P, c_int64, c_int64, c_int64, c_int64, _P],
    "allset_pma_bwd_src_ld": [c_int, c_int, c_int64, _P, _P, _P, _P, _P, c_int64, _P, c_int64, _P, c_int64, c_float, _P, c_int64,
                              _P, c_int64, c_int64, c_int64, c_int64, _P],
    "allset_pma_attention": [_P, _P, _P, _P, _P, c_float, _P, c_int64, c_int64, _P],
    "allset_pma_bwd_stats": [c_int, _P, c_int64, _P, c_int64, _P, _P, _P, c_int64, c_int64, c_int64, _P],
    "allset_pma_bwd_src": [c_int, _P, _P, _P, _P, c_int64, _P, c_int64, _P, c_float, _P, c_int64, _P,
                           c_int64, c_int64, c_int64, c_int64, _P],
    "allset_pma_bwd_src_ex": [c_int, c_int, c_int64, _P, _P, _P, _P, _P, c_int64, _P, c_int64, _P, c_float, _P, c_int64, _P,
                              c_int64, c_int64, c_int64, c_int64, _P],
    "allset_ln_fwd": [_P, c_int64, _P, _P, c_float, c_int, c_float, c_uint64, _P, c_int64, _P, c_int64, c_int64, _P, _P],
    "allset_ln_bwd_partials": [c_int64, c_int64, POINTER(c_int64)],
    "allset_ln_bwd": [_P, c_int64, _P, c_int64, _P, _P, c_int, c_float, c_uint64, _P, c_int64, _P, c_int64,
                      c_int64, c_int64, _P, _P],
    "allset_relu_dropout_fwd": [_P, c_float, c_uint64, _P, c_int64, _P, _P],
    "allset_relu_dropout_bwd": [_P, _P, c_float, _P, c_int64, _P],
    "allset_wgrad_slices": [c_int64, c_int64, c_int64, POINTER(c_int64)],
    "allset_wgrad": [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int64, c_int64, c_int64, _P],
    "allset_wgrad_bf16_slices": [c_int64, c_int64, c_int64, POINTER(c_int64)],
    "allset_wgrad_bf16": [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int64, c_int64, c_int64, _P],
    "allset_reduce_partials": [_P, c_int64, c_int64, _P, _P, _P],
    "allset_wgrad_fused_ex": [_P, c_int64, _P, c_int64, c_float, _P, c_int64, _P, _P, _P, c_int, c_float, c_uint64, _P, c_int64, c_int,
                              c_int64, c_int64, c_int64, c_int64, _P, _P, _P],
    "allset_wgrad_f16x3_supported": [c_int64, c_int64],
    "allset_wgrad_f16x3_slices": [c_int64, c_int64, c_int64, POINTER(c_int64)],
    "allset_wgrad_f16x3": [_P, c_int64, _P, c_float, _P, c_int64, _P, _P, _P, c_int, c_float, c_uint64, _P, c_int64, c_int,
                           c_int64, c_int64, c_int64, c_int64, _P, _P],
    "allset_wgrad_fused": [_P, c_int64, _P, c_int64, c_float, _P, c_int64, _P, _P, _P, c_int, c_float, c_uint64, _P, _P,
                           c_int64, c_int64, c_int64, c_int64, _P, _P, _P],
    "allset_fused_linear_supported": [c_int64, c_int64],
    "allset_fused_linear_bwd_partials": [c_int64, POINTER(c_int64)],
    "allset_fused_linear_bwd": [_P, c_int64, _P, c_int64, c_float, _P, _P, c_int64, _P, _P, c_int, c_float, c_uint64, _P,
                                c_int64, _P, c_int64, c_int64, c_int64, c_int64, _P, _P, _P, c_int64, _P, _P, _P],
    "allset_fused_linear_fwd": [_P, c_int64, _P, _P, c_float, c_int, c_float, c_uint64, _P, _P, c_int, c_float, c_uint64,
                                _P, c_int64, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P],
    "allset_fused_linear_mask_words": [c_int64, c_int64],
    "allset_fused_linear_bwd_all_supported": [c_int64, c_int64, c_int, c_int, c_int, c_int, c_int],
    "allset_fused_linear_bwd_all_slices": [c_int64, POINTER(c_int64)],
    "allset_fused_linear_bwd_all_slices_for": [c_int64, c_int64, c_int64, c_int, POINTER(c_int64)],
    "allset_fused_linear_bwd_all": [_P, c_int64, _P, c_float, _P, _P, c_int64, _P, _P, _P, c_int, c_float, c_uint64, _P, c_int64,
                                    _P, _P, _P, c_int64, c_int64, c_int64, c_int64, _P, _P, c_int64, c_int64, _P],
    "allset_fused_linear_bwd_all_nm": [_P, c_int64, _P, c_float, _P, _P, c_int64, _P, _P, _P, c_int, c_int, c_float, c_uint64, _P, c_int64,
                                       _P, _P, _P, c_int64, c_int64, c_int64, c_int64, _P, c_int64, _P],
    "allset_fused_linear_fwd_nm": [_P, c_int64, _P, _P, c_float, c_int, c_int, c_float, c_uint64, _P, _P, c_int, c_float, c_uint64, _P,
                                   c_int64, _P, c_int64, c_int64, c_int64, _P, _P, _P],
    "allset_col_moments_supported": [c_int64],
    "allset_col_moments_slices": [c_int64, POINTER(c_int64)],
    "allset_col_moments": [_P, c_int64, c_int64, c_int64, c_int, _P, _P, c_int64, _P],
    "allset_col_moments2": [_P, c_int64, c_int64, c_int64, c_int, _P, c_int64, _P],
    "allset_col_affine_add": [_P, c_int64, _P, c_int64, _P, _P, c_int, c_int64, c_int64, _P],
    "allset_reduce_partials_batch_max": [],
    "allset_reduce_partials_batchable": [c_int64, c_int64],
    "allset_reduce_partials_batched": [_P, _P, _P, _P, _P, c_int64, _P],
    "allset_reduce_partials_batched_ex": [_P, _P, _P, _P, _P, c_int64, _P, _P, c_int64, _P],
    "allset_reduce_partials_batched_ex2": [_P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, c_int64, _P],
    "allset_reduce_partials_is_tree": [c_int64, c_int64],
    "allset_reduce_partials_batch_max_counters": [],
    "allset_sparse_ln_linear_supported": [c_int64],
    "allset_sparse_ln_linear_slices": [],
    "allset_fold_ln_linear_t": [_P, c_int64, _P, _P, _P, c_int64, c_int64, _P, _P],
    "allset_sparse_ln_linear_fwd": [_P, _P, _P, c_int64, c_int64, _P, c_int64, c_float, c_float, c_uint64, _P, _P, c_int64, _P, _P, _P],
    "allset_sparse_ln_linear_bwd": [_P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int64, c_int64, _P, c_int64, _P, _P],
    "allset_unfold_ln_linear_ex": [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int64, _P, c_int64, _P, _P, _P, _P, c_int64, _P],
    "allset_sparse_linear_supported": [c_int64, c_int64],
    "allset_sparse_linear_pitch": [c_int64, c_int64],
    "allset_sparse_linear_wt": [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, c_int64, _P, _P],
    "allset_sparse_linear_fwd": [_P, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, c_float, c_uint64, _P, _P, c_int64, _P, _P, _P],
    "allset_sparse_linear_bwd": [_P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, _P, _P, _P],
    "allset_linear_narrow_supported": [c_int64, c_int64],
    "allset_linear_narrow_slices": [c_int64, POINTER(c_int64)],
    "allset_linear_narrow_bwd": [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, c_int64, _P],
    "allset_input_linear_k": [c_int64],
    "allset_input_linear_supported": [c_int64],
    "allset_xhat_rows": [_P, c_int64, c_int64, c_int64, c_float, c_float, c_uint64, _P, _P, c_int64, _P],
    "allset_fold_ln_linear": [_P, c_int64, _P, _P, _P, c_int64, c_int64, _P, c_int64, _P],
    "allset_unfold_ln_linear": [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int64, _P, c_int64, _P, _P, _P, _P],
    "allset_fused_linear_bwd_all_aux_supported": [c_int64, c_int64],
    "allset_fused_linear_bwd_all_aux": [_P, c_int64, _P, _P, c_int64, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int64, c_int64,
                                        c_int64, _P],
    "allset_fused_linear_bwd_pma_tail_supported": [c_int64, c_int64, c_int64],
    "allset_fused_linear_bwd_pma_tail": [_P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, _P, c_int64, _P, c_int64, _P, c_int64, c_int64, _P, _P, _P,
                                         c_int64, c_int64, c_int64, c_int64, _P],
    "allset_fused_linear_bwd_ln_pro_supported": [c_int64, c_int64],
    "allset_fused_linear_bwd_ln_pro": [_P, c_int64, _P, c_int64, _P, _P, _P, c_int, c_float, c_uint64, _P, _P, _P, _P, c_int64, c_int, _P, c_int64,
                                       _P, c_int64, _P, c_int64, c_int64, c_int64, c_int64, c_int64, _P],
    "allset_fused_linear_blocked_supported": [c_int64, c_int64],
    "allset_fused_linear_tail_supported": [c_int64, c_int64],
    "allset_fused_linear_fwd_ln_side": [_P, c_int64, _P, _P, _P, c_float, _P, _P, c_int, _P, c_int64, _P, c_int64, _P, c_int64, c_int64,
                                        c_int64, _P],
    "allset_fused_linear_fwd_res_ln": [_P, c_int64, c_int, _P, _P, c_int, _P, c_int64, _P, _P, c_float, c_int, c_float, c_uint64, _P, _P,
                                       c_int64, _P, c_int64, _P, _P, c_int64, c_int64, c_int64, _P],
    "allset_fused_linear_arith_supported": [c_int, c_int64, c_int64, c_int, c_int, c_int],
    "allset_fused_linear_fwd_ex": [_P, c_int64, c_int64, _P, _P, c_float, c_int, c_int, c_float, c_uint64, _P, _P, c_int, c_float, c_uint64,
                                   _P, c_int64, c_int64, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, c_int, _P],
    "allset_fused_linear_bwd_all_ex": [_P, c_int64, c_int64, _P, c_float, _P, _P, c_int64, c_int64, _P, _P, _P, c_int, c_int, c_float,
                                       c_uint64, _P, c_int64, c_int64, _P, _P, _P, c_int64, c_int64, c_int64, c_int64, _P, _P, c_int64,
                                       c_int64, c_int, _P],
    "allset_fused_linear_fwd_blocked": [_P, c_int64, c_int64, _P, _P, c_float, c_int, c_float, c_uint64, _P, _P, c_int, c_float, c_uint64,
                                        _P, c_int64, c_int64, _P, c_int64, c_int64, c_int64, _P, _P, _P],
    "allset_fused_linear_bwd_all_blocked": [_P, c_int64, c_int64, _P, c_float, _P, _P, c_int64, c_int64, _P, _P, _P, c_int, c_float,
                                            c_uint64, _P, c_int64, c_int64, _P, _P, _P, c_int64, c_int64, c_int64, c_int64, _P, c_int64,
                                            _P],
    "allset_gemm_x6_supported": [c_int64, c_int64],
    "allset_gemm_x6_plane_bytes": [c_int64, c_int64],
    "allset_gemm_x6_planes": [_P, c_int64, c_int, _P, c_int64, c_int64, _P],
    "allset_row_stats": [_P, c_int64, c_int, c_float, _P, c_int64, c_int64, _P],
    "allset_gemm_x6": [_P, c_int64, _P, c_int64, c_float, c_int, _P, _P, _P, c_float, c_uint64, _P, _P, c_int, c_float, c_uint64,
                       _P, c_int64, c_int64, c_int64, c_int64, _P, _P],
    "allset_gemm_x6_lnb_partials": [c_int64],
    "allset_gemm_wide": [c_int, _P, c_int64, _P, c_int64, _P, c_float, c_int, _P, _P, _P, c_float, c_uint64, _P, _P, c_int, c_float, c_uint64,
                         _P, _P, c_float, c_int, _P, c_int64, c_int64, c_int64, c_int64, _P, _P],
    "allset_gemm_wide_sgn_supported": [c_int, c_int64, c_int64],
    "allset_gemm_wide_sgn": [c_int, _P, c_int64, _P, c_int64, _P, c_float, _P, _P, c_int64, _P, c_int64, c_int64, c_int64, c_int64, _P, _P],
    "allset_gemm_wide_lnb": [c_int, _P, c_int64, _P, c_int64, _P, c_float, _P, _P, c_int64, _P, _P, c_int, c_float, c_uint64, _P, c_int64,
                             _P, c_int64, c_int64, c_int64, c_int64, _P, _P],
    "allset_gemm_f16x3_plane_bytes": [c_int64, c_int64],
    "allset_gemm_f16x3_planes": [_P, c_int64, c_int, _P, c_int64, c_int64, _P],
    "allset_gemm_f16x3_planes_batch_max": [],
    "allset_gemm_f16x3_planes_batched": [_P, _P, _P, _P, _P, _P, c_int64, _P],
    "allset_gemm_f16x3": [_P, c_int64, _P, c_int64, c_float, c_int, _P, _P, _P, c_float, c_uint64, _P, _P, c_int, c_float, c_uint64,
                          _P, c_int64, c_int64, c_int64, c_int64, _P, _P],
    "allset_gemm_f16x3_lnb": [_P, c_int64, _P, c_int64, c_float, _P, _P, c_int64, _P, _P, c_int, c_float, c_uint64, _P, c_int64, _P,
                              c_int64, c_int64, c_int64, c_int64, _P, _P],
    "allset_gemm_x6_lnb": [_P, c_int64, _P, c_int64, c_float, _P, _P, c_int64, _P, _P, c_int, c_float, c_uint64, _P, c_int64, _P,
                           c_int64, c_int64, c_int64, c_int64, _P, _P],
    "allset_block_transpose": [_P, _P, c_int64, c_int64, c_int64, c_int64, c_int, _P],
    "allset_pma_merge_pack": [_P, c_int64, _P, _P, _P, _P, c_int64, c_int64, c_int64, c_int64, _P],
    "allset_ln_bf16_supported": [c_int64],
    "allset_ln_fwd_bf16": [_P, c_int64, _P, _P, c_float, c_int, c_float, c_uint64, _P, c_int64, _P, c_int64, c_int64, _P, _P],
    "allset_ln_bwd_bf16_partials": [c_int64, c_int64, POINTER(c_int64)],
    "allset_ln_bwd_bf16": [_P, c_int64, _P, c_int64, _P, _P, c_int, c_float, c_uint64, _P, c_int64, _P, c_int64,
                           c_int64, c_int64, _P, _P],
    "allset_ln_res_fwd_bf16": [_P, c_int64, _P, _P, c_int64, _P, _P, c_float, c_int, c_float, c_uint64, _P, c_int64, _P, c_int64,
                               c_int64, _P, _P],
    "allset_ln_res_bwd_bf16": [_P, c_int64, _P, c_int64, _P, _P, c_int64, _P, _P, _P, c_int, c_float, c_uint64, _P, c_int64, _P,
                               c_int64, c_int64, c_int64, _P, _P],
    "allset_ln_res_supported": [c_int64],
    "allset_ln_res_fwd": [_P, c_int64, _P, _P, c_int64, _P, _P, c_float, c_int, c_float, c_uint64, _P, c_int64, _P, c_int64,
                          c_int64, _P, _P],
    "allset_ln_res_bwd_partials": [c_int64, c_int64, POINTER(c_int64)],
    "allset_wgrad_bf16_ex": [_P, c_int64, _P, c_int64, _P, c_int64, c_int, c_int64, c_int64, c_int64, c_int64, _P],
    "allset_reduce_partials_ex": [_P, c_int64, c_int64, c_int64, _P, c_int, _P, _P],
    "allset_adam_max_tensors": [],
    "allset_adam_step_dtype": [c_int, _P, _P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, _P],
    "allset_adam_step": [_P, _P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, _P],
    "allset_pma_fold_fwd": [_P, _P, _P, _P, _P, c_int64, c_int64, c_int64, _P],
    "allset_pma_fold_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int64, _P],
    "allset_split_metrics": [_P, c_int64, _P, _P, _P, c_int64, c_int64, c_int64, _P],
    "allset_nll_partials": [c_int64, POINTER(c_int64)],
    "allset_nll_logsoftmax_fwd": [_P, c_int64, _P, _P, c_float, _P, c_int64, c_int64, c_int64, _P],
    "allset_nll_logsoftmax_fwd_total": [_P, c_int64, _P, _P, c_float, _P, c_int64, _P, _P, c_int64, c_int64, _P],
    "allset_nll_logsoftmax_bwd": [_P, c_int64, _P, _P, c_float, _P, _P, c_int64, c_int64, c_int64, _P],
    "allset_linear_bf16_supported": [c_int64, c_int64],
    "allset_linear_bf16_fwd": [_P, c_int64, _P, _P, c_int, _P, _P, _P, _P, c_int64, c_int64, c_int64, c_int64, _P],
    "allset_linear_bf16_bwd": [_P, c_int64, _P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int64,
                               c_int64, _P],
    "allset_linear_bf16_mask_pitch": [c_int64],
    "allset_linear_bf16_fwd_mask": [_P, c_int64, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int64, _P],
    "allset_linear_bf16_bwd_bits": [_P, c_int64, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int64, c_int64, _P],
    "allset_wgrad_bf16_ex2_supported": [c_int64, c_int64, c_int, c_int],
    "allset_wgrad_bf16_ex2": [_P, c_int64, _P, _P, _P, c_int64, _P, c_int64, c_int, c_int64, c_int64, c_int64, c_int64, _P],
    "allset_pma_fold_fwd_bf16": [_P, _P, _P, _P, _P, c_int64, c_int64, c_int64, _P],
    "allset_pma_fold_bwd_bf16": [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int64, _P],
    "allset_ln_res_bwd_pma_bf16_supported": [c_int64, c_int64],
    "allset_ln_res_bwd_pma_bf16": [_P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int64, _P, _P, _P,
                                   c_int64, _P],
    "allset_ln_res_bwd_pma_supported": [c_int64, c_int64],
    "allset_ln_res_bwd_pma": [_P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int64, _P, _P, _P,
                              c_int64, _P],
    "allset_ln_res_bwd": [_P, c_int64, _P, c_int64, _P, _P, c_int64, _P, _P, _P, c_int, c_float, c_uint64, _P, c_int64, _P,
                          c_int64, c_int64, c_int64, _P, _P],
}
EXPORTED_SYMBOLS = sorted(list(SIGNATURES) + ["allset_last_error"])


class AllSetHipError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Load (once) and return the library; raises if it is absent or has the wrong ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AllSetHipError(
            f"{LIB_PATH} not found: build it with `python -m allset_amd.build` "
            "(allset_amd has no CPU / eager fallback for the aggregation path)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.allset_fused_linear_mask_words.restype = c_int64
    lib.allset_gemm_x6_plane_bytes.restype = c_int64
    lib.allset_gemm_f16x3_plane_bytes.restype = c_int64
    lib.allset_gemm_x6_lnb_partials.restype = c_int64
    lib.allset_input_linear_k.restype = c_int64
    lib.allset_linear_bf16_mask_pitch.restype = c_int64
    lib.allset_sparse_linear_pitch.restype = c_int64
    lib.allset_last_error.argtypes = []
    lib.allset_last_error.restype = c_char_p
    got = lib.allset_version()
    if got != ABI_VERSION:
        raise AllSetHipError(f"liballset_hip.so ABI version {got}, python binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().allset_last_error().decode("utf-8", "replace")
        raise AllSetHipError(f"{what} failed with status {rc}: {msg}")


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


class _NullContext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_CONTEXT = _NullContext()


def on_device(device):
    """``torch.cuda.device(device)`` only when ``device`` is not already current (one process per GPU: it always is, and the
    context manager costs ~3 us of Python per launch -- a tenth of an eager dataset-scale step)."""
    if device is None or torch.cuda.current_device() == device.index:
        return _NULL_CONTEXT
    return torch.cuda.device(device)


def stream_of(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_device(*tensors) -> torch.device:
    """All tensors must live on one ROCm device; CPU tensors are refused (no fallback)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise AllSetHipError(
                "allset_amd aggregation kernels need ROCm device tensors; got a CPU tensor "
                "(there is no CPU fallback -- the CPU restatement lives in oracle/ and is test-only)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise AllSetHipError(f"tensors on different devices: {dev} vs {t.device}")
    return dev
