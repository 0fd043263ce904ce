"""Differentiable aggregation operators backed by the HIP kernels (``torch.autograd.Function``
wrappers over ``ops.py``).  These two functions are the whole "propagate" step of the reference:

* ``deepsets_aggregate``  == ``HalfNLHconv.propagate`` -> ``message`` -> ``aggregate``  (layers.py:633-656)
* ``pma_aggregate``       == ``PMA.propagate`` -> ``message`` -> ``aggregate``           (layers.py:145,168-194)

Backward passes are kernels too (same segreduce kernel on the transposed CSR; one-gather-pass PMA
backward) -- no autograd graph over per-incidence temporaries exists.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch.autograd.function import once_differentiable

from . import _lib, ops
from ._lib import MAX, MEAN, MIN, REDUCE_CODES, SUM
from .incidence import Incidence

Tensor = torch.Tensor


def _variant(csr, kind: str, n_rows: int, x: Tensor, heads: int = 1) -> int:
    """Short-row kernel (2) only where it is built: 16-byte packets covering the whole row in one chunk."""
    es = x.element_size()
    wide = 16 // es
    d = x.shape[1]
    ok = (d % wide == 0 and d <= 64 * wide and x.stride(0) % wide == 0 and x.data_ptr() % 16 == 0
          and (d // max(heads, 1)) % wide == 0)
    if not ok:
        return 1
    if d * es <= 128 and csr.max_deg <= 512:
        # rows of at most one cache line (the column-sharded layer, dist.py): a wavefront per row leaves most lanes
        # idle; the short-row kernel packs 64 / (d / wide) rows into each (profiles/r01_colshard_kernels.txt)
        return 2
    return csr.variant(kind, n_rows)


def _sizes(csr, x: Tensor, heads: int = 1):
    """``CSR.sizes`` (long-row list + compacted short rows, ops.SizeSplit) where it pays and the short-row kernel exists: rows of
    at most one cache line.  ``ALLSET_SIZE_SPLIT=0`` turns it off."""
    import os
    if getattr(csr, "sizes", None) is None or os.environ.get("ALLSET_SIZE_SPLIT", "1") == "0":
        return None
    es = x.element_size()
    wide = 16 // es
    d = x.shape[1]
    ok = (d % wide == 0 and d * es <= 128 and x.stride(0) % wide == 0 and x.data_ptr() % 16 == 0 and (d // max(heads, 1)) % wide == 0)
    return csr.sizes if ok else None


def _split(csr, x: Tensor, heads: int = 1) -> int:
    """``CSR.short_tail`` if the short-row kernel exists for this feature layout, else -1 (single launch)."""
    if csr.short_tail <= 0:
        return -1
    es = x.element_size()
    wide = 16 // es
    d = x.shape[1]
    ok = (d % wide == 0 and d <= 64 * wide and x.stride(0) % wide == 0 and x.data_ptr() % 16 == 0
          and (d // max(heads, 1)) % wide == 0)
    return csr.short_tail if ok else -1


def _check_rows(x: Tensor, inc: Incidence) -> None:
    """The gathered matrix must cover every source id (PyG's index_select would raise otherwise) and
    may not have more rows than the transposed CSR (its backward produces one row per CSR row)."""
    if not (inc.src_extent <= x.shape[0] <= inc.n_src):
        raise ValueError(f"source matrix has {x.shape[0]} rows; incidence needs between {inc.src_extent} "
                         f"and {inc.n_src}")


class _RouteWeights(torch.autograd.Function):
    """``w[perm]`` for a PERMUTATION ``perm`` (edge-list order -> CSR order of a trainable per-incidence weight, LearnMask):
    the backward of a permutation is the gather by its inverse -- torch's advanced-indexing backward is a sort-based
    ``index_put`` (several radix-sort passes over nnz keys per call, ~1 ms at nnz = 16M)."""

    @staticmethod
    def forward(ctx, w, perm, inv_perm):
        ctx.save_for_backward(inv_perm)
        return w.index_select(0, perm)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (inv_perm,) = ctx.saved_tensors
        return g.index_select(0, inv_perm), None, None


def _routed(norm: Tensor, inc: Incidence, dst: bool) -> Tensor:
    """``norm`` (edge-list order, requires grad) in the order of ``inc.by_dst`` / ``inc.by_src``.

    A NON-LEAF norm (SetGNN's ``Importance * norm``, a fresh tensor per forward) caches the routed copy on itself per
    (CSR object, version, grad mode): the V->E and E->V convs of a layer share one routing and autograd sums their gradients; the
    cache dies with that forward's tensor.  A LEAF (an ``nn.Parameter`` handed straight to ``deepsets_aggregate``) is never
    cached: it outlives the forward, the optimizer updates it in place, a hit would replay the first call's values and graph
    (and keep it alive through AccumulateGrad -> tensor -> cache).  Routing is one gather, 0.2 ms at nnz = 16M."""
    csr = inc.by_dst if dst else inc.by_src
    perm, inv = (inc.perm_dst_long(), inc.inv_perm_dst()) if dst else (inc.perm_src_long(), inc.inv_perm_src())
    if norm.grad_fn is None:
        return _RouteWeights.apply(norm.reshape(-1).to(torch.float32), perm, inv)
    cache = norm.__dict__.setdefault("_allset_routed", {})
    key = (id(csr), norm._version, torch.is_grad_enabled())
    hit = cache.get(key)
    if hit is None:
        hit = cache[key] = _RouteWeights.apply(norm.reshape(-1).to(torch.float32), perm, inv)
    return hit


class _SegReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, w_dst: Optional[Tensor], w_src: Optional[Tensor], inc: Incidence, reduce: int):
        csr = inc.by_dst
        ext = reduce in (MAX, MIN)
        out, arg = ops.segreduce(reduce, csr.rowptr, csr.col, w_dst, x, inc.n_dst, want_arg=ext,
                                 variant=1 if ext else _variant(csr, "segreduce", inc.n_dst, x), row_order=csr.row_order,
                                 split=_split(csr, x))
        need_gw = w_dst is not None and ctx.needs_input_grad[1]
        ctx.inc, ctx.reduce, ctx.n_s, ctx.need_gw = inc, reduce, x.shape[0], need_gw
        ctx.save_for_backward(x if need_gw else None, w_dst, w_src, arg)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout: Tensor):
        x, w_dst, w_src, arg = ctx.saved_tensors
        inc, reduce = ctx.inc, ctx.reduce
        T = inc.by_src
        gx = gw = None
        gout = gout.contiguous()
        if reduce in (MAX, MIN) or ctx.need_gw:
            if gout.dtype != torch.float32:
                raise _lib.AllSetHipError("max/min backward and weight gradients are fp32-only; bf16 storage covers "
                                          "sum/mean and the PMA path")
        if ctx.needs_input_grad[0]:
            if w_dst is not None and w_src is None:       # differentiable weights: route on the fly
                w_src = w_dst[inc.pos_dst_of_src().long()]
            if reduce in (SUM, MEAN):
                if reduce == MEAN:
                    inv = inc.inv_count_by_src()
                    w_src = inv if w_src is None else w_src * inv
                gx, _ = ops.segreduce(SUM, T.rowptr, T.col, w_src, gout, ctx.n_s,
                                      variant=_variant(T, "segreduce", ctx.n_s, gout), row_order=T.row_order,
                                      split=_split(T, gout) if ctx.n_s == T.n_rows else -1)
            else:
                gx = ops.segmax_bwd(T.rowptr, T.col, inc.pos_dst_of_src(), w_src, arg, gout, ctx.n_s)
        if ctx.need_gw:
            csr = inc.by_dst
            gw = ops.sddmm_rowdot(reduce, csr.rowptr, csr.col, x, gout, arg)
        return gx, gw, None, None, None


def deepsets_aggregate(x: Tensor, inc: Incidence, norm: Optional[Tensor] = None, aggr: str = "add") -> Tensor:
    """``out[t] = reduce_{i: dst_i = t} norm_i * x[src_i]`` for ``aggr`` in add|sum|mean|max|min.

    ``norm`` is per incidence in the caller's edge-list order (int64 ones in the reference default,
    preprocessing.py:454); ``None`` or all-ones skips the weight stream.  Output has ``inc.n_dst`` rows.
    """
    if aggr not in REDUCE_CODES:
        raise ValueError(f"unknown aggr {aggr!r}")
    _lib.require_device(x)
    _check_rows(x, inc)
    if norm is not None and norm.requires_grad:
        # differentiable routing (LearnMask): edge-list order -> the two CSR orders, ONCE per norm tensor and CSR -- every conv of a
        # forward gets the same ``Importance * norm`` object (models.py:451-452) and V->E / E->V share the two CSRs, so a two-layer
        # model routes twice per forward and twice per backward instead of three gathers per aggregation
        w_dst = _routed(norm, inc, True)
        w_src = _routed(norm, inc, False).detach()        # (only the input gradient reads it; the weight gradient flows through w_dst)
    else:
        w_dst, w_src = inc.weights(norm)
    return _SegReduce.apply(x, w_dst, w_src, inc, REDUCE_CODES[aggr])


def _colocate(V: Tensor, heads: int) -> bool:
    """Feature rows of at most half a cache line (the column-sharded layer's d/P slices): put the per-row scalars the
    kernels gather -- logits forward, (M, delta) backward -- into the same 128-byte line as the row, so an incidence
    costs one cache-line request instead of two (profiles/r01_colshard_kernels.txt)."""
    row = V.shape[1] * V.element_size()
    return V.is_cuda and row % 16 == 0 and row + 8 * heads <= 128 and row <= 64 and V.shape[0] >= 4096


def _beside(rows: Tensor, small_cols: int) -> Tuple[Tensor, Tensor]:
    """A 128-byte-pitched buffer holding a copy of ``rows`` [n, d] and room for ``small_cols`` float32 per row behind it.
    Returns (view of the row copy [n, d], float32 view [n, small_cols]) -- both row-strided views of the one buffer."""
    n, d = rows.shape
    rb = d * rows.element_size()
    buf = torch.empty((n, 128), dtype=torch.uint8, device=rows.device)
    rv = buf[:, :rb].view(rows.dtype)
    rv.copy_(rows)
    return rv, buf[:, rb:rb + 4 * small_cols].view(torch.float32)


class _PmaAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, V: Tensor, alpha: Tensor, inc: Incidence, heads: int, slope: float):
        csr = inc.by_dst
        split = _split(csr, V, heads)
        Vg, ag = V, alpha
        if _colocate(V, heads) and split <= 0:
            Vg, ag = _beside(V, heads)
            ag.copy_(alpha)
        out, m, l = ops.pma_fwd(csr.rowptr, csr.col, ag, Vg, heads, slope, inc.n_dst,
                                variant=_variant(csr, "pma_fwd", inc.n_dst, V, heads), row_order=csr.row_order,
                                split=split, sizes=_sizes(csr, V, heads) if split <= 0 else None)
        ctx.inc, ctx.slope = inc, slope
        ctx.save_for_backward(V, alpha, out, m, l)
        ctx.mark_non_differentiable(m, l)
        ctx.set_materialize_grads(False)     # (m, l never carry a gradient: no [n, H] zero-fills for them in every backward)
        return out, m, l

    @staticmethod
    @once_differentiable
    def backward(ctx, gout: Tensor, _gm, _gl):
        if gout is None:
            return None, None, None, None, None
        V, alpha, out, m, l = ctx.saved_tensors
        T = ctx.inc.by_src
        gout = gout.contiguous()
        H = alpha.shape[1]
        split = _split(T, V, H)
        if _colocate(gout, H) and split <= 0:
            gout, sv = _beside(gout, 2 * H)
            stats = ops.pma_bwd_stats(out, gout, m, l, stats=sv.unflatten(1, (H, 2)))
        else:
            stats = ops.pma_bwd_stats(out, gout, m, l)
        gV, galpha = ops.pma_bwd_src(T.rowptr, T.col, alpha, V, gout, stats, ctx.slope,
                                     variant=_variant(T, "pma_bwd_src", V.shape[0], V, H),
                                     row_order=T.row_order, split=split, sizes=_sizes(T, V, H) if split <= 0 else None)
        return gV, galpha, None, None, None


class _PmaPoolLn0(torch.autograd.Function):
    """``LayerNorm_{gamma,beta}( pma_pool(V, alpha) + att_r )`` -- the pooling and the first LayerNorm of the PMA tail
    (reference layers.py:145-154) as ONE autograd node, so that the backward statistics of the pooling
    (``{m + log l, <out, gout>}`` per target and head) are written by the LayerNorm-backward kernel that already holds ``out``
    and its gradient in registers, instead of by a separate pass over both (allset_pma_bwd_stats: 0.22 ms per direction at 1M x 128)."""

    @staticmethod
    def forward(ctx, V, alpha, inc, heads, slope, att_r, gamma, beta, eps):
        from . import dense
        csr = inc.by_dst
        pooled, m, l = ops.pma_fwd(csr.rowptr, csr.col, alpha, V, heads, slope, inc.n_dst,
                                   variant=_variant(csr, "pma_fwd", inc.n_dst, V, heads), row_order=csr.row_order,
                                   split=_split(csr, V, heads), sizes=_sizes(csr, V, heads))
        cb = att_r.reshape(-1)
        y, stats = dense.ln_res_fwd(pooled, cb, None, gamma, beta, eps, False, 0.0, 0, None)
        ctx.inc, ctx.slope, ctx.cshape = inc, slope, att_r.shape
        ctx.params = (gamma, beta, att_r)           # (the objects themselves: dense.deferred_param_grads assigns their .grad)
        ctx.save_for_backward(V, alpha, pooled, m, l, cb, stats, gamma, beta)
        ctx.mark_non_differentiable(m, l)
        ctx.set_materialize_grads(False)     # (m, l never carry a gradient: no [n, H] zero-fills for them in every backward)
        return y, m, l

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gm, _gl):
        from . import dense
        if gy is None:
            return (None,) * 9
        V, alpha, pooled, m, l, cb, stats, gamma, beta = ctx.saved_tensors
        need = ctx.needs_input_grad
        dfr = dense._Deferred.active and need[5] and need[6] and need[7] and pooled.dtype == torch.float32
        g_pooled, dg, db, dc, pstats = dense.ln_res_bwd_pma(gy.contiguous(), pooled, cb, stats, gamma, beta, m, l,
                                                            defer_to=ctx.params if dfr else None)
        T = ctx.inc.by_src
        H = alpha.shape[1]
        gV, galpha = ops.pma_bwd_src(T.rowptr, T.col, alpha, V, g_pooled, pstats, ctx.slope,
                                     variant=_variant(T, "pma_bwd_src", V.shape[0], V, H), row_order=T.row_order,
                                     split=_split(T, V, H), sizes=_sizes(T, V, H))
        return gV, galpha, None, None, None, (dc.reshape(ctx.cshape) if dc is not None else None), dg, db, None


class _PmaPoolTail(torch.autograd.Function):
    """Pooling + the whole PMA tail (reference layers.py:145-157) as ONE autograd node: ``pma_fwd`` and the two tail kernels of
    ``dense.pma_tail_fwd`` forward; backward as ``_PmaPoolLn0`` (the pooling's backward statistics come out of ln0's backward
    pass) with ln1's backward reading the saved sum."""

    @staticmethod
    def forward(ctx, V, alpha, inc, heads, slope, att_r, g0, b0, eps0, w1, b1, w2, b2, g1, bt1, eps1, relu_post, p):
        from . import dense
        csr = inc.by_dst
        pooled, m, l = ops.pma_fwd(csr.rowptr, csr.col, alpha, V, heads, slope, inc.n_dst,
                                   variant=_variant(csr, "pma_fwd", inc.n_dst, V, heads), row_order=csr.row_order,
                                   split=_split(csr, V, heads), sizes=_sizes(csr, V, heads))
        y, saved, cfg = dense.pma_tail_fwd(pooled, att_r.reshape(-1), g0, b0, eps0, w1, b1, w2, b2, g1, bt1, eps1, relu_post, p)
        ctx.inc, ctx.slope, ctx.cshape, ctx.cfg = inc, slope, att_r.shape, cfg
        ctx.params = (att_r, g0, b0, w1, b1, w2, b2, g1, bt1)     # (the objects themselves: dense.deferred_param_grads assigns their .grad)
        ctx.save_for_backward(V, alpha, m, l, *saved)
        ctx.mark_non_differentiable(m, l)
        ctx.set_materialize_grads(False)     # (m, l never carry a gradient: no [n, H] zero-fills for them in every backward)
        return y, m, l

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gm, _gl):
        from . import dense
        if gy is None:
            return (None,) * 18
        V, alpha, m, l = ctx.saved_tensors[:4]
        g_pooled, dc, dg0, db0, gw1, gb1, gw2, gb2, dg1, db1, pstats = dense.pma_tail_bwd(
            ctx.saved_tensors[4:], ctx.cfg, gy, m, l, params=dense.tail_defer_params(ctx, (5, 6, 7, 9, 10, 11, 12, 13, 14)))
        T = ctx.inc.by_src
        H = alpha.shape[1]
        gV, galpha = ops.pma_bwd_src(T.rowptr, T.col, alpha, V, g_pooled, pstats, ctx.slope,
                                     variant=_variant(T, "pma_bwd_src", V.shape[0], V, H), row_order=T.row_order,
                                     split=_split(T, V, H), sizes=_sizes(T, V, H))
        return (gV, galpha, None, None, None, (dc.reshape(ctx.cshape) if dc is not None else None), dg0, db0, None, gw1, gb1, gw2, gb2,
                dg1, db1, None, None, None)


def pma_pool_tail(V: Tensor, alpha: Tensor, inc: Incidence, heads: int, negative_slope: float, att_r: Tensor, g0, b0, eps0, w1, b1, w2, b2,
                  g1, bt1, eps1, relu_post: bool, p: float) -> Tuple[Tensor, Tensor, Tensor]:
    """``(tail(pool(V, alpha)), m, l)``; see :class:`_PmaPoolTail` (callers check ``pma_pool_ln0_supported`` and
    ``dense.pma_tail_supported``)."""
    _lib.require_device(V, alpha)
    _check_rows(V, inc)
    if alpha.dtype != torch.float32:
        alpha = alpha.float()
    return _PmaPoolTail.apply(V, alpha, inc, int(heads), float(negative_slope), att_r, g0, b0, float(eps0), w1, b1, w2, b2, g1, bt1,
                              float(eps1), bool(relu_post), float(p))


def pma_pool_ln0_supported(V: Tensor, heads: int) -> bool:
    from . import dense
    d = V.shape[1]
    return (V.is_cuda and V.dtype in (torch.float32, torch.bfloat16) and dense.ln_res_supported(d, V.dtype)
            and dense.ln_res_bwd_pma_supported(d, heads, V.dtype) and not _colocate(V, heads))


def pma_pool_ln0(V: Tensor, alpha: Tensor, inc: Incidence, heads: int, negative_slope: float, att_r: Tensor, gamma: Tensor,
                 beta: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    """``(LayerNorm(pool(V, alpha) + att_r), m, l)``; see :class:`_PmaPoolLn0`."""
    _lib.require_device(V, alpha)
    _check_rows(V, inc)
    if alpha.dtype != torch.float32:
        alpha = alpha.float()
    return _PmaPoolLn0.apply(V, alpha, inc, int(heads), float(negative_slope), att_r, gamma, beta, float(eps))


def pma_aggregate(V: Tensor, alpha: Tensor, inc: Incidence, heads: int, negative_slope: float = 0.2
                  ) -> Tuple[Tensor, Tensor, Tensor]:
    """Softmax-attention pooling: ``out[t,h,:] = sum_i softmax_i(leaky_relu(alpha[src_i,h])) * V[src_i,h,:]``.

    ``V``: [n_src, heads*C], ``alpha``: [n_src, heads] (pre-activation).  Returns
    ``(out [n_dst, heads*C], m [n_dst, heads], l [n_dst, heads])``; empty targets give 0.
    """
    _lib.require_device(V, alpha)
    _check_rows(V, inc)
    if alpha.dtype != torch.float32:          # logits and softmax statistics are always fp32 (bf16 is storage only)
        alpha = alpha.float()
    return _PmaAggregate.apply(V, alpha, inc, int(heads), float(negative_slope))


def pma_attention_weights(alpha: Tensor, m: Tensor, l: Tensor, inc: Incidence, negative_slope: float = 0.2) -> Tensor:
    """Per-incidence attention weights [nnz, heads] in the caller's edge-list order
    (reference ``PMA.forward(..., return_attention_weights=True)``, layers.py:159-162)."""
    csr = inc.by_dst
    p_csr = ops.pma_attention(csr.rowptr, csr.col, alpha.detach(), m, l, float(negative_slope))
    p = torch.empty_like(p_csr)
    p[csr.perm.long()] = p_csr
    return p
