"""CPU oracle for AllSet's vertex<->hyperedge aggregation path.  TEST INFRASTRUCTURE ONLY.

This file is a *restatement* (plain torch-CPU, functional, no nn.Module, no PyG, no
torch_scatter) of what the reference computes on the path
``SetGNN.forward -> HalfNLHconv.forward -> {PMA.forward | propagate/message/aggregate}``.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it -- as the checker / the timed CPU comparator, never as the thing shipped.  The product
package ``allset_amd`` must not (and does not) import anything from ``oracle/``.

Pinning status
--------------
The reference (jianhao2016/AllSet) ships no tests, fixtures or golden vectors, and
the arithmetic of the path lives in third-party wheels that are absent from /root/reference and
from this image: ``torch-scatter==2.0.4`` (``scatter``), ``torch-geometric==1.6.3``
(``MessagePassing.propagate``, ``utils.softmax``), pins at reference README.md:18-22.
What *is* pinned: every function below is checked (``oracle/gen_golden.py``, run in the build
container) against the reference's own ``src/layers.py`` / ``src/models.py`` executed under
``oracle/ref_shim.py`` -- forward outputs and all input/parameter gradients, max-abs-diff <= 1e-6 --
and the resulting vectors are committed under ``tests/golden/``.  What is *not* pinned by any
reference-owned artefact: the published semantics of the two absent wheels, which
``ref_shim.py`` restates (dim_size = index.max()+1, empty segment -> 0, mean = sum/clamp(count,1),
softmax denominator +1e-16).  For that seam: **parity unpinned** (see DESIGN.md "Oracle").

The arithmetic is fp32 unless the caller passes fp64 tensors; eval-mode semantics (dropout is the
identity; BatchNorm uses running statistics) unless a ``drop`` callable is passed: then every
``F.dropout(..., training=True)`` site of the reference is visited in the reference's order and
``drop(x, p)`` stands for it (``ExplicitDropout``: keep masks given by the caller, so a product run
whose masks are known can be checked in TRAINING mode).  Autograd is plain torch autograd over these
ops, which is what the reference relies on as well.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------
# Operator seam: what torch_scatter.scatter / torch_geometric.utils.softmax do [external]
# --------------------------------------------------------------------------------------


def scatter(src: Tensor, index: Tensor, dim_size: Optional[int] = None, reduce: str = "sum") -> Tensor:
    """Segment-reduce rows of ``src`` ([nnz, ...]) into ``out[index[i]]`` along dim 0.

    Restates ``torch_scatter.scatter(src, index, dim=0, reduce=...)`` as called at reference
    layers.py:194 (PMA.aggregate) and layers.py:656 (HalfNLHconv.aggregate).  Both call sites
    omit ``dim_size`` so the row count defaults to ``index.max()+1``.
    Rows that receive nothing are 0 for every reduce mode.  ``mean`` divides by
    ``count.clamp(min=1)``.  ``max``/``min`` back-propagate to the extremal element.
    """
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    tail = src.shape[1:]
    flat = src.reshape(src.shape[0], -1)
    idx = index.view(-1, 1).expand_as(flat)
    out = torch.zeros((dim_size, flat.shape[1]), dtype=src.dtype, device=src.device)
    if reduce in ("sum", "add"):
        out = out.scatter_add(0, idx, flat)
    elif reduce == "mean":
        out = out.scatter_add(0, idx, flat)
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        cnt = cnt.scatter_add(0, index, torch.ones_like(index, dtype=src.dtype)).clamp(min=1)
        out = out / cnt.view(-1, 1)
    elif reduce in ("max", "min"):
        out = out.scatter_reduce(0, idx, flat, "amax" if reduce == "max" else "amin", include_self=False)
    else:
        raise ValueError(f"unknown reduce {reduce!r}")
    return out.reshape((dim_size,) + tuple(tail))


def segment_softmax(src: Tensor, index: Tensor, num_nodes: Optional[int] = None) -> Tensor:
    """Softmax of ``src`` ([nnz, H]) over entries sharing ``index``.

    Restates ``torch_geometric.utils.softmax(src, index, ptr=None, num_nodes)`` (PyG 1.6.3) as
    called at reference layers.py:174: subtract the per-segment max, exp, divide by the per-segment
    sum plus 1e-16.
    """
    n = int(index.max()) + 1 if num_nodes is None else int(num_nodes)
    seg_max = scatter(src, index, n, "max")
    e = (src - seg_max.index_select(0, index)).exp()
    seg_sum = scatter(e, index, n, "sum")
    return e / (seg_sum.index_select(0, index) + 1e-16)


# --------------------------------------------------------------------------------------
# The two aggregations (the hot path proper)
# --------------------------------------------------------------------------------------


def deepsets_aggregate(x: Tensor, edge_index: Tensor, norm: Tensor, aggr: str = "add") -> Tensor:
    """out[t] = reduce_{i: dst_i = t} norm_i * x[src_i]  -- reference layers.py:633,638-656.

    ``propagate`` lifts ``x_j = x.index_select(0, edge_index[0])`` (PyG ``__lift__``), ``message``
    multiplies by ``norm.view(-1, 1)`` (int64 ones by default, promoted to fp32),
    ``aggregate`` is ``scatter(..., reduce=aggr)`` without ``dim_size``.
    """
    x_j = x.index_select(0, edge_index[0])
    msg = norm.view(-1, 1) * x_j
    return scatter(msg, edge_index[1], None, aggr)


def pma_aggregate(x_v: Tensor, alpha_r: Tensor, edge_index: Tensor, negative_slope: float = 0.2
                  ) -> Tuple[Tensor, Tensor]:
    """Softmax-attention pooling with a source-only logit -- reference layers.py:145,168-194.

    ``x_v`` is [n_s, H, C], ``alpha_r`` [n_s, H].  Per incidence: leaky_relu(alpha_r[src]), softmax
    over the incidences of the same target, weighted sum of ``x_v[src]``.  Attention dropout is
    hard-wired to 0 in the reference (layers.py:63).  Returns (out [n_t, H, C], p [nnz, H]).
    """
    src, dst = edge_index[0], edge_index[1]
    a = F.leaky_relu(alpha_r.index_select(0, src), negative_slope)
    p = segment_softmax(a, dst, int(dst.max()) + 1)
    out = scatter(x_v.index_select(0, src) * p.unsqueeze(-1), dst, None, "add")
    return out, p


# --------------------------------------------------------------------------------------
# Dense tail and module-level composition, driven by a reference-layout state_dict
# --------------------------------------------------------------------------------------


class ExplicitDropout:
    """``F.dropout(x, p, training=True)`` with the keep mask SUPPLIED: ``y = x * keep / (1 - p)`` (torch's scaling).

    ``masks``: bool tensors in the order the reference's training forward meets its dropout sites (models.py:473,477,481;
    layers.py:577,632).  Sites with p == 0 draw nothing and consume nothing, as in torch.  ``used`` counts consumed masks
    so a caller can assert that product and oracle agree on the NUMBER of sites too."""

    def __init__(self, masks):
        self.masks = list(masks)
        self.used = 0

    def __call__(self, x: Tensor, p: float) -> Tensor:
        if p <= 0.0:
            return x
        if self.used >= len(self.masks):
            raise IndexError(f"dropout site {self.used}: no mask left (shape {tuple(x.shape)}, p = {p})")
        keep = self.masks[self.used]
        if tuple(keep.shape) != tuple(x.shape):
            raise ValueError(f"dropout site {self.used}: mask {tuple(keep.shape)} for a tensor {tuple(x.shape)}")
        self.used += 1
        return x * keep.to(x.dtype) / (1.0 - p)



def _norm_apply(sd: Dict[str, Tensor], key: str, x: Tensor, kind: str, training: bool = False) -> Tensor:
    if key + ".weight" not in sd:
        return x  # nn.Identity slot
    if kind == "ln":
        return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], 1e-5)
    if kind == "bn":  # eval mode: running statistics; training mode: batch statistics (running buffers not updated here)
        if training:
            return F.batch_norm(x, None, None, sd[key + ".weight"], sd[key + ".bias"], True, 0.0, 1e-5)
        return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                            sd[key + ".weight"], sd[key + ".bias"], False, 0.0, 1e-5)
    raise ValueError(kind)


def mlp_forward(sd: Dict[str, Tensor], prefix: str, x: Tensor, normalization: str = "ln", drop=None, p: float = 0.0) -> Tensor:
    """reference layers.py:571-579: norm0 -> [Linear -> ReLU -> norm -> dropout] x (L-1) -> Linear.
    ``drop`` / ``p``: training mode, the MLP's own dropout rate (layers.py:577)."""
    n_lin = 0
    while f"{prefix}lins.{n_lin}.weight" in sd:
        n_lin += 1
    training = drop is not None
    x = _norm_apply(sd, f"{prefix}normalizations.0", x, normalization, training)
    for i in range(n_lin - 1):
        x = F.linear(x, sd[f"{prefix}lins.{i}.weight"], sd[f"{prefix}lins.{i}.bias"])
        x = F.relu(x)
        x = _norm_apply(sd, f"{prefix}normalizations.{i + 1}", x, normalization, training)
        if training:
            x = drop(x, p)
    last = n_lin - 1
    return F.linear(x, sd[f"{prefix}lins.{last}.weight"], sd[f"{prefix}lins.{last}.bias"])


def pma_forward(sd: Dict[str, Tensor], prefix: str, x: Tensor, edge_index: Tensor, heads: int,
                return_attention_weights: bool = False):
    """reference layers.py:120-166 (PMA.forward)."""
    w_k, b_k = sd[prefix + "lin_K.weight"], sd[prefix + "lin_K.bias"]
    w_v, b_v = sd[prefix + "lin_V.weight"], sd[prefix + "lin_V.bias"]
    att_r = sd[prefix + "att_r"]                       # [1, H, C]
    H = heads
    C = w_k.shape[0] // H
    x_k = F.linear(x, w_k, b_k).view(-1, H, C)
    x_v = F.linear(x, w_v, b_v).view(-1, H, C)
    alpha_r = (x_k * att_r).sum(dim=-1)                # [n_s, H]
    out, p = pma_aggregate(x_v, alpha_r, edge_index, 0.2)
    out = out + att_r                                  # seed + multihead  (layers.py:153)
    out = out.reshape(-1, H * C)
    out = F.layer_norm(out, (H * C,), sd[prefix + "ln0.weight"], sd[prefix + "ln0.bias"], 1e-5)
    ff = mlp_forward(sd, prefix + "rFF.", out, "None")
    out = F.layer_norm(out + F.relu(ff), (H * C,), sd[prefix + "ln1.weight"], sd[prefix + "ln1.bias"], 1e-5)
    if return_attention_weights:
        return out, (edge_index, p)
    return out


def halfnlhconv_forward(sd: Dict[str, Tensor], prefix: str, x: Tensor, edge_index: Tensor, norm: Tensor,
                        aggr: str, attention: bool, heads: int, normalization: str, drop=None, p: float = 0.0) -> Tensor:
    """reference layers.py:623-636 (HalfNLHconv.forward); eval mode unless ``drop`` is given (``p`` = the conv's
    ``dropout``, which is also its MLPs' rate, layers.py:603-604; PMA has no dropout site, layers.py:63,76-80)."""
    if attention:
        return pma_forward(sd, prefix + "prop.", x, edge_index, heads)
    has_mlp = (prefix + "f_enc.lins.0.weight") in sd           # num_layers == 0 -> nn.Identity (Q8)
    if has_mlp:
        x = mlp_forward(sd, prefix + "f_enc.", x, normalization, drop, p)
    x = F.relu(x)
    if drop is not None:
        x = drop(x, p)                                         # layers.py:632
    x = deepsets_aggregate(x, edge_index, norm, aggr)
    if has_mlp:
        x = mlp_forward(sd, prefix + "f_dec.", x, normalization, drop, p)
    return F.relu(x)


def setgnn_forward(sd: Dict[str, Tensor], args: SimpleNamespace, x: Tensor, edge_index: Tensor,
                   norm: Tensor, collect: Optional[dict] = None, drop=None) -> Tensor:
    """reference models.py:450-484 (both the GPR branch :457-471 and the plain one); eval mode (dropouts are identities)
    unless ``drop`` is given: then training mode with ``drop(x, p)`` at every dropout site, in the reference's order
    (input dropout 0.2 :473; per conv the MLPs' inner sites and the conv's own, layers.py:577,632; after every relu(conv)
    :477,481 / :461,466; the classifier's inner sites).

    Unlike the reference this does not mutate ``edge_index`` in place (Q2): the hyperedge ids are
    re-based on a copy.  ``collect`` (optional dict) receives the intermediate conv outputs.
    """
    if getattr(args, "LearnMask", False):
        norm = sd["Importance"] * norm
    ei = torch.stack([edge_index[0], edge_index[1] - edge_index[1].min()], dim=0)
    rev = torch.stack([ei[1], ei[0]], dim=0)
    attention = bool(args.PMA)
    nl = args.normalization
    p = float(args.dropout)
    d_ = (lambda t, q: t) if drop is None else drop
    if getattr(args, "GPR", False):
        xs = [F.relu(mlp_forward(sd, "MLP.", x, nl, drop, p))]
        for i in range(args.All_num_layers):
            x = halfnlhconv_forward(sd, f"V2EConvs.{i}.", x, ei, norm, args.aggregate, attention, args.heads, nl, drop, p)
            if collect is not None:
                collect[f"v2e{i}"] = x
            x = d_(F.relu(x), p)
            x = halfnlhconv_forward(sd, f"E2VConvs.{i}.", x, rev, norm, args.aggregate, attention, args.heads, nl, drop, p)
            if collect is not None:
                collect[f"e2v{i}"] = x
            x = F.relu(x)
            xs.append(x)
            x = d_(x, p)
        x = torch.stack(xs, dim=-1)
        x = F.linear(x, sd["GPRweights.weight"]).squeeze()
        return mlp_forward(sd, "classifier.", x, nl, drop, p)
    x = d_(x, 0.2)                                                           # models.py:473
    for i in range(args.All_num_layers):
        x = halfnlhconv_forward(sd, f"V2EConvs.{i}.", x, ei, norm, args.aggregate, attention, args.heads, nl, drop, p)
        if collect is not None:
            collect[f"v2e{i}"] = x          # raw conv output (what a forward hook on the conv sees)
        x = d_(F.relu(x), p)
        x = halfnlhconv_forward(sd, f"E2VConvs.{i}.", x, rev, norm, args.aggregate, attention, args.heads, nl, drop, p)
        if collect is not None:
            collect[f"e2v{i}"] = x
        x = d_(F.relu(x), p)
    return mlp_forward(sd, "classifier.", x, nl, drop, p)


# --------------------------------------------------------------------------------------
# Aggregation-only V->E->V pass used as the timed CPU baseline (bench.py cpu_baseline leg)
# --------------------------------------------------------------------------------------


def v2e2v_aggregation_fwd_bwd(x: Tensor, edge_index: Tensor, norm: Tensor, aggr: str = "add") -> Tensor:
    """One V->E->V aggregation forward + backward with no dense tail: exactly the
    index_select / mul / scatter_add_ / autograd sequence torch_scatter 2.0.4 dispatches to for the
    reference DeepSets branch (layers.py:633-656), twice (models.py:475,478).  Returns grad wrt x."""
    x = x.detach().requires_grad_(True)
    rev = torch.stack([edge_index[1], edge_index[0]], dim=0)
    e = deepsets_aggregate(x, edge_index, norm, aggr)
    v = deepsets_aggregate(e, rev, norm, aggr)
    v.backward(torch.ones_like(v))
    return x.grad
