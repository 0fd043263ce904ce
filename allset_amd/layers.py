"""The reference's layer-level module surface for the AllSet path, on MI355X kernels.

Same class names, constructor signatures, ``forward`` arguments and ``state_dict`` layout as
reference ``src/layers.py`` (``MLP`` :496, ``PMA`` :42, ``HalfNLHconv`` :582), so a checkpoint or a
``train.py``-style caller moves over unchanged.  What differs is underneath: ``propagate`` is not a
PyG message-passing template over [nnz, d] temporaries but one call into ``functional.py`` (HIP
kernels over a CSR built once), and the dense tail (LayerNorm / Linear / ReLU / dropout) runs on the
fused fp32-MFMA kernels of ``dense.py`` for device fp32 tensors (plain torch modules otherwise: CPU
tensors in host-side tests, bf16, BatchNorm).

``edge_index`` may be the reference's int64 ``[2, nnz]`` tensor (converted once and cached on tensor
identity + version) or a prebuilt :class:`allset_amd.incidence.Incidence`.
"""
from __future__ import annotations

import math
from typing import Tuple, Optional, Union

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dense
from . import functional as AF
from .incidence import Incidence, cached_incidence

Tensor = torch.Tensor
EdgeIndex = Union[Tensor, Incidence]


def glorot(tensor: Optional[Tensor]) -> None:
    """U(-a, a), a = sqrt(6 / (fan_in + fan_out))  -- reference layers.py:31-34."""
    if tensor is not None:
        bound = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
        with torch.no_grad():
            tensor.uniform_(-bound, bound)


def zeros(tensor: Optional[Tensor]) -> None:
    if tensor is not None:
        with torch.no_grad():
            tensor.zero_()


def _as_incidence(edge_index: EdgeIndex, n_src: int) -> Incidence:
    if isinstance(edge_index, Incidence):
        return edge_index
    return cached_incidence(edge_index, n_src=n_src)      # n_dst = index.max()+1, the reference's rule (Q1)


def _on_hip(x: Tensor) -> bool:
    """Device fp32 tensors take the HIP dense-tail kernels (allset_amd/dense.py); anything else (CPU tensors in
    unit tests / gloo tests, other dtypes) runs the same math as plain torch modules."""
    return x.is_cuda and x.dtype == torch.float32


def relu_dropout(x: Tensor, p: float, training: bool) -> Tensor:
    """``dropout(relu(x))`` -- one fused pass on the device."""
    p = float(p) if training else 0.0
    if _on_hip(x):
        return dense.relu_dropout(x, p)
    return F.dropout(F.relu(x), p=p, training=training)


def _layer_norm(norm: nn.LayerNorm, x: Tensor, relu_in: bool = False, p: float = 0.0) -> Tensor:
    if _on_hip(x) and norm.elementwise_affine and norm.bias is not None:
        return dense.layer_norm(x, norm.weight, norm.bias, norm.eps, relu_in, p)
    if (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and norm.elementwise_affine and norm.bias is not None
            and norm.weight.dtype == torch.bfloat16 and dense.ln_bf16_supported(x.shape[1])):
        return dense.layer_norm(x, norm.weight, norm.bias, norm.eps, relu_in, p)     # bf16 in / out, fp32 arithmetic
    y = norm(F.relu(x) if relu_in else x)
    return F.dropout(y, p=p, training=p > 0.0)


def _norm_module(nm: nn.Module, x: Tensor) -> Tensor:
    """Identity / BatchNorm1d slot of an MLP; BatchNorm goes through ``dense.batch_norm`` so that a sharded layer
    (``dist.py``) can give it the statistics of the whole batch instead of one rank's block of rows."""
    if isinstance(nm, nn.modules.batchnorm._BatchNorm):
        return dense.batch_norm(nm, x)
    return nm(x)


def _linear(lin: nn.Linear, x: Tensor, relu_out: bool = False) -> Tensor:
    if dense.linear_bf16_supported(x, lin.weight, lin.bias):
        return dense.linear_bf16(x, lin.weight, lin.bias, relu_out)     # bf16 regime: one kernel, relu in its epilogue
    if relu_out:
        return F.relu(_linear(lin, x))
    if _on_hip(x) or (x.is_cuda and x.dtype == torch.bfloat16 and lin.weight.dtype == torch.bfloat16):
        return dense.linear(x, lin.weight, lin.bias)     # library GEMMs forward / backward-data, split-K MFMA weight gradient
    return lin(x)


def _make_norm(kind: str, width: int) -> nn.Module:
    if kind == "bn":
        return nn.BatchNorm1d(width)
    if kind == "ln":
        return nn.LayerNorm(width)
    return nn.Identity()


class MLP(nn.Module):
    """``norm0 -> [Linear -> ReLU -> norm -> dropout] x (L-1) -> Linear``  (reference layers.py:496-579).

    ``Normalization`` in {'bn','ln','None'}; ``InputNorm`` selects whether slot 0 normalises the input.
    Parameter names: ``lins.{i}``, ``normalizations.{i}``.
    """

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers,
                 dropout=.5, Normalization='bn', InputNorm=False):
        super().__init__()
        assert Normalization in ['bn', 'ln', 'None']
        self.InputNorm = InputNorm
        self.dropout = dropout
        widths = [in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels]
        self.lins = nn.ModuleList(nn.Linear(widths[i], widths[i + 1]) for i in range(num_layers))
        norms = [_make_norm(Normalization if InputNorm else 'None', in_channels)]
        norms += [_make_norm(Normalization, hidden_channels) for _ in range(num_layers - 1)]
        self.normalizations = nn.ModuleList(norms)
        self._raw_input = False          # set by the owning model on the MLP that consumes data.x (see _wide_input)

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()
        for norm in self.normalizations:
            if not isinstance(norm, nn.Identity):
                norm.reset_parameters()

    def _fusable(self, x: Tensor) -> bool:
        """Every Linear can take the fused kernel (norm -> Linear -> activation in one pass): device fp32, widths the
        kernel is built for, LayerNorm/None normalisation (BatchNorm stays on torch)."""
        if not _on_hip(x) or x.dim() != 2:
            return False
        if not all(isinstance(nm, (nn.LayerNorm, nn.Identity)) for nm in self.normalizations):
            return False
        p = float(self.dropout) if self.training else 0.0
        for i, lin in enumerate(self.lins):
            if lin.bias is None:
                return False
            if not (dense.fused_linear_supported(lin.in_features, lin.out_features) or
                    dense.wide_linear_supported(lin.in_features, lin.out_features, isinstance(self.normalizations[i], nn.LayerNorm),
                                                i > 0, p if i > 0 else 0.0)):
                return False
        return True

    def _bn_foldable(self, x: Tensor) -> bool:
        """BatchNorm1d slots in EVAL mode (running statistics) are per-column affine maps ``x * a + b`` -- in MLP.forward's order
        (norm0 -> Linear; Linear -> relu -> norm -> dropout -> Linear, reference layers.py:571-579) each one sits directly in front
        of a Linear and folds into its weight and bias.  The whole MLP then takes the fused Linear kernels with no normalisation
        prologue at all (round 3; training mode needs batch statistics and their backward and stays on torch)."""
        if self.training or not _on_hip(x) or x.dim() != 2 or x.dtype != torch.float32:
            return False
        bns = [nm for nm in self.normalizations if not isinstance(nm, nn.Identity)]
        if not bns or not all(isinstance(nm, nn.modules.batchnorm._BatchNorm) and nm.running_mean is not None and nm.affine for nm in bns):
            return False
        return all(lin.bias is not None and dense.fused_linear_supported(lin.in_features, lin.out_features) for lin in self.lins)

    def _bn_trainable(self, x: Tensor) -> bool:
        """BatchNorm1d slots with BATCH statistics (training mode; round 4): each one, with the relu before it and the dropout
        behind it, is the operand prologue of the Linear that follows -- two column reductions + a per-column affine map inside the
        fused Linear kernel, and one extra in-place pass in the backward for the statistics' own gradient (``dense.bn_linear``,
        csrc/batchnorm.hip).  Not inside a sharded scope (cross-rank / padded-row statistics keep ``dense.batch_norm``)."""
        if not _on_hip(x) or x.dim() != 2 or x.dtype != torch.float32:
            return False
        scope = getattr(dense._sync_bn_state, "scope", None)
        if scope is not None:             # a sharded layer's scope: only the trivial one (one rank, every local row a real row)
            import torch.distributed as dist
            valid, group = scope
            if valid < x.shape[0] or (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
                return False
        if os.environ.get("ALLSET_BN_TORCH", "0") == "1":           # comparison arm (tools, bench variants): torch's BatchNorm kernels
            return False
        bns = [nm for nm in self.normalizations if not isinstance(nm, nn.Identity)]
        if not bns or not all(isinstance(nm, nn.BatchNorm1d) for nm in bns):
            return False
        if not all(nm.training or (nm.running_mean is None and nm.running_var is None) for nm in bns):
            return False
        if x.shape[0] < 2:
            return False
        for nm, lin in zip(self.normalizations, self.lins):
            if lin.bias is None or not dense.fused_linear_supported(lin.in_features, lin.out_features):
                return False
            if not isinstance(nm, nn.Identity) and not dense.bn_linear_supported(nm, lin, x if lin is self.lins[0] else None):
                return False
        return True

    def _wide_input(self, x: Tensor) -> bool:
        """The FIRST MLP of a model at dataset scale (round 4): a LayerNorm over raw features (widths the resident-weight kernels do
        not take) in front of its first Linear, on an input that needs no gradient -- ``dense.input_norm_linear`` (one GEMM each
        way, no [n, d] input gradient); the remaining Linears take the fused kernels as in ``_fusable``."""
        if not _on_hip(x) or x.dim() != 2:
            return False
        # "needs no gradient" must mean "is the raw feature matrix": with autograd on, a tensor without requires_grad is one; under
        # no_grad EVERY activation looks like that, so there only the MLP that a model marked as its first (``_raw_input``, set by
        # SetGNN for the module that consumes data.x) takes this route -- every other MLP keeps the kernels its training step uses.
        if not torch.is_grad_enabled() and not self._raw_input:
            return False
        n0, lin0 = self.normalizations[0], self.lins[0]
        if not (isinstance(n0, nn.LayerNorm) and n0.elementwise_affine and n0.bias is not None):
            return False
        if dense.fused_linear_supported(lin0.in_features, lin0.out_features):
            return False
        if not dense.input_norm_linear_supported(x, n0.weight, n0.bias, lin0.weight, lin0.bias):
            return False
        p = float(self.dropout) if self.training else 0.0
        for nm, lin in zip(self.normalizations[1:], self.lins[1:]):
            if lin.bias is None or not isinstance(nm, (nn.LayerNorm, nn.Identity)):
                return False
            if not (dense.fused_linear_supported(lin.in_features, lin.out_features) or
                    dense.wide_linear_supported(lin.in_features, lin.out_features, isinstance(nm, nn.LayerNorm), True, p)):
                return False
        return True

    def takes_pre_dropout(self, x: Tensor) -> bool:
        """``forward(x, _pre=p)`` can apply the dropout in FRONT of the MLP (models.py:473) inside its first kernel."""
        return self._wide_input(x)

    @staticmethod
    def _fold_bn(bn, lin) -> Tuple[Tensor, Tensor]:
        """(W', b') with ``lin(bn(x)) == x @ W'^T + b'`` for an eval-mode BatchNorm (differentiable w.r.t. W, b, gamma, beta)."""
        a = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * a
        return lin.weight * a.unsqueeze(0), lin.bias + lin.weight @ shift

    def _resident(self) -> bool:
        """Every Linear fits the LDS-resident-weight kernels (widths <= 128): the joint PMA nodes build on those."""
        return all(dense.fused_linear_supported(lin.in_features, lin.out_features) for lin in self.lins)

    def blockable(self, x: Tensor, cb: int) -> bool:
        """The first / last Linear can read / write a COLUMN-BLOCKED tensor of block width ``cb`` (dist.py's exchange layout)."""
        if cb < 4 or (cb & (cb - 1)) or not (_on_hip(x) and x.dtype == torch.float32):
            return False
        first, last = self.lins[0], self.lins[-1]
        return (all(isinstance(nm, (nn.LayerNorm, nn.Identity)) for nm in self.normalizations) and
                all(lin.bias is not None for lin in self.lins) and
                all(dense.blocked_linear_supported(lin.in_features, lin.out_features) for lin in (first, last)) and
                all(dense.fused_linear_supported(lin.in_features, lin.out_features) for lin in self.lins) and
                cb <= first.in_features // 2 and cb <= last.out_features // 2)

    def forward(self, x, _post: Optional[float] = None, _in_cb: int = 0, _out_cb: int = 0, _pre: float = 0.0):
        """``_post`` (internal): also apply ``dropout_p(relu(.))`` to the output -- the activation its callers
        (``HalfNLHconv``) put right after the MLP -- inside the last Linear's epilogue.  ``_in_cb`` / ``_out_cb`` (internal, only
        after ``blockable``): the input / output is column-blocked [(C / cb) * n, cb].  ``_pre`` (internal, only after
        ``takes_pre_dropout``): a dropout in front of the MLP, applied inside its first kernel."""
        p = float(self.dropout) if self.training else 0.0
        post_p = (float(_post) if self.training else 0.0) if _post is not None else None
        if _pre and not self._wide_input(x):
            raise ValueError("MLP.forward(_pre=...) needs takes_pre_dropout(x)")
        if not (_in_cb or _out_cb) and self._wide_input(x):
            n0, lin0 = self.normalizations[0], self.lins[0]
            x = dense.input_norm_linear(x, n0.weight, n0.bias, lin0.weight, lin0.bias, n0.eps, float(_pre) if self.training else 0.0)
            last = len(self.lins) - 1
            if last == 0:
                return x if post_p is None else relu_dropout(x, _post, self.training)
            for i in range(1, last + 1):
                nm, lin = self.normalizations[i], self.lins[i]
                ln = nm if isinstance(nm, nn.LayerNorm) else None
                ro = i == last and post_p is not None
                x = dense.fused_norm_linear(
                    x, ln.weight if ln is not None else None, ln.bias if ln is not None else None, lin.weight, lin.bias,
                    ln.eps if ln is not None else 1e-5, relu_in=True, p_in=p, relu_out=ro, p_out=post_p if ro else 0.0)
            return x
        if _in_cb or _out_cb:
            last = len(self.lins) - 1
            for i, lin in enumerate(self.lins):
                nm = self.normalizations[i]
                ln = nm if isinstance(nm, nn.LayerNorm) else None
                is_last = i == last
                x = dense.fused_norm_linear(
                    x, ln.weight if ln is not None else None, ln.bias if ln is not None else None, lin.weight, lin.bias,
                    ln.eps if ln is not None else 1e-5, relu_in=i > 0, p_in=p if i > 0 else 0.0,
                    relu_out=is_last and post_p is not None, p_out=post_p if (is_last and post_p is not None) else 0.0,
                    in_cb=_in_cb if i == 0 else 0, out_cb=_out_cb if is_last else 0)
            return x
        if self._bn_foldable(x):
            last = len(self.lins) - 1
            for i, lin in enumerate(self.lins):
                nm = self.normalizations[i]
                w, b = (lin.weight, lin.bias) if isinstance(nm, nn.Identity) else self._fold_bn(nm, lin)
                is_last = i == last
                x = dense.fused_norm_linear(x, None, None, w, b, 1e-5, relu_in=i > 0, p_in=0.0,
                                            relu_out=is_last and post_p is not None, p_out=0.0)
            return x
        if self._fusable(x):
            last = len(self.lins) - 1
            stats = None
            for i, lin in enumerate(self.lins):
                nm = self.normalizations[i]
                ln = nm if isinstance(nm, nn.LayerNorm) else None
                is_last = i == last
                # two consecutive 256-wide Linears on the tiled path: the first one's epilogue writes the row statistics the second
                # one's LayerNorm prologue needs (dense.fused_norm_linear emit_stats) -- no allset_row_stats pass in between
                nxt = self.normalizations[i + 1] if not is_last else None
                emit = None
                if (isinstance(nxt, nn.LayerNorm) and not (_in_cb or _out_cb)
                        and dense.wide_stats_chain_supported(lin.weight.shape[1], lin.weight.shape[0], ln is not None, i > 0, p if i > 0 else 0.0)
                        and dense.wide_linear_supported(self.lins[i + 1].weight.shape[1], self.lins[i + 1].weight.shape[0], True, True, p)):
                    emit = (nxt.eps, True)
                x = dense.fused_norm_linear(
                    x, ln.weight if ln is not None else None, ln.bias if ln is not None else None, lin.weight, lin.bias,
                    ln.eps if ln is not None else 1e-5, relu_in=i > 0, p_in=p if i > 0 else 0.0,
                    relu_out=is_last and post_p is not None, p_out=post_p if (is_last and post_p is not None) else 0.0,
                    stats_in=stats if ln is not None else None, emit_stats=emit)
                stats = None
                if emit is not None:
                    x, stats = x
            return x
        if self._bn_trainable(x):
            last = len(self.lins) - 1
            for i, lin in enumerate(self.lins):
                nm = self.normalizations[i]
                ro = i == last and post_p is not None
                po = post_p if ro else 0.0
                if isinstance(nm, nn.Identity):
                    x = dense.fused_norm_linear(x, None, None, lin.weight, lin.bias, 1e-5, relu_in=i > 0, p_in=p if i > 0 else 0.0,
                                                relu_out=ro, p_out=po)
                else:               # relu -> BatchNorm (batch statistics) -> dropout -> Linear: one column-affine prologue
                    x = dense.bn_linear(nm, lin, x, relu_in=i > 0, p_in=p if i > 0 else 0.0, relu_out=ro, p_out=po)
            return x
        n0 = self.normalizations[0]
        x = _layer_norm(n0, x) if isinstance(n0, nn.LayerNorm) else _norm_module(n0, x)
        for i, lin in enumerate(self.lins[:-1]):
            nxt = self.normalizations[i + 1]
            if isinstance(nxt, nn.Identity) and p == 0.0 and dense.linear_bf16_supported(x, lin.weight, lin.bias):
                x = _linear(lin, x, relu_out=True)     # bf16 regime: the relu is the Linear's epilogue
                continue
            a = _linear(lin, x)
            if isinstance(nxt, nn.LayerNorm):          # relu -> LayerNorm -> dropout: one kernel
                x = _layer_norm(nxt, a, relu_in=True, p=p)
            elif isinstance(nxt, nn.Identity):         # relu -> dropout: one kernel
                x = relu_dropout(a, p, self.training)
            else:                                      # BatchNorm1d: torch
                x = dense.hash_dropout(_norm_module(nxt, F.relu(a)), p, self.training)
        if post_p == 0.0 and dense.linear_bf16_supported(x, self.lins[-1].weight, self.lins[-1].bias):
            return _linear(self.lins[-1], x, relu_out=True)
        x = _linear(self.lins[-1], x)
        return x if post_p is None else relu_dropout(x, _post, self.training)


class PMA(nn.Module):
    """Pooling by multi-head attention with a learned seed (reference layers.py:42-199).

    ``K = lin_K(x)``, ``V = lin_V(x)``, ``alpha[s,h] = <K[s,h,:], att_r[h,:]>``; per target the softmax
    of ``leaky_relu(alpha)`` over its incidences pools ``V``; then ``+att_r``, ``ln0``,
    ``ln1(z + relu(rFF(z)))``.  ``concat``, ``dropout`` and ``bias`` are accepted and ignored exactly as
    in the reference (attention dropout is hard-wired to 0 there, layers.py:63; bias is always None).

    ``fold_alpha`` (default True): since ``alpha`` is linear in ``x``, it is computed as
    ``x @ (att_r . W_K)^T + att_r . b_K`` -- an [in, H] mat-vec instead of the full [n, d] K projection
    (SURVEY K6); gradients of ``att_r``/``lin_K`` follow by autograd through that contraction.
    """

    def __init__(self, in_channels, hid_dim, out_channels, num_layers, heads=1, concat=True,
                 negative_slope=0.2, dropout=0.0, bias=False, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.hidden = hid_dim // heads
        self.out_channels = out_channels
        self.heads = heads
        self.concat = concat
        self.negative_slope = negative_slope
        self.dropout = 0.
        self.aggr = 'add'
        self.fold_alpha = kwargs.pop("fold_alpha", True)
        self._raw_input = False          # set by the owning model on the conv that consumes data.x (see _sparse_rows)
        self.lin_K = nn.Linear(in_channels, self.heads * self.hidden)
        self.lin_V = nn.Linear(in_channels, self.heads * self.hidden)
        self.att_r = nn.Parameter(torch.empty(1, heads, self.hidden))
        self.rFF = MLP(in_channels=self.heads * self.hidden, hidden_channels=self.heads * self.hidden,
                       out_channels=out_channels, num_layers=num_layers, dropout=.0, Normalization='None')
        self.ln0 = nn.LayerNorm(self.heads * self.hidden)
        self.ln1 = nn.LayerNorm(self.heads * self.hidden)
        self.register_parameter('bias', None)
        self._alpha = None
        self.reset_parameters()

    def reset_parameters(self):
        # weights only: the reference never re-initialises lin_K / lin_V biases
        glorot(self.lin_K.weight)
        glorot(self.lin_V.weight)
        self.rFF.reset_parameters()
        self.ln0.reset_parameters()
        self.ln1.reset_parameters()
        nn.init.xavier_uniform_(self.att_r)

    def _fold(self) -> Tuple[Tensor, Tensor]:
        """``(w [H, in], b [H])`` with ``alpha = x w^T + b``: one kernel each way for fp32 device parameters."""
        H, C = self.heads, self.hidden
        Wk, bk = self.lin_K.weight, self.lin_K.bias
        if (Wk.is_cuda and Wk.dtype in (torch.float32, torch.bfloat16) and self.att_r.dtype == Wk.dtype
                and (bk is None or bk.dtype == Wk.dtype)):
            return dense.pma_fold(Wk, bk, self.att_r)       # fp32, or the bf16 regime: one kernel each way
        w = (Wk.view(H, C, -1) * self.att_r.view(H, C, 1)).sum(dim=1)     # [H, in]
        b = (bk.view(H, C) * self.att_r.view(H, C)).sum(dim=1)           # [H]
        return w, b

    def _logits(self, x: Tensor) -> Tensor:
        H, C = self.heads, self.hidden
        if self.fold_alpha:
            w, b = self._fold()
            # [n,in] x [in,H]: the weight gradient of this skinny Linear is a [H x n] x [n x in] product that the
            # library tiles badly (1.5 ms at n = 1M); dense.linear routes it to the split-K MFMA kernel
            hip = _on_hip(x) or (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16)
            return dense.linear(x, w, b) if hip else F.linear(x, w, b)      # (odd H / widths: the padded weight gradient)
        return (self.lin_K(x).view(-1, H, C) * self.att_r).sum(dim=-1)

    def _sparse_rows(self, x: Tensor):
        """Raw bag-of-words features in a training step: the projection runs from their non-zeros (``dense.sparse_pma_project``)."""
        if not (self.fold_alpha and _on_hip(x) and self.lin_V.weight.dtype == torch.float32 and self.att_r.dtype == torch.float32):
            return None
        if x.dim() != 2 or x.shape[1] < 256:
            return None
        # "needs no gradient" must mean "is the raw feature matrix" (MLP._wide_input has the same rule): with autograd on, a tensor
        # without requires_grad is one; a no-grad forward qualifies only on the conv the model marked as the consumer of data.x, and
        # only where the caller declared the features constant (dense.constant_features: no in-place overwrite behind a graph)
        if torch.is_grad_enabled():
            if x.requires_grad:
                return None
        elif not (self._raw_input and dense.constant_features_active()):
            return None
        if not dense.sparse_linear_supported(self.lin_V.out_features, self.heads):
            return None
        return dense.sparse_rows(x)

    def takes_pre_dropout(self, x: Tensor) -> bool:
        """``forward(x, _pre=p)`` can apply the dropout in FRONT of the conv (models.py:473) inside the projection's kernel."""
        return self._sparse_rows(x) is not None

    def project(self, x: Tensor, _pre: float = 0.0) -> Tuple[Tensor, Tensor]:
        """``(x_V, alpha_r)``: the value projection and the (folded) logits of ``x``.  ``_pre`` (internal, only after
        ``takes_pre_dropout``): the dropout in front of the conv."""
        H, C = self.heads, self.hidden
        sp = self._sparse_rows(x)
        if sp is not None:
            # raw sparse features without gradient (Citeseer: 32 of 3703 entries per row): sums over the non-zeros, both consumers of x
            # and the input dropout in one gather kernel each way
            w, b = self._fold()
            return dense.sparse_pma_project(x, sp, self.lin_V.weight, self.lin_V.bias, w, b, float(_pre) if self.training else 0.0)
        if _pre:
            raise ValueError("PMA.project(_pre=...) needs takes_pre_dropout(x)")
        fusable = _on_hip(x) and dense.fused_linear_supported(self.lin_V.in_features, self.lin_V.out_features)
        wide = _on_hip(x) and self.lin_V.bias is not None and dense.wide_linear_supported(
            self.lin_V.in_features, self.lin_V.out_features, False)
        if fusable and self.fold_alpha and dense.x6_active():
            # one autograd node for both consumers of x (bf16x6 kernels; the branches' gradients are summed in-kernel)
            w, b = self._fold()
            return dense.pma_project(x, self.lin_V.weight, self.lin_V.bias, w, b)
        if (self.fold_alpha and H <= 4 and x.dim() == 2 and x.shape[1] % 8 == 0
                and dense.linear_bf16_supported(x, self.lin_V.weight, self.lin_V.bias) and self.att_r.dtype == torch.bfloat16):
            # bf16 regime: the logits are four fp32 auxiliary columns of the value projection's kernel
            w, b = self._fold()
            return dense.pma_project_bf16(x, self.lin_V.weight, self.lin_V.bias, w, b)
        x_V = (dense.fused_norm_linear(x, None, None, self.lin_V.weight, self.lin_V.bias) if (fusable or wide)
               else _linear(self.lin_V, x))
        return x_V, self._logits(x)

    def tail(self, pooled: Tensor, _post: Optional[float] = None, _ln0_done: bool = False) -> Tensor:
        """``+att_r -> ln0 -> ln1(z + relu(rFF(z)))`` (reference layers.py:153-157) on pooled [n_t, H*C].
        ``_post`` (internal): also the ``relu -> dropout(p)`` SetGNN wraps around the conv, in ln1's pass.
        ``_ln0_done`` (internal): ``pooled`` already is ``ln0(pooled + att_r)`` (the joint pooling + ln0 node of ``forward``)."""
        H, C = self.heads, self.hidden
        hip = _on_hip(pooled) or (pooled.is_cuda and pooled.dtype == torch.bfloat16 and self.ln0.weight.dtype == torch.bfloat16)
        if not _ln0_done and self._tail_fused(pooled):
            ff = self.rFF          # round 5: ln0 / ln1 inside the two rFF Linears -- two forward kernels for the whole tail
            return dense.pma_tail(pooled, self.att_r, self.ln0.weight, self.ln0.bias, self.ln0.eps, ff.lins[0].weight, ff.lins[0].bias,
                                  ff.lins[1].weight, ff.lins[1].bias, self.ln1.weight, self.ln1.bias, self.ln1.eps,
                                  _post is not None, float(_post or 0.0) if self.training else 0.0)
        if hip and dense.ln_res_supported(H * C, pooled.dtype) and self.ln0.bias is not None and self.ln1.bias is not None:
            # the seed add rides in ln0's pass, the residual add (and the conv's relu -> dropout) in ln1's
            out = pooled if _ln0_done else dense.layer_norm_res(pooled, self.att_r, None, self.ln0.weight, self.ln0.bias, self.ln0.eps)
            ff = self.rFF
            if (dense.x6_active() and len(ff.lins) == 2 and ff._fusable(out) and ff._resident()
                    and all(isinstance(nm, nn.Identity) for nm in ff.normalizations)
                    and ff.lins[1].out_features == H * C):
                # the whole residual block as one autograd node (gradient branches of `out` summed in a kernel)
                return dense.pma_residual_ff(out, ff.lins[0].weight, ff.lins[0].bias, ff.lins[1].weight, ff.lins[1].bias,
                                             self.ln1.weight, self.ln1.bias, self.ln1.eps, _post is not None, float(_post or 0.0))
            if (len(ff.lins) == 2 and all(isinstance(nm, nn.Identity) for nm in ff.normalizations)
                    and ff.lins[1].out_features == H * C and all(lin.bias is not None for lin in ff.lins)
                    and all(dense.linear_bf16_supported(out, lin.weight, lin.bias) for lin in ff.lins)):
                # bf16 regime: the same block as one autograd node on the bf16 Linear kernels
                return dense.pma_residual_ff_bf16(out, ff.lins[0].weight, ff.lins[0].bias, ff.lins[1].weight, ff.lins[1].bias,
                                                  self.ln1.weight, self.ln1.bias, self.ln1.eps, _post is not None, float(_post or 0.0))
            z = ff(out, _post=0.0)                                      # relu(rFF(.)) in rFF's last fused epilogue
            return dense.layer_norm_res(out, None, z, self.ln1.weight, self.ln1.bias, self.ln1.eps,
                                        relu_out=_post is not None, p=float(_post or 0.0))
        out = (pooled.view(-1, H, C) + self.att_r).view(-1, H * C)     # seed + multihead (layers.py:153)
        out = _layer_norm(self.ln0, out)
        out = _layer_norm(self.ln1, out + self.rFF(out, _post=0.0))
        return out if _post is None else relu_dropout(out, _post, True)

    def _tail_fused(self, pooled: Tensor) -> bool:
        """The tail runs on ``dense.pma_tail`` (fp32 device tensors, every width 128, a 2-layer rFF without norms, affine LayerNorms,
        the fp16x3 arithmetic available)."""
        ff = self.rFF
        return (_on_hip(pooled) and pooled.dim() == 2 and len(ff.lins) == 2 and all(isinstance(nm, nn.Identity) for nm in ff.normalizations)
                and self.ln0.elementwise_affine and self.ln1.elementwise_affine and self.ln0.bias is not None and self.ln1.bias is not None
                and self.att_r.dtype == torch.float32
                and dense.pma_tail_supported(pooled, self.heads * self.hidden, ff.lins[0].weight, ff.lins[1].weight))

    def pool_tail(self, x_V: Tensor, alpha_r: Tensor, inc: Incidence, _post: Optional[float] = None):
        """``tail(pool(x_V, alpha_r))`` plus the softmax statistics: ``(out [n_t, H*C], m, l)`` (reference layers.py:145-157)."""
        H = self.heads
        hip = _on_hip(x_V) or (x_V.is_cuda and x_V.dtype == torch.bfloat16 and self.ln0.weight.dtype == torch.bfloat16
                               and self.att_r.dtype == torch.bfloat16)
        if _on_hip(x_V) and AF.pma_pool_ln0_supported(x_V, H) and self._tail_fused(x_V):
            ff = self.rFF
            return AF.pma_pool_tail(x_V, alpha_r, inc, H, self.negative_slope, self.att_r, self.ln0.weight, self.ln0.bias, self.ln0.eps,
                                    ff.lins[0].weight, ff.lins[0].bias, ff.lins[1].weight, ff.lins[1].bias, self.ln1.weight, self.ln1.bias,
                                    self.ln1.eps, _post is not None, float(_post or 0.0) if self.training else 0.0)
        if (hip and self.ln0.bias is not None and self.ln1.bias is not None and self.ln0.elementwise_affine
                and AF.pma_pool_ln0_supported(x_V, H)):
            # pooling + seed add + ln0 as one autograd node: the pooling's backward statistics come out of ln0's backward pass
            out, m, l = AF.pma_pool_ln0(x_V, alpha_r, inc, H, self.negative_slope, self.att_r, self.ln0.weight, self.ln0.bias,
                                        self.ln0.eps)
            return self.tail(out, _post, _ln0_done=True), m, l
        out, m, l = AF.pma_aggregate(x_V, alpha_r, inc, H, self.negative_slope)
        return self.tail(out, _post), m, l

    def forward(self, x, edge_index: EdgeIndex, size=None, return_attention_weights=None, _post: Optional[float] = None,
                _pre: float = 0.0):
        assert x.dim() == 2, 'Static graphs not supported in `GATConv`.'
        H = self.heads
        inc = _as_incidence(edge_index, x.shape[0])
        x_V, alpha_r = self.project(x, _pre)                  # [n_s, H*C], [n_s, H]
        out, m, l = self.pool_tail(x_V, alpha_r, inc, _post)
        if isinstance(return_attention_weights, bool):
            alpha = AF.pma_attention_weights(alpha_r, m, l, inc, self.negative_slope)
            return out, (edge_index, alpha)
        return out

    def __repr__(self):
        return '{}({}, {}, heads={})'.format(self.__class__.__name__, self.in_channels, self.out_channels, self.heads)


class HalfNLHconv(nn.Module):
    """One direction (V->E or E->V) of an AllSet layer (reference layers.py:582-658).

    ``attention=True``: delegate to :class:`PMA` (AllSetTransformer).  Otherwise Deep Sets:
    ``relu(f_dec(aggregate(dropout(relu(f_enc(x))))))`` with ``aggr`` in add|sum|mean|max|min and
    per-incidence ``norm`` weights.  ``num_layers == 0`` makes ``f_enc``/``f_dec`` identities.
    """

    def __init__(self, in_dim, hid_dim, out_dim, num_layers, dropout, Normalization='bn', InputNorm=False,
                 heads=1, attention=True):
        super().__init__()
        self.attention = attention
        self.dropout = dropout
        if self.attention:
            self.prop = PMA(in_dim, hid_dim, out_dim, num_layers, heads=heads)
        elif num_layers > 0:
            self.f_enc = MLP(in_dim, hid_dim, hid_dim, num_layers, dropout, Normalization, InputNorm)
            self.f_dec = MLP(hid_dim, hid_dim, out_dim, num_layers, dropout, Normalization, InputNorm)
        else:
            self.f_enc = nn.Identity()
            self.f_dec = nn.Identity()

    def reset_parameters(self):
        if self.attention:
            self.prop.reset_parameters()
        else:
            for f in (self.f_enc, self.f_dec):
                if not isinstance(f, nn.Identity):
                    f.reset_parameters()

    def takes_pre_dropout(self, x: Tensor) -> bool:
        if self.attention:
            return self.prop.takes_pre_dropout(x)
        return isinstance(self.f_enc, MLP) and self.f_enc.takes_pre_dropout(x)

    def forward(self, x, edge_index: EdgeIndex, norm, aggr='add', _post_dropout: Optional[float] = None, _pre_dropout: float = 0.0):
        """``_post_dropout`` (internal, used by ``SetGNN``): also apply the ``relu -> dropout(p)`` that
        ``SetGNN.forward`` wraps around every conv (models.py:475-481) inside the conv's last fused pass.  ``_pre_dropout``
        (internal, only after ``takes_pre_dropout``): the dropout ``SetGNN.forward`` puts in front of the first conv (models.py:473)."""
        post = _post_dropout is not None
        if _pre_dropout and not (self.takes_pre_dropout(x) and (aggr is not None or self.attention)):
            raise ValueError("HalfNLHconv.forward(_pre_dropout=...) needs takes_pre_dropout(x)")
        if self.attention:
            if post and self.training:      # training: the conv's relu -> dropout rides in ln1's pass (PMA.tail)
                return self.prop(x, edge_index, _post=float(_post_dropout), _pre=float(_pre_dropout))
            x = self.prop(x, edge_index, _pre=float(_pre_dropout))    # eval: keep the raw PMA output observable (forward hooks), relu separately
            return relu_dropout(x, _post_dropout, self.training) if post else x
        if aggr is None:
            raise ValueError("aggr was not passed!")
        x = self._mlp_act(self.f_enc, x, self.dropout, pre=_pre_dropout)
        inc = _as_incidence(edge_index, x.shape[0])
        x = AF.deepsets_aggregate(x, inc, norm, aggr)
        # relu(f_dec(.)); SetGNN's outer relu is idempotent on it, so its dropout can ride in the same pass
        return self._mlp_act(self.f_dec, x, _post_dropout if post else 0.0)

    def _mlp_act(self, mlp, x, p, in_cb: int = 0, out_cb: int = 0, pre: float = 0.0):
        """``dropout_p(relu(mlp(x)))``; the activation rides in the MLP's last fused kernel when there is one.
        ``in_cb`` / ``out_cb``: column-blocked input / output (only after ``mlp.blockable``; dist.py)."""
        if isinstance(mlp, MLP):
            return mlp(x, _post=p, _in_cb=in_cb, _out_cb=out_cb, _pre=pre)
        return relu_dropout(mlp(x), p, self.training)
