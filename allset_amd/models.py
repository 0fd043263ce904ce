"""``SetGNN`` -- the ``nn.Module`` surface ``train.py`` talks to (reference src/models.py:295-484).

AllSetTransformer and AllDeepSets are both this class; ``args.PMA`` selects the attention path
(reference train.py:30-42).  Constructor reads the same ``args`` attributes, registers the same
sub-modules under the same names (including the never-applied ``bnV2Es`` / ``bnE2Vs`` BatchNorms, which
still live in the ``state_dict``) and ``forward(data)`` takes the same ``data.x`` /
``data.edge_index`` / ``data.norm``.

What is different underneath: the bipartite incidence is converted once into two device CSRs
(:mod:`allset_amd.incidence`) and cached on the identity of ``data.edge_index``; the per-forward
``edge_index[1].min()`` host sync, the stacked reversed index and the ``index.max()+1`` syncs of the
reference (models.py:453-456, layers.py:174,656) do not exist after the first call.
"""
from __future__ import annotations

import weakref
from typing import Dict, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Linear, Parameter

from . import dense
from .incidence import Incidence
from .layers import MLP, HalfNLHconv, relu_dropout


class _WeightedSum(torch.autograd.Function):
    """``sum_k w[0, k] * xs[k]`` -- the GPR readout (reference models.py:468-470) as one autograd node: L + 1 element-wise passes
    forward (``mul`` then ``addcmul``), and backward one scaled copy per term plus one dot product per weight, where the chain
    ``x = x + xs[k] * w[0, k]`` cost two passes per term forward and three + a reduction per term backward."""

    @staticmethod
    def forward(ctx, w, *xs):
        ctx.save_for_backward(w, *xs)
        out = xs[0] * w[0, 0]
        for k in range(1, len(xs)):
            out = torch.addcmul(out, xs[k], w[0, k])
        return out

    @staticmethod
    def backward(ctx, g):
        w, *xs = ctx.saved_tensors
        g = g.contiguous()
        gw = None
        if ctx.needs_input_grad[0]:
            gf = g.reshape(-1)
            gw = torch.stack([torch.dot(gf, x.reshape(-1).to(gf.dtype)) for x in xs]).reshape(1, -1).to(w.dtype)
        gxs = [g * w[0, k] if ctx.needs_input_grad[1 + k] else None for k in range(len(xs))]
        return (gw, *gxs)


class SetGNN(nn.Module):
    def __init__(self, args, norm=None):
        super().__init__()
        self.All_num_layers = args.All_num_layers
        self.dropout = args.dropout
        self.aggr = args.aggregate
        self.NormLayer = args.normalization
        self.InputNorm = args.deepset_input_norm
        self.GPR = args.GPR
        self.LearnMask = args.LearnMask

        self.V2EConvs = nn.ModuleList()
        self.E2VConvs = nn.ModuleList()
        self.bnV2Es = nn.ModuleList()
        self.bnE2Vs = nn.ModuleList()

        if self.LearnMask:
            self.Importance = Parameter(torch.ones(norm.size()))

        def conv(in_dim):
            return HalfNLHconv(in_dim=in_dim, hid_dim=args.MLP_hidden, out_dim=args.MLP_hidden,
                               num_layers=args.MLP_num_layers, dropout=self.dropout,
                               Normalization=self.NormLayer, InputNorm=self.InputNorm,
                               heads=args.heads, attention=args.PMA)

        def head(in_dim):
            return MLP(in_channels=in_dim, hidden_channels=args.Classifier_hidden, out_channels=args.num_classes,
                       num_layers=args.Classifier_num_layers, dropout=self.dropout,
                       Normalization=self.NormLayer, InputNorm=False)

        if self.All_num_layers == 0:
            self.classifier = head(args.num_features)
        else:
            for i in range(self.All_num_layers):
                self.V2EConvs.append(conv(args.num_features if i == 0 else args.MLP_hidden))
                self.bnV2Es.append(nn.BatchNorm1d(args.MLP_hidden))
                self.E2VConvs.append(conv(args.MLP_hidden))
                self.bnE2Vs.append(nn.BatchNorm1d(args.MLP_hidden))
            if self.GPR:
                self.MLP = MLP(in_channels=args.num_features, hidden_channels=args.MLP_hidden,
                               out_channels=args.MLP_hidden, num_layers=args.MLP_num_layers,
                               dropout=self.dropout, Normalization=self.NormLayer, InputNorm=False)
                self.GPRweights = Linear(self.All_num_layers + 1, 1, bias=False)
            self.classifier = head(args.MLP_hidden)

        # the modules that consume data.x itself (layers.MLP._wide_input: the raw-feature route also under no_grad)
        if self.All_num_layers == 0:
            self.classifier._raw_input = True
        else:
            if isinstance(getattr(self.V2EConvs[0], "f_enc", None), MLP):
                self.V2EConvs[0].f_enc._raw_input = True
            if getattr(self.V2EConvs[0], "attention", False) and not self.GPR:
                self.V2EConvs[0].prop._raw_input = True
            if self.GPR:
                self.MLP._raw_input = True

        self._inc_cache: Dict[Tuple, Tuple[weakref.ref, Tuple[Incidence, Incidence]]] = {}

    def reset_parameters(self):
        for group in (self.V2EConvs, self.E2VConvs, self.bnV2Es, self.bnE2Vs):
            for layer in group:
                layer.reset_parameters()
        self.classifier.reset_parameters()
        if self.GPR:
            self.MLP.reset_parameters()
            self.GPRweights.reset_parameters()
        if self.LearnMask:
            nn.init.ones_(self.Importance)

    # ---- incidence handling ---------------------------------------------------------------------
    def _incidences(self, edge_index: torch.Tensor, n_v: int) -> Tuple[Incidence, Incidence]:
        """(V->E, E->V) incidences for ``data.edge_index``; built on first sight of the tensor.

        Mirrors reference models.py:453-454: hyperedge ids are re-based IN PLACE so that they start at
        0 (train.py hands them over starting at n_V; after the first call the tensor is unchanged, as in
        the reference where subtracting min()==0 is a no-op).
        """
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), n_v)
        hit = self._inc_cache.get(key)
        if hit is not None and hit[0]() is edge_index:        # same live tensor object, not merely a reused address
            return hit[1]
        if edge_index.numel() > 0:
            cidx = int(edge_index[1].min())          # one-time host sync (reference: every forward)
            if cidx != 0:
                edge_index[1] -= cidx                # reference side effect, kept
        v2e = Incidence.from_edge_index(edge_index, n_src=n_v)    # n_E = max hyperedge id + 1 (Q1)
        e2v = v2e.reversed()                                      # n_V' = max vertex id + 1 (Q1)
        self._inc_cache.clear()
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), n_v)
        self._inc_cache[key] = (weakref.ref(edge_index), (v2e, e2v))
        return v2e, e2v

    def _prefetch_planes(self, x: torch.Tensor) -> None:
        """At dataset scale every launch is a link of the step's dependent chain: the fp16 plane images of all wide (256 / 512-wide)
        Linears of the convs -- W for the forward, W^T too when a backward will follow -- in one launch up front (dense.prefetch_wide_planes;
        at benchmark scale the ten 5-us launches do not matter and the images are built where they are used)."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.shape[0] <= 65536):
            return
        ws = getattr(self, "_wide_weights", None)
        if ws is None:
            ws = []
            for conv in list(self.V2EConvs) + list(self.E2VConvs):
                for m in conv.modules():
                    if isinstance(m, nn.Linear) and dense.wide_linear_supported(m.in_features, m.out_features, False):
                        ws.append(m.weight)
            self._wide_weights = ws
        # (always: an empty list drops whatever an earlier forward left unconsumed)
        dense.prefetch_wide_planes(ws, with_transposed=torch.is_grad_enabled() and any(w.requires_grad for w in ws))

    def forward(self, data):
        """``data.x`` [n_V, F] float32, ``data.edge_index`` int64 [2, nnz] (row 0 vertex ids, row 1
        hyperedge ids), ``data.norm`` [nnz] per-incidence weights.  Returns vertex logits."""
        x, edge_index, norm = data.x, data.edge_index, data.norm
        if self.LearnMask:
            norm = self.Importance * norm
        v2e, e2v = self._incidences(edge_index, x.shape[0])
        self._prefetch_planes(x)
        if self.GPR:
            xs = [F.relu(self.MLP(x))]
            for i in range(len(self.V2EConvs)):
                x = self.V2EConvs[i](x, v2e, norm, self.aggr, _post_dropout=self.dropout)   # relu + dropout fused
                x = self.E2VConvs[i](x, e2v, norm, self.aggr)
                if self.E2VConvs[i].attention:     # (the Deep Sets conv ends in relu(f_dec(.)) already: layers.py, reference layers.py:634)
                    x = F.relu(x)
                xs.append(x)
                x = relu_dropout(x, self.dropout, self.training)     # x >= 0 already: the dropout alone, from the counter hash (no mask tensor)
            # reference models.py:468-470: x = GPRweights(stack(xs, -1)).squeeze() -- a Linear(L+1 -> 1) over the stacked layer
            # outputs, i.e. a weighted sum of them.  Written as that sum: as a matmul it is an [n*d, L+1] x [L+1, 1] product,
            # which the library runs as a strided-batched GEMM with K = L+1 -- 19 SECONDS per step at n = 1M, d = 128
            # (tools/model_step_profile.py with MODEL_ARGS=All_num_layers=2,GPR=1); same arithmetic, same parameter.
            return self.classifier(_WeightedSum.apply(self.GPRweights.weight, *xs))
        # hard-coded input dropout (models.py:473); on raw features without gradient it rides in the first conv's first kernel
        pre = 0.2 if (self.training and len(self.V2EConvs) and self.V2EConvs[0].takes_pre_dropout(x)) else 0.0
        if not pre:
            x = F.dropout(x, p=0.2, training=self.training)
        for i in range(len(self.V2EConvs)):
            # x = dropout(relu(conv(x))) (models.py:475-481); relu + dropout ride in the conv's last fused pass
            x = self.V2EConvs[i](x, v2e, norm, self.aggr, _post_dropout=self.dropout, _pre_dropout=pre if i == 0 else 0.0)
            x = self.E2VConvs[i](x, e2v, norm, self.aggr, _post_dropout=self.dropout)
        return self.classifier(x)
