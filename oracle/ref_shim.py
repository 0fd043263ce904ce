"""Run the *real* reference modules (``/root/reference/src/layers.py``, ``models.py``) in the build
container.  TEST INFRASTRUCTURE ONLY -- container-only: /root/reference does not exist on the GPU box.

The reference imports ``torch_scatter``, ``torch_geometric`` and ``ipdb`` (layers.py:22-25,
utils.py:3), none of which is installed here.  ``install()`` registers small stand-in modules in
``sys.modules`` that restate the behaviour of the pinned releases (torch-scatter 2.0.4,
torch-geometric 1.6.3; README.md:18-22) for exactly the entry points the path touches:

* ``torch_scatter.scatter`` / ``scatter_add``       (call sites layers.py:194,656)
* ``torch_geometric.utils.softmax``                 (call site  layers.py:174)
* ``torch_geometric.nn.conv.MessagePassing``        (``propagate`` -> ``message`` -> ``aggregate``;
                                                      call sites layers.py:145,633)

The stand-ins deliberately share no code with ``oracle/allset_oracle.py`` so that agreement between
the two is evidence, not tautology.  Nothing here is copied from the reference or from the
third-party wheels; the semantics are those documented for the pinned versions [external].
"""
from __future__ import annotations

import inspect
import os
import sys
import types
import warnings
from typing import Optional, Tuple

import torch

REFERENCE_SRC = "/root/reference/src"


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_SRC, "layers.py"))


# ---- torch_scatter stand-in ------------------------------------------------------------


def _expand_index(index: torch.Tensor, src: torch.Tensor, dim: int) -> torch.Tensor:
    shape = [1] * src.dim()
    shape[dim] = -1
    return index.view(shape).expand_as(src)


def _scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    assert out is None, "out= is not used on the AllSet path"
    dim = dim % src.dim()
    if dim_size is None:
        dim_size = (int(index.max()) + 1) if index.numel() > 0 else 0
    full_shape = list(src.shape)
    full_shape[dim] = dim_size
    idx = _expand_index(index, src, dim)
    base = src.new_zeros(full_shape)
    if reduce in ("sum", "add"):
        return base.scatter_add_(dim, idx, src)
    if reduce == "mean":
        total = base.scatter_add_(dim, idx, src)
        ones = torch.ones(index.shape[0], dtype=src.dtype, device=src.device)
        count = src.new_zeros(dim_size).scatter_add_(0, index, ones).clamp_(min=1)
        cshape = [1] * src.dim()
        cshape[dim] = dim_size
        return total / count.view(cshape)
    if reduce == "max":
        return base.scatter_reduce(dim, idx, src, "amax", include_self=False)
    if reduce == "min":
        return base.scatter_reduce(dim, idx, src, "amin", include_self=False)
    raise ValueError(reduce)


def _scatter_add(src, index, dim=-1, out=None, dim_size=None):
    return _scatter(src, index, dim, out, dim_size, "sum")


# ---- torch_geometric stand-ins ------------------------------------------------------------


def _pyg_softmax(src, index, ptr=None, num_nodes=None):
    assert ptr is None
    n = int(index.max()) + 1 if num_nodes is None else int(num_nodes)
    shifted = src - _scatter(src, index, 0, None, n, "max").index_select(0, index)
    num = shifted.exp()
    den = _scatter(num, index, 0, None, n, "sum").index_select(0, index)
    return num / (den + 1e-16)


class _MessagePassing(torch.nn.Module):
    """Template-method core of PyG 1.6.3's MessagePassing for a Tensor ``edge_index`` and
    ``flow='source_to_target'``: arguments of ``message``/``aggregate`` that end in ``_j`` are
    gathered with ``edge_index[0]``, ``_i`` with ``edge_index[1]``; ``index`` is ``edge_index[1]``."""

    _reserved = ("edge_index", "adj_t", "edge_index_i", "edge_index_j", "size", "size_i", "size_j",
                 "ptr", "index", "dim_size")

    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        assert flow == "source_to_target"
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        size = [None, None] if size is None else list(size)
        msg_params = list(inspect.signature(self.message).parameters)
        agg_params = list(inspect.signature(self.aggregate).parameters)[1:]
        pool = {}
        for name in msg_params + agg_params:
            if name in self._reserved:
                continue
            if name.endswith("_j") or name.endswith("_i"):
                side = 0 if name.endswith("_j") else 1
                t = kwargs[name[:-2]]
                n = t.size(self.node_dim)
                if size[side] is None:
                    size[side] = n
                elif size[side] != n:
                    raise ValueError("size mismatch in propagate")
                pool[name] = t.index_select(self.node_dim, edge_index[side])
            elif name in kwargs:
                pool[name] = kwargs[name]
        size_i = size[1] if size[1] is not None else size[0]
        size_j = size[0] if size[0] is not None else size[1]
        pool.update(edge_index_j=edge_index[0], edge_index_i=edge_index[1], index=edge_index[1],
                    ptr=None, adj_t=None, size=size, size_i=size_i, size_j=size_j, dim_size=size_i)
        msg = self.message(**{k: pool[k] for k in msg_params if k in pool})
        return self.aggregate(msg, **{k: pool[k] for k in agg_params if k in pool})

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        return _scatter(inputs, index, dim=self.node_dim, dim_size=dim_size, reduce=self.aggr)


def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = False


def install() -> None:
    """Register the stand-in modules (idempotent)."""
    global _installed
    if _installed:
        return
    _module("torch_scatter", scatter=_scatter, scatter_add=_scatter_add)
    tg = _module("torch_geometric")
    tg.nn = _module("torch_geometric.nn")
    tg.nn.conv = _module("torch_geometric.nn.conv", MessagePassing=_MessagePassing,
                         GCNConv=type("GCNConv", (), {}), GATConv=type("GATConv", (), {}))
    _module("torch_geometric.nn.conv.gcn_conv", gcn_norm=None)      # imported (never called) by preprocessing.py:20
    tg.utils = _module("torch_geometric.utils", softmax=_pyg_softmax)
    tg.typing = _module("torch_geometric.typing", Adj=torch.Tensor, Size=Optional[Tuple[int, int]],
                        OptTensor=Optional[torch.Tensor])
    _module("ipdb")
    _installed = True


def import_reference_preprocessing():
    """The reference's ``preprocessing`` module (host-side edge-list surgery, preprocessing.py:394-469,22-144)."""
    if not available():
        raise RuntimeError("/root/reference is not present (this only works in the build container)")
    install()
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import preprocessing as ref_pre   # noqa: E402
    return ref_pre


class _Data:
    """torch_geometric.data.Data as the reference's loaders use it (load_other_datasets.py:81-85,113-117): an attribute bag
    constructed from keyword tensors."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def _coalesce(index, value, m, n, op="add"):
    """torch_sparse.coalesce (pinned torch-sparse 0.6.x, reference README.md:18-22) for ``value=None``: the edge list sorted
    by (row, col) with duplicates removed.  [external: restated from the library's documented behaviour]"""
    assert value is None
    key = torch.unique(index[0] * int(n) + index[1])                     # sorted
    return torch.stack([key // int(n), key % int(n)]), None


def import_reference_loaders():
    """The reference's ``load_other_datasets`` module (raw-file readers, load_other_datasets.py:32-391)."""
    if not available():
        raise RuntimeError("/root/reference is not present (this only works in the build container)")
    install()
    tg = sys.modules["torch_geometric"]
    tg.data = _module("torch_geometric.data", Data=_Data)
    _module("torch_sparse", coalesce=_coalesce)
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import load_other_datasets as ref_load   # noqa: E402
    return ref_load


def import_reference():
    """Return the reference's ``(layers, models)`` modules, imported from /root/reference/src."""
    if not available():
        raise RuntimeError("/root/reference is not present (this only works in the build container)")
    install()
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", SyntaxWarning)   # `is` with a literal, layers.py:568,617,619
        import layers as ref_layers   # noqa: E402
        import models as ref_models   # noqa: E402
    return ref_layers, ref_models
