"""AllSet's edge-list preprocessing (the step immediately before the hot path) as vectorised
tensor programs that run wherever the ids live (ROCm device or host).

Same function names, ``data`` attributes and results as reference ``src/preprocessing.py``:

* ``ExtractV2E``        (:394-409)  [V|E ; E|V] block edge list -> the V->E half, sorted by vertex id
* ``Add_Self_Loops``    (:412-448)  one new singleton hyperedge per vertex that is not already alone in one
* ``norm_contruction``  (:451-469)  per-incidence ``norm``: 'all_one' (int64 ones) or 'deg_half_sym'
* ``expand_edge_index`` (:22-144)   "exclude-self" expansion: hyperedge e of size k becomes k hyperedges e_i,
                                     e_i containing every member except the i-th

The reference implements these with Python loops over vertices / hyperedges and ``i not in list`` scans --
O(n_V * k), minutes at 1M vertices.  Here each is a handful of sorts / bincounts / prefix sums.  Where the
reference's result depends on an unstable sort (``torch.sort`` / ``argsort`` ties, :398,446,141) the order within
equal vertex ids is unspecified there; these versions use stable sorts, i.e. they return one of the orders the
reference may return (tests compare per-vertex multisets).
"""
from __future__ import annotations

import torch

Tensor = torch.Tensor


def _first(v) -> int:
    """``data.n_x[0]`` / ``data.num_hyperedges[0]`` may be tensors, arrays, lists or plain ints."""
    try:
        v = v[0]
    except (TypeError, IndexError):
        pass
    return int(v)


def _sort_by_vertex(edge_index: Tensor) -> Tensor:
    order = torch.argsort(edge_index[0], stable=True)
    return edge_index[:, order].to(torch.int64)


def ExtractV2E(data):
    """Keep the vertex->hyperedge half of a ``[V|E ; E|V]`` edge list (reference preprocessing.py:394-409)."""
    edge_index = _sort_by_vertex(data.edge_index)
    num_nodes = _first(data.n_x)
    num_hyperedges = _first(data.num_hyperedges)
    if not ((num_nodes + num_hyperedges - 1) == int(data.edge_index[0].max())):
        print('num_hyperedges does not match! 1')
        return
    # sorted by row 0: the V->E half is the prefix whose source id is a vertex id
    cidx = int(torch.searchsorted(edge_index[0].contiguous(), torch.tensor(num_nodes, device=edge_index.device)))
    data.edge_index = edge_index[:, :cidx].contiguous()
    return data


def Add_Self_Loops(data):
    """Append a new singleton hyperedge for every vertex that is not already the only member of some hyperedge
    (reference preprocessing.py:412-448).  New ids continue after the largest hyperedge id, in increasing vertex
    order; ``data.totedges`` is set as in the reference."""
    edge_index = data.edge_index
    num_nodes = _first(data.n_x)
    num_hyperedges = _first(data.num_hyperedges)
    if not ((num_nodes + num_hyperedges - 1) == int(edge_index[1].max())):
        print('num_hyperedges does not match! 2')
        return
    dev = edge_index.device
    e_min = int(edge_index[1].min())
    sizes = torch.bincount(edge_index[1] - e_min)
    alone = sizes[edge_index[1] - e_min] == 1                       # incidences of size-1 hyperedges
    skip = torch.zeros(num_nodes, dtype=torch.bool, device=dev)
    skip[edge_index[0][alone]] = True
    new_v = (~skip).nonzero().reshape(-1)
    new_e = int(edge_index[1].max()) + 1 + torch.arange(new_v.numel(), device=dev, dtype=torch.int64)
    n_skipped = int(alone.sum())                                    # the reference counts list entries (:440)
    data.totedges = num_hyperedges + num_nodes - n_skipped
    edge_index = torch.cat([edge_index, torch.stack([new_v, new_e])], dim=1)
    data.edge_index = _sort_by_vertex(edge_index).contiguous()
    return data


def norm_contruction(data, option='all_one', TYPE='V2E'):
    """Per-incidence weights (reference preprocessing.py:451-469; the reference's spelling is kept).
    'all_one': int64 ones (the default train.py uses).  'deg_half_sym': D_v^-1/2 * D_e^-1/2."""
    if TYPE == 'V2E':
        if option == 'all_one':
            data.norm = torch.ones_like(data.edge_index[0])
        elif option == 'deg_half_sym':
            v, e = data.edge_index[0], data.edge_index[1]
            cidx = e.min()
            Vdeg = torch.bincount(v).to(torch.float32)
            HEdeg = torch.bincount(e - cidx).to(torch.float32)
            data.norm = Vdeg.pow(-0.5)[v] * HEdeg.pow(-0.5)[e - cidx]
    elif TYPE == 'V2V':
        raise NotImplementedError("TYPE='V2V' (gcn_norm for the clique-expansion baselines) is outside the AllSet path")
    return data


def expand_edge_index(data, edge_th=0):
    """"Exclude-self" expansion (reference preprocessing.py:22-144; ``--exclude_self`` in train.py).

    Hyperedge e = {n_1..n_k} (k > 1) becomes k hyperedges e_1..e_k with e_i = e minus n_i, i.e. node n_j is
    connected to every e_i with i != j; a size-1 hyperedge is kept as one hyperedge.  New hyperedge ids are
    consecutive from ``n_x`` in the order (original hyperedge id, member position); hyperedges larger than
    ``edge_th`` (> 0) are dropped without consuming ids.  Result sorted by node id."""
    edge_index = data.edge_index
    dev = edge_index.device
    num_nodes = _first(data.n_x)
    num_edges = int(data.totedges) if hasattr(data, 'totedges') else _first(data.num_hyperedges)
    v, e = edge_index[0], edge_index[1] - num_nodes
    valid = (e >= 0) & (e < num_edges)
    v, e = v[valid], e[valid]
    order = torch.argsort(e, stable=True)                            # members of a hyperedge in edge-list order
    v, e = v[order], e[order]
    sizes = torch.bincount(e, minlength=num_edges)
    keep_e = sizes > 0
    if edge_th > 0:
        keep_e &= sizes <= edge_th
    ids_used = torch.where(keep_e, sizes, torch.zeros_like(sizes))   # a kept hyperedge of size k consumes k ids
    base = num_nodes + torch.cumsum(ids_used, 0) - ids_used         # first new id of each original hyperedge
    start = torch.cumsum(sizes, 0) - sizes
    pos = torch.arange(e.numel(), device=dev) - start[e]             # member position j inside its hyperedge
    keep_inc = keep_e[e]
    v, e, pos = v[keep_inc], e[keep_inc], pos[keep_inc]
    k = sizes[e]
    # every (hyperedge, member j) emits k candidates i = 0..k-1; drop i == j unless the hyperedge is a singleton
    rep_v = v.repeat_interleave(k)
    rep_e = e.repeat_interleave(k)
    rep_j = pos.repeat_interleave(k)
    first = torch.cumsum(k, 0) - k
    i = torch.arange(rep_v.numel(), device=dev) - first.repeat_interleave(k)
    keep = (i != rep_j) | (sizes[rep_e] == 1)
    new_v = rep_v[keep]
    new_e = base[rep_e[keep]] + i[keep]
    data.edge_index = _sort_by_vertex(torch.stack([new_v, new_e])).contiguous()
    return data
