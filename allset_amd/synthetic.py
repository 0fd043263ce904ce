"""Seeded synthetic hypergraphs of stated |V| / |E| / degree, generated on the device
(BASELINE.json configs[2..4]).

Conventions follow what the reference's loaders + preprocessing hand to ``SetGNN``: members of a
hyperedge are distinct (the loaders ``coalesce`` duplicates away, load_other_datasets.py:178-181),
incidences are emitted sorted by vertex id (preprocessing.py:398,446-447), ``norm`` is all-ones int64
(preprocessing.py:454).  Hyperedge ids start at ``e_base`` (0 by default; ``n_v`` reproduces the layout
train.py passes, which ``SetGNN.forward`` re-bases).
"""
from __future__ import annotations

from types import SimpleNamespace


import torch


def _distinct_members(n_rows: int, k: int, n_v: int, gen: torch.Generator, device) -> torch.Tensor:
    """[n_rows, k] vertex ids, uniform, distinct within each row (rejection on the rare duplicate rows)."""
    members = torch.randint(n_v, (n_rows, k), generator=gen, device=device, dtype=torch.int64)
    if k > n_v:
        raise ValueError("hyperedge size exceeds the number of vertices")
    for _ in range(64):
        s, _ = members.sort(dim=1)
        dup = (s[:, 1:] == s[:, :-1]).any(dim=1)
        n_dup = int(dup.sum())
        if n_dup == 0:
            break
        members[dup] = torch.randint(n_v, (n_dup, k), generator=gen, device=device, dtype=torch.int64)
    return members


def hyperedge_sizes(n_e: int, degree: float, dist: str, gen: torch.Generator, device, max_degree: int = 4096,
                    zipf_a: float = 2.0) -> torch.Tensor:
    """int64[n_e] hyperedge sizes.  'fixed': all == degree.  'poisson': Poisson(degree) clipped to >= 1.
    'zipf': truncated power law P(s) ~ s^-a on [1, max_degree], rescaled so the mean is ~degree."""
    if dist == "fixed":
        return torch.full((n_e,), int(degree), dtype=torch.int64, device=device)
    if dist == "poisson":
        lam = torch.full((n_e,), float(degree), device=device)
        return torch.poisson(lam, generator=gen).clamp_(min=1).to(torch.int64)
    if dist == "zipf":
        s = torch.arange(1, max_degree + 1, device=device, dtype=torch.float64)
        pmf = s.pow(-zipf_a)
        pmf /= pmf.sum()
        draw = torch.multinomial(pmf.float(), n_e, replacement=True, generator=gen) + 1      # in [1, max_degree]
        mean = float(draw.double().mean())
        scaled = (draw.double() * (degree / mean)).round().clamp_(1, max_degree)
        return scaled.to(torch.int64)
    raise ValueError(f"unknown degree distribution {dist!r}")


def random_hypergraph(n_v: int, n_e: int, degree: float = 16, seed: int = 0, device="cuda", dist: str = "fixed",
                      max_degree: int = 4096, e_base: int = 0, e_offset: int = 0, sort_by_vertex: bool = True,
                      zipf_a: float = 2.0, locality: float = 0.0, home=None) -> SimpleNamespace:
    """``n_e`` hyperedges over ``n_v`` vertices.  Returns ``SimpleNamespace(edge_index, norm, n_v, n_e,
    nnz, seed, dist)`` with ``edge_index`` int64 [2, nnz] (row 0 vertex ids, row 1 hyperedge ids
    ``e_base + e_offset + [0, n_e)``) on ``device``.

    ``e_offset`` lets one rank of a sharded run generate only its own block of hyperedges (their members
    still range over the global vertex set).

    ``locality`` in [0, 1) with ``home = (block, n_blocks)``: each membership is re-drawn, with that probability, from the
    home block of vertices ``[block * n_v / n_blocks, (block + 1) * n_v / n_blocks)`` -- a hypergraph whose partitions have few
    boundary vertices (what the row partition's boundary-vertex exchange, ``allset_amd.dist.Halo``, is for).  0 = the uniform
    hypergraph of the benchmark.  (Re-drawn members may repeat inside a hyperedge; duplicates are coalesced.)
    """
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    if dist == "fixed":
        k = int(degree)
        members = _distinct_members(n_e, k, n_v, gen, device)
        v = members.reshape(-1)
        e = torch.arange(n_e, device=device, dtype=torch.int64).repeat_interleave(k)
    else:
        sizes = hyperedge_sizes(n_e, degree, dist, gen, device, max_degree, zipf_a).clamp_(max=n_v)
        e = torch.arange(n_e, device=device, dtype=torch.int64).repeat_interleave(sizes)
        v = torch.randint(n_v, (int(e.numel()),), generator=gen, device=device, dtype=torch.int64)
        key = torch.unique(e * n_v + v)            # coalesce duplicate memberships (as the reference loaders do)
        e, v = key // n_v, key % n_v
    if locality > 0.0:
        if home is None:
            raise ValueError("locality needs home = (block, n_blocks)")
        blk, nb = int(home[0]), int(home[1])
        per = n_v // nb
        local = torch.rand(v.shape, generator=gen, device=device) < float(locality)
        redraw = blk * per + torch.randint(per, v.shape, generator=gen, device=device, dtype=torch.int64)
        v = torch.where(local, redraw, v)
        key = torch.unique(e * n_v + v)
        e, v = key // n_v, key % n_v
    if sort_by_vertex:
        order = torch.argsort(v, stable=True)
        v, e = v[order], e[order]
    edge_index = torch.stack([v, e + (e_base + e_offset)], dim=0).contiguous()
    norm = torch.ones(edge_index.shape[1], dtype=torch.int64, device=device)
    return SimpleNamespace(edge_index=edge_index, norm=norm, n_v=n_v, n_e=n_e, nnz=int(edge_index.shape[1]),
                           seed=int(seed), dist=dist)
