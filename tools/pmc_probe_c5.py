#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, one counter per pass) at the BASELINE configs[4] per-GPU
shape: |V| = |E| = 250k, truncated-Zipf hyperedge sizes <= 4096, d = 256, bf16 storage, heads 4 -- the bench's own hypergraph
(``bench.py --dtype bf16 --d 256 --model pma --degree-dist zipf --n-per-gpu 250000``, seed 20260928 + 1).  3 launches each of
pma_fwd / pma_bwd_src in BOTH directions of the layer (V -> E and E -> V): the per-launch average is what bench.py's line reports."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allset_amd import Incidence, ops
from allset_amd.synthetic import random_hypergraph

dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 20260929
n, d, H, R = 250_000, 256, 4, 3
hg = random_hypergraph(n, n, 16, seed=seed, device=dev, dist="zipf")
inc = Incidence.from_edge_index(hg.edge_index, n_src=hg.n_v, n_dst=hg.n_e)
x = torch.randn(hg.n_v, d, device=dev).to(torch.bfloat16)
alpha = torch.randn(hg.n_v, H, device=dev)
gout = torch.randn(hg.n_e, d, device=dev).to(torch.bfloat16)
torch.cuda.synchronize()
from allset_amd.functional import _variant
for _ in range(R):
    out, m, l = ops.pma_fwd(inc.by_dst.rowptr, inc.by_dst.col, alpha, x, H, 0.2, hg.n_e,
                            variant=_variant(inc.by_dst, "pma_fwd", hg.n_e, x, H), row_order=inc.by_dst.row_order)
rev = inc.reversed(n_dst=hg.n_v)                 # E -> V: rows = vertices, sources = hyperedges (the layer's second half)
xe = torch.randn(hg.n_e, d, device=dev).to(torch.bfloat16)
ae = torch.randn(hg.n_e, H, device=dev)
gv = torch.randn(hg.n_v, d, device=dev).to(torch.bfloat16)
for _ in range(R):
    out2, m2, l2 = ops.pma_fwd(rev.by_dst.rowptr, rev.by_dst.col, ae, xe, H, 0.2, hg.n_v,
                               variant=_variant(rev.by_dst, "pma_fwd", hg.n_v, xe, H), row_order=rev.by_dst.row_order)
stats2 = ops.pma_bwd_stats(out2, gv, m2, l2)
for _ in range(R):
    ops.pma_bwd_src(rev.by_src.rowptr, rev.by_src.col, ae, xe, gv, stats2, 0.2,
                    variant=_variant(rev.by_src, "pma_bwd_src", hg.n_e, xe, H), row_order=rev.by_src.row_order)
stats = ops.pma_bwd_stats(out, gout, m, l)
for _ in range(R):
    ops.pma_bwd_src(inc.by_src.rowptr, inc.by_src.col, alpha, x, gout, stats, 0.2,
                    variant=_variant(inc.by_src, "pma_bwd_src", hg.n_v, x, H), row_order=inc.by_src.row_order)
torch.cuda.synchronize()
print("pmc_probe_c5 done: nnz", hg.nnz)
