#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, one counter per pass) that calibrate
and measure HBM traffic of segreduce_kernel.  Launch order (each kernel launched 3x):
  1. copy      : torch clone of a 1 GiB f32 tensor             -> known 1 GiB read + 1 GiB write
  2. noreuse   : segreduce over an identity-like incidence (row t gathers rows 16t..16t+15 of a 16M x 128
                 table): every gathered byte is compulsory     -> known nnz*512 B read, n_t*512 B write
  3. c3_v2e    : segreduce, V->E of the bench hypergraph (|V|=|E|=1M, deg 16, d=128)
  4. c3_e2v    : segreduce, E->V (the transposed CSR)
  5. pma       : pma_fwd, pma_bwd_stats, pma_bwd_src on the same hypergraph (heads 4) -- the AllSetTransformer passes
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allset_amd import Incidence, ops
from allset_amd.synthetic import random_hypergraph

dev = torch.device("cuda:0")
R = 3
a = torch.randn(256 * 1024 * 1024, device=dev)
for _ in range(R):
    b = a.clone()
del a, b
n_t, k, d = 1_000_000, 16, 128
table = torch.randn(n_t * k, d, device=dev)
rowptr = (torch.arange(n_t + 1, device=dev, dtype=torch.int64) * k).to(torch.int32)
col = torch.arange(n_t * k, device=dev, dtype=torch.int32)
for _ in range(R):
    ops.segreduce(0, rowptr, col, None, table, n_t)
del table
hg = random_hypergraph(1_000_000, 1_000_000, 16, seed=20260929, device=dev)
inc = Incidence.from_edge_index(hg.edge_index, n_src=hg.n_v, n_dst=hg.n_e)
x = torch.randn(hg.n_v, d, device=dev)
y = torch.randn(hg.n_e, d, device=dev)
torch.cuda.synchronize()
for _ in range(R):
    ops.segreduce(0, inc.by_dst.rowptr, inc.by_dst.col, None, x, hg.n_e)
for _ in range(R):
    ops.segreduce(0, inc.by_src.rowptr, inc.by_src.col, None, y, hg.n_v)
torch.cuda.synchronize()
H = 4
alpha = torch.randn(hg.n_v, H, device=dev)
gout = torch.randn(hg.n_e, d, device=dev)
for _ in range(R):
    out, m, l = ops.pma_fwd(inc.by_dst.rowptr, inc.by_dst.col, alpha, x, H, 0.2, hg.n_e, variant=1, row_order=inc.by_dst.row_order)
for _ in range(R):
    stats = ops.pma_bwd_stats(out, gout, m, l)
for _ in range(R):
    ops.pma_bwd_src(inc.by_src.rowptr, inc.by_src.col, alpha, x, gout, stats, 0.2, variant=1, row_order=inc.by_src.row_order)
torch.cuda.synchronize()
print("pmc_probe done")
