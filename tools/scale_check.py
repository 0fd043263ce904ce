#!/usr/bin/env python
"""One-off scale check beyond the bench size: |V| = |E| = 8M, 16 members per hyperedge (nnz = 128M), d = 128 -- the
global problem of BASELINE configs[3] on ONE GPU.  Exercises 64-bit offsets (n*d = 2^30 elements, 4 GiB matrices) with
the size-independent properties of tests/test_gpu_fullsize.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allset_amd import Incidence, deepsets_aggregate, pma_aggregate
from allset_amd.synthetic import random_hypergraph
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
t0 = time.time()
hg = random_hypergraph(n, n, 16, seed=77, device=dev)
v2e = Incidence.from_edge_index(hg.edge_index, n_src=n, n_dst=n)
e2v = v2e.reversed(n_dst=n)
torch.cuda.synchronize(); print(f"built nnz={hg.nnz} in {time.time()-t0:.1f}s  max_deg E/V = {v2e.by_dst.max_deg}/{v2e.by_src.max_deg}")
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(n, 128, device=dev, generator=g)
y = torch.randn(n, 128, device=dev, generator=g)
e = deepsets_aggregate(x, v2e, None, "add")
deg_v = (v2e.by_src.rowptr[1:] - v2e.by_src.rowptr[:-1]).double()
lhs, rhs = e.double().sum(0), (x.double() * deg_v[:, None]).sum(0)
print("conservation max rel err:", float(((lhs - rhs).abs() / (rhs.abs() + 1)).max()))
xt = deepsets_aggregate(y, e2v, None, "add")
a, b = (e.double() * y.double()).sum(), (x.double() * xt.double()).sum()
print("adjoint rel err:", abs(float(a - b)) / float(a.abs() + b.abs()))
# last rows are reached correctly (offsets past 4 GiB)
row = n - 1
s, t = int(v2e.by_dst.rowptr[row]), int(v2e.by_dst.rowptr[row + 1])
ref = x[v2e.by_dst.col[s:t].long()].sum(0)
print("last-row max abs err:", float((e[row] - ref).abs().max()))
alpha = torch.randn(n, 4, device=dev, generator=g)
out, m, l = pma_aggregate(torch.ones(n, 128, device=dev), alpha, v2e, 4, 0.2)
print("pma convexity max err:", float((out - 1).abs().max()), " min l:", float(l.min()))
xr = x.clone().requires_grad_(True)
o2, _, _ = pma_aggregate(xr, alpha.clone().requires_grad_(True), v2e, 4, 0.2)
o2.sum().backward()
print("pma grad column sums (should equal n per column: each target row's weights sum to 1):", float(xr.grad.sum(0).mean()), n)
print("peak memory GB:", torch.cuda.max_memory_allocated() / 2**30)
