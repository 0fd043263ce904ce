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

# ---- dense tail at the same row count (8M x 128): sampled rows against float64, additivity of the weight gradient
from allset_amd import dense
n8 = 8_000_000 + 5
g = torch.Generator(device=dev).manual_seed(3)
xx = torch.randn(n8, 128, device=dev, generator=g)
W = torch.randn(128, 128, device=dev, generator=g) / 128 ** 0.5
bb = torch.randn(128, device=dev, generator=g)
gam = 1 + 0.2 * torch.randn(128, device=dev, generator=g); bet = 0.3 * torch.randn(128, device=dev, generator=g)
yy, st = dense.fused_linear_fwd(xx, W, bb, gam, bet, 1e-5, False, 0.0, 0, True, 0.0, 0)
rows = torch.cat([torch.arange(0, 32, device=dev), torch.randint(0, n8, (2000,), device=dev, generator=g), torch.arange(n8 - 32, n8, device=dev)])
ref = torch.relu(torch.nn.functional.layer_norm(xx[rows].double(), (128,), gam.double(), bet.double(), 1e-5) @ W.double().t() + bb.double())
print("dense fwd @8M rows: max abs err on sampled rows", float((yy[rows].double() - ref).abs().max()))
GG = torch.randn(n8, 128, device=dev, generator=g)
gx, dg, db = dense.fused_linear_bwd(GG, yy, 0.0, W, xx, st, gam, False, 0.0, 0)
xr = xx[rows].double().requires_grad_(True)
r2 = torch.relu(torch.nn.functional.layer_norm(xr, (128,), gam.double(), bet.double(), 1e-5) @ W.double().t() + bb.double())
(r2 * GG[rows].double()).sum().backward()
print("dense bwd @8M rows: max abs err on sampled rows", float((gx[rows].double() - xr.grad).abs().max()))
gw, gb = dense.wgrad(GG, xx)
h = n8 // 2 + 7
gw1, _ = dense.wgrad(GG[:h], xx[:h]); gw2, _ = dense.wgrad(GG[h:], xx[h:])
print("wgrad additivity @8M rows: rel err", float((gw1 + gw2 - gw).abs().max() / gw.abs().max()))
print("peak memory GB:", torch.cuda.max_memory_allocated() / 2 ** 30)
