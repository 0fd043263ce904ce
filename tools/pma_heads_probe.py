#!/usr/bin/env python
"""pma_fwd / pma_bwd_src time vs head count at fixed d = 128 (does the per-incidence logit gather, 4*H bytes from a
separate table, explain the gap to the plain segment-sum?)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from allset_amd import ops, synthetic
from allset_amd.incidence import Incidence
dev = torch.device("cuda:0")
n, d = 1_000_000, 128
hgr = synthetic.random_hypergraph(n, n, 16, seed=1, device=dev)
inc = Incidence.from_edge_index(hgr.edge_index, n_src=n)
csr, T = inc.by_dst, inc.by_src
V = torch.randn(n, d, device=dev)
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(it):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return statistics.median(ts)
print(f"segreduce sum            {timeit(lambda: ops.segreduce(0, csr.rowptr, csr.col, None, V, n)):.3f} ms")
for H in (1, 2, 4, 8, 32):
    alpha = torch.randn(n, H, device=dev)
    t1 = timeit(lambda: ops.pma_fwd(csr.rowptr, csr.col, alpha, V, H, 0.2, n, variant=1))
    out, m, l = ops.pma_fwd(csr.rowptr, csr.col, alpha, V, H, 0.2, n, variant=1)
    G = torch.randn(n, d, device=dev)
    st = ops.pma_bwd_stats(out, G, m, l)
    t2 = timeit(lambda: ops.pma_bwd_src(T.rowptr, T.col, alpha, V, G, st, 0.2))
    print(f"H={H:2d}: pma_fwd {t1:.3f} ms   pma_bwd_src {t2:.3f} ms")
