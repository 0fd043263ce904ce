#!/usr/bin/env python
"""Does slicing the gather by columns make the gathered table fit the 256 MiB Infinity Cache?  Times segreduce(sum) on
the C3 shape as one d=128 pass vs 2 x d=64 vs 4 x d=32 column slices (same total algorithmic bytes except the index
re-reads)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from allset_amd import ops, synthetic
from allset_amd.incidence import Incidence
dev = torch.device("cuda:0")
n, deg, d = 1_000_000, 16, 128
hgr = synthetic.random_hypergraph(n, n, deg, seed=1, device=dev)
inc = Incidence.from_edge_index(hgr.edge_index, n_src=n)
csr = inc.by_dst
x = torch.randn(n, d, device=dev)
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(it):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return statistics.median(ts)
full = timeit(lambda: ops.segreduce(0, csr.rowptr, csr.col, None, x, n))
print(f"1 x d=128: {full:.3f} ms")
for parts in (2, 4):
    w = d // parts
    t = timeit(lambda: [ops.segreduce(0, csr.rowptr, csr.col, None, x[:, k * w:(k + 1) * w], n) for k in range(parts)])
    print(f"{parts} x d={w}: {t:.3f} ms total")
xs = x[:, :64].contiguous()
print(f"1 x d=64 contiguous table (256 MB): {timeit(lambda: ops.segreduce(0, csr.rowptr, csr.col, None, xs, n)):.3f} ms")
xs = x[:, :32].contiguous()
print(f"1 x d=32 contiguous table (128 MB): {timeit(lambda: ops.segreduce(0, csr.rowptr, csr.col, None, xs, n)):.3f} ms")
