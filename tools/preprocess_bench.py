#!/usr/bin/env python
"""Wall time of the preprocessing chain (ExtractV2E -> Add_Self_Loops -> norm_contruction('deg_half_sym') -> CSR pair) at
the bench scale (|V| = |E| = 1M, nnz = 16M), on the device and on the host, next to one training step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from types import SimpleNamespace
import torch
from allset_amd import preprocessing as P, synthetic
from allset_amd.incidence import Incidence
n = 1_000_000
hg = synthetic.random_hypergraph(n, n, 16, seed=3, device="cuda")
v, e = hg.edge_index[0], hg.edge_index[1] - hg.edge_index[1].min() + n
block = torch.cat([torch.stack([v, e]), torch.stack([e, v])], dim=1)           # the [V|E ; E|V] list the loaders produce
block = block[:, torch.randperm(block.shape[1], device=block.device)]
for dev in ("cuda", "cpu"):
    b = block.to(dev)
    for rep in range(2):
        data = SimpleNamespace(edge_index=b.clone(), n_x=[n], num_hyperedges=[n])
        if dev == "cuda": torch.cuda.synchronize()
        t0 = time.perf_counter()
        data = P.ExtractV2E(data); t1 = time.perf_counter()
        data = P.Add_Self_Loops(data); t2 = time.perf_counter()
        data = P.norm_contruction(data, option="deg_half_sym")
        if dev == "cuda": torch.cuda.synchronize()
        t3 = time.perf_counter()
        t4 = t3
        if dev == "cuda":
            ei = data.edge_index.clone(); ei[1] -= ei[1].min()
            inc = Incidence.from_edge_index(ei, n_src=n); inc.reversed(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"{dev:5s} ExtractV2E {1e3*(t1-t0):8.1f} ms  Add_Self_Loops {1e3*(t2-t1):8.1f} ms  norm {1e3*(t3-t2):8.1f} ms"
          + (f"  CSR pair {1e3*(t4-t3):8.1f} ms" if dev == "cuda" else "") + f"   nnz after self-loops {data.edge_index.shape[1]}")
