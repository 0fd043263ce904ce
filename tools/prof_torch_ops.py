#!/usr/bin/env python
"""Which torch (non-library) ops remain in a PMA layer step: torch.profiler table of aten ops with input shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from allset_amd import HalfNLHconv, dist as adist, synthetic
dev = torch.device("cuda:0")
n, d = 500_000, 128
hgr = synthetic.random_hypergraph(n, n, 16, seed=1, device=dev)
hg = adist.ShardedHypergraph(hgr.edge_index, n, n, 1, 0, norm=hgr.norm).build_incidences()
v2e = HalfNLHconv(d, d, d, 2, 0.5, "ln", True, heads=4, attention=True).to(dev).train()
e2v = HalfNLHconv(d, d, d, 2, 0.5, "ln", True, heads=4, attention=True).to(dev).train()
x = torch.randn(n, d, device=dev).requires_grad_(True); G = torch.randn(n, d, device=dev)
def step():
    x.grad = None
    for p in list(v2e.parameters()) + list(e2v.parameters()): p.grad = None
    out = adist.sharded_pma_layer(v2e, e2v, x, hg, dropout=0.5, training=True)
    out.backward(G)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 30]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f"{e.key:28s} n={e.count:3d} dev_us={e.device_time_total:9.0f}  shapes={str(e.input_shapes)[:110]}")
