#!/usr/bin/env python
"""Which torch (non-library) ops remain in a PMA layer step: torch.profiler table of aten ops with input shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from allset_amd import HalfNLHconv, dist as adist, synthetic
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=500_000); ap.add_argument("--d", type=int, default=128)
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"]); ap.add_argument("--degree-dist", default="fixed")
ap.add_argument("--all", action="store_true", help="every device kernel, not only aten ops")
ap.add_argument("--model", default="pma", choices=["pma", "deepsets"])
a = ap.parse_args()
dev = torch.device("cuda:0")
n, d = a.n, a.d
tdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
hgr = synthetic.random_hypergraph(n, n, 16, seed=1, device=dev, dist=a.degree_dist)
hg = adist.ShardedHypergraph(hgr.edge_index, n, n, 1, 0, norm=hgr.norm).build_incidences()
attn = a.model == "pma"
v2e = HalfNLHconv(d, d, d, 2, 0.5, "ln", True, heads=4, attention=attn).to(dev).to(tdt).train()
e2v = HalfNLHconv(d, d, d, 2, 0.5, "ln", True, heads=4, attention=attn).to(dev).to(tdt).train()
x = torch.randn(n, d, device=dev).to(tdt).requires_grad_(True); G = torch.randn(n, d, device=dev).to(tdt)
def step():
    x.grad = None
    for p in list(v2e.parameters()) + list(e2v.parameters()): p.grad = None
    out = (adist.sharded_pma_layer(v2e, e2v, x, hg, dropout=0.5, training=True) if attn else
           adist.sharded_deepsets_layer(v2e, e2v, x, hg, aggr="add", dropout=0.5, training=True))
    out.backward(G)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 2]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    print(f"{e.key:28s} n={e.count:3d} dev_us={e.device_time_total:9.0f}  shapes={str(e.input_shapes)[:110]}")
if a.all:
    ks = [e for e in prof.key_averages() if e.device_time_total > 20 and not e.key.startswith("aten::")]
    ks.sort(key=lambda e: -e.device_time_total)
    tot = sum(e.device_time_total for e in ks)
    print(f"--- device kernels, total {tot:.0f} us")
    for e in ks[:40]:
        print(f"{e.key[:90]:90s} n={e.count:3d} dev_us={e.device_time_total:9.0f}")
