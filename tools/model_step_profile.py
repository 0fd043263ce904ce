#!/usr/bin/env python
"""One full SetGNN training step (models.SetGNN as train.py drives it: input dropout, V->E->V layers, classifier, loss, Adam) at
the headline scale -- the layer bench (bench.py) times one HalfNLHconv pair; this is the user-facing path around it.
usage: model_step_profile.py [deepsets|pma] [n] [features]      (run under rocprofv3 --kernel-trace --stats for the inventory)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from types import SimpleNamespace
import cases
from allset_amd import SetGNN
from allset_amd.losses import nll_log_softmax, split_mask
from allset_amd.optim import FusedAdam
from allset_amd.synthetic import random_hypergraph
mode = sys.argv[1] if len(sys.argv) > 1 else "deepsets"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
f = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda:0")
over = {}
for kv in os.environ.get("MODEL_ARGS", "").split(","):        # e.g. MODEL_ARGS=All_num_layers=2,GPR=1,LearnMask=1
    if "=" in kv:
        k, v = kv.split("=")
        over[k] = int(v) if v.lstrip("-").isdigit() else v
        if k in ("GPR", "LearnMask"):
            over[k] = bool(int(v))
args = cases.make_args("pma_h4" if mode == "pma" else "ds_add", f, 128, 10, **over)
hg = random_hypergraph(n, n, 16, seed=3, device=dev)
ei = hg.edge_index.clone()
ei[1] += n                                                     # the reference's layout: hyperedge ids follow the vertex ids
data = SimpleNamespace(x=torch.randn(n, f, device=dev), edge_index=ei, norm=torch.ones(ei.shape[1], device=dev),
                       y=torch.randint(0, 10, (n,), device=dev))
model = (SetGNN(args, data.norm) if getattr(args, "LearnMask", False) else SetGNN(args)).to(dev)
if os.environ.get("MODEL_DTYPE") == "bf16":                    # bf16 storage end to end (BASELINE configs[4] regime)
    model = model.to(torch.bfloat16)
    data.x = data.x.to(torch.bfloat16)
model.reset_parameters()
opt = FusedAdam(model.parameters(), lr=1e-3)
idx = torch.randperm(n, device=dev)[: n // 2]
mask, cnt = split_mask(idx, n), idx.numel()


def step():
    model.train(); opt.zero_grad(set_to_none=True)
    loss = nll_log_softmax(model(data), data.y, mask, cnt)
    loss.backward(); opt.step()


for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 10
for _ in range(K): step()
torch.cuda.synchronize()
print(f"{mode} SetGNN step at |V|=|E|={n}, {f} features, hidden 128: {(time.perf_counter() - t0) / K * 1e3:.2f} ms")
