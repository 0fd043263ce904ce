#!/usr/bin/env python
"""Is a replayed dataset-scale training step bound by the GPU or by the host's graph launch?  Host time per ``replay()`` call (no
synchronisation), wall time per replay back to back, and wall time per replay with a synchronisation after each one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from types import SimpleNamespace
import cases
from allset_amd import SetGNN
from allset_amd.graphs import GraphedTrainStep
from allset_amd.optim import FusedAdam
from allset_amd.losses import nll_log_softmax
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["cora_ds_add", "citeseer_pma_h4"]:
    case = cases.build_case(name)
    model = SetGNN(case["args"]).to(dev); model.reset_parameters()
    data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(dev), edge_index=torch.from_numpy(case["edge_index"]).to(dev), norm=torch.from_numpy(case["norm"]).to(dev))
    n = data.x.shape[0]
    y = torch.randint(0, case["args"].num_classes, (n,), device=dev); ones = torch.ones(n, device=dev)
    g = GraphedTrainStep(model, data, lambda out: nll_log_softmax(out, y, ones, n), FusedAdam(model.parameters(), lr=1e-3))
    for _ in range(20): g()
    torch.cuda.synchronize()
    N = 300
    t0 = time.perf_counter()
    for _ in range(N): g()
    t_host = (time.perf_counter() - t0) / N
    torch.cuda.synchronize(); t_b2b = (time.perf_counter() - t0) / N
    t0 = time.perf_counter()
    for _ in range(N):
        g(); torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t0) / N
    print(f"{name:18s} host call {t_host*1e6:7.1f} us   back to back {t_b2b*1e6:7.1f} us   with a sync after each {t_sync*1e6:7.1f} us")
