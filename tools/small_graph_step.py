#!/usr/bin/env python
"""Per-step wall time of a full training step (fwd + loss + bwd + Adam) and of an eval forward at dataset scale
(Cora-/Citeseer-shaped parity cases): launch-bound regime."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from types import SimpleNamespace
import cases
from allset_amd import SetGNN
dev = torch.device("cuda:0")
import gc
for name in ("cora_ds_add", "citeseer_pma_h4"):
    case = cases.build_case(name)
    model = SetGNN(case["args"]).to(dev)
    model.reset_parameters()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(dev), edge_index=torch.from_numpy(case["edge_index"]).to(dev),
                           norm=torch.from_numpy(case["norm"]).to(dev))
    n = data.x.shape[0]
    y = torch.randint(0, case["args"].num_classes, (n,), device=dev)
    gc.collect(); gc.freeze()      # (generation-2 passes over the test fixtures' object graph otherwise land in the eager loops: 4x)
    def step():
        model.train(); opt.zero_grad()
        out = F.log_softmax(model(data), dim=1)
        loss = F.nll_loss(out, y); loss.backward(); opt.step()
    def evalf():
        model.eval()
        with torch.no_grad():
            return model(data)
    for fn, label in ((step, "train step"), (evalf, "eval forward")):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"{name:18s} {label:13s} {dt*1e3:7.3f} ms")
    # the same two loops as hipGraph replays (allset_amd/graphs.py)
    from allset_amd.graphs import GraphedForward, GraphedTrainStep
    from allset_amd.optim import FusedAdam
    opt_c = FusedAdam(model.parameters(), lr=1e-3)
    from allset_amd.losses import nll_log_softmax
    ones = torch.ones(n, device=dev)
    gstep = GraphedTrainStep(model, data, lambda out: nll_log_softmax(out, y, ones, n), opt_c)   # the driver's loss (allset_amd/train.py)
    gfwd = GraphedForward(model, data)
    gfwd_c = GraphedForward(model, data, constant_features=True)      # as allset_amd/train.py captures its per-epoch evaluation
    for fn, label in ((gstep, "train step (graph)"), (gfwd, "eval forward (graph)"), (gfwd_c, "eval forward (graph, constant features)")):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(f"{name:18s} {label:20s} {dt*1e3:7.3f} ms")
