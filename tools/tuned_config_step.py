#!/usr/bin/env python
"""Graphed training step / eval forward at the reference's TUNED AllSetTransformer widths (run_AllSetTransformer.sh: Cora MLP_hidden 256,
4 heads; Citeseer 512, 8 heads) on the Cora- / Citeseer-shaped stand-in data, with the first conv's projection from the non-zeros of the
bag-of-words rows (dense.sparse_pma_project) and -- for comparison -- with that path switched off (library GEMMs over the raw rows)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from types import SimpleNamespace
import cases
from allset_amd import SetGNN, dense
from allset_amd.graphs import GraphedTrainStep, GraphedForward
from allset_amd.optim import FusedAdam
from allset_amd.losses import nll_log_softmax
dev = torch.device("cuda:0")
CONFIGS = [("cora_ds_add", dict(MLP_hidden=256, heads=4), "Cora shape, AllSetTransformer 256 / 4 heads"),
           ("citeseer_pma_h4", dict(MLP_hidden=512, heads=8), "Citeseer shape, AllSetTransformer 512 / 8 heads"),
           ("citeseer_pma_h4", dict(MLP_hidden=256, heads=8), "Citeseer shape, AllSetTransformer 256 / 8 heads")]
real = dense.sparse_linear_supported
for name, over, label in CONFIGS:
    case = cases.build_case(name)
    ref_args = cases.build_case("citeseer_pma_h4")["args"]
    args = SimpleNamespace(**{**vars(ref_args), "num_features": case["x"].shape[1], "num_classes": case["args"].num_classes, **over})
    for sparse in (True, False):
        dense.sparse_linear_supported = real if sparse else (lambda *a: False)
        torch.manual_seed(0)
        model = SetGNN(args).to(dev); model.reset_parameters()
        data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(dev), edge_index=torch.from_numpy(case["edge_index"]).clone().to(dev),
                               norm=torch.from_numpy(case["norm"]).to(dev))
        n = data.x.shape[0]
        y = torch.randint(0, args.num_classes, (n,), device=dev); ones = torch.ones(n, device=dev)
        g = GraphedTrainStep(model, data, lambda out: nll_log_softmax(out, y, ones, n), FusedAdam(model.parameters(), lr=1e-3))
        gf = GraphedForward(model, data, constant_features=True)
        res = []
        for fn in (g, gf):
            for _ in range(10): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): fn()
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 200)
        print(f"{label:52s} {'from the non-zeros' if sparse else 'library GEMMs     '}  train step {res[0]*1e3:6.3f} ms   eval forward {res[1]*1e3:6.3f} ms")
        del g, gf, model
dense.sparse_linear_supported = real
