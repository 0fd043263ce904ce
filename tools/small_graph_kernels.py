#!/usr/bin/env python
"""Kernel inventory of ONE graphed training step at dataset scale (run under `rocprofv3 --kernel-trace --stats`):
captures the step of a parity case as a hipGraph, replays it REPLAYS times; calls / REPLAYS = launches per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from types import SimpleNamespace
import cases
from allset_amd import SetGNN
from allset_amd.graphs import GraphedTrainStep
REPLAYS = int(os.environ.get("REPLAYS", "200"))
name = sys.argv[1] if len(sys.argv) > 1 else "cora_ds_add"
dev = torch.device("cuda:0")
case = cases.build_case(name)
# MODEL_ARGS="MLP_hidden=512,heads=8": the reference's tuned widths on the same stand-in data (run_AllSetTransformer.sh)
if os.environ.get("MODEL_FROM"):      # MODEL_FROM=citeseer_pma_h4: that case's model arguments on THIS case's data (an AllSetTransformer on the Cora-shaped features)
    src = cases.build_case(os.environ["MODEL_FROM"])["args"]
    src.num_features, src.num_classes = case["args"].num_features, case["args"].num_classes
    case["args"] = src
for kv in filter(None, os.environ.get("MODEL_ARGS", "").split(",")):
    k, v = kv.split("=")
    setattr(case["args"], k, type(getattr(case["args"], k))(v))
model = SetGNN(case["args"]).to(dev)
model.reset_parameters()
data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(dev), edge_index=torch.from_numpy(case["edge_index"]).to(dev),
                       norm=torch.from_numpy(case["norm"]).to(dev))
y = torch.randint(0, case["args"].num_classes, (data.x.shape[0],), device=dev)
from allset_amd.optim import FusedAdam
opt = FusedAdam(model.parameters(), lr=1e-3)
from allset_amd.losses import nll_log_softmax
ones = torch.ones(data.x.shape[0], device=dev)
g = GraphedTrainStep(model, data, lambda out: nll_log_softmax(out, y, ones, data.x.shape[0]), opt)
torch.cuda.synchronize()
for _ in range(REPLAYS):
    g()
torch.cuda.synchronize()
print("replays", REPLAYS)
