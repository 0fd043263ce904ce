import sys, copy, torch, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from types import SimpleNamespace
import cases
from allset_amd import SetGNN, dense, graphs
from allset_amd.optim import FusedAdam
from allset_amd.losses import nll_log_softmax
device = torch.device("cuda:0")
case = cases.build_case("cora_ds_add")
data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(device), edge_index=torch.from_numpy(case["edge_index"]).to(device), norm=torch.from_numpy(case["norm"]).to(device))
n = data.x.shape[0]
y = torch.randint(0, case["args"].num_classes, (n,), device=device)
ones = torch.ones(n, device=device)
loss_fn = lambda out: nll_log_softmax(out, y, ones, n)
for mode in ("nobump", "bump", "bump_norestore"):
    torch.manual_seed(0)
    m1 = SetGNN(case["args"]).to(device); m1.reset_parameters()
    o1 = FusedAdam(m1.parameters(), lr=1e-3)
    graphs.FusedAdam = type("X", (), {}) if mode == "nobump" else FusedAdam
    g = graphs.GraphedTrainStep(m1, data, loss_fn, o1, train_mode=False, restore=(mode != "bump_norestore"))
    ls = [float(g()) for _ in range(4)]
    p0 = next(iter(m1.parameters()))
    print(mode, ls, "p0.grad max", float(p0.grad.abs().max()), "steps", sorted(set(float(s["step"]) for s in o1.state.values())))
