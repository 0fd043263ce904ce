import sys, copy, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from types import SimpleNamespace
import cases
from allset_amd import SetGNN
from allset_amd.graphs import GraphedTrainStep
from allset_amd.optim import FusedAdam
from allset_amd.losses import nll_log_softmax
device = torch.device("cuda:0")
case = cases.build_case("cora_ds_add")
torch.manual_seed(0)
m1 = SetGNN(case["args"]).to(device); m1.reset_parameters()
m2 = copy.deepcopy(m1)
data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(device), edge_index=torch.from_numpy(case["edge_index"]).to(device), norm=torch.from_numpy(case["norm"]).to(device))
n = data.x.shape[0]
y = torch.randint(0, case["args"].num_classes, (n,), device=device)
ones = torch.ones(n, device=device)
loss_fn = lambda out: nll_log_softmax(out, y, ones, n)
o1 = FusedAdam(m1.parameters(), lr=1e-3); o2 = FusedAdam(m2.parameters(), lr=1e-3)
gstep = GraphedTrainStep(m1, data, loss_fn, o1, train_mode=False)
print("steps after capture", sorted(set(float(st["step"]) for st in o1.state.values())))
m2.train(False)
d2 = SimpleNamespace(x=data.x, edge_index=data.edge_index.clone(), norm=data.norm)
for it in range(4):
    l1 = gstep()
    o2.zero_grad(); l2 = loss_fn(m2(d2)); l2.backward(); o2.step()
    bad = [(k, float((a - b).abs().max())) for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()) if not torch.equal(a, b)]
    gbad = [(k, float((a.grad - b.grad).abs().max())) for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()) if a.grad is not None and b.grad is not None and not torch.equal(a.grad, b.grad)]
    print(it, float(l1), float(l2), torch.equal(l1, l2.detach()), "param diffs", bad[:4], "grad diffs", gbad[:4],
          "steps", sorted(set(float(st["step"]) for st in o1.state.values())), sorted(set(float(st["step"]) for st in o2.state.values())))
p0 = next(iter(m1.parameters()))
st = o1.state[p0]
print("exp_avg max", float(st["exp_avg"].abs().max()), "exp_avg_sq max", float(st["exp_avg_sq"].abs().max()), "grad max", float(p0.grad.abs().max()), "step", float(st["step"]))
# torch Adam variant
m3 = copy.deepcopy(m2); m3.train(False)
o3 = torch.optim.Adam(m3.parameters(), lr=1e-3, capturable=True)
g3 = GraphedTrainStep(m3, data, loss_fn, o3, train_mode=False)
print("torch adam graphed losses", [float(g3()) for _ in range(4)])
