import sys, copy, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from types import SimpleNamespace
import cases
from allset_amd import SetGNN, dense
from allset_amd.optim import FusedAdam
from allset_amd.losses import nll_log_softmax
device = torch.device("cuda:0")
case = cases.build_case("cora_ds_add")
torch.manual_seed(0)
m1 = SetGNN(case["args"]).to(device); m1.reset_parameters(); m1.train(False)
data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(device), edge_index=torch.from_numpy(case["edge_index"]).to(device), norm=torch.from_numpy(case["norm"]).to(device))
n = data.x.shape[0]
y = torch.randint(0, case["args"].num_classes, (n,), device=device)
ones = torch.ones(n, device=device)
loss_fn = lambda out: nll_log_softmax(out, y, ones, n)
o1 = FusedAdam(m1.parameters(), lr=1e-3)
counter = torch.zeros(1, dtype=torch.int64, device=device)
one = torch.ones((), device=device)
for it in range(3):
    o1.zero_grad(set_to_none=True)
    loss = loss_fn(m1(data))
    with dense.deferred_param_grads(bump_i64=counter, bump_f32=o1.step_counters):
        loss.backward(one)
    torch.cuda.synchronize()
    gm = {k: float(p.grad.abs().max()) for k, p in m1.named_parameters() if p.grad is not None}
    print(it, float(loss), "grads", list(gm.items())[:3], "n", len(gm), "zero grads", [k for k, v in gm.items() if v == 0.0][:5])
    o1.step(counters_advanced=True)
    torch.cuda.synchronize()
    p0 = next(iter(m1.parameters()))
    print("   steps", sorted(set(float(s["step"]) for s in o1.state.values())), "exp_avg", float(o1.state[p0]["exp_avg"].abs().max()), "counter", int(counter))
