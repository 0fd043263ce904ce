import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from types import SimpleNamespace
from torch.profiler import profile, ProfilerActivity
import cases
from allset_amd import SetGNN, dense
from allset_amd.optim import FusedAdam
from allset_amd.losses import nll_log_softmax
name = sys.argv[1]
dev = torch.device("cuda:0")
case = cases.build_case(name)
model = SetGNN(case["args"]).to(dev); model.reset_parameters()
data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(dev), edge_index=torch.from_numpy(case["edge_index"]).to(dev), norm=torch.from_numpy(case["norm"]).to(dev))
n = data.x.shape[0]
y = torch.randint(0, case["args"].num_classes, (n,), device=dev)
ones = torch.ones(n, device=dev)
opt = FusedAdam(model.parameters(), lr=1e-3)
def step():
    model.train(); opt.zero_grad(set_to_none=True)
    loss = nll_log_softmax(model(data), y, ones, n)
    with dense.deferred_param_grads():
        loss.backward()
    opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 1]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:30]:
    print(f"{e.key:28s} n={e.count:3d} dev_us={e.device_time_total:8.1f}  shapes={str(e.input_shapes)[:120]}")
