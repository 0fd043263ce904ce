"""Host-side cost of one eager Cora-shaped training step: cProfile of 200 steps, top functions by cumulative time."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from types import SimpleNamespace
import cases
from allset_amd import SetGNN
dev = torch.device("cuda:0")
case = cases.build_case("cora_ds_add")
model = SetGNN(case["args"]).to(dev); model.reset_parameters()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(dev), edge_index=torch.from_numpy(case["edge_index"]).to(dev), norm=torch.from_numpy(case["norm"]).to(dev))
y = torch.randint(0, case["args"].num_classes, (data.x.shape[0],), device=dev)
def step():
    model.train(); opt.zero_grad(set_to_none=True)
    loss = F.nll_loss(F.log_softmax(model(data), dim=1), y); loss.backward(); opt.step()
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime")
st.print_stats(22)
