"""Kernel launches of one eager Cora-shaped AllDeepSets training step, by kernel name (run under rocprofv3 --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from types import SimpleNamespace
import cases
from allset_amd import SetGNN
dev = torch.device("cuda:0")
case = cases.build_case(sys.argv[1] if len(sys.argv) > 1 else "cora_ds_add")
model = SetGNN(case["args"]).to(dev); model.reset_parameters()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(dev), edge_index=torch.from_numpy(case["edge_index"]).to(dev), norm=torch.from_numpy(case["norm"]).to(dev))
y = torch.randint(0, case["args"].num_classes, (data.x.shape[0],), device=dev)
for _ in range(20):
    model.train(); opt.zero_grad(set_to_none=True)
    loss = F.nll_loss(F.log_softmax(model(data), dim=1), y); loss.backward(); opt.step()
torch.cuda.synchronize()
