"""Leaf path (dense.input_norm_linear) vs general path of the SAME model with the SAME dropout masks (deterministic seed sequence;
the general path's torch input dropout replaced by the hash dropout of the same seed)."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from types import SimpleNamespace
import cases
from allset_amd import SetGNN, dense, models
dev = torch.device("cuda:0")
ctr = [0]
def draw():
    ctr[0] += 1
    return 1000003 * ctr[0]
dense._draw_seed = draw
def hash_dropout(x, p=0.5, training=True):
    if not training or p == 0.0: return x
    return x * dense.dropout_scale(x.shape, p, draw(), x.device)
models.F = SimpleNamespace(**{k: getattr(models.F, k) for k in dir(models.F) if not k.startswith("__")})
models.F.dropout = hash_dropout
for over in (dict(MLP_num_layers=3, MLP_hidden=128), dict(MLP_num_layers=3), dict(MLP_hidden=128)):
    case = cases.build_case("cora_ds_add")
    args = SimpleNamespace(**{**vars(case["args"]), **over})
    for attempt in range(3):
        torch.manual_seed(case["seed"] + attempt)
        model = SetGNN(args); model.reset_parameters(); model.train().to(dev)
        res = []
        for leaf in (True, False):
            model.zero_grad(set_to_none=True)
            ctr[0] = 0
            x = torch.from_numpy(case["x"]).to(dev).requires_grad_(not leaf)
            data = SimpleNamespace(x=x, edge_index=torch.from_numpy(case["edge_index"]).clone().to(dev), norm=torch.from_numpy(case["norm"]).to(dev))
            out = model(data)
            G = torch.from_numpy(cases.cotangent("cora_ds_add", out.shape)).to(dev)
            (out * G).sum().backward()
            res.append((out.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, ctr[0]))
        (oa, ga, ca), (ob, gb, cb) = res
        worst = max((float((ga[k] - gb[k]).abs().max() / gb[k].abs().max().clamp(min=1e-12)), k) for k in ga)
        print(over, attempt, "draws", ca, cb, "logits", float((oa - ob).abs().max() / ob.abs().max()), "worst grad rel-to-max", worst)
        if worst[0] > 1e-4:
            for k in ga:
                print("     ", k, float((ga[k] - gb[k]).abs().max() / gb[k].abs().max().clamp(min=1e-12)))
