"""Debug: ShardedSetGNN with BatchNorm (training mode) on one rank with forced collectives, GPU vs the float64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ALLSET_FORCE_COLLECTIVES"] = "1"
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
import numpy as np, torch, torch.distributed as dist
import cases
from allset_amd import SetGNN, dist as adist
from oracle import allset_oracle as oracle
dev = torch.device("cuda:0")
dist.init_process_group("gloo", rank=0, world_size=1)
N_V, N_E, NNZ, d = 301, 187, 2400, 64
rng = np.random.default_rng(7)
pairs = sorted({(int(rng.integers(N_V)), int(rng.integers(N_E))) for _ in range(NNZ)} | {(0, e) for e in range(N_E)})
ei = torch.tensor(pairs, dtype=torch.int64).t().contiguous()
x = torch.from_numpy(rng.standard_normal((N_V, d)).astype(np.float32))
for scheme in sys.argv[1:] or ["cols", "rows"]:
    args = cases.make_args("ds_add", d, 64, 5, All_num_layers=2, normalization="bn", dropout=0.0)
    torch.manual_seed(11)
    model = SetGNN(args).train().to(dev)
    adist._rank_dropout = lambda t, p, training: t
    ones = torch.ones(ei.shape[1], dtype=torch.int64)
    if scheme == "cols":
        hg = adist.ColumnShardedHypergraph(ei.to(dev), N_V, N_E, 1, 0, norm=ones.to(dev), chunks=1).build_incidences()
    else:
        hg = adist.ShardedHypergraph(ei.to(dev), N_V, N_E, 1, 0, norm=ones.to(dev), inc_ids=torch.arange(ei.shape[1], device=dev)).build_incidences()
    sharded = adist.ShardedSetGNN(model, hg)
    sd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.cpu().clone()) for k, v in model.state_dict().items()}
    out = sharded(x.to(dev))
    cot = torch.linspace(-1.0, 1.0, N_V * out.shape[1]).view(N_V, -1)
    (out * cot.to(dev)).sum().backward()
    for t in sd.values():
        if t.is_floating_point():
            t.requires_grad_(True)
    ref = oracle.setgnn_forward(sd, args, x.double(), ei, ones, drop=lambda t, p: t)
    (ref * cot.double()).sum().backward()
    print(scheme, "logits err", float((out.detach().cpu().double() - ref.detach()).abs().max()))
    for k, p in model.named_parameters():
        if p.grad is not None and sd[k].grad is not None:
            e = float((p.grad.cpu().double() - sd[k].grad).abs().max()); s = float(sd[k].grad.abs().max())
            print(f"   {k:45s} err {e:.3e} scale {s:.3e}")
dist.destroy_process_group()
