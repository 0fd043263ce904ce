"""Debug: two ranks on one GPU (gloo), ShardedSetGNN with BatchNorm in training mode vs the float64 oracle, per-parameter errors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
N_V, N_E, NNZ, d = 301, 187, int(os.environ.get('DBG_NNZ', '2400')), 64

def problem():
    rng = np.random.default_rng(7)
    pairs = sorted({(int(rng.integers(N_V)), int(rng.integers(N_E))) for _ in range(NNZ)} | {(0, e) for e in range(N_E)})
    ei = torch.tensor(pairs, dtype=torch.int64).t().contiguous()
    if os.environ.get("DBG_NORM"): rng.uniform(0.5, 1.5, size=ei.shape[1])
    x = torch.from_numpy(rng.standard_normal((N_V, d)).astype(np.float32))
    return ei, x

def worker(rank, world, q, cfgs):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
    torch.cuda.set_device(0); dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for scheme, norm_kind in cfgs:
        one(rank, world, q, scheme, norm_kind, dev)
    dist.barrier(); dist.destroy_process_group()

def one(rank, world, q, scheme, norm_kind, dev):
    import torch.distributed as dist
    import cases
    from allset_amd import SetGNN, dist as adist
    ei, x = problem()
    args = cases.make_args("ds_add", d, 64, 5, All_num_layers=2, normalization=norm_kind, dropout=0.0)
    torch.manual_seed(11)
    model = SetGNN(args).train().to(dev)
    if os.environ.get("DBG_UNFUSED"):
        from allset_amd import layers
        layers.MLP._fusable = lambda self, x: False
    adist._rank_dropout = lambda t, p, training: t
    ones = torch.ones(ei.shape[1], dtype=torch.int64)
    if scheme == "cols":
        hg = adist.ColumnShardedHypergraph(ei.to(dev), N_V, N_E, world, rank, norm=ones.to(dev), chunks=1).build_incidences()
    else:
        owner = adist.partition_hyperedges(torch.bincount(ei[1], minlength=N_E), world, "contiguous")
        loc, gids = adist.local_shard(ei, owner, rank)
        keep = owner[ei[1]] == rank
        hg = adist.ShardedHypergraph(loc.to(dev), N_V, gids.numel(), world, rank, norm=ones[keep].to(dev), inc_ids=keep.nonzero().reshape(-1).to(dev)).build_incidences()
    sharded = adist.ShardedSetGNN(model, hg)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    xp = torch.cat([x, x.new_zeros(hg.n_v_pad - N_V, d)])
    out = sharded(xp[hg.v_lo:hg.v_hi].to(dev))
    live = max(0, min(hg.v_hi, N_V) - hg.v_lo)
    cot = torch.linspace(-1.0, 1.0, N_V * out.shape[1]).view(N_V, -1)[hg.v_lo:hg.v_lo + live].to(dev)
    (out[:live] * cot).sum().backward()
    sharded.allreduce_grads()
    q.put((rank, scheme, norm_kind, out.detach().cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in model.named_parameters() if p.grad is not None}, {k: v.numpy() for k, v in sd.items()}))

if __name__ == "__main__":
    import torch.multiprocessing as mp
    import cases
    from oracle import allset_oracle as oracle
    cfgs = [tuple(c.split(":")) for c in sys.argv[1:]] or [("rows", "bn"), ("cols", "bn")]
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, q, cfgs)) for r in range(2)]
    [p.start() for p in ps]
    allres = [q.get(timeout=300) for _ in range(2 * len(cfgs))]
    [p.join(60) for p in ps]
    for scheme, kind in cfgs:
        res = sorted([r for r in allres if r[1] == scheme and r[2] == kind], key=lambda t: t[0])
        res = [(r[0], r[3], r[4], r[5]) for r in res]
        ei, x = problem()
        args = cases.make_args("ds_add", d, 64, 5, All_num_layers=2, normalization=kind, dropout=0.0)
        sd = {k: (torch.from_numpy(v).double() if torch.from_numpy(v).is_floating_point() else torch.from_numpy(v).clone()) for k, v in res[0][3].items()}
        for t in sd.values():
            if t.is_floating_point(): t.requires_grad_(True)
        ref = oracle.setgnn_forward(sd, args, x.double(), ei, torch.ones(ei.shape[1], dtype=torch.int64), drop=lambda t, p: t)
        cot = torch.linspace(-1.0, 1.0, N_V * ref.shape[1]).view(N_V, -1)
        (ref * cot.double()).sum().backward()
        got = torch.cat([torch.from_numpy(r[1]) for r in res])[:N_V]
        print(scheme, kind, "logits err", float((got.double() - ref.detach()).abs().max()), flush=True)
        for k, g in res[0][2].items():
            if sd[k].grad is not None:
                e = float((torch.from_numpy(g).double() - sd[k].grad).abs().max()); s_ = float(sd[k].grad.abs().max())
                if e > 1e-3 * max(s_, 1.0): print(f"   BAD {k:45s} err {e:.3e} scale {s_:.3e}")
