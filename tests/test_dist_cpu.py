"""CPU, world_size 2 over gloo: the hyperedge-sharded layer (partition + all-gather / reduce-scatter exchange
+ replicated-parameter gradient all-reduce) reproduces the unsharded layer.  The local aggregation is the
oracle here (the product passes the HIP kernels through the same `aggregate=` hook), so this exercises exactly
the N>1 control flow bench.py runs under torchrun on RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_aggregate(x, inc, norm, aggr):
    from oracle import allset_oracle as oracle
    ei, n_dst = inc
    if ei.shape[1] == 0:                      # a rank without any incidence: zeros (kept in the graph so backward reaches x with zeros)
        return x.new_zeros(n_dst, x.shape[1]) + 0.0 * x.sum()
    out = oracle.deepsets_aggregate(x, ei, norm, aggr)
    if out.shape[0] < n_dst:
        out = torch.cat([out, out.new_zeros(n_dst - out.shape[0], out.shape[1])])
    return out


def _problem(world):
    rng = np.random.default_rng(42)
    n_v, n_e, d = 37, 23, 16                       # n_v not divisible by 2 -> exercises padding
    pairs = sorted({(int(rng.integers(n_v)), int(rng.integers(n_e))) for _ in range(260)} | {(0, e) for e in range(n_e)})
    ei = torch.tensor(pairs, dtype=torch.int64).t().contiguous()
    norm = torch.from_numpy(rng.uniform(0.5, 1.5, size=ei.shape[1]).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((n_v, d)).astype(np.float32))
    G = torch.from_numpy(rng.standard_normal((n_v, d)).astype(np.float32))
    return n_v, n_e, d, ei, norm, x, G


def _convs(d):
    from allset_amd import HalfNLHconv
    torch.manual_seed(3)
    a = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, attention=False).eval()
    b = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, attention=False).eval()
    return a, b


def _worker(rank, world, port, aggr, method, q, halo=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from allset_amd import dist as adist
        n_v, n_e, d, ei, norm, x, G = _problem(world)
        sizes = torch.bincount(ei[1], minlength=n_e)
        owner = adist.partition_hyperedges(sizes, world, method)
        loc, gids = adist.local_shard(ei, owner, rank)
        keep = owner[ei[1]] == rank
        hg = adist.ShardedHypergraph(loc, n_v, gids.numel(), world, rank, norm=norm[keep], halo=halo)
        hg.v2e = (loc, hg.n_e_local)
        hg.e2v = (torch.stack([loc[1], loc[0]]), hg.n_v_pad)
        if halo:            # the compact incidence of the boundary-vertex exchange, in the tuple form the oracle aggregation takes
            hloc = hg.halo_edge_index()
            hg.halo_v2e = (hloc, hg.n_e_local)
            hg.halo_e2v = (torch.stack([hloc[1], hloc[0]]), hg.halo.n_needed)
            assert hg.halo.n_needed == int(torch.unique(loc[0]).numel()) and sum(hg.halo.need_counts) == hg.halo.n_needed
        a, b = _convs(d)
        xp = torch.cat([x, x.new_zeros(hg.n_v_pad - n_v, d)])
        Gp = torch.cat([G, G.new_zeros(hg.n_v_pad - n_v, d)])
        xo = xp[hg.v_lo:hg.v_hi].clone().requires_grad_(True)
        out = adist.sharded_deepsets_layer(a, b, xo, hg, aggr=aggr, aggregate=_oracle_aggregate)
        (out * Gp[hg.v_lo:hg.v_hi]).sum().backward()
        params = list(a.parameters()) + list(b.parameters())
        adist.allreduce_grads(params)
        q.put((rank, out.detach().numpy().copy(), xo.grad.numpy().copy(), [p.grad.numpy().copy() for p in params], hg.v_lo, hg.v_hi))  # by value
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("aggr,method,world", [("add", "contiguous", 2), ("mean", "lpt", 2), ("add", "lpt", 4), ("mean", "contiguous", 3),
                                               ("max", "contiguous", 2), ("min", "lpt", 3), ("max", "lpt", 4)])
def test_sharded_layer_with_boundary_vertex_exchange_equals_unsharded(aggr, method, world):
    """The row partition with ``halo=True``: only the rows of the vertices a rank's hyperedges touch are exchanged (all-to-all with
    per-peer counts + fixed-order sums) -- same outputs, input and parameter gradients as the unsharded layer, at 2, 3 and 4 ranks
    (3: vertex blocks with padding); ``max`` / ``min``: the key merge through the owners (one winner per vertex and feature)."""
    test_sharded_layer_equals_unsharded(aggr, method, world, halo=True)


@pytest.mark.parametrize("aggr,method", [("add", "contiguous"), ("mean", "lpt"), ("max", "contiguous"), ("min", "lpt")])
def test_sharded_layer_equals_unsharded(aggr, method, world=2, halo=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, aggr, method, q, halo)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    results = [(r, torch.from_numpy(o), torch.from_numpy(g), [torch.from_numpy(t) for t in pg], lo, hi) for r, o, g, pg, lo, hi in results]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # unsharded reference with the same modules, via the oracle aggregation
    import torch.nn.functional as F
    n_v, n_e, d, ei, norm, x, G = _problem(world)
    a, b = _convs(d)
    xr = x.clone().requires_grad_(True)
    h = F.relu(a.f_enc(xr))
    e = F.relu(a.f_dec(_oracle_aggregate(h, (ei, n_e), norm, aggr)))
    g = F.relu(b.f_enc(e))
    v = F.relu(b.f_dec(_oracle_aggregate(g, (torch.stack([ei[1], ei[0]]), n_v), norm, aggr)))
    (v * G).sum().backward()
    ref_pg = [p.grad for p in list(a.parameters()) + list(b.parameters())]

    out = torch.cat([r[1] for r in results])[:n_v]
    gx = torch.cat([r[2] for r in results])[:n_v]
    torch.testing.assert_close(out, v.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gx, xr.grad, rtol=1e-5, atol=1e-5)
    # padded vertex rows (beyond n_v) go through the dense tail too; with a zero cotangent they only touch
    # parameter grads through LayerNorm/bias paths multiplied by zero -> parameter grads must match
    for got, exp in zip(results[0][3], ref_pg):
        torch.testing.assert_close(got, exp, rtol=1e-4, atol=1e-5)
    for r in range(1, world):
        for got, got1 in zip(results[0][3], results[r][3]):
            torch.testing.assert_close(got, got1, rtol=0, atol=0)      # all ranks hold the same summed grads


def test_halo_exchange_volume_follows_locality():
    """What the boundary-vertex exchange buys: rows received per gather on a hypergraph WITH locality (every hyperedge draws 15 of
    its 16 members from its rank's own vertex block) against one without (members uniform over all vertices)."""
    from allset_amd import dist as adist
    from allset_amd.synthetic import random_hypergraph
    world, n_loc = 4, 2000
    for locality, lo, hi in ((0.0, 0.95, 1.0), (15 / 16, 0.0, 0.35)):
        hg = random_hypergraph(world * n_loc, n_loc, 16, seed=3, device="cpu", locality=locality, home=(1, world))   # rank 1's block
        ids = hg.edge_index[0]
        halo = adist.Halo(ids, world * n_loc, 1, 0)                      # (world 1: no collective; the counts are per OWNER block below)
        owner = torch.div(halo.needed, n_loc, rounding_mode="floor")
        foreign = int((owner != 1).sum())
        frac = foreign / (3 * n_loc)                                     # share of the other ranks' rows this rank would ask for
        assert lo <= frac <= hi, (locality, frac)


def test_partition_is_nnz_balanced_and_complete():
    from allset_amd import dist as adist
    rng = np.random.default_rng(0)
    sizes = torch.from_numpy(np.minimum(rng.zipf(1.8, size=5000), 4096).astype(np.int64))
    for method in ("contiguous", "lpt"):
        owner = adist.partition_hyperedges(sizes, 8, method)
        assert owner.min() >= 0 and owner.max() <= 7 and owner.numel() == 5000
        load = torch.zeros(8, dtype=torch.int64).index_add_(0, owner, sizes)
        assert int(load.sum()) == int(sizes.sum())
        if method == "lpt":
            assert float(load.max()) <= 1.05 * float(load.float().mean()) + 4096
    lo, hi, pad = adist.vertex_block(37, 2, 1)
    assert (lo, hi, pad) == (19, 38, 38)


def test_local_shard_renumbers_and_covers():
    from allset_amd import dist as adist
    ei = torch.tensor([[0, 1, 2, 2, 3, 4], [0, 0, 1, 2, 2, 3]])
    owner = torch.tensor([0, 1, 0, 1])
    l0, g0 = adist.local_shard(ei, owner, 0)
    l1, g1 = adist.local_shard(ei, owner, 1)
    assert g0.tolist() == [0, 2] and g1.tolist() == [1, 3]
    assert l0.tolist() == [[0, 1, 2, 3], [0, 0, 1, 1]] and l1.tolist() == [[2, 4], [0, 1]]
    assert l0.shape[1] + l1.shape[1] == ei.shape[1]


# ---------------------------------------------------------------------------------------------
# sharded PMA (AllSetTransformer): cross-shard softmax merge
# ---------------------------------------------------------------------------------------------

class TorchPmaKernels:
    """torch-CPU stand-ins for the four local PMA primitives (same contracts as the HIP kernels)."""

    @staticmethod
    def aggregate(V, alpha, inc, heads, slope):
        from oracle import allset_oracle as oracle
        ei, n_dst = inc
        out, _ = oracle.pma_aggregate(V.view(V.shape[0], heads, -1), alpha, ei, slope)
        out = out.reshape(out.shape[0], -1)
        if out.shape[0] < n_dst:
            out = torch.cat([out, out.new_zeros(n_dst - out.shape[0], out.shape[1])])
        return out

    @staticmethod
    def fwd(V, alpha, inc, heads, slope):
        ei, n_dst = inc
        src, dst = ei[0], ei[1]
        a = torch.nn.functional.leaky_relu(alpha[src], slope)
        idx = dst.view(-1, 1).expand_as(a)
        m = torch.full((n_dst, heads), float("-inf")).scatter_reduce(0, idx, a, "amax", include_self=True)
        e = torch.exp(a - m[dst])
        l = torch.zeros(n_dst, heads).index_add_(0, dst, e)
        C = V.shape[1] // heads
        num = torch.zeros(n_dst, heads, C).index_add_(0, dst, V.view(-1, heads, C)[src] * e.unsqueeze(-1))
        out = torch.where(l.unsqueeze(-1) > 0, num / (l.unsqueeze(-1) + 1e-16), torch.zeros_like(num))
        m = torch.where(l > 0, m, torch.zeros_like(m))
        return out.reshape(n_dst, -1), m, l

    @staticmethod
    def bwd_stats(out, gout, m, l):
        H = m.shape[1]
        delta = (out * gout).view(out.shape[0], H, -1).sum(-1)
        M = torch.where(l > 0, m + torch.log(l + 1e-16), torch.full_like(m, 3.0e38))
        return torch.stack([M, delta], dim=-1)

    @staticmethod
    def bwd_src(inc, alpha, V, gout, stats, slope):
        ei, n_dst = inc
        src, dst = ei[0], ei[1]
        H = alpha.shape[1]
        C = V.shape[1] // H
        a = torch.nn.functional.leaky_relu(alpha, slope)
        p = torch.exp(a[src] - stats[dst, :, 0])                                   # [nnz, H]
        g = gout.view(-1, H, C)[dst]
        gV = torch.zeros(V.shape[0], H, C).index_add_(0, src, p.unsqueeze(-1) * g)
        gp = (V.view(-1, H, C)[src] * g).sum(-1)
        ga = torch.zeros_like(alpha).index_add_(0, src, p * (gp - stats[dst, :, 1]))
        ga = ga * torch.where(alpha > 0, torch.ones_like(alpha), torch.full_like(alpha, slope))
        return gV.reshape(V.shape[0], -1), ga


def _pma_convs(d, H):
    from allset_amd import HalfNLHconv
    torch.manual_seed(5)
    a = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=H, attention=True).eval()
    b = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=H, attention=True).eval()
    return a, b


def _pma_worker(rank, world, port, q, halo=False, H=4):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from allset_amd import dist as adist
        n_v, n_e, d, ei, _, x, G = _problem(world)
        owner = adist.partition_hyperedges(torch.bincount(ei[1], minlength=n_e), world, "contiguous")
        loc, gids = adist.local_shard(ei, owner, rank)
        hg = adist.ShardedHypergraph(loc, n_v, gids.numel(), world, rank, halo=halo)
        hg.v2e = (loc, hg.n_e_local)
        hg.e2v = (torch.stack([loc[1], loc[0]]), hg.n_v_pad)
        if halo:
            hloc = hg.halo_edge_index()
            hg.halo_v2e = (hloc, hg.n_e_local)
            hg.halo_e2v = (torch.stack([hloc[1], hloc[0]]), hg.halo.n_needed)
        a, b = _pma_convs(d, H)
        xp = torch.cat([x, x.new_zeros(hg.n_v_pad - n_v, d)])
        Gp = torch.cat([G, G.new_zeros(hg.n_v_pad - n_v, d)])
        xo = xp[hg.v_lo:hg.v_hi].clone().requires_grad_(True)
        out = adist.sharded_pma_layer(a, b, xo, hg, kernels=TorchPmaKernels)
        (out * Gp[hg.v_lo:hg.v_hi]).sum().backward()
        params = list(a.parameters()) + list(b.parameters())
        adist.allreduce_grads(params)
        q.put((rank, out.detach().numpy().copy(), xo.grad.numpy().copy(), [p.grad.numpy().copy() for p in params]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 4), (3, 1), (4, 2)])
def test_sharded_pma_layer_with_boundary_vertex_exchange_equals_unsharded(world, H):
    """The PMA row partition with ``halo=True``: [V | logits] rows of the touched vertices only in V->E; in E->V the (m, l, o) merge
    through the owners (per-vertex maximum of the local softmax maxima gathered back, weighted sums scattered to the owners) instead
    of an all-reduce(max) + reduce-scatter over the whole vertex range."""
    test_sharded_pma_layer_equals_unsharded(world, True, H)


def test_sharded_pma_layer_equals_unsharded(world=2, halo=False, H=4):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pma_worker, args=(r, world, port, q, halo, H)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    import torch.nn.functional as F
    n_v, n_e, d, ei, _, x, G = _problem(world)
    a, b = _pma_convs(d, H)

    def pma_module(p, xin, e_idx, n_dst):
        H, C = p.heads, p.hidden
        o = TorchPmaKernels.aggregate(p.lin_V(xin), p._logits(xin), (e_idx, n_dst), H, 0.2)
        o = (o.view(-1, H, C) + p.att_r).view(-1, H * C)
        o = p.ln0(o)
        return p.ln1(o + F.relu(p.rFF(o)))

    xr = x.clone().requires_grad_(True)
    e = F.relu(pma_module(a.prop, xr, ei, n_e))
    v = F.relu(pma_module(b.prop, e, torch.stack([ei[1], ei[0]]), n_v))
    (v * G).sum().backward()
    ref_pg = [p.grad for p in list(a.parameters()) + list(b.parameters())]
    out = torch.cat([torch.from_numpy(r[1]) for r in results])[:n_v]
    gx = torch.cat([torch.from_numpy(r[2]) for r in results])[:n_v]
    torch.testing.assert_close(out, v.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gx, xr.grad, rtol=1e-4, atol=1e-5)
    for got, exp in zip(results[0][3], ref_pg):
        torch.testing.assert_close(torch.from_numpy(got), exp, rtol=1e-4, atol=2e-5)


# ---------------------------------------------------------------------------------------------
# model level: ShardedSetGNN == SetGNN (oracle) on the owned vertex blocks
# ---------------------------------------------------------------------------------------------

def _model_worker(rank, world, port, mode, q, columns=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import cases
        from allset_amd import SetGNN, dist as adist
        n_v, n_e, d, ei, _, x, G = _problem(world)
        args = cases.make_args(mode, d, 32, 5, All_num_layers=2)
        torch.manual_seed(11)
        model = SetGNN(args).eval()
        if columns:            # column-sharded aggregation, `columns` chunks in the overlapped exchange
            hg = adist.ColumnShardedHypergraph(ei, n_v, n_e, world, rank, norm=torch.ones(ei.shape[1], dtype=torch.int64),
                                               chunks=columns)
            hg.v2e = (ei, hg.n_e_pad)
            hg.e2v = (torch.stack([ei[1], ei[0]]), hg.n_v_pad)
        else:
            owner = adist.partition_hyperedges(torch.bincount(ei[1], minlength=n_e), world, "contiguous")
            loc, gids = adist.local_shard(ei, owner, rank)
            keep = owner[ei[1]] == rank
            hg = adist.ShardedHypergraph(loc, n_v, gids.numel(), world, rank, norm=torch.ones(int(keep.sum()), dtype=torch.int64))
            hg.v2e = (loc, hg.n_e_local)
            hg.e2v = (torch.stack([loc[1], loc[0]]), hg.n_v_pad)
        sharded = adist.ShardedSetGNN(model, hg, aggregate=_oracle_aggregate, kernels=TorchPmaKernels)
        xp = torch.cat([x, x.new_zeros(hg.n_v_pad - n_v, d)])
        out = sharded(xp[hg.v_lo:hg.v_hi])
        q.put((rank, out.detach().numpy().copy(), {k: v.numpy().copy() for k, v in model.state_dict().items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,columns", [("ds_add", 0), ("pma_h4", 0), ("ds_add", 1), ("pma_h4", 2)])
def test_sharded_setgnn_equals_oracle(mode, columns):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cases
    from oracle import allset_oracle as oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, world, port, mode, q, columns)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_v, n_e, d, ei, _, x, G = _problem(world)
    args = cases.make_args(mode, d, 32, 5, All_num_layers=2)
    sd = {k: torch.from_numpy(v) for k, v in results[0][2].items()}
    ref = oracle.setgnn_forward(sd, args, x, ei, torch.ones(ei.shape[1], dtype=torch.int64))
    got = torch.cat([torch.from_numpy(r[1]) for r in results])[:n_v]
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5)


def _gpr_mask_worker(rank, world, port, columns, q):
    """GPR + LearnMask through ShardedSetGNN (reference models.py:451-452,457-471): logits of the owned rows and the
    all-reduced gradient of the replicated per-incidence ``Importance``."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import cases
        from allset_amd import SetGNN, dist as adist
        n_v, n_e, d, ei, norm, x, G = _problem(world)
        args = cases.make_args("ds_add", d, 32, 5, All_num_layers=2, GPR=True, LearnMask=True)
        torch.manual_seed(11)
        model = SetGNN(args, norm).eval()
        with torch.no_grad():
            model.Importance.copy_(torch.linspace(0.5, 1.5, ei.shape[1]))
        if columns:
            hg = adist.ColumnShardedHypergraph(ei, n_v, n_e, world, rank, norm=norm, chunks=columns)
            hg.v2e = (ei, hg.n_e_pad)
            hg.e2v = (torch.stack([ei[1], ei[0]]), hg.n_v_pad)
        else:
            owner = adist.partition_hyperedges(torch.bincount(ei[1], minlength=n_e), world, "contiguous")
            loc, gids = adist.local_shard(ei, owner, rank)
            keep = owner[ei[1]] == rank
            hg = adist.ShardedHypergraph(loc, n_v, gids.numel(), world, rank, norm=norm[keep], inc_ids=keep.nonzero().reshape(-1))
            hg.v2e = (loc, hg.n_e_local)
            hg.e2v = (torch.stack([loc[1], loc[0]]), hg.n_v_pad)
        sharded = adist.ShardedSetGNN(model, hg, aggregate=_oracle_aggregate, kernels=TorchPmaKernels)
        xp = torch.cat([x, x.new_zeros(hg.n_v_pad - n_v, d)])
        out = sharded(xp[hg.v_lo:hg.v_hi])
        live = max(0, min(hg.v_hi, n_v) - hg.v_lo)                 # pad rows carry no cotangent
        cot = torch.linspace(-1.0, 1.0, n_v * out.shape[1]).view(n_v, -1)[hg.v_lo:hg.v_lo + live]
        (out[:live] * cot).sum().backward()
        sharded.allreduce_grads()
        q.put((rank, out.detach().numpy().copy(), model.Importance.grad.numpy().copy(),
               model.GPRweights.weight.grad.numpy().copy(), {k: v.numpy().copy() for k, v in model.state_dict().items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("columns", [0, 1, 2])
def test_sharded_gpr_learnmask_equals_oracle(columns):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cases
    from oracle import allset_oracle as oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpr_mask_worker, args=(r, world, port, columns, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_v, n_e, d, ei, norm, x, G = _problem(world)
    args = cases.make_args("ds_add", d, 32, 5, All_num_layers=2, GPR=True, LearnMask=True)
    sd = {k: torch.from_numpy(v).requires_grad_(torch.from_numpy(v).is_floating_point()) for k, v in results[0][4].items()}
    ref = oracle.setgnn_forward(sd, args, x, ei, norm)
    cot = torch.linspace(-1.0, 1.0, n_v * ref.shape[1]).view(n_v, -1)
    (ref * cot).sum().backward()
    got = torch.cat([torch.from_numpy(r[1]) for r in results])[:n_v]
    torch.testing.assert_close(got, ref.detach(), rtol=1e-4, atol=1e-5)
    for r in results:                                  # after the all-reduce every rank holds the full gradient
        torch.testing.assert_close(torch.from_numpy(r[2]), sd["Importance"].grad, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(torch.from_numpy(r[3]), sd["GPRweights.weight"].grad, rtol=1e-4, atol=1e-5)


def _bn_worker(rank, world, port, columns, q):
    """``Normalization='bn'`` (the reference MLP's default, layers.py:499-562) through ShardedSetGNN in TRAINING mode: batch
    statistics over the real rows of both ranks; dropouts off so that the oracle's training mode is comparable."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import cases
        from allset_amd import SetGNN, dist as adist
        n_v, n_e, d, ei, _, x, G = _problem(world)
        args = cases.make_args("ds_add", d, 32, 5, All_num_layers=2, normalization="bn", dropout=0.0, Classifier_num_layers=2)
        torch.manual_seed(11)
        model = SetGNN(args).train()
        sd0 = {k: v.clone().numpy() for k, v in model.state_dict().items()}
        adist._rank_dropout = lambda t, p, training: t            # the hard-wired input dropout (models.py:473) off
        ones = torch.ones(ei.shape[1], dtype=torch.int64)
        if columns == "hybrid":                                    # 2 target groups x world / 2 column groups (round 5)
            cg, _ = adist.hybrid_groups(world, 2, rank)
            hg = adist.ColumnShardedHypergraph(ei, n_v, n_e, world, rank, norm=ones, row_groups=2, col_group=cg)
            (e1, n1, k1), (e2, n2, k2) = hg.target_slices()
            hg.v2e, hg.e2v, hg.ids_v2e, hg.ids_e2v = (e1, n1), (e2, n2), k1, k2
        elif columns:
            hg = adist.ColumnShardedHypergraph(ei, n_v, n_e, world, rank, norm=ones, chunks=columns)
            hg.v2e = (ei, hg.n_e_pad)
            hg.e2v = (torch.stack([ei[1], ei[0]]), hg.n_v_pad)
        else:
            owner = adist.partition_hyperedges(torch.bincount(ei[1], minlength=n_e), world, "contiguous")
            loc, gids = adist.local_shard(ei, owner, rank)
            keep = owner[ei[1]] == rank
            hg = adist.ShardedHypergraph(loc, n_v, gids.numel(), world, rank, norm=torch.ones(int(keep.sum()), dtype=torch.int64))
            hg.v2e = (loc, hg.n_e_local)
            hg.e2v = (torch.stack([loc[1], loc[0]]), hg.n_v_pad)
        sharded = adist.ShardedSetGNN(model, hg, aggregate=_oracle_aggregate, kernels=TorchPmaKernels)
        xp = torch.cat([x, x.new_zeros(hg.n_v_pad - n_v, d)])
        out = sharded(xp[hg.v_lo:hg.v_hi])
        cot = torch.linspace(-1.0, 1.0, hg.n_v_pad * out.shape[1]).view(hg.n_v_pad, -1)[hg.v_lo:hg.v_hi].clone()
        cot[max(0, n_v - hg.v_lo):] = 0.0                         # pad rows carry no loss
        (out * cot).sum().backward()
        sharded.allreduce_grads()
        grads = {k: p.grad.numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
        q.put((rank, out.detach().numpy().copy(), sd0, grads, {k: v.numpy().copy() for k, v in model.state_dict().items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("columns", [0, 1, "hybrid"])
def test_sharded_batchnorm_uses_the_statistics_of_the_whole_batch(columns):
    """(``hybrid``: the whole two-layer model through ShardedSetGNN on four ranks = 2 target groups x 2 column groups; the batch
    statistics run over the WORLD group, the exchanges over the world and the column groups.)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cases
    from oracle import allset_oracle as oracle
    world = 4 if columns == "hybrid" else 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, world, port, columns, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_v, n_e, d, ei, _, x, G = _problem(world)
    args = cases.make_args("ds_add", d, 32, 5, All_num_layers=2, normalization="bn", dropout=0.0, Classifier_num_layers=2)
    sd = {k: torch.from_numpy(v).clone() for k, v in results[0][2].items()}
    for t in sd.values():
        if t.is_floating_point():
            t.requires_grad_(True)
    ref = oracle.setgnn_forward(sd, args, x, ei, torch.ones(ei.shape[1], dtype=torch.int64), drop=lambda t, p: t)   # training mode, no dropout
    got = torch.cat([torch.from_numpy(r[1]) for r in results])[:n_v]
    torch.testing.assert_close(got, ref.detach(), rtol=1e-4, atol=1e-4 * float(ref.detach().abs().max()))    # (atol relative to the tensor's scale: tests/util.py)
    n_pad = sum(r[1].shape[0] for r in results)
    cot = torch.linspace(-1.0, 1.0, n_pad * ref.shape[1]).view(n_pad, -1)[:n_v]
    (ref * cot).sum().backward()
    for r in results:                                              # all-reduced: every rank holds the full gradient
        for k, g in r[3].items():
            exp = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])      # (never-applied modules: zero-filled by allreduce_grads)
            torch.testing.assert_close(torch.from_numpy(g), exp, rtol=1e-3, atol=1e-4 * max(1.0, float(exp.abs().max())), msg=lambda m, k=k: f"{k}: {m}")
    # running statistics: identical on every rank, moved towards the whole batch's statistics exactly as torch moves them
    for r in results[1:]:
        for k in results[0][4]:
            np.testing.assert_array_equal(results[0][4][k], r[4][k])
    key = "V2EConvs.0.f_enc.normalizations.0"
    rm = torch.from_numpy(results[0][4][key + ".running_mean"])
    torch.testing.assert_close(rm, 0.1 * x.mean(0), rtol=1e-4, atol=1e-6)                      # InputNorm slot sees x itself
    rv = torch.from_numpy(results[0][4][key + ".running_var"])
    torch.testing.assert_close(rv, 0.9 + 0.1 * x.var(0, unbiased=True), rtol=1e-4, atol=1e-6)
    assert int(results[0][4][key + ".num_batches_tracked"]) == 1


def _zero_fill_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from allset_amd import dist as adist
        # the zero-gradient substitution of allreduce_grads: rank 1 has no gradient for one parameter
        p0, p1 = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2))
        p0.grad = torch.full((3,), float(rank + 1))
        if rank == 0:
            p1.grad = torch.full((2,), 5.0)
        adist.allreduce_grads([p0, p1])
        q.put((rank, p0.grad.tolist(), p1.grad.tolist()))
    finally:
        dist.destroy_process_group()


def test_missing_grads_are_zero_filled():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero_fill_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, g0, g1 in results:
        assert g0 == [3.0, 3.0, 3.0] and g1 == [5.0, 5.0]


# ---------------------------------------------------------------------------------------------------------------
# cross-shard max: one winner per (vertex, feature), lowest rank on exact ties, vertices without incidences give 0
# ---------------------------------------------------------------------------------------------------------------

def _merge_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from allset_amd import dist as adist
        n_v, d = 6, 3
        hg = adist.ShardedHypergraph(torch.zeros((2, 0), dtype=torch.int64), n_v, 0, world, rank)
        # rows: 0 only rank 0 has it; 1 only rank 1; 2 both, rank 1 larger; 3 both, exact tie; 4 nobody; 5 both, negative values
        part = torch.tensor([[[1., -2., 3.], [0., 0., 0.], [1., 1., 1.], [5., -0.5, 0.], [0., 0., 0.], [-4., -1., -9.]],
                             [[0., 0., 0.], [-7., 8., 9.], [2., 0.5, 4.], [5., -0.5, 0.], [0., 0., 0.], [-3., -2., -9.]]])[rank]
        has = torch.tensor([[True, False, True, True, False, True], [False, True, True, True, False, True]])[rank]
        part = part.clone().requires_grad_(True)
        out = adist._ShardedExtremeMerge.apply(part, has, hg, None, False)
        G = torch.arange(1, 1 + out.numel(), dtype=torch.float32).view_as(out) + 100 * rank
        (out * G).sum().backward()
        q.put((rank, out.detach().numpy().copy(), part.grad.numpy().copy(), hg.v_lo, hg.v_hi))
    finally:
        dist.destroy_process_group()


def test_sharded_max_merge_picks_one_winner():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merge_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out = np.concatenate([res[0][1], res[1][1]])
    expect = np.array([[1., -2., 3.], [-7., 8., 9.], [2., 1., 4.], [5., -0.5, 0.], [0., 0., 0.], [-3., -1., -9.]], dtype=np.float32)
    assert np.array_equal(out, expect)
    # cotangent of row v, feature c (as seen by the owner of v): rank 0 owns rows 0-2, rank 1 rows 3-5
    G = np.concatenate([np.arange(1, 10, dtype=np.float32).reshape(3, 3), np.arange(1, 10, dtype=np.float32).reshape(3, 3) + 100])
    g0, g1 = res[0][2], res[1][2]
    win0 = np.array([[1, 1, 1], [0, 0, 0], [0, 1, 0], [1, 1, 1], [0, 0, 0], [0, 1, 1]], dtype=bool)     # ties -> rank 0
    win1 = np.array([[0, 0, 0], [1, 1, 1], [1, 0, 1], [0, 0, 0], [0, 0, 0], [1, 0, 0]], dtype=bool)
    assert np.array_equal(g0, np.where(win0, G, 0)) and np.array_equal(g1, np.where(win1, G, 0))


# ---------------------------------------------------------------------------------------------------------------
# column-sharded aggregation: full incidence on every rank, d/P columns each, all-to-all layout changes
# ---------------------------------------------------------------------------------------------------------------

def _col_worker(rank, world, port, kind, arg, q, chunks=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from allset_amd import dist as adist
        n_v, n_e, d, ei, norm, x, G = _problem(world)
        hg = adist.ColumnShardedHypergraph(ei, n_v, n_e, world, rank, norm=norm if kind == "ds" else None, chunks=chunks)
        hg.v2e = (ei, hg.n_e_pad)
        hg.e2v = (torch.stack([ei[1], ei[0]]), hg.n_v_pad)
        xp = torch.cat([x, x.new_zeros(hg.n_v_pad - n_v, d)])
        Gp = torch.cat([G, G.new_zeros(hg.n_v_pad - n_v, d)])
        xo = xp[hg.v_lo:hg.v_hi].clone().requires_grad_(True)
        if kind == "ds":
            a, b = _convs(d)
            out = adist.colsharded_deepsets_layer(a, b, xo, hg, aggr=arg, aggregate=_oracle_aggregate, chunks=chunks)
        else:
            a, b = _pma_convs(d, arg)
            out = adist.colsharded_pma_layer(a, b, xo, hg, kernels=TorchPmaKernels, chunks=chunks)
        (out * Gp[hg.v_lo:hg.v_hi]).sum().backward()
        params = list(a.parameters()) + list(b.parameters())
        adist.allreduce_grads(params)
        q.put((rank, out.detach().numpy().copy(), xo.grad.numpy().copy(), [p.grad.numpy().copy() for p in params]))
    finally:
        dist.destroy_process_group()


def _run_col(kind, arg, chunks=1):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_col_worker, args=(r, world, port, kind, arg, q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.parametrize("aggr,chunks", [("add", 1), ("mean", 1), ("max", 1), ("min", 1), ("add", 4), ("max", 3)])
def test_colsharded_deepsets_layer_equals_unsharded(aggr, chunks):
    """chunks > 1: the overlapped exchange (K all-to-alls in flight, tokens through autograd); the owned blocks are padded
    to a multiple of the chunk count (37 vertices -> 2 x 20 rows for 4 chunks)."""
    import torch.nn.functional as F
    results = _run_col("ds", aggr, chunks)
    n_v, n_e, d, ei, norm, x, G = _problem(2)
    a, b = _convs(d)
    xr = x.clone().requires_grad_(True)
    h = F.relu(a.f_enc(xr))
    e = F.relu(a.f_dec(_oracle_aggregate(h, (ei, n_e), norm, aggr)))
    g = F.relu(b.f_enc(e))
    v = F.relu(b.f_dec(_oracle_aggregate(g, (torch.stack([ei[1], ei[0]]), n_v), norm, aggr)))
    (v * G).sum().backward()
    ref_pg = [p.grad for p in list(a.parameters()) + list(b.parameters())]
    out = torch.cat([torch.from_numpy(r[1]) for r in results])[:n_v]
    gx = torch.cat([torch.from_numpy(r[2]) for r in results])[:n_v]
    torch.testing.assert_close(out, v.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gx, xr.grad, rtol=1e-5, atol=1e-5)
    # padded hyperedge rows (beyond n_e) run through the dense tail with zero cotangent AND are never gathered: their
    # contribution to parameter gradients is exactly zero
    for got, exp in zip(results[0][3], ref_pg):
        torch.testing.assert_close(torch.from_numpy(got), exp, rtol=1e-4, atol=1e-5)
    for got, got1 in zip(results[0][3], results[1][3]):
        np.testing.assert_array_equal(got, got1)


@pytest.mark.parametrize("H,chunks", [(4, 1), (1, 1), (4, 4), (1, 2)])   # H=4: two whole heads per rank; 1: a shared head
def test_colsharded_pma_layer_equals_unsharded(H, chunks):
    import torch.nn.functional as F
    results = _run_col("pma", H, chunks)
    n_v, n_e, d, ei, _, x, G = _problem(2)
    a, b = _pma_convs(d, H)

    def pma_module(p, xin, e_idx, n_dst):
        C = p.hidden
        o = TorchPmaKernels.aggregate(p.lin_V(xin), p._logits(xin), (e_idx, n_dst), H, 0.2)
        o = (o.view(-1, H, C) + p.att_r).view(-1, H * C)
        o = p.ln0(o)
        return p.ln1(o + F.relu(p.rFF(o)))

    xr = x.clone().requires_grad_(True)
    e = F.relu(pma_module(a.prop, xr, ei, n_e))
    v = F.relu(pma_module(b.prop, e, torch.stack([ei[1], ei[0]]), n_v))
    (v * G).sum().backward()
    ref_pg = [p.grad for p in list(a.parameters()) + list(b.parameters())]
    out = torch.cat([torch.from_numpy(r[1]) for r in results])[:n_v]
    gx = torch.cat([torch.from_numpy(r[2]) for r in results])[:n_v]
    torch.testing.assert_close(out, v.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gx, xr.grad, rtol=1e-4, atol=1e-5)
    for got, exp in zip(results[0][3], ref_pg):
        torch.testing.assert_close(torch.from_numpy(got), exp, rtol=1e-4, atol=2e-5)


# ---------------------------------------------------------------------------------------------------------------
# hybrid partition (round 5): R target groups x C column groups -- gloo worlds 4 (2 x 2) and 8 (2 x 4)
# ---------------------------------------------------------------------------------------------------------------

def _hybrid_worker(rank, world, port, kind, arg, q, row_groups=2, learn_mask=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from allset_amd import dist as adist
        n_v, n_e, d, ei, norm, x, G = _problem(world)
        cg, gg = adist.hybrid_groups(world, row_groups, rank)
        hg = adist.ColumnShardedHypergraph(ei, n_v, n_e, world, rank, norm=norm if kind == "ds" else None, row_groups=row_groups,
                                           col_group=cg, gather_group=gg)
        assert hg.hybrid and hg.col_world == world // row_groups and hg.rank == hg.group_a * hg.col_world + hg.slice_b
        (ei1, n1, k1), (ei2, n2, k2) = hg.target_slices()          # the tuple form the oracle aggregation takes
        hg.v2e, hg.e2v, hg.ids_v2e, hg.ids_e2v = (ei1, n1), (ei2, n2), k1, k2
        # every incidence is kept by exactly one target group per direction
        cnt = torch.tensor([k1.numel(), k2.numel()])
        dist.all_reduce(cnt)
        assert cnt.tolist() == [ei.shape[1] * hg.col_world, ei.shape[1] * hg.col_world]
        xp = torch.cat([x, x.new_zeros(hg.n_v_pad - n_v, d)])
        Gp = torch.cat([G, G.new_zeros(hg.n_v_pad - n_v, d)])
        xo = xp[hg.v_lo:hg.v_hi].clone().requires_grad_(True)
        imp = None
        if kind == "ds":
            a, b = _convs(d)
            nrm = norm
            if learn_mask:                                           # a replicated per-incidence parameter: gradient = sum over ranks
                imp = torch.nn.Parameter(torch.linspace(0.5, 1.5, ei.shape[1]))
                nrm = imp * norm
            out = adist.colsharded_deepsets_layer(a, b, xo, hg, aggr=arg, aggregate=_oracle_aggregate, norm=nrm)
        else:
            a, b = _pma_convs(d, arg)
            out = adist.colsharded_pma_layer(a, b, xo, hg, kernels=TorchPmaKernels)
        (out * Gp[hg.v_lo:hg.v_hi]).sum().backward()
        params = list(a.parameters()) + list(b.parameters()) + ([imp] if imp is not None else [])
        adist.allreduce_grads(params)
        q.put((rank, out.detach().numpy().copy(), xo.grad.numpy().copy(), [p.grad.numpy().copy() for p in params]))
    finally:
        dist.destroy_process_group()


def _run_hybrid(world, kind, arg, learn_mask=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hybrid_worker, args=(r, world, port, kind, arg, q, 2, learn_mask)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.parametrize("world,aggr,mask", [(4, "add", False), (4, "max", False), (8, "add", True), (8, "mean", False)])
def test_hybrid_deepsets_layer_equals_unsharded(world, aggr, mask):
    """2 target groups x (world / 2) column groups against the unsharded layer: outputs, input gradient, parameter gradients
    (identical on every rank after the all-reduce) and -- ``mask`` -- the gradient of a replicated per-incidence weight parameter
    (LearnMask's Importance, reference models.py:451-452), of which every rank sees its directions' halves."""
    import torch.nn.functional as F
    results = _run_hybrid(world, "ds", aggr, mask)
    n_v, n_e, d, ei, norm, x, G = _problem(world)
    a, b = _convs(d)
    imp = torch.nn.Parameter(torch.linspace(0.5, 1.5, ei.shape[1])) if mask else None
    nrm = imp * norm if mask else norm
    xr = x.clone().requires_grad_(True)
    h = F.relu(a.f_enc(xr))
    e = F.relu(a.f_dec(_oracle_aggregate(h, (ei, n_e), nrm, aggr)))
    g = F.relu(b.f_enc(e))
    v = F.relu(b.f_dec(_oracle_aggregate(g, (torch.stack([ei[1], ei[0]]), n_v), nrm, aggr)))
    (v * G).sum().backward()
    ref_pg = [p.grad for p in list(a.parameters()) + list(b.parameters())] + ([imp.grad] if mask else [])
    out = torch.cat([torch.from_numpy(r[1]) for r in results])[:n_v]
    gx = torch.cat([torch.from_numpy(r[2]) for r in results])[:n_v]
    torch.testing.assert_close(out, v.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gx, xr.grad, rtol=1e-5, atol=1e-5)
    for got, exp in zip(results[0][3], ref_pg):
        torch.testing.assert_close(torch.from_numpy(got), exp, rtol=1e-4, atol=1e-5)
    for r in results[1:]:
        for got, got0 in zip(r[3], results[0][3]):
            np.testing.assert_array_equal(got, got0)


@pytest.mark.parametrize("world,H", [(4, 4), (8, 4), (8, 1)])      # 2 / 1 whole heads per column rank; one head shared by all four
def test_hybrid_pma_layer_equals_unsharded(world, H):
    import torch.nn.functional as F
    results = _run_hybrid(world, "pma", H)
    n_v, n_e, d, ei, _, x, G = _problem(world)
    a, b = _pma_convs(d, H)

    def pma_module(p, xin, e_idx, n_dst):
        C = p.hidden
        o = TorchPmaKernels.aggregate(p.lin_V(xin), p._logits(xin), (e_idx, n_dst), H, 0.2)
        o = (o.view(-1, H, C) + p.att_r).view(-1, H * C)
        o = p.ln0(o)
        return p.ln1(o + F.relu(p.rFF(o)))

    xr = x.clone().requires_grad_(True)
    e = F.relu(pma_module(a.prop, xr, ei, n_e))
    v = F.relu(pma_module(b.prop, e, torch.stack([ei[1], ei[0]]), n_v))
    (v * G).sum().backward()
    ref_pg = [p.grad for p in list(a.parameters()) + list(b.parameters())]
    out = torch.cat([torch.from_numpy(r[1]) for r in results])[:n_v]
    gx = torch.cat([torch.from_numpy(r[2]) for r in results])[:n_v]
    torch.testing.assert_close(out, v.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gx, xr.grad, rtol=1e-4, atol=1e-5)
    for got, exp in zip(results[0][3], ref_pg):
        torch.testing.assert_close(torch.from_numpy(got), exp, rtol=1e-4, atol=2e-5)


def test_head_slots_and_exchange_volume():
    from allset_amd import dist as adist
    assert adist.head_slots(4, 2) == (2, [0, 1, 2, 3])
    assert adist.head_slots(4, 4) == (1, [0, 1, 2, 3])
    assert adist.head_slots(4, 8) == (1, [0, 0, 1, 1, 2, 2, 3, 3])
    assert adist.head_slots(1, 2) == (1, [0, 0])
    with pytest.raises(ValueError):
        adist.head_slots(4, 3)
    n, d = 8_000_000, 128
    rows = adist.exchange_bytes_per_rank("rows", 8, n, n, d)
    cols = adist.exchange_bytes_per_rank("columns", 8, n, n, d)
    assert rows == 4 * cols                              # (n_V + n_E) / P vs n_V with n_E == n_V, P = 8
    assert adist.exchange_bytes_per_rank("columns", 2, n, n, d) == adist.exchange_bytes_per_rank("rows", 2, n, n, d)
    assert adist.exchange_bytes_per_rank("rows", 1, n, n, d) == 0


def test_choose_sharding():
    from allset_amd import dist as adist
    assert adist.choose_sharding(1, 128) == "rows" and adist.choose_sharding(2, 128) == "columns"
    assert adist.choose_sharding(4, 128) == "columns" and adist.choose_sharding(8, 128) == "columns"
    assert adist.choose_sharding(8, 128, heads=4) == "columns" and adist.choose_sharding(8, 256, heads=4, elem=2) == "columns"
    assert adist.choose_sharding(8, 64) == "rows"            # 8 fp32 columns = 32-byte rows
    assert adist.choose_sharding(8, 100) == "rows" and adist.choose_sharding(4, 128, heads=3) == "rows"


# ---------------------------------------------------------------------------------------------------------------
# column scheme at random shapes, 2 ranks: one spawn, many configurations
# ---------------------------------------------------------------------------------------------------------------

def _rand_cfgs():
    rng = np.random.default_rng(2026)
    cfgs = []
    for i in range(24):
        pma = bool(i % 2)
        H = int(rng.choice([1, 2, 4])) if pma else 1
        cfgs.append(dict(n_v=int(rng.integers(5, 60)), n_e=int(rng.integers(3, 40)), nnz=int(rng.integers(20, 400)),
                         d=int(rng.choice([8, 16, 32])), pma=pma, H=H, aggr=str(rng.choice(["add", "mean", "max", "min"])),
                         chunks=int(rng.choice([1, 2, 3])), seed=int(rng.integers(1 << 30))))
    return cfgs


def _rand_problem(c):
    rng = np.random.default_rng(c["seed"])
    pairs = sorted({(int(rng.integers(c["n_v"])), int(rng.integers(c["n_e"]))) for _ in range(c["nnz"])})
    ei = torch.tensor(pairs, dtype=torch.int64).t().contiguous()
    x = torch.from_numpy(rng.standard_normal((c["n_v"], c["d"])).astype(np.float32))
    G = torch.from_numpy(rng.standard_normal((c["n_v"], c["d"])).astype(np.float32))
    return ei, x, G


def _rand_convs(c):
    from allset_amd import HalfNLHconv
    torch.manual_seed(c["seed"] % 1000)
    d = c["d"]
    return (HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=c["H"], attention=c["pma"]).eval(),
            HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=c["H"], attention=c["pma"]).eval())


def _col_random_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from allset_amd import dist as adist
        res = []
        for c in _rand_cfgs():
            ei, x, G = _rand_problem(c)
            hg = adist.ColumnShardedHypergraph(ei, c["n_v"], c["n_e"], world, rank, chunks=c["chunks"],
                                               norm=torch.ones(ei.shape[1], dtype=torch.int64))
            hg.v2e = (ei, hg.n_e_pad)
            hg.e2v = (torch.stack([ei[1], ei[0]]), hg.n_v_pad)
            a, b = _rand_convs(c)
            xp = torch.cat([x, x.new_zeros(hg.n_v_pad - c["n_v"], c["d"])])
            Gp = torch.cat([G, G.new_zeros(hg.n_v_pad - c["n_v"], c["d"])])
            xo = xp[hg.v_lo:hg.v_hi].clone().requires_grad_(True)
            if c["pma"]:
                out = adist.colsharded_pma_layer(a, b, xo, hg, kernels=TorchPmaKernels, chunks=c["chunks"])
            else:
                out = adist.colsharded_deepsets_layer(a, b, xo, hg, aggr=c["aggr"], aggregate=_oracle_aggregate, chunks=c["chunks"])
            (out * Gp[hg.v_lo:hg.v_hi]).sum().backward()
            res.append((out.detach().numpy().copy(), xo.grad.numpy().copy()))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_colsharded_layers_random_configurations_two_ranks():
    """24 random (sizes, width, heads, aggregation, chunk count) configurations through the 2-rank column scheme in one spawn:
    d/2 columns per rank, a head shared by both ranks when H = 1, padded owned blocks, chunked exchange."""
    import torch.nn.functional as F
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_col_random_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i, c in enumerate(_rand_cfgs()):
        ei, x, G = _rand_problem(c)
        a, b = _rand_convs(c)
        xr = x.clone().requires_grad_(True)
        rev = torch.stack([ei[1], ei[0]])
        if c["pma"]:
            def pm(p, xin, e_idx, n_dst):
                H, C = p.heads, p.hidden
                o = TorchPmaKernels.aggregate(p.lin_V(xin), p._logits(xin), (e_idx, n_dst), H, 0.2)
                o = p.ln0((o.view(-1, H, C) + p.att_r).view(-1, H * C))
                return p.ln1(o + F.relu(p.rFF(o)))
            v = F.relu(pm(b.prop, F.relu(pm(a.prop, xr, ei, c["n_e"])), rev, c["n_v"]))
        else:
            h = F.relu(a.f_enc(xr))
            e = F.relu(a.f_dec(_oracle_aggregate(h, (ei, c["n_e"]), torch.ones(ei.shape[1], dtype=torch.int64), c["aggr"])))
            g = F.relu(b.f_enc(e))
            v = F.relu(b.f_dec(_oracle_aggregate(g, (rev, c["n_v"]), torch.ones(ei.shape[1], dtype=torch.int64), c["aggr"])))
        (v * G).sum().backward()
        out = torch.cat([torch.from_numpy(results[r][i][0]) for r in range(world)])[:c["n_v"]]
        gx = torch.cat([torch.from_numpy(results[r][i][1]) for r in range(world)])[:c["n_v"]]
        torch.testing.assert_close(out, v.detach(), rtol=1e-4, atol=1e-5, msg=lambda m: f"config {i} {c}: {m}")
        if c["aggr"] in ("add", "mean") or c["pma"]:
            torch.testing.assert_close(gx, xr.grad, rtol=1e-4, atol=1e-4, msg=lambda m: f"config {i} {c}: {m}")


# ---------------------------------------------------------------------------------------------------------------
# opt-in bf16 wire format (SURVEY section 7 mitigation (b)): fp32 everywhere except on the wire, fp32 sums
# ---------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("scheme", ["rows", "columns"])
def test_bf16_wire_format_within_its_restated_tolerance(scheme, monkeypatch):
    """ALLSET_WIRE_DTYPE=bf16: every exchanged activation is rounded once to bf16 (2^-9 relative), sums stay fp32 (the row
    scheme's reduce-scatter becomes an all-to-all of pieces + a local fp32 sum).  Restated tolerance for one V->E->V layer:
    2e-2 of the output scale and 1e-1 of the input-gradient scale (max norm; the gradient passes four LayerNorm backwards over
    16 columns, which amplify the wire's 2^-9 noise) -- against 1e-5 for the exact wire."""
    import torch.nn.functional as F
    monkeypatch.setenv("ALLSET_WIRE_DTYPE", "bf16")            # read by allset_amd.dist at import, i.e. in the spawned ranks
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    if scheme == "rows":
        procs = [ctx.Process(target=_worker, args=(r, world, port, "add", "contiguous", q)) for r in range(world)]
    else:
        procs = [ctx.Process(target=_col_worker, args=(r, world, port, "ds", "add", q, 1)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_v, n_e, d, ei, norm, x, G = _problem(world)
    a, b = _convs(d)
    xr = x.clone().requires_grad_(True)
    h = F.relu(a.f_enc(xr))
    e = F.relu(a.f_dec(_oracle_aggregate(h, (ei, n_e), norm, "add")))
    g = F.relu(b.f_enc(e))
    v = F.relu(b.f_dec(_oracle_aggregate(g, (torch.stack([ei[1], ei[0]]), n_v), norm, "add")))
    (v * G).sum().backward()
    out = torch.cat([torch.from_numpy(np.asarray(r[1])) for r in results])[:n_v]
    gx = torch.cat([torch.from_numpy(np.asarray(r[2])) for r in results])[:n_v]
    err_o = float((out - v.detach()).abs().max()) / float(v.detach().abs().max())
    err_g = float((gx - xr.grad).abs().max()) / float(xr.grad.abs().max())
    assert 1e-6 < err_o < 2e-2 and err_g < 1e-1, (err_o, err_g)       # really rounded on the wire, and within the restated tolerance


def test_bf16_wire_with_the_boundary_vertex_exchange_keeps_the_softmax_statistics_in_fp32(monkeypatch):
    """ALLSET_WIRE_DTYPE=bf16 + ``halo=True`` + PMA: the gathered gradient rows are rounded on the wire, the softmax statistics
    {m + log l, delta} of the backward travel in fp32 in their own all-to-all (as the forward's maxima do) -- rounded to 2^-9 they
    would put exp(a - M) off by ~1 % and the backward on other maxima than the forward.  The layer stays inside the bf16
    wire's restated tolerance, and the wire really is narrow (the result differs from the exact one)."""
    import torch.nn.functional as F
    monkeypatch.setenv("ALLSET_WIRE_DTYPE", "bf16")            # read by allset_amd.dist at import, i.e. in the spawned ranks
    world, H = 2, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pma_worker, args=(r, world, port, q, True, H)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_v, n_e, d, ei, _, x, G = _problem(world)
    a, b = _pma_convs(d, H)

    def pma_module(p, xin, e_idx, n_dst):
        Hh, C = p.heads, p.hidden
        o = TorchPmaKernels.aggregate(p.lin_V(xin), p._logits(xin), (e_idx, n_dst), Hh, 0.2)
        o = p.ln0((o.view(-1, Hh, C) + p.att_r).view(-1, Hh * C))
        return p.ln1(o + F.relu(p.rFF(o)))
    xr = x.clone().requires_grad_(True)
    v = F.relu(pma_module(b.prop, F.relu(pma_module(a.prop, xr, ei, n_e)), torch.stack([ei[1], ei[0]]), n_v))
    (v * G).sum().backward()
    out = torch.cat([torch.from_numpy(r[1]) for r in results])[:n_v]
    gx = torch.cat([torch.from_numpy(r[2]) for r in results])[:n_v]
    err_o = float((out - v.detach()).abs().max()) / float(v.detach().abs().max())
    err_g = float((gx - xr.grad).abs().max()) / float(xr.grad.abs().max())
    assert 1e-6 < err_o < 2e-2 and err_g < 1e-1, (err_o, err_g)


def test_halo_gather_and_scatter_are_transposes_single_rank():
    """``Halo`` without a process group (world 1): gather = index_select by the touched vertices, scatter_add its transpose
    (<gather(x), y> == <x, scatter_add(y)>), scatter_max the per-owned-row maximum; an empty incidence gives empty tables."""
    from allset_amd import dist as adist
    g = torch.Generator().manual_seed(0)
    n_v, d = 29, 8
    ids = torch.randint(0, n_v, (60,), generator=g)
    halo = adist.Halo(ids, n_v, 1, 0)
    assert torch.equal(halo.needed, torch.unique(ids)) and torch.equal(halo.needed[halo.compact_ids], ids)
    x, y = torch.randn(n_v, d, generator=g), torch.randn(halo.n_needed, d, generator=g)
    gx, sy = halo._gather(x), halo._scatter_add(y)
    assert gx.shape == (halo.n_needed, d) and sy.shape == (n_v, d)
    torch.testing.assert_close((gx * y).sum(), (x * sy).sum(), rtol=1e-5, atol=1e-5)
    assert torch.equal(gx, x[halo.needed])
    untouched = torch.ones(n_v, dtype=torch.bool); untouched[halo.needed] = False
    assert float(sy[untouched].abs().max()) == 0.0
    m = halo._scatter_max(y[:, :3].contiguous())
    assert torch.equal(m[halo.needed], y[:, :3]) and float(m[untouched].abs().max()) == 0.0
    # autograd mirrors
    xa = x.clone().requires_grad_(True)
    (adist.halo_gather(xa, halo) * y).sum().backward()
    torch.testing.assert_close(xa.grad, sy)
    ya = y.clone().requires_grad_(True)
    (adist.halo_scatter_add(ya, halo) * x).sum().backward()
    torch.testing.assert_close(ya.grad, gx)
    # a rank without any incidence
    empty = adist.Halo(torch.zeros(0, dtype=torch.int64), n_v, 1, 0)
    assert empty.n_needed == 0 and empty._gather(x).shape == (0, d) and float(empty._scatter_add(torch.zeros(0, d)).abs().sum()) == 0.0


def _halo_model_worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import cases
        from allset_amd import SetGNN, dist as adist
        n_v, n_e, d, ei, _, x, G = _problem(world)
        args = cases.make_args(mode, d, 32, 5, All_num_layers=2)
        torch.manual_seed(11)
        model = SetGNN(args).double().eval()
        owner = adist.partition_hyperedges(torch.bincount(ei[1], minlength=n_e), world, "lpt")
        loc, gids = adist.local_shard(ei, owner, rank)
        keep = owner[ei[1]] == rank
        hg = adist.ShardedHypergraph(loc, n_v, gids.numel(), world, rank, norm=torch.ones(int(keep.sum()), dtype=torch.int64), halo=True)
        hg.v2e = (loc, hg.n_e_local)
        hg.e2v = (torch.stack([loc[1], loc[0]]), hg.n_v_pad)
        hloc = hg.halo_edge_index()
        hg.halo_v2e = (hloc, hg.n_e_local)
        hg.halo_e2v = (torch.stack([hloc[1], hloc[0]]), hg.halo.n_needed)
        sharded = adist.ShardedSetGNN(model, hg, aggregate=_oracle_aggregate, kernels=TorchPmaKernels)
        xp = torch.cat([x, x.new_zeros(hg.n_v_pad - n_v, d)]).double()
        out = sharded(xp[hg.v_lo:hg.v_hi])
        live = max(0, min(hg.v_hi, n_v) - hg.v_lo)
        cot = torch.linspace(-1.0, 1.0, n_v * out.shape[1]).view(n_v, -1)[hg.v_lo:hg.v_lo + live].double()
        (out[:live] * cot).sum().backward()
        sharded.allreduce_grads()
        q.put((rank, out.detach().numpy().copy(), {k: v.numpy().copy() for k, v in model.state_dict().items()},
               {k: p.grad.numpy().copy() for k, p in model.named_parameters() if p.grad is not None}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ds_mean", "ds_add"])       # (the PMA merge keeps its statistics in fp32 by design: layer-level tests above)
def test_sharded_setgnn_with_halo_logits_and_gradients_equal_oracle_float64(mode):
    """Two layers through ``ShardedSetGNN`` on the row partition with the boundary-vertex exchange, everything in float64 (so that
    a mismatch is an algorithm error, not rounding or a relu kink): logits of the owned rows and every all-reduced parameter
    gradient against the oracle's."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cases
    from oracle import allset_oracle as oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_model_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_v, n_e, d, ei, _, x, G = _problem(world)
    args = cases.make_args(mode, d, 32, 5, All_num_layers=2)
    sd = {k: torch.from_numpy(v) for k, v in results[0][2].items()}
    for t in sd.values():
        if t.is_floating_point():
            t.requires_grad_(True)
    ref = oracle.setgnn_forward(sd, args, x.double(), ei, torch.ones(ei.shape[1], dtype=torch.int64))
    (ref * torch.linspace(-1.0, 1.0, n_v * ref.shape[1]).view(n_v, -1).double()).sum().backward()
    got = torch.cat([torch.from_numpy(r[1]) for r in results])[:n_v]
    torch.testing.assert_close(got, ref.detach(), rtol=1e-9, atol=1e-10)
    for r in range(world):
        for k, gnp in results[r][3].items():
            if sd[k].grad is not None:
                torch.testing.assert_close(torch.from_numpy(gnp), sd[k].grad, rtol=1e-7, atol=1e-9, msg=lambda m, k=k: f"{k} (rank {r}): {m}")


def _halo_random_worker(rank, world, port, q):
    """Several odd-shaped problems through the row partition with the boundary-vertex exchange on one process group: fewer
    hyperedges than ranks (a rank without any incidence), vertex counts that do not divide, untouched vertices, a hub vertex in
    every hyperedge, weighted incidences, mean."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from allset_amd import dist as adist
        res = []
        for ci, (n_v, n_e, nnz, aggr, method, hub) in enumerate(_HALO_RANDOM):
            rng = np.random.default_rng(100 + ci)
            pairs = {(int(rng.integers(n_v // 2)), int(rng.integers(n_e))) for _ in range(nnz)}      # upper half of the vertices untouched
            if hub:
                pairs |= {(3, e) for e in range(n_e)}
            ei = torch.tensor(sorted(pairs), dtype=torch.int64).t().contiguous()
            norm = torch.from_numpy(rng.uniform(0.5, 1.5, size=ei.shape[1]).astype(np.float32))
            d = 8
            x = torch.from_numpy(rng.standard_normal((n_v, d)).astype(np.float32))
            sizes = torch.bincount(ei[1], minlength=n_e)
            owner = adist.partition_hyperedges(sizes, world, method)
            loc, gids = adist.local_shard(ei, owner, rank)
            keep = owner[ei[1]] == rank
            hg = adist.ShardedHypergraph(loc, n_v, gids.numel(), world, rank, norm=norm[keep], halo=True)
            hg.v2e = (loc, hg.n_e_local)
            hg.e2v = (torch.stack([loc[1], loc[0]]), hg.n_v_pad)
            hloc = hg.halo_edge_index()
            hg.halo_v2e = (hloc, hg.n_e_local)
            hg.halo_e2v = (torch.stack([hloc[1], hloc[0]]), hg.halo.n_needed)
            a, b = _convs(d)
            xp = torch.cat([x, x.new_zeros(hg.n_v_pad - n_v, d)])
            xo = xp[hg.v_lo:hg.v_hi].clone().requires_grad_(True)
            out = adist.sharded_deepsets_layer(a, b, xo, hg, aggr=aggr, aggregate=_oracle_aggregate)
            cot = torch.linspace(-1, 1, hg.n_v_pad * d).view(hg.n_v_pad, d)[hg.v_lo:hg.v_hi]
            (out * cot).sum().backward()
            res.append((out.detach().numpy().copy(), xo.grad.numpy().copy(), hg.n_v_pad, hg.halo.n_needed))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


# n_v, n_e, nnz, aggr, partition method, hub vertex
_HALO_RANDOM = [(23, 2, 30, "add", "contiguous", False), (41, 7, 90, "mean", "lpt", True), (64, 19, 200, "add", "lpt", True),
                (17, 5, 25, "mean", "contiguous", False)]


@pytest.mark.parametrize("world", [3, 5])
def test_halo_exchange_on_odd_shaped_problems(world):
    import torch.nn.functional as F
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_random_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for ci, (n_v, n_e, nnz, aggr, method, hub) in enumerate(_HALO_RANDOM):
        rng = np.random.default_rng(100 + ci)
        pairs = {(int(rng.integers(n_v // 2)), int(rng.integers(n_e))) for _ in range(nnz)}
        if hub:
            pairs |= {(3, e) for e in range(n_e)}
        ei = torch.tensor(sorted(pairs), dtype=torch.int64).t().contiguous()
        norm = torch.from_numpy(rng.uniform(0.5, 1.5, size=ei.shape[1]).astype(np.float32))
        d = 8
        x = torch.from_numpy(rng.standard_normal((n_v, d)).astype(np.float32))
        a, b = _convs(d)
        n_pad = results[0][ci][2]
        xr = torch.cat([x, x.new_zeros(n_pad - n_v, d)]).requires_grad_(True)     # the pad rows go through the dense tail too
        h = F.relu(a.f_enc(xr))
        e = F.relu(a.f_dec(_oracle_aggregate(h, (ei, n_e), norm, aggr)))
        g = F.relu(b.f_enc(e))
        v = F.relu(b.f_dec(_oracle_aggregate(g, (torch.stack([ei[1], ei[0]]), n_pad), norm, aggr)))
        (v * torch.linspace(-1, 1, n_pad * d).view(n_pad, d)).sum().backward()
        out = torch.cat([torch.from_numpy(results[r][ci][0]) for r in range(world)])
        gx = torch.cat([torch.from_numpy(results[r][ci][1]) for r in range(world)])
        torch.testing.assert_close(out, v.detach(), rtol=1e-5, atol=1e-5, msg=lambda m: f"config {ci}: {m}")
        torch.testing.assert_close(gx, xr.grad, rtol=1e-4, atol=1e-5, msg=lambda m: f"config {ci} grad: {m}")
        assert any(results[r][ci][3] == 0 for r in range(world)) or n_e >= world      # (config 0: some rank has no incidence at all)
