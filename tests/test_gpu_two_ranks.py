"""GPU, world_size 2 on ONE device: two real ranks run the product's sharded layers -- the HIP kernels behind the real
``torch.autograd.Function``s of ``allset_amd/dist.py`` -- and exchange through gloo, device tensors staged through the host
(``dist._host_staged``; RCCL refuses two ranks on one GPU, and the driver's box has one).  What this covers that neither the
2-rank gloo tests on CPU (oracle as the local aggregate) nor the 1-rank RCCL tests (no peer) do: partial sums, (m, l, o)
merges, extreme-key merges, column slices and the asynchronous chunked exchange computed by the HIP path and combined with a
peer's.  Both ranks run every configuration in one spawn; the parent compares with the unsharded HIP layer and with the CPU
oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

N_V, N_E, NNZ = 301, 187, 2600            # odd counts: padded owned blocks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# kind, scheme, aggr / heads, partition method, chunks, width
LAYER_CFGS = [
    ("ds", "rows", "add", "contiguous", 1, 64), ("ds", "rows", "mean", "lpt", 1, 64), ("ds", "rows", "max", "contiguous", 1, 64),
    ("ds", "rows", "min", "lpt", 1, 32),
    # the row partition with the boundary-vertex exchange (dist.Halo): only the rows the local hyperedges touch travel
    ("ds", "rows+halo", "add", "contiguous", 1, 64), ("ds", "rows+halo", "mean", "lpt", 1, 128), ("ds", "rows+halo", "max", "lpt", 1, 64),
    ("ds", "rows+halo", "min", "contiguous", 1, 32),
    ("pma", "rows+halo", 4, "contiguous", 1, 64), ("pma", "rows+halo", 1, "lpt", 1, 64),
    ("ds", "cols", "add", None, 1, 64), ("ds", "cols", "mean", None, 4, 64), ("ds", "cols", "max", None, 1, 128),
    ("ds", "cols", "add", None, 4, 128),
    ("pma", "rows", 4, "contiguous", 1, 64), ("pma", "rows", 1, "lpt", 1, 64),
    ("pma", "cols", 4, None, 1, 64), ("pma", "cols", 1, None, 1, 64), ("pma", "cols", 4, None, 4, 128), ("pma", "cols", 1, None, 4, 64),
]
# mode, scheme, chunks, model kwargs
MODEL_CFGS = [
    ("ds_add", "rows", 1, dict(GPR=True, LearnMask=True)), ("ds_add", "cols", 1, dict(GPR=True, LearnMask=True)),
    ("ds_add", "cols", 4, dict(GPR=True, LearnMask=True)), ("pma_h4", "rows", 1, {}), ("pma_h4", "cols", 4, {}),
    ("ds_mean", "cols", 1, {}),
    # (ds_mean through rows + halo is checked exactly, in float64, by tests/test_dist_cpu.py: on this example's data the fp32 HIP
    #  run lands on the other side of a relu kink -- one weight gradient moves by 3e-4 of its scale)
    ("ds_add", "rows+halo", 1, dict(GPR=True, LearnMask=True)), ("pma_h4", "rows+halo", 1, {}),
    # the reference MLP's default normalisation, TRAINING mode (batch statistics over both ranks' real rows), dropouts off
    ("ds_add", "rows", 1, dict(normalization="bn", dropout=0.0)), ("ds_add", "cols", 1, dict(normalization="bn", dropout=0.0)),
]


def _problem(d, seed=42):
    rng = np.random.default_rng(seed)
    pairs = sorted({(int(rng.integers(N_V)), int(rng.integers(N_E))) for _ in range(NNZ)} | {(0, e) for e in range(N_E)})
    ei = torch.tensor(pairs, dtype=torch.int64).t().contiguous()
    norm = torch.from_numpy(rng.uniform(0.5, 1.5, size=ei.shape[1]).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((N_V, d)).astype(np.float32))
    G = torch.from_numpy(rng.standard_normal((N_V, d)).astype(np.float32))
    return ei, norm, x, G


def _model_seed(kw):
    """Data seed of a model configuration.  BatchNorm in training mode couples all rows, so ONE relu input within fp32 rounding
    of zero moves every parameter gradient by percents (a kink: either side's gradient is right) -- and with ~250k relu inputs
    most random examples have one: of the seeds 7..19, a 1e-6 perturbation of x moves the float64 oracle's own gradient by
    0.5-4 % for all but 11 and 17.  The bn-train configurations use 11, and the test below asserts that the example is stable."""
    return 11 if kw.get("normalization") == "bn" else 7


def _convs(kind, arg, d):
    from allset_amd import HalfNLHconv
    torch.manual_seed(3)
    attn = kind == "pma"
    H = arg if attn else 1
    return (HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=H, attention=attn).eval(),
            HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=H, attention=attn).eval())


_HYB = {}


def _shard(adist, scheme, ei, norm, world, rank, method, chunks, dev):
    if scheme == "hybrid":          # 2 target groups x world / 2 column groups (round 5); the groups are created once per process
        if world not in _HYB:
            _HYB[world] = adist.hybrid_groups(world, 2, rank)
        cg, gg = _HYB[world]
        return adist.ColumnShardedHypergraph(ei.to(dev), N_V, N_E, world, rank, norm=None if norm is None else norm.to(dev),
                                             row_groups=2, col_group=cg, gather_group=gg).build_incidences()
    if scheme == "cols":
        return adist.ColumnShardedHypergraph(ei.to(dev), N_V, N_E, world, rank, norm=None if norm is None else norm.to(dev),
                                             chunks=chunks).build_incidences()
    owner = adist.partition_hyperedges(torch.bincount(ei[1], minlength=N_E), world, method)
    loc, gids = adist.local_shard(ei, owner, rank)
    keep = owner[ei[1]] == rank
    return adist.ShardedHypergraph(loc.to(dev), N_V, gids.numel(), world, rank, norm=None if norm is None else norm[keep].to(dev),
                                   inc_ids=keep.nonzero().reshape(-1).to(dev), halo=scheme == "rows+halo").build_incidences()


def _worker(rank, world, port, q, which="two"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (os.path.dirname(HERE), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if which == "tiny":             # 11 vertices, 5 hyperedges over 8 ranks: ranks without hyperedges, one-row vertex blocks
            globals().update(N_V=TINY[0], N_E=TINY[1], NNZ=TINY[2])
        _run_configs(rank, world, dev, q, *((LAYER_CFGS, MODEL_CFGS) if which == "two" else
                                          ((LAYER_CFGS_TINY, []) if which == "tiny" else (LAYER_CFGS_8, []))))
    except BaseException:                       # report instead of leaving the parent waiting for a result that never comes
        import traceback
        q.put((rank, "ERROR\n" + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run_configs(rank, world, dev, q, layer_cfgs, model_cfgs):
    if True:
        import cases
        from allset_amd import SetGNN, dist as adist
        res = {}
        for cfg in layer_cfgs:
            kind, scheme, arg, method, chunks, d = cfg
            ei, norm, x, G = _problem(d)
            hg = _shard(adist, scheme, ei, norm if kind == "ds" else None, world, rank, method, chunks, dev)
            a, b = _convs(kind, arg, d)
            a.to(dev); b.to(dev)
            xp = torch.cat([x, x.new_zeros(hg.n_v_pad - N_V, d)])
            Gp = torch.cat([G, G.new_zeros(hg.n_v_pad - N_V, d)])
            xo = xp[hg.v_lo:hg.v_hi].to(dev).requires_grad_(True)
            if kind == "ds":
                layer = adist.colsharded_deepsets_layer if scheme in ("cols", "hybrid") else adist.sharded_deepsets_layer
                out = layer(a, b, xo, hg, aggr=arg, **({"chunks": chunks} if scheme == "cols" else {}))
            else:
                layer = adist.colsharded_pma_layer if scheme in ("cols", "hybrid") else adist.sharded_pma_layer
                out = layer(a, b, xo, hg, **({"chunks": chunks} if scheme == "cols" else {}))
            (out * Gp[hg.v_lo:hg.v_hi].to(dev)).sum().backward()
            params = list(a.parameters()) + list(b.parameters())
            adist.allreduce_grads(params)
            res[("layer",) + cfg] = (out.detach().cpu().numpy(), xo.grad.cpu().numpy(), [p.grad.cpu().numpy() for p in params])
        for cfg in model_cfgs:
            mode, scheme, chunks, kw = cfg
            d = 64
            ei, norm, x, G = _problem(d, seed=_model_seed(kw))
            args = cases.make_args(mode, d, 64, 5, All_num_layers=2, **kw)
            torch.manual_seed(11)
            model = SetGNN(args, norm if kw.get("LearnMask") else None).eval()
            keep_rank_dropout = adist._rank_dropout
            if kw.get("normalization") == "bn":
                model.train()
                adist._rank_dropout = lambda t, p, training: t        # the hard-wired input dropout (models.py:473) off
            if kw.get("LearnMask"):
                with torch.no_grad():
                    model.Importance.copy_(torch.linspace(0.5, 1.5, ei.shape[1]))
            model.to(dev)
            nrm = norm if kw.get("LearnMask") else torch.ones(ei.shape[1], dtype=torch.int64)
            hg = _shard(adist, scheme, ei, nrm, world, rank, "contiguous", chunks, dev)
            sharded = adist.ShardedSetGNN(model, hg)
            xp = torch.cat([x, x.new_zeros(hg.n_v_pad - N_V, d)])
            out = sharded(xp[hg.v_lo:hg.v_hi].to(dev))
            live = max(0, min(hg.v_hi, N_V) - hg.v_lo)
            cot = torch.linspace(-1.0, 1.0, N_V * out.shape[1]).view(N_V, -1)[hg.v_lo:hg.v_lo + live].to(dev)
            (out[:live] * cot).sum().backward()
            sharded.allreduce_grads()
            adist._rank_dropout = keep_rank_dropout
            grads = {k: p.grad.cpu().numpy() for k, p in model.named_parameters() if p.grad is not None}
            res[("model", mode, scheme, chunks, tuple(sorted(kw)))] = (
                out.detach().cpu().numpy(), grads, {k: v.cpu().numpy() for k, v in model.state_dict().items()})
        q.put((rank, res))


# The N = 8 shapes with eight real ranks: d / P = 16 columns per rank (64-byte gather rows, column blocks of 16 in the Linear kernels),
# PMA with two ranks sharing each of 4 heads and its logits riding in the gathered row's line, 8 pieces per all-to-all, chunks.
LAYER_CFGS_8 = [
    ("ds", "cols", "add", None, 1, 128), ("ds", "cols", "mean", None, 2, 128), ("ds", "cols", "max", None, 1, 128),
    ("pma", "cols", 4, None, 1, 128), ("pma", "cols", 4, None, 2, 128), ("pma", "cols", 1, None, 1, 128),
    ("ds", "rows", "add", "lpt", 1, 64), ("ds", "rows", "max", "contiguous", 1, 64), ("pma", "rows", 4, "contiguous", 1, 64),
    ("ds", "rows+halo", "add", "lpt", 1, 64), ("ds", "rows+halo", "mean", "contiguous", 1, 128), ("pma", "rows+halo", 4, "lpt", 1, 64),
    ("ds", "rows+halo", "max", "lpt", 1, 64),
    # the hybrid partition: 2 target groups x 4 column groups -- at d = 128 rank (a, b) gathers rows of 32 columns (128 bytes) for
    # half of the targets; the repack-free blocked exchange inside the column group + the all-gather across the two groups
    ("ds", "hybrid", "add", None, 1, 128), ("ds", "hybrid", "max", None, 1, 128), ("ds", "hybrid", "mean", None, 1, 64),
    ("pma", "hybrid", 4, None, 1, 128), ("pma", "hybrid", 1, None, 1, 128),
]


TINY = (11, 5, 14)
LAYER_CFGS_TINY = [("ds", "rows", "add", "contiguous", 1, 64), ("ds", "rows+halo", "add", "contiguous", 1, 64),
                   ("ds", "rows+halo", "mean", "lpt", 1, 64), ("ds", "rows+halo", "max", "lpt", 1, 64), ("ds", "cols", "add", None, 1, 128), ("ds", "cols", "max", None, 1, 128),
                   ("pma", "rows", 4, "contiguous", 1, 64), ("pma", "rows+halo", 4, "contiguous", 1, 64), ("pma", "cols", 4, None, 1, 128)]


def _spawn_ranks(world, which):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, which)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(world):
            r, payload = q.get(timeout=300)
            assert not isinstance(payload, str), f"rank {r}: {payload}"
            results[r] = payload
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    return results


@pytest.fixture(scope="module")
def eight_ranks():
    return _spawn_ranks(8, "eight")


@pytest.fixture(scope="module")
def eight_ranks_tiny():
    return _spawn_ranks(8, "tiny")


@pytest.mark.parametrize("cfg", LAYER_CFGS_TINY, ids=lambda c: "-".join(str(t) for t in c))
def test_eight_ranks_on_a_tiny_hypergraph(cfg, eight_ranks_tiny, device, monkeypatch):
    """More ranks than hyperedges: 11 vertices (blocks of two rows, the last ranks' blocks all padding) and 5 hyperedges over EIGHT
    ranks -- ranks without a single incidence, empty compact tables, zero-row MLP calls -- through both partitions and the
    boundary-vertex exchange, against the unsharded HIP layer and the oracle."""
    import sys
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, "N_V", TINY[0]); monkeypatch.setattr(mod, "N_E", TINY[1]); monkeypatch.setattr(mod, "NNZ", TINY[2])
    _check_layer(cfg, eight_ranks_tiny, 8, device)


@pytest.fixture(scope="module")
def two_ranks():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(world):
            r, payload = q.get(timeout=420)
            assert not isinstance(payload, str), f"rank {r}: {payload}"
            results[r] = payload
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    return results


@pytest.mark.parametrize("cfg", LAYER_CFGS_8, ids=lambda c: "-".join(str(t) for t in c))
def test_eight_rank_hip_layer_equals_unsharded_hip_layer_and_oracle(cfg, eight_ranks, device):
    _check_layer(cfg, eight_ranks, 8, device)


@pytest.mark.parametrize("cfg", LAYER_CFGS, ids=lambda c: "-".join(str(t) for t in c))
def test_two_rank_hip_layer_equals_unsharded_hip_layer_and_oracle(cfg, two_ranks, device):
    _check_layer(cfg, two_ranks, 2, device)


def _check_layer(cfg, two_ranks, world, device):
    import torch.nn.functional as F
    from allset_amd import Incidence
    from oracle import allset_oracle as oracle
    kind, scheme, arg, method, chunks, d = cfg
    ei, norm, x, G = _problem(d)
    key = ("layer",) + cfg
    out = torch.cat([torch.from_numpy(two_ranks[r][key][0]) for r in range(world)])[:N_V]
    gx = torch.cat([torch.from_numpy(two_ranks[r][key][1]) for r in range(world)])[:N_V]
    pg = [torch.from_numpy(t) for t in two_ranks[0][key][2]]
    for r in range(1, world):
        for g0, g1 in zip(two_ranks[0][key][2], two_ranks[r][key][2]):
            np.testing.assert_array_equal(g0, g1)                  # after the all-reduce every rank holds the same gradients

    # (a) the unsharded layer on the HIP kernels
    a, b = _convs(kind, arg, d)
    a.to(device); b.to(device)
    inc = Incidence.from_edge_index(ei.to(device), n_src=N_V, n_dst=N_E)
    xr = x.to(device).requires_grad_(True)
    ag = arg if kind == "ds" else "add"
    nrm = norm.to(device) if kind == "ds" else None
    ref = F.relu(b(F.relu(a(xr, inc, nrm, ag)), inc.reversed(n_dst=N_V), nrm, ag))
    (ref * G.to(device)).sum().backward()
    torch.testing.assert_close(out, ref.detach().cpu(), rtol=1e-4, atol=1e-4)
    ties = kind == "ds" and arg in ("max", "min")                  # exact ties may route a gradient to another incidence
    if not ties:
        torch.testing.assert_close(gx, xr.grad.cpu(), rtol=1e-4, atol=1e-4 * max(1.0, float(xr.grad.abs().max())))
        for got, p in zip(pg, list(a.parameters()) + list(b.parameters())):
            torch.testing.assert_close(got, p.grad.cpu(), rtol=1e-3, atol=1e-3 * max(1.0, float(p.grad.abs().max())))

    # (b) the CPU oracle (restatement of the reference's layer) with the same parameters
    a, b = _convs(kind, arg, d)
    sd = {f"V2EConvs.0.{k}": v.detach().clone() for k, v in a.state_dict().items()}
    sd.update({f"E2VConvs.0.{k}": v.detach().clone() for k, v in b.state_dict().items()})
    H = arg if kind == "pma" else 1
    onorm = norm if kind == "ds" else torch.ones(ei.shape[1])
    xo = x.clone().requires_grad_(True)
    e = oracle.halfnlhconv_forward(sd, "V2EConvs.0.", xo, ei, onorm, ag, kind == "pma", H, "ln")
    if e.shape[0] < N_E:
        e = torch.cat([e, e.new_zeros(N_E - e.shape[0], d)])
    v = oracle.halfnlhconv_forward(sd, "E2VConvs.0.", F.relu(e), torch.stack([ei[1], ei[0]]), onorm, ag, kind == "pma", H, "ln")
    v = F.relu(v)
    (v * G[:v.shape[0]]).sum().backward()
    torch.testing.assert_close(out[:v.shape[0]], v.detach(), rtol=1e-4, atol=1e-4)
    if not ties:
        torch.testing.assert_close(gx, xo.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(xo.grad.abs().max())))


@pytest.mark.parametrize("cfg", MODEL_CFGS, ids=lambda c: "-".join(str(t) for t in c[:3]) + ("-gpr-mask" if c[3].get("GPR") else "") +
                         ("-bn-train" if c[3].get("normalization") == "bn" else ""))
def test_two_rank_sharded_setgnn_equals_oracle(cfg, two_ranks):
    import cases
    from oracle import allset_oracle as oracle
    mode, scheme, chunks, kw = cfg
    key = ("model", mode, scheme, chunks, tuple(sorted(kw)))
    d = 64
    ei, norm, x, G = _problem(d, seed=_model_seed(kw))
    args = cases.make_args(mode, d, 64, 5, All_num_layers=2, **kw)
    # the yardstick is the oracle in float64: two layers of stacked LayerNorms put the fp32 oracle itself ~1e-3 of the gradient
    # scale away from it, which is no statement about the sharding
    sd = {k: (torch.from_numpy(v).double() if torch.from_numpy(v).is_floating_point() else torch.from_numpy(v).clone())
          for k, v in two_ranks[0][key][2].items()}
    for t in sd.values():
        if t.is_floating_point():
            t.requires_grad_(True)
    nrm = norm.double() if kw.get("LearnMask") else torch.ones(ei.shape[1], dtype=torch.int64)
    bn_train = kw.get("normalization") == "bn"                 # training mode without dropouts: batch statistics
    ref = oracle.setgnn_forward(sd, args, x.double(), ei, nrm, drop=(lambda t, p: t) if bn_train else None)
    if bn_train:                                               # the example is not on a relu kink at fp32 rounding (see _model_seed)
        def oracle_grads(xin):
            sd2 = {k: (t.detach().clone().requires_grad_(True) if t.is_floating_point() else t.clone()) for k, t in sd.items()}
            r2 = oracle.setgnn_forward(sd2, args, xin, ei, nrm, drop=lambda t, p: t)
            (r2 * torch.linspace(-1.0, 1.0, N_V * r2.shape[1]).view(N_V, -1).double()).sum().backward()
            return {k: t.grad for k, t in sd2.items() if t.is_floating_point() and t.grad is not None}
        g0 = oracle_grads(x.double())
        gen = torch.Generator().manual_seed(0)
        for _ in range(3):
            g1 = oracle_grads(x.double() + 1e-6 * torch.randn(x.shape, generator=gen, dtype=torch.float64))
            gscale = max(float(v.abs().max()) for v in g0.values())
            assert max(float((g0[k] - g1[k]).abs().max()) for k in g0) <= 1e-4 * gscale, "the example sits on a relu kink"
    cot = torch.linspace(-1.0, 1.0, N_V * ref.shape[1]).view(N_V, -1)
    (ref * cot.double()).sum().backward()
    got = torch.cat([torch.from_numpy(two_ranks[r][key][0]) for r in range(2)])[:N_V]
    torch.testing.assert_close(got.double(), ref.detach(), rtol=1e-4, atol=1e-4)
    scale = max(float(t.grad.abs().max()) for t in sd.values() if t.requires_grad and t.grad is not None)
    for r in range(2):
        grads = two_ranks[r][key][1]
        for k, t in sd.items():
            if t.requires_grad and t.grad is not None and k in grads:
                torch.testing.assert_close(torch.from_numpy(grads[k]).double(), t.grad, rtol=1e-3, atol=1e-4 * max(scale, 1.0),
                                           msg=lambda m, k=k: f"{k} (rank {r}): {m}")


def test_bench_two_ranks_one_gpu_gloo():
    """bench.py --gpus 2 as the driver launches it, except for the wire: ALLSET_DIST_BACKEND=gloo (host-staged collectives, both
    ranks on the one device).  Both partitions, blocking and chunked (asynchronous) column exchange."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    for model, chunks in (("deepsets", 1), ("deepsets", 2), ("pma", 2)):
        env = dict(os.environ, ALLSET_DIST_BACKEND="gloo")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--n-per-gpu", "20000", "--model", model, "--pipeline-chunks", str(chunks)]
        res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, res.stdout[-2000:]
        line = json.loads(lines[0])
        assert line["n_gpus"] == 2 and "gloo" in line["config"]["collectives"]
        parts = line["partitions"]
        assert {"rows", "columns"} <= set(parts) and all("error" not in parts[k] for k in ("rows", "columns")), parts
        assert line["config"]["nnz"] == 2 * 20000 * 16
        for k in ("rows", "columns"):
            assert abs(parts[k]["value"] - line["config"]["nnz"] * 128 / (parts[k]["ms_per_step"] * 1e-3)) <= 1e-6 * parts[k]["value"]


def test_bench_bare_command_spawns_its_own_ranks():
    """``python bench.py --gpus 2`` with NO launcher around it (no RANK / WORLD_SIZE): bench.py re-launches itself under
    torch.distributed.run and the bare command prints the one line."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ALLSET_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n-per-gpu", "20000", "--d", "128",
           "--chunk-entry", "2"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["nnz"] == 2 * 20000 * 16
    parts = line["partitions"]
    meta = ("note", "value_note", "fastest_exact")
    assert [k for k in parts if k not in meta] == ["rows", "columns", "columns+chunks2", "rows+bf16wire"]
    assert all("error" not in parts[k] for k in ("rows", "columns", "columns+chunks2", "rows+bf16wire")), parts
    exact = ["rows", "columns", "columns+chunks2"]
    best = min(exact, key=lambda k: parts[k]["ms_per_step"])
    # default --shard rows (round 5): `value` is the north star's partition, the fastest exact execution a label
    assert parts["rows"]["is_value"] and line["value"] == parts["rows"]["value"] and line["config"]["partition"] == "rows"
    assert parts["fastest_exact"] == best
    assert any("all_to_all" in k for k in line["preflight"]["collectives"])
    early = [l for l in res.stderr.splitlines() if l.startswith("[bench] early line")]
    assert len(early) == 1 and json.loads(early[0].split(": ", 1)[1])["partitions"]["rows"]["is_value"]


def test_bench_hung_second_region_keeps_the_first_regions_line():
    """Rank 1 parks at the start of the column partition: rank 0 blocks in its first exchange.  The watchdog prints the line with
    the row partition's result and both ranks exit 0."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ALLSET_DIST_BACKEND="gloo", ALLSET_BENCH_TEST_HANG="columns:1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n-per-gpu", "20000",
           "--region-timeout", "8", "--preflight", "off"]       # (first region: 3 x 8 s)
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    parts = line["partitions"]
    assert parts["rows"]["is_value"] and line["value"] == parts["rows"]["value"] and "timeout" in parts["columns"]["error"]
    assert line["roofline"]["launches"] > 0 and line["config"]["partition"] == "rows"


@pytest.mark.parametrize("model", ["deepsets", "pma"])
def test_bench_eight_ranks_one_gpu_gloo(model):
    """The N = 8 shapes of the job with eight real ranks (one device, gloo): d / P = 16 columns per rank -- 64-byte gather rows, the
    short-row kernels, PMA with two ranks sharing a head (P > H) and its logits riding in the row's line, column-blocked Linear
    operands of width 16, the chunked exchange with 8 pieces per all-to-all.  Every region must finish and the exact partitions must
    describe the same job."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ALLSET_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--n-per-gpu", "8000",
           "--model", model, "--chunk-entry", "2", "--region-timeout", "300"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    parts = line["partitions"]
    # rows first, then the hybrid partition (2 target groups x 4 column groups: 32 columns per rank, 128-byte gather rows; its two
    # process groups are created inside the run), then the pure column partition and its variants
    keys = ["rows", "hybrid2x4", "columns", "columns+chunks2", "rows+bf16wire"]
    assert [k for k in parts if k not in ("note", "value_note", "fastest_exact")] == keys
    assert all("error" not in parts[k] and parts[k]["ms_per_step"] > 0 for k in keys), parts
    assert "2 target groups x 4 column groups" in parts["hybrid2x4"]["parallelism"] and parts["rows"]["is_value"]
    for k in ("rows", "hybrid2x4", "columns"):      # the same job under every exact partition
        assert abs(parts[k]["value"] - line["config"]["nnz"] * 128 / (parts[k]["ms_per_step"] * 1e-3)) <= 1e-6 * parts[k]["value"]
    assert line["n_gpus"] == 8 and line["config"]["nnz"] == 8 * 8000 * 16 and line["config"]["n_v"] == 64000
    assert "column-shard x8" in parts["columns"]["parallelism"] and "d/8 columns" in parts["columns"]["parallelism"]
    assert len(line["preflight"]["collectives"]) >= 6


def test_bench_two_ranks_locality_variant_with_halo_exchange():
    """``bench.py --gpus 2 --locality 0.9`` (gloo, one device): the row partition runs the boundary-vertex exchange on the HIP path --
    index_select of the asked-for rows, all-to-all with per-peer counts, the segment-sum kernel as the fixed-order scatter-add."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ALLSET_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n-per-gpu", "20000",
           "--locality", "0.9", "--chunk-entry", "0", "--no-wire-entry"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    parts = line["partitions"]
    assert "boundary-vertex exchange" in parts["rows"]["parallelism"] and "error" not in parts["rows"] and "error" not in parts["columns"]
    assert "VARIANT workload" in line["config"]["workload"]


def test_halo_layer_on_a_rank_without_incidences_single_process(device):
    """``ShardedHypergraph(halo=True)`` whose local incidence is EMPTY (fewer hyperedges than ranks): compact table of zero rows,
    aggregates of zero rows, the owned block's output is what the dense tail makes of zero sums -- on the HIP path, no collective
    (world 1)."""
    from allset_amd import dist as adist
    a, b = _convs("ds", "add", 64)
    a.to(device); b.to(device)
    empty = torch.zeros(2, 0, dtype=torch.int64, device=device)
    hg = adist.ShardedHypergraph(empty, 40, 0, 1, 0, halo=True).build_incidences()
    assert hg.halo.n_needed == 0
    x = torch.randn(40, 64, device=device, requires_grad=True)
    out = adist.sharded_deepsets_layer(a, b, x, hg, aggr="add")
    out.sum().backward()
    ref = torch.relu(b.f_dec(torch.zeros(40, 64, device=device)))
    torch.testing.assert_close(out.detach(), ref.detach(), rtol=1e-5, atol=1e-5)
    assert float(x.grad.abs().max()) == 0.0
    hg2 = adist.ShardedHypergraph(empty, 40, 0, 1, 0, halo=True).build_incidences()
    c, e = _convs("pma", 4, 64)
    c.to(device); e.to(device)
    x2 = torch.randn(40, 64, device=device, requires_grad=True)
    out2 = adist.sharded_pma_layer(c, e, x2, hg2)
    out2.sum().backward()
    assert out2.shape == (40, 64) and torch.isfinite(out2).all()


@pytest.mark.parametrize("method", ["AllDeepSets", "AllSetTransformer"])
def test_sharded_training_example_reproduces_the_single_gpu_loss_curve(method):
    """examples/sharded_train.py (the multi-GPU usage of INTEGRATION.md 3b): ten epochs of the reference's training-loop body with
    dropouts off on ONE rank and on TWO ranks (gloo, one device) under each partition -- the loss sequence is a deterministic function
    of the initial weights and must not depend on the number of ranks or on the partition."""
    import re
    import subprocess
    root = os.path.dirname(HERE)
    script = os.path.join(root, "examples", "sharded_train.py")
    base = [script, "--method", method, "--epochs", "10", "--eval-mode", "--n-v", "1500"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}

    def losses(out):
        return [float(v) for v in re.findall(r"loss ([0-9.]+)", out)]

    one = subprocess.run([sys.executable] + base + ["--partition", "rows"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    ref = losses(one.stdout)
    assert len(ref) == 10 and ref[-1] < 0.7 * ref[0]
    env2 = dict(env, ALLSET_DIST_BACKEND="gloo")
    # the three two-rank runs side by side (six processes sharing the device: each is launch-bound and small)
    procs = {}
    for part in ("rows", "rows+halo", "columns"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + base + ["--partition", part]
        procs[part] = subprocess.Popen(cmd, cwd=root, env=env2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    outs = {}
    try:
        for part, pr in procs.items():
            outs[part] = pr.communicate(timeout=300) + (pr.returncode,)
    finally:
        for pr in procs.values():
            if pr.poll() is None:
                pr.kill()
    for part, (so, se, rc) in outs.items():
        assert rc == 0, part + so[-2000:] + se[-2000:]
        got = losses(so)
        assert len(got) == 10
        np.testing.assert_allclose(got[:4], ref[:4], rtol=2e-4, err_msg=part)         # plain parity before Adam amplifies rounding
        np.testing.assert_allclose(got, ref, rtol=2e-2, err_msg=part)
