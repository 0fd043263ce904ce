"""GPU: BASELINE.json configs[3] and configs[4] at their FULL global sizes on one MI355X (288 GB holds both), through
size-independent properties -- the oracle would take many minutes here.

configs[3]: |V| = |E| = 8M, 16 members per hyperedge (nnz = 128M), d = 128, AllSetTransformer (PMA, 4 heads): the global
problem the 8 ranks shard.  n*d = 2^30 elements = 4 GiB per matrix: every row offset past 2^31 bytes is exercised.
configs[4]: |V| = |E| = 2M, truncated-Zipf hyperedge sizes <= 4096 (mean 16), d = 256, bf16 storage, PMA; plus the
nnz-balanced `lpt` partition into 8 hyperedge bins the config asks for.
The 8-rank execution itself is covered by world-size-2 gloo tests (tests/test_dist_cpu.py) and the 1-rank RCCL group
(tests/test_gpu_bench_cli.py); no multi-GPU box is available to the test suite."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3(device):
    from allset_amd import Incidence
    from allset_amd.synthetic import random_hypergraph
    n = 8_000_000
    hg = random_hypergraph(n, n, 16, seed=77, device=device)
    v2e = Incidence.from_edge_index(hg.edge_index, n_src=n, n_dst=n)
    yield n, hg, v2e
    del hg, v2e
    torch.cuda.empty_cache()


def test_configs3_global_size_aggregation_properties(c3, device):
    from allset_amd import deepsets_aggregate
    n, hg, v2e = c3
    assert hg.nnz == 128_000_000 and v2e.by_dst.max_deg == 16
    e2v = v2e.reversed(n_dst=n)
    g = torch.Generator(device=device).manual_seed(1)
    x = torch.randn(n, 128, device=device, generator=g)
    y = torch.randn(n, 128, device=device, generator=g)
    e = deepsets_aggregate(x, v2e, None, "add")
    deg_v = (v2e.by_src.rowptr[1:] - v2e.by_src.rowptr[:-1]).double()
    lhs, rhs = e.double().sum(0), (x.double() * deg_v[:, None]).sum(0)
    scale = (x.double().abs() * deg_v[:, None]).sum(0)                           # the sums cancel: measure against sum |terms|
    assert float(((lhs - rhs).abs() / scale).max()) < 1e-7                       # conservation: a checksum of checksums
    xt = deepsets_aggregate(y, e2v, None, "add")
    a, b = (e.double() * y.double()).sum(), (x.double() * xt.double()).sum()
    assert abs(float(a - b)) <= 1e-6 * float(a.abs() + b.abs())                  # <A x, y> = <x, A^T y>
    # rows whose byte offset is past 4 GiB are gathered and written correctly
    for row in (n - 1, n - 12345, 4_194_304 + 17, 5_000_000):
        s, t = int(v2e.by_dst.rowptr[row]), int(v2e.by_dst.rowptr[row + 1])
        ref = x[v2e.by_dst.col[s:t].long()].double().sum(0)
        torch.testing.assert_close(e[row].double(), ref, rtol=1e-5, atol=1e-5)


def test_configs3_global_size_pma_properties(c3, device):
    from allset_amd import pma_aggregate
    n, hg, v2e = c3
    H = 4
    g = torch.Generator(device=device).manual_seed(2)
    alpha = torch.randn(n, H, device=device, generator=g)
    out, m, l = pma_aggregate(torch.ones(n, 128, device=device), alpha, v2e, H, 0.2)
    assert float((out - 1).abs().max()) < 1e-5 and float(l.min()) >= 1.0 - 1e-5      # convex combination; max term exp(0)
    del out
    V = torch.randn(n, 128, device=device, generator=g).requires_grad_(True)
    ar = alpha.clone().requires_grad_(True)
    o2, _, _ = pma_aggregate(V, ar, v2e, H, 0.2)
    # sampled targets against a float64 segment softmax
    for row in (0, n // 2 + 3, n - 1):
        s, t = int(v2e.by_dst.rowptr[row]), int(v2e.by_dst.rowptr[row + 1])
        src = v2e.by_dst.col[s:t].long()
        p = torch.softmax(torch.nn.functional.leaky_relu(alpha[src].double(), 0.2), dim=0)          # [deg, H]
        ref = (p[:, :, None] * V.detach()[src].double().view(-1, H, 32)).sum(0).reshape(-1)
        torch.testing.assert_close(o2[row].double(), ref, rtol=1e-5, atol=1e-5)
    o2.sum().backward()
    # every target's weights sum to 1 per head, so d(sum out)/dV summed over sources = number of targets, per column
    assert abs(float(V.grad.double().sum(0).mean()) - n) < 1e-3 * n
    # softmax shift invariance: per target and head the logit gradients of its incidences sum to 0, so divided by the
    # leaky-relu slope they sum to 0 over all sources too
    assert bool(torch.isfinite(ar.grad).all())
    ga = ar.grad.double() / torch.where(alpha > 0, 1.0, 0.2).double()
    assert float((ga.sum(0).abs() / ga.abs().sum(0)).max()) < 1e-6


def test_configs3_row_count_dense_tail_one_pass_backward(device):
    """The dense tail at the configs[3] row count (8M + 5 rows x 128): forward and the one-pass backward on sampled rows
    against float64, and additivity of the weight gradient over a row split."""
    from allset_amd import dense
    n8, d = 8_000_005, 128
    g = torch.Generator(device=device).manual_seed(3)
    x = torch.randn(n8, d, device=device, generator=g)
    W = torch.randn(d, d, device=device, generator=g) / d ** 0.5
    b = torch.randn(d, device=device, generator=g)
    gam = 1 + 0.2 * torch.randn(d, device=device, generator=g)
    bet = 0.3 * torch.randn(d, device=device, generator=g)
    G = torch.randn(n8, d, device=device, generator=g)
    mask = torch.empty(dense.activation_mask_words(n8, d), dtype=torch.int32, device=device)
    y, st = dense.fused_linear_fwd(x, W, b, gam, bet, 1e-5, False, 0.0, 0, True, 0.0, 0, None, mask)
    rows = torch.cat([torch.arange(0, 32, device=device), torch.randint(0, n8, (2000,), device=device, generator=g),
                      torch.arange(n8 - 32, n8, device=device)])
    xr = x[rows].double().requires_grad_(True)
    ref = torch.relu(torch.nn.functional.layer_norm(xr, (d,), gam.double(), bet.double(), 1e-5) @ W.double().t() + b.double())
    torch.testing.assert_close(y[rows].double(), ref.detach(), rtol=1e-4, atol=1e-5)
    (ref * G[rows].double()).sum().backward()
    gx, dg, db, gw, gb = dense.fused_linear_bwd_all(G, mask, 0.0, W, x, st, gam, bet, False, 0.0, 0)
    torch.testing.assert_close(gx[rows].double(), xr.grad, rtol=1e-4, atol=1e-5)
    h = n8 // 2 + 7
    words_h = dense.activation_mask_words(h - h % 16, d)       # mask blocks are 16 rows: split on a block boundary
    h = h - h % 16
    _, _, _, gw1, gb1 = dense.fused_linear_bwd_all(G[:h], mask[:words_h], 0.0, W, x[:h], st[:h], gam, bet, False, 0.0, 0)
    _, _, _, gw2, gb2 = dense.fused_linear_bwd_all(G[h:], mask[words_h:], 0.0, W, x[h:], st[h:], gam, bet, False, 0.0, 0)
    torch.testing.assert_close(gw1 + gw2, gw, rtol=1e-4, atol=1e-5 * float(gw.abs().max()))
    torch.testing.assert_close(gb1 + gb2, gb, rtol=1e-4, atol=1e-5 * float(gb.abs().max()))


@pytest.fixture(scope="module")
def c4(device):
    from allset_amd import Incidence
    from allset_amd.synthetic import random_hypergraph
    n = 2_000_000
    hg = random_hypergraph(n, n, 16, seed=99, device=device, dist="zipf", max_degree=4096)
    v2e = Incidence.from_edge_index(hg.edge_index, n_src=n, n_dst=n)
    yield n, hg, v2e
    del hg, v2e
    torch.cuda.empty_cache()


def test_configs4_zipf_bf16_pma_properties(c4, device):
    """Power-law sizes up to 4096, d = 256, bf16 storage (fp32 accumulation and softmax statistics)."""
    from allset_amd import deepsets_aggregate, pma_aggregate, ops
    n, hg, v2e = c4
    csr = v2e.by_dst
    assert 3000 < csr.max_deg <= 4096 and 10 < hg.nnz / n < 22
    assert csr.row_order is not None                     # skewed sizes: the long-rows-first processing order is on
    H, d = 4, 256
    g = torch.Generator(device=device).manual_seed(5)
    alpha = torch.randn(n, H, device=device, generator=g)
    ones = torch.ones(n, d, device=device, dtype=torch.bfloat16)
    out, m, l = pma_aggregate(ones, alpha, v2e, H, 0.2)
    nonempty = csr.rowptr[1:] > csr.rowptr[:-1]
    assert out.dtype == torch.bfloat16 and float((out[nonempty].float() - 1).abs().max()) < 1e-2     # convexity, to bf16 rounding
    V = torch.randn(n, d, device=device, generator=g).to(torch.bfloat16)
    o1, m1, l1 = ops.pma_fwd(csr.rowptr, csr.col, alpha, V, H, 0.2, n, row_order=csr.row_order)
    o0, m0, l0 = ops.pma_fwd(csr.rowptr, csr.col, alpha, V, H, 0.2, n, row_order=None)
    assert torch.equal(o1, o0) and torch.equal(m1, m0) and torch.equal(l1, l0)       # the order changes speed only
    # the longest hyperedge against float64 (fp32 accumulation over ~4000 bf16 rows; result rounded to bf16)
    row = int(torch.argmax(csr.rowptr[1:] - csr.rowptr[:-1]))
    s, t = int(csr.rowptr[row]), int(csr.rowptr[row + 1])
    src = csr.col[s:t].long()
    p = torch.softmax(torch.nn.functional.leaky_relu(alpha[src].double(), 0.2), dim=0)
    ref = (p[:, :, None] * V[src].double().view(-1, H, d // H)).sum(0).reshape(-1)
    torch.testing.assert_close(o1[row].double(), ref, rtol=2e-2, atol=2e-2 * float(ref.abs().max()) + 1e-3)
    # conservation of the bf16 segment sum (fp32 accumulation): checksum of checksums within bf16 output rounding
    e = deepsets_aggregate(V, v2e, None, "add")
    deg_v = (v2e.by_src.rowptr[1:] - v2e.by_src.rowptr[:-1]).double()
    lhs, rhs = e.double().sum(0), (V.double() * deg_v[:, None]).sum(0)
    scale = (V.double().abs() * deg_v[:, None]).sum(0)
    assert float(((lhs - rhs).abs() / scale).max()) < 2e-3


def test_configs4_load_balanced_hyperedge_bins(c4, device):
    """'load-balanced hyperedge bins, 8 GPUs': the lpt partition balances incidences, not hyperedge counts, and the
    shards tile the hypergraph exactly."""
    from allset_amd import dist as adist
    n, hg, v2e = c4
    sizes = (v2e.by_dst.rowptr[1:] - v2e.by_dst.rowptr[:-1]).to(torch.int64)
    owner = adist.partition_hyperedges(sizes.cpu(), 8, "lpt").to(device)
    load = torch.bincount(owner, weights=sizes.double(), minlength=8)
    assert float(load.max() - load.min()) <= 4096                   # greedy LPT: within one longest job
    assert int(load.sum()) == hg.nnz
    nnz = 0
    for r in (0, 7):
        loc, gids = adist.local_shard(hg.edge_index, owner, r)
        nnz += loc.shape[1]
        assert int(loc[1].max()) == gids.numel() - 1 and loc.shape[1] == int(load[r])
    assert nnz == int(load[0] + load[7])


# ---------------------------------------------------------------------------------------------------------------------
# the whole LAYER at the two configurations' sizes (round 3): both partitions' layer functions on one rank, train-free
# ---------------------------------------------------------------------------------------------------------------------

def _float64_layer_rows(v2e_conv, e2v_conv, x, by_dst_v2e, by_src_v2e, sample_v, heads, in_dtype=torch.float64):
    """Rows ``sample_v`` of ``relu(E2V(relu(V2E(x))))`` (AllSetTransformer layer, eval mode) in float64 through the CPU oracle
    on the induced sub-hypergraph: all hyperedges incident to the sampled vertices, all members of those hyperedges."""
    from oracle import allset_oracle as oracle
    rp_s, col_s = by_src_v2e.rowptr, by_src_v2e.col            # vertex -> its hyperedges
    rp_d, col_d = by_dst_v2e.rowptr, by_dst_v2e.col            # hyperedge -> its members
    es = torch.unique(torch.cat([col_s[int(rp_s[v]):int(rp_s[v + 1])].long() for v in sample_v.tolist()])).cpu()
    mem = [col_d[int(rp_d[e]):int(rp_d[e + 1])].long().cpu() for e in es.tolist()]
    sample_v = sample_v.cpu()
    vs = torch.unique(torch.cat(mem + [sample_v]))
    vid = {int(v): i for i, v in enumerate(vs.tolist())}
    src = torch.tensor([vid[int(v)] for m in mem for v in m.tolist()], dtype=torch.int64)
    dst = torch.repeat_interleave(torch.arange(len(mem)), torch.tensor([m.numel() for m in mem]))
    ei = torch.stack([src, dst])
    sd = {f"V2EConvs.0.{k}": v.detach().to("cpu", torch.float64) for k, v in v2e_conv.state_dict().items()}
    sd.update({f"E2VConvs.0.{k}": v.detach().to("cpu", torch.float64) for k, v in e2v_conv.state_dict().items()})
    xs = x[vs.to(x.device)].to("cpu", torch.float64)
    ones = torch.ones(ei.shape[1], dtype=torch.float64)
    e = torch.relu(oracle.halfnlhconv_forward(sd, "V2EConvs.0.", xs, ei, ones, "add", True, heads, "ln"))
    # E -> V restricted to the sampled vertices: incidences (hyperedge e, vertex v in the sample)
    sv = {int(v): i for i, v in enumerate(sample_v.tolist())}
    keep = torch.tensor([int(vs[s]) in sv for s in src.tolist()])
    rev = torch.stack([dst[keep], torch.tensor([sv[int(vs[s])] for s in src[keep].tolist()], dtype=torch.int64)])
    v = torch.relu(oracle.halfnlhconv_forward(sd, "E2VConvs.0.", e, rev, ones[:rev.shape[1]], "add", True, heads, "ln"))
    return v            # row i = vertex sample_v[i]   (every sampled vertex has >= 1 hyperedge in these generators)


def _layer_both_partitions(n, hg, d, heads, dtype, device, seed):
    from allset_amd import HalfNLHconv
    from allset_amd import dist as adist
    torch.manual_seed(seed)
    a = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=heads, attention=True).to(device).to(dtype).eval()
    b = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=heads, attention=True).to(device).to(dtype).eval()
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn(n, d, device=device, generator=g).to(dtype)
    G = torch.randn(n, d, device=device, generator=g).to(dtype)
    rows = adist.ShardedHypergraph(hg.edge_index, n, n, 1, 0).build_incidences()
    cols = adist.ColumnShardedHypergraph(hg.edge_index, n, n, 1, 0).build_incidences()
    res = {}
    for name, fn, h in (("rows", adist.sharded_pma_layer, rows), ("columns", adist.colsharded_pma_layer, cols)):
        outs = []
        for rep in range(2):
            xs = x.clone().requires_grad_(True)
            out = fn(a, b, xs, h)
            out.backward(G)
            outs.append((out.detach(), xs.grad.detach(), [p.grad.detach().clone() for p in list(a.parameters()) + list(b.parameters())]))
            for p in list(a.parameters()) + list(b.parameters()):
                p.grad = None
        o0, o1 = outs
        assert torch.equal(o0[0], o1[0]) and torch.equal(o0[1], o1[1])                    # no atomics anywhere: bitwise repeatable
        assert all(torch.equal(p0, p1) for p0, p1 in zip(o0[2], o1[2]))
        assert bool(torch.isfinite(o0[0]).all()) and bool(torch.isfinite(o0[1]).all())
        res[name] = o0
        del outs, o1
    return a, b, x, rows, res


def test_configs3_full_layer_both_partitions(c3, device):
    """One AllSetTransformer layer step (forward + backward, eval mode) at |V| = |E| = 8M, nnz = 128M, d = 128, 4 heads through
    BOTH partitions' layer functions on one rank: finite, bitwise repeatable, the two partitions agree, sampled output rows
    match a float64 recomputation of the layer (CPU oracle on the induced sub-hypergraph)."""
    n, hg, v2e = c3
    a, b, x, rows, res = _layer_both_partitions(n, hg, 128, 4, torch.float32, device, 31)
    (o_r, gx_r, pg_r), (o_c, gx_c, pg_c) = res["rows"], res["columns"]
    torch.testing.assert_close(o_c, o_r, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gx_c, gx_r, rtol=1e-4, atol=1e-4 * float(gx_r.abs().max()))
    for p_c, p_r in zip(pg_c, pg_r):
        torch.testing.assert_close(p_c, p_r, rtol=1e-3, atol=1e-3 * max(float(p_r.abs().max()), 1e-3))
    sample = torch.tensor([0, 1_234_567, 4_194_305, n - 1])
    ref = _float64_layer_rows(a, b, x, rows.v2e.by_dst, rows.v2e.by_src, sample, 4)
    torch.testing.assert_close(o_r[sample.to(device)].double().cpu(), ref, rtol=1e-4, atol=1e-4)


def test_configs4_full_layer_both_partitions_bf16(c4, device):
    """The same at configs[4]: |V| = |E| = 2M, truncated-Zipf sizes <= 4096, d = 256, bf16 end to end (fp32 accumulation)."""
    n, hg, v2e = c4
    a, b, x, rows, res = _layer_both_partitions(n, hg, 256, 4, torch.bfloat16, device, 41)
    (o_r, gx_r, pg_r), (o_c, gx_c, pg_c) = res["rows"], res["columns"]
    # bf16 storage: one rounding of O(1) LayerNorm outputs is 2^-8 relative; the two code paths round at different places
    torch.testing.assert_close(o_c.float(), o_r.float(), rtol=4e-2, atol=4e-2)
    assert float((gx_c.float() - gx_r.float()).abs().mean()) <= 2e-2 * float(gx_r.float().abs().mean()) + 1e-6
    deg = rows.v2e.by_src.rowptr[1:] - rows.v2e.by_src.rowptr[:-1]
    cand = torch.nonzero((deg > 0) & (deg < 40)).reshape(-1)
    sample = cand[torch.tensor([0, cand.numel() // 3, cand.numel() - 1])].cpu()
    ref = _float64_layer_rows(a, b, x, rows.v2e.by_dst, rows.v2e.by_src, sample, 4)
    got = o_r[sample.to(device)].double().cpu()
    assert float((got - ref).abs().max()) <= 6e-2 * max(1.0, float(ref.abs().max()))
