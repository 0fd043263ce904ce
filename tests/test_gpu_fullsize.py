"""GPU: size-independent properties at BASELINE.json's full single-GPU size (configs[2]: |V| = |E| = 1M,
mean degree 16, d = 128), where the CPU oracle would take minutes: conservation (a checksum of checksums),
linearity, agreement of the transposed pass with the adjoint identity <A x, y> = <x, A^T y>, permutation
invariance within segments, and softmax normalisation for PMA."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(device):
    from allset_amd.synthetic import random_hypergraph
    from allset_amd import Incidence
    hg = random_hypergraph(n_v=1_000_000, n_e=1_000_000, degree=16, seed=1234, device=device)
    inc = Incidence.from_edge_index(hg.edge_index, n_src=hg.n_v, n_dst=hg.n_e)
    return hg, inc


def test_sum_conservation_and_adjoint_identity(big, device):
    from allset_amd import deepsets_aggregate
    hg, v2e = big
    e2v = v2e.reversed()
    d = 128
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.randn(hg.n_v, d, device=device, generator=g)
    y = torch.randn(hg.n_e, d, device=device, generator=g)
    e = deepsets_aggregate(x, v2e, None, "add")
    assert tuple(e.shape) == (hg.n_e, d)
    # conservation: sum_t out[t,:] == sum_s deg(s) * x[s,:]   (fp64 accumulation of both checksums)
    deg_v = (v2e.by_src.rowptr[1:] - v2e.by_src.rowptr[:-1]).double()
    lhs = e.double().sum(0)
    rhs = (x.double() * deg_v[:, None]).sum(0)
    torch.testing.assert_close(lhs, rhs, rtol=1e-6, atol=1e-2)
    # adjoint: <A x, y> == <x, A^T y>, A^T applied by the same kernel on the transposed CSR
    xt = deepsets_aggregate(y, e2v, None, "add")
    a = (e.double() * y.double()).sum()
    b = (x.double() * xt.double()).sum()
    assert abs(float(a - b)) <= 1e-6 * float(a.abs() + b.abs()) + 1e-3
    # autograd backward of V->E is that same transposed pass
    xr = x.clone().requires_grad_(True)
    (deepsets_aggregate(xr, v2e, None, "add") * y).sum().backward()
    torch.testing.assert_close(xr.grad, xt, rtol=0, atol=0)


def test_linearity_mean_and_max_bounds(big, device):
    from allset_amd import deepsets_aggregate
    hg, v2e = big
    g = torch.Generator(device=device).manual_seed(8)
    x = torch.randn(hg.n_v, 128, device=device, generator=g)
    z = torch.randn(hg.n_v, 128, device=device, generator=g)
    s = deepsets_aggregate(2.0 * x - 0.5 * z, v2e, None, "add")
    t = 2.0 * deepsets_aggregate(x, v2e, None, "add") - 0.5 * deepsets_aggregate(z, v2e, None, "add")
    torch.testing.assert_close(s, t, rtol=1e-4, atol=1e-4)
    mean = deepsets_aggregate(x, v2e, None, "mean")
    mx = deepsets_aggregate(x, v2e, None, "max")
    mn = deepsets_aggregate(x, v2e, None, "min")
    assert bool((mn <= mean + 1e-5).all()) and bool((mean <= mx + 1e-5).all())
    cnt = (v2e.by_dst.rowptr[1:] - v2e.by_dst.rowptr[:-1]).clamp(min=1).float()
    torch.testing.assert_close(mean * cnt[:, None], deepsets_aggregate(x, v2e, None, "add"), rtol=1e-4, atol=1e-4)


def test_pma_is_a_convex_combination(big, device):
    from allset_amd import pma_aggregate
    hg, v2e = big
    H = 4
    g = torch.Generator(device=device).manual_seed(9)
    alpha = torch.randn(hg.n_v, H, device=device, generator=g)
    ones = torch.ones(hg.n_v, 128, device=device)
    out, m, l = pma_aggregate(ones, alpha, v2e, H, 0.2)
    nonempty = (v2e.by_dst.rowptr[1:] > v2e.by_dst.rowptr[:-1])
    torch.testing.assert_close(out[nonempty], torch.ones_like(out[nonempty]), rtol=1e-5, atol=1e-5)
    assert bool((l[nonempty] >= 1.0 - 1e-5).all())               # the max element contributes exp(0) = 1
    V = torch.randn(hg.n_v, 128, device=device, generator=g)
    out, _, _ = pma_aggregate(V, alpha, v2e, H, 0.2)
    from allset_amd import deepsets_aggregate
    assert bool((out <= deepsets_aggregate(V, v2e, None, "max") + 1e-4).all())
    assert bool((out >= deepsets_aggregate(V, v2e, None, "min") - 1e-4).all())


def test_dense_tail_full_size_properties(device):
    """The fused Linear kernels at the bench shape (n = 1M rows, 128 -> 128), checked through properties that need no
    reference run: sampled rows against float64, additivity of the weight gradient over a row split, and the adjoint
    identity <J x, g> = <x, J^T g> between forward and backward-data of the plain Linear."""
    from allset_amd import dense
    n, d = 1_000_003, 128                      # not a multiple of the 16 / 32-row tiles
    g = torch.Generator(device=device).manual_seed(11)
    x = torch.randn(n, d, device=device, generator=g)
    W = torch.randn(d, d, device=device, generator=g) / d ** 0.5
    b = torch.randn(d, device=device, generator=g)
    gamma = 1 + 0.2 * torch.randn(d, device=device, generator=g)
    beta = 0.3 * torch.randn(d, device=device, generator=g)
    G = torch.randn(n, d, device=device, generator=g)
    rows = torch.cat([torch.arange(0, 64, device=device), torch.randint(0, n, (4000,), device=device, generator=g),
                      torch.arange(n - 64, n, device=device)])
    # forward with the LayerNorm prologue and a relu epilogue, sampled rows in float64
    y, st = dense.fused_linear_fwd(x, W, b, gamma, beta, 1e-5, False, 0.0, 0, True, 0.0, 0)
    xs = x[rows].double()
    ref = torch.relu(torch.nn.functional.layer_norm(xs, (d,), gamma.double(), beta.double(), 1e-5) @ W.double().t() + b.double())
    torch.testing.assert_close(y[rows].double(), ref, rtol=1e-4, atol=1e-5)
    # weight gradient is additive over a split of the rows
    gw, gb = dense.wgrad(G, x)
    h = 499_999
    gw1, gb1 = dense.wgrad(G[:h], x[:h])
    gw2, gb2 = dense.wgrad(G[h:], x[h:])
    scale = float(gw.abs().max())
    torch.testing.assert_close(gw1 + gw2, gw, rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(gb1 + gb2, gb, rtol=1e-4, atol=1e-5 * float(gb.abs().max()))
    # the identity as weight returns the input bit for bit in the strict arithmetic (bf16x6: x = h + m + l exactly, and they sum
    # back exactly); the default at this shape since round 5 is fp16x3 with a scale per row: two fp16 planes carry 22 of the 24
    # significant bits of the row's LARGEST element -- |y - x| <= 2^-21 max|x[r, :]|
    with dense.arithmetic("strict"):
        yi, _ = dense.fused_linear_fwd(x, torch.eye(d, device=device), None)
    assert torch.equal(yi, x)
    yi, _ = dense.fused_linear_fwd(x, torch.eye(d, device=device), None)
    assert bool(((yi - x).abs() <= 2.0 ** -21 * x.abs().max(1, keepdim=True).values).all())
    # adjoint identity between the plain forward and its backward-data
    yp, _ = dense.fused_linear_fwd(x, W, None)
    gx, _, _ = dense.fused_linear_bwd(G, None, 0.0, W, x, None, None, False, 0.0, 0)
    lhs = (yp.double() * G.double()).sum()
    rhs = (x.double() * gx.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-6 * float(lhs.abs() + rhs.abs()) + 1e-2


def test_wide_gemm_full_size_properties(device):
    """The tiled bf16x6 GEMM of the 256-wide layers at bench scale (n = 1M rows), through properties that need no
    reference run: with the identity as weight it returns its input BIT FOR BIT (x = h + m + l exactly, and the six
    products that survive a 0/1 weight sum back to x exactly in fp32); the transposed-planes GEMM is the adjoint of the
    plain one; sampled rows of the LayerNorm-prologue / relu-epilogue form against float64."""
    from allset_amd import dense
    n, d = 1_000_003, 256                      # not a multiple of the 128-row tile
    g = torch.Generator(device=device).manual_seed(13)
    x = torch.randn(n, d, device=device, generator=g) * torch.exp(torch.randn(n, 1, device=device, generator=g))
    eye = torch.eye(d, device=device)
    y = dense.gemm_x6(x, dense.gemm_x6_planes(eye, False, f16=False), d, None)          # the exact split (strict arithmetic)
    assert torch.equal(y, x)
    # the default since round 5: two fp16 planes, A scaled per row from a maximum read one tile ahead -- 22 of the 24 significant
    # bits of the row's largest element
    y = dense.gemm_x6(x, dense.gemm_x6_planes(eye, False, f16=True), d, None)
    assert bool(((y - x).abs() <= 2.0 ** -21 * x.abs().max(1, keepdim=True).values).all())
    W = torch.randn(d, d, device=device, generator=g) / d ** 0.5
    b = torch.randn(d, device=device, generator=g)
    G = torch.randn(n, d, device=device, generator=g)
    yp = dense.gemm_x6(x, dense.gemm_x6_planes(W, False), d, None)              # x W^T
    gx = dense.gemm_x6(G, dense.gemm_x6_planes(W, True), d, None)               # G W
    lhs = (yp.double() * G.double()).sum()
    rhs = (x.double() * gx.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-6 * float(lhs.abs() + rhs.abs()) + 1e-2
    gamma = 1 + 0.2 * torch.randn(d, device=device, generator=g)
    beta = 0.3 * torch.randn(d, device=device, generator=g)
    rows = torch.cat([torch.arange(0, 64, device=device), torch.randint(0, n, (4000,), device=device, generator=g),
                      torch.arange(n - 64, n, device=device)])
    st = dense.row_stats(x, False, 1e-5)
    yl = dense.gemm_x6(x, dense.gemm_x6_planes(W, False), d, b, stats=st, gamma=gamma, beta=beta, relu_out=True)
    ref = torch.relu(torch.nn.functional.layer_norm(x[rows].double(), (d,), gamma.double(), beta.double(), 1e-5)
                     @ W.double().t() + b.double())
    torch.testing.assert_close(yl[rows].double(), ref, rtol=1e-4, atol=1e-5)


def test_pma_tail_full_size_rows_and_gradients(device):
    """The two-kernel PMA tail (``dense.pma_tail``: ln0 / ln1 inside the rFF Linears, reference layers.py:153-157) at the bench's
    size, n = 1M rows: everything in it is row-local for fixed parameters, so SAMPLED rows of the output and of the input gradient
    are checked against float64 (first / last 64 rows and 4000 random ones: stage tails of the persistent workgroups included); the
    parameter gradients -- sums over all rows -- against the unfused chain in the strict arithmetic, and with the conv's relu ->
    dropout in the epilogue the kept share and the exact zeros are checked."""
    import torch.nn.functional as F
    from allset_amd import dense
    n = 1_000_003
    g = torch.Generator(device=device).manual_seed(21)
    mk = lambda *s, sc=1.0, off=0.0: (torch.randn(*s, device=device, generator=g) * sc + off).requires_grad_(True)
    pooled = mk(n, 128, sc=2.0)
    att = mk(1, 4, 32, sc=0.5)
    g0, b0, g1, b1 = mk(128, sc=0.2, off=1.0), mk(128, sc=0.3), mk(128, sc=0.2, off=1.0), mk(128, sc=0.3)
    w1, w2 = mk(128, 128, sc=128 ** -0.5), mk(128, 128, sc=128 ** -0.5)
    bb1, bb2 = mk(128, sc=0.1), mk(128, sc=0.1)
    sign = torch.where(torch.arange(128, device=device) % 2 == 0, 5.0, -5.0)
    with torch.no_grad():                       # ~2.6e8 relu inputs: off the kinks (tests/cases.py kinkfree_biases); the second Linear's
        bb1 += sign; bb2 += 4 * sign            # input relu(. + 5) is not unit-scale (its Linear term has a deviation of ~3.5): +- 20
    G = torch.randn(n, 128, device=device, generator=g)
    params = [pooled, att, g0, b0, w1, bb1, w2, bb2, g1, b1]
    y = dense.pma_tail(pooled, att, g0, b0, 1e-5, w1, bb1, w2, bb2, g1, b1, 1e-5, False, 0.0)
    (y * G).sum().backward()
    got = [t.grad.clone() for t in params]
    rows = torch.cat([torch.arange(0, 64, device=device), torch.randint(0, n, (4000,), device=device, generator=g),
                      torch.arange(n - 64, n, device=device)])
    P = pooled.detach()[rows].double().requires_grad_(True)
    pd = [t.detach().double() for t in params[1:]]
    A, G0, B0, W1, BB1, W2, BB2, G1, B1 = pd
    out = F.layer_norm(P + A.reshape(1, -1), (128,), G0, B0, 1e-5)
    ref = F.layer_norm(out + F.relu(F.linear(F.relu(F.linear(out, W1, BB1)), W2, BB2)), (128,), G1, B1, 1e-5)
    (ref * G[rows].double()).sum().backward()
    torch.testing.assert_close(y.detach()[rows].double(), ref.detach(), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(got[0][rows].double(), P.grad, rtol=1e-4, atol=1e-4 * float(P.grad.abs().max()))
    # parameter gradients: the unfused chain in the strict arithmetic on the same tensors
    for t in params:
        t.grad = None
    with dense.arithmetic("strict"):
        o2 = dense.layer_norm_res(pooled, att.reshape(-1), None, g0, b0, 1e-5)
        y2 = dense.pma_residual_ff(o2, w1, bb1, w2, bb2, g1, b1, 1e-5, False, 0.0)
        (y2 * G).sum().backward()
    for nm, a, t in zip(["att_r", "ln0.w", "ln0.b", "w1", "b1", "w2", "b2", "ln1.w", "ln1.b"], got[1:], params[1:]):
        # (sums of 1M signed terms in two different fp32 orders; measured against float64 both paths are within 4e-6 of the largest
        #  element, tools/tail_diag_fullsize.py -- with the relus off their kinks: at +- 5 on the second bias a dozen of the 1.3e8
        #  second-layer relu inputs sat within rounding of zero and these same gradients differed by 6e-4, the two paths flipping
        #  different units)
        scale = max(float(t.grad.abs().max()), 1e-6)
        assert float((a - t.grad).abs().max()) <= 2e-5 * scale, (nm, float((a - t.grad).abs().max()), scale)
    # with the conv's relu -> dropout(0.5) in the epilogue: half of the positive outputs survive, scaled by 2; nothing else changes
    with torch.no_grad():
        y0 = dense.pma_tail(pooled, att, g0, b0, 1e-5, w1, bb1, w2, bb2, g1, b1, 1e-5, True, 0.0)
        yd = dense.pma_tail(pooled, att, g0, b0, 1e-5, w1, bb1, w2, bb2, g1, b1, 1e-5, True, 0.5)
    kept = yd != 0
    assert torch.equal(yd[kept], (2.0 * y0)[kept]) and not bool((kept & (y0 <= 0)).any())
    share = float(kept[y0 > 0].float().mean())
    assert abs(share - 0.5) < 2e-3, share
