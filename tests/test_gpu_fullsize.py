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
