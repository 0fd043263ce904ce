"""GPU: kernel-level parity of every C-ABI entry point against the oracle on seeded random incidences,
across feature widths (vector and scalar paths, multi-chunk rows), degree shapes (empty, singleton,
duplicates, > 64 and 4096-long segments) and head configurations (incl. heads that straddle chunks)."""
import numpy as np
import pytest
import torch

from oracle import allset_oracle as oracle

pytestmark = pytest.mark.gpu
RTOL = ATOL = 1e-4


def make_incidence(rng, n_s, n_t, nnz, long_row=0, sort=False):
    src = rng.integers(0, n_s, size=nnz)
    dst = rng.integers(0, n_t, size=nnz)
    if n_t > 3:
        dst[dst == 1] = 0                       # row 1 empty (interior empty segment)
    if long_row:
        src = np.concatenate([src, rng.integers(0, n_s, size=long_row)])
        dst = np.concatenate([dst, np.full(long_row, min(2, n_t - 1))])
    if sort:
        o = np.argsort(src, kind="stable")
        src, dst = src[o], dst[o]
    return torch.from_numpy(np.stack([src, dst]).astype(np.int64))


def test_csr_build_matches_numpy(device):
    from allset_amd import Incidence
    rng = np.random.default_rng(0)
    for n_s, n_t, nnz in ((1, 1, 1), (7, 5, 40), (1000, 300, 20000), (50, 70000, 300000)):
        ei = make_incidence(rng, n_s, n_t, nnz)
        inc = Incidence.from_edge_index(ei.to(device), n_src=n_s, n_dst=n_t)
        for csr, keys, vals, nr in ((inc.by_dst, ei[1], ei[0], n_t), (inc.by_src, ei[0], ei[1], n_s)):
            order = np.argsort(keys.numpy(), kind="stable")
            np.testing.assert_array_equal(csr.perm.cpu().numpy(), order)          # stable: ties keep edge order
            np.testing.assert_array_equal(csr.col.cpu().numpy(), vals.numpy()[order])
            rp = np.concatenate([[0], np.cumsum(np.bincount(keys.numpy(), minlength=nr))])
            np.testing.assert_array_equal(csr.rowptr.cpu().numpy(), rp)
        pos = inc.pos_dst_of_src().cpu().numpy()
        np.testing.assert_array_equal(inc.by_dst.perm.cpu().numpy()[pos], inc.by_src.perm.cpu().numpy())


def test_csr_build_rejects_out_of_range_ids(device):
    from allset_amd import Incidence
    ei = torch.tensor([[0, 1, 5], [0, 1, 1]], dtype=torch.int64, device=device)
    with pytest.raises(ValueError):
        Incidence.from_edge_index(ei, n_src=3)
    empty = Incidence.from_edge_index(torch.zeros((2, 0), dtype=torch.int64, device=device), n_src=4, n_dst=3)
    assert empty.nnz == 0 and empty.by_dst.rowptr.cpu().tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("aggr", ["add", "mean", "max", "min"])
@pytest.mark.parametrize("d", [1, 3, 4, 20, 64, 100, 128, 256, 516])
def test_deepsets_aggregate_fwd_bwd(aggr, d, device):
    from allset_amd import Incidence, deepsets_aggregate
    rng = np.random.default_rng(d * 7 + len(aggr))
    n_s, n_t = 300, 120
    ei = make_incidence(rng, n_s, n_t, 1500, long_row=200 if d <= 128 else 70)
    nnz = ei.shape[1]
    for weighted in (False, True):
        norm = torch.from_numpy(rng.uniform(0.2, 2.0, size=nnz).astype(np.float32)) if weighted \
            else torch.ones(nnz, dtype=torch.int64)
        x = torch.from_numpy(rng.standard_normal((n_s, d)).astype(np.float32))
        G = torch.from_numpy(rng.standard_normal((n_t, d)).astype(np.float32))
        xo = x.clone().requires_grad_(True)
        no = norm.clone().requires_grad_(True) if weighted else norm
        ref = oracle.deepsets_aggregate(xo, ei, no, aggr)
        ref = torch.cat([ref, ref.new_zeros(n_t - ref.shape[0], d)])       # oracle sizes by max+1 (Q1)
        (ref * G).sum().backward()

        inc = Incidence.from_edge_index(ei.to(device), n_src=n_s, n_dst=n_t)
        xg = x.to(device).requires_grad_(True)
        ng = norm.to(device).requires_grad_(True) if weighted else norm.to(device)
        out = deepsets_aggregate(xg, inc, ng, aggr)
        (out * G.to(device)).sum().backward()
        torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=RTOL, atol=ATOL)
        torch.testing.assert_close(xg.grad.cpu(), xo.grad, rtol=RTOL, atol=ATOL)
        if weighted:
            torch.testing.assert_close(ng.grad.cpu(), no.grad, rtol=RTOL, atol=ATOL * 10)
        assert float(out[1].detach().abs().max()) == 0.0                             # empty segment -> exactly 0


def test_fixed_float_norm_uses_cached_routing(device):
    """Non-differentiable float norm (deg_half_sym, preprocessing.py:456-463): weights are routed once."""
    from allset_amd import Incidence, deepsets_aggregate
    rng = np.random.default_rng(5)
    ei = make_incidence(rng, 64, 32, 500)
    norm = torch.from_numpy(rng.uniform(0.1, 1.0, size=500).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((64, 32)).astype(np.float32))
    inc = Incidence.from_edge_index(ei.to(device), n_src=64, n_dst=32)
    ng = norm.to(device)
    for aggr in ("add", "mean", "max"):
        xo = x.clone().requires_grad_(True)
        ref = oracle.deepsets_aggregate(xo, ei, norm, aggr)
        ref = torch.cat([ref, ref.new_zeros(32 - ref.shape[0], 32)])
        ref.square().sum().backward()
        xg = x.to(device).requires_grad_(True)
        out = deepsets_aggregate(xg, inc, ng, aggr)
        out.square().sum().backward()
        torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=RTOL, atol=ATOL)
        torch.testing.assert_close(xg.grad.cpu(), xo.grad, rtol=RTOL, atol=ATOL)
    assert len(inc._wcache) == 1


def test_leaf_norm_parameter_two_optimizer_steps(device):
    """a persistent LEAF norm that requires grad (an nn.Parameter handed straight to deepsets_aggregate)
    must be re-routed on every call -- the optimizer updates it in place between calls, and the routed copy of the first call
    carries that call's graph.  Two SGD steps against the oracle's own two steps; also under no_grad first (a cached no-grad hit
    would drop the weight gradient)."""
    from allset_amd import Incidence, deepsets_aggregate
    rng = np.random.default_rng(17)
    ei = make_incidence(rng, 64, 32, 500)
    x = torch.from_numpy(rng.standard_normal((64, 32)).astype(np.float32))
    w0 = torch.from_numpy(rng.uniform(0.5, 1.5, size=500).astype(np.float32))
    inc = Incidence.from_edge_index(ei.to(device), n_src=64, n_dst=32)
    wg = torch.nn.Parameter(w0.clone().to(device))
    wo = torch.nn.Parameter(w0.clone())
    xg = x.to(device)
    with torch.no_grad():
        deepsets_aggregate(xg, inc, wg, "add")                     # must not poison the later differentiable calls
    og, oo = torch.optim.SGD([wg], lr=0.05), torch.optim.SGD([wo], lr=0.05)
    for step in range(3):
        og.zero_grad(); oo.zero_grad()
        out = deepsets_aggregate(xg, inc, wg, "add")
        ref = oracle.deepsets_aggregate(x, ei, wo, "add")
        ref = torch.cat([ref, ref.new_zeros(32 - ref.shape[0], 32)])
        torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=RTOL, atol=ATOL, msg=lambda m: f"step {step}: {m}")
        out.square().sum().backward()
        ref.square().sum().backward()
        torch.testing.assert_close(wg.grad.cpu(), wo.grad, rtol=RTOL, atol=ATOL, msg=lambda m: f"step {step} grad: {m}")
        og.step(); oo.step()
    assert "_allset_routed" not in wg.__dict__                    # nothing cached on a leaf
    assert float((wg.detach().cpu() - w0).abs().max()) > 1e-3       # the steps did move it


def test_persistent_float_ones_norm_skips_the_weight_stream_from_its_second_use(device):
    """``torch.ones(nnz)`` kept by the caller is probed on its SECOND use (a per-forward temporary never is):
    from then on the kernels skip the weight stream like for the reference's int64 ones."""
    from allset_amd import Incidence
    rng = np.random.default_rng(3)
    ei = make_incidence(rng, 64, 32, 500)
    inc = Incidence.from_edge_index(ei.to(device), n_src=64, n_dst=32)
    ones = torch.ones(500, device=device)
    first = inc.weights(ones)
    assert first[0] is not None and first[1] is not None            # first sight: routed unseen (no host sync)
    assert inc.weights(ones) == (None, None)                         # second use of the same live tensor: probed
    assert inc.weights(ones) == (None, None)
    other = torch.full((500,), 0.5, device=device)
    inc.weights(other)
    w = inc.weights(other)
    assert w[0] is not None and float(w[0][0]) == 0.5
    ones.mul_(2.0)                                                   # in-place update: a new version, routed again
    w2 = inc.weights(ones); w2 = inc.weights(ones)
    assert w2[0] is not None and float(w2[0][0]) == 2.0


def test_max_ties_go_to_first_incidence(device):
    """Documented tie rule: arg-extremum = smallest CSR position = first in edge order."""
    from allset_amd import Incidence, ops
    ei = torch.tensor([[0, 1, 2, 3, 4, 5], [0, 0, 0, 0, 1, 1]], dtype=torch.int64, device=device)
    inc = Incidence.from_edge_index(ei, n_src=6, n_dst=3)
    x = torch.tensor([[1.0], [5.0], [5.0], [2.0], [-3.0], [-3.0]], device=device).repeat(1, 4).contiguous()
    out, arg = ops.segreduce(2, inc.by_dst.rowptr, inc.by_dst.col, None, x, 3, want_arg=True)
    assert out.cpu().tolist() == [[5.0] * 4, [-3.0] * 4, [0.0] * 4]
    assert arg.cpu().tolist() == [[1] * 4, [4] * 4, [-1] * 4]
    out, arg = ops.segreduce(3, inc.by_dst.rowptr, inc.by_dst.col, None, x, 3, want_arg=True)
    assert out.cpu().tolist() == [[1.0] * 4, [-3.0] * 4, [0.0] * 4] and arg.cpu().tolist()[0] == [0] * 4


@pytest.mark.parametrize("H,C", [(1, 4), (1, 64), (4, 32), (8, 16), (4, 8), (3, 5), (2, 192), (1, 520), (16, 4), (5, 12)])
def test_pma_aggregate_fwd_bwd(H, C, device):
    from allset_amd import Incidence, pma_aggregate, pma_attention_weights
    rng = np.random.default_rng(H * 100 + C)
    n_s, n_t = 200, 90
    ei = make_incidence(rng, n_s, n_t, 1200, long_row=300 if H * C <= 256 else 80, sort=True)
    V = torch.from_numpy(rng.standard_normal((n_s, H, C)).astype(np.float32))
    alpha = torch.from_numpy((3.0 * rng.standard_normal((n_s, H))).astype(np.float32))
    alpha[3, 0] = 0.0                                            # leaky_relu'(0) = slope (PyTorch convention)
    G = torch.from_numpy(rng.standard_normal((n_t, H, C)).astype(np.float32))
    Vo, ao = V.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    ref, p_ref = oracle.pma_aggregate(Vo, ao, ei, 0.2)
    ref = torch.cat([ref, ref.new_zeros(n_t - ref.shape[0], H, C)])
    (ref * G).sum().backward()

    inc = Incidence.from_edge_index(ei.to(device), n_src=n_s, n_dst=n_t)
    Vg = V.reshape(n_s, H * C).to(device).requires_grad_(True)
    ag = alpha.to(device).requires_grad_(True)
    out, m, l = pma_aggregate(Vg, ag, inc, H, 0.2)
    (out * G.reshape(n_t, H * C).to(device)).sum().backward()
    torch.testing.assert_close(out.detach().cpu().view(n_t, H, C), ref.detach(), rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(Vg.grad.cpu().view(n_s, H, C), Vo.grad, rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(ag.grad.cpu(), ao.grad, rtol=RTOL, atol=ATOL)
    p = pma_attention_weights(ag.detach(), m, l, inc, 0.2)
    torch.testing.assert_close(p.cpu(), p_ref.detach(), rtol=RTOL, atol=1e-6)
    assert float(out[1].detach().abs().max()) == 0.0 and float(l[1].abs().max()) == 0.0    # empty target


def test_pma_extreme_logits_are_stable(device):
    """Online softmax must survive logits that overflow a naive exp (max-subtraction semantics)."""
    from allset_amd import Incidence, pma_aggregate
    ei = torch.tensor([[0, 1, 2, 3], [0, 0, 0, 1]], dtype=torch.int64)
    V = torch.arange(16, dtype=torch.float32).view(4, 4)
    alpha = torch.tensor([[200.0], [-500.0], [199.0], [-1e4]])
    ref, _ = oracle.pma_aggregate(V.view(4, 1, 4), alpha, ei, 0.2)
    inc = Incidence.from_edge_index(ei.to(device), n_src=4, n_dst=2)
    out, _, _ = pma_aggregate(V.to(device), alpha.to(device), inc, 1, 0.2)
    assert torch.isfinite(out).all()
    torch.testing.assert_close(out.cpu().view(2, 1, 4), ref, rtol=RTOL, atol=ATOL)


def test_strided_inputs_use_leading_dimension(device):
    """Row-major matrices with ld > d are consumed in place (ABI takes an explicit leading dimension)."""
    from allset_amd import Incidence, deepsets_aggregate
    rng = np.random.default_rng(9)
    ei = make_incidence(rng, 50, 20, 300)
    big = torch.from_numpy(rng.standard_normal((50, 96)).astype(np.float32)).to(device)
    x = big[:, 32:64]                                              # ld = 96, 16-byte aligned view
    inc = Incidence.from_edge_index(ei.to(device), n_src=50, n_dst=20)
    ref = oracle.deepsets_aggregate(x.cpu(), ei, torch.ones(300, dtype=torch.int64), "add")
    out = deepsets_aggregate(x, inc, None, "add")
    torch.testing.assert_close(out.cpu()[:ref.shape[0]], ref, rtol=RTOL, atol=ATOL)
    x2 = big[:, 1:33]                                              # misaligned view -> scalar kernel path
    ref2 = oracle.deepsets_aggregate(x2.cpu(), ei, torch.ones(300, dtype=torch.int64), "add")
    out2 = deepsets_aggregate(x2, inc, None, "add")
    torch.testing.assert_close(out2.cpu()[:ref2.shape[0]], ref2, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("d,H", [(256, 8), (128, 4), (64, 1), (24, 3), (520, 1)])
def test_bf16_storage_fp32_accumulate(d, H, device):
    """BASELINE configs[4]: bf16 storage, fp32 accumulation.  Inputs are rounded to bf16 once; the oracle runs in
    fp32 on those rounded values, so the only differences are the final bf16 rounding of the outputs
    (rel 2^-8) -- tolerance 1e-2 relative to the tensor scale."""
    from allset_amd import Incidence, deepsets_aggregate, pma_aggregate
    rng = np.random.default_rng(d + H)
    n_s, n_t = 250, 100
    ei = make_incidence(rng, n_s, n_t, 1400, long_row=150)
    x = torch.from_numpy(rng.standard_normal((n_s, d)).astype(np.float32)).bfloat16()
    G = torch.from_numpy(rng.standard_normal((n_t, d)).astype(np.float32)).bfloat16()
    inc = Incidence.from_edge_index(ei.to(device), n_src=n_s, n_dst=n_t)

    def close(a, b):
        scale = float(b.abs().max())
        torch.testing.assert_close(a.float().cpu(), b, rtol=2e-2, atol=1e-2 * scale)

    for aggr in ("add", "mean"):
        xo = x.float().requires_grad_(True)
        ref = oracle.deepsets_aggregate(xo, ei, torch.ones(ei.shape[1], dtype=torch.int64), aggr)
        ref = torch.cat([ref, ref.new_zeros(n_t - ref.shape[0], d)])
        (ref * G.float()).sum().backward()
        xg = x.to(device).requires_grad_(True)
        out = deepsets_aggregate(xg, inc, None, aggr)
        assert out.dtype == torch.bfloat16
        out.backward(G.to(device))
        close(out.detach(), ref.detach())
        close(xg.grad, xo.grad)
    out_max = deepsets_aggregate(x.to(device), inc, None, "max")                 # forward-only in bf16
    ref_max = oracle.deepsets_aggregate(x.float(), ei, torch.ones(ei.shape[1], dtype=torch.int64), "max")
    close(out_max[:ref_max.shape[0]], ref_max)

    C = d // H
    alpha = torch.from_numpy((2.0 * rng.standard_normal((n_s, H))).astype(np.float32))
    Vo, ao = x.float().view(n_s, H, C).clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    ref, _ = oracle.pma_aggregate(Vo, ao, ei, 0.2)
    ref = torch.cat([ref, ref.new_zeros(n_t - ref.shape[0], H, C)])
    (ref * G.float().view(n_t, H, C)).sum().backward()
    Vg, ag = x.to(device).requires_grad_(True), alpha.to(device).requires_grad_(True)
    out, m, l = pma_aggregate(Vg, ag, inc, H, 0.2)
    assert out.dtype == torch.bfloat16 and m.dtype == torch.float32
    out.backward(G.to(device))
    close(out.detach().view(n_t, H, C), ref.detach())
    close(Vg.grad.view(n_s, H, C), Vo.grad)
    close(ag.grad, ao.grad)


@pytest.mark.parametrize("d", [4, 20, 64, 128, 256])
@pytest.mark.parametrize("mean_deg", [0.7, 2.0, 12.0])
def test_short_row_kernel_matches_row_per_wave_and_oracle(d, mean_deg, device):
    """allset_segreduce_fwd_ex variant 2 (several consecutive rows per half-wave, one incidence stream) against
    variant 1 and the oracle: empty rows, singletons, a 300-long row, weights, mean, bf16."""
    from allset_amd import Incidence, ops
    rng = np.random.default_rng(int(d * 10 + mean_deg * 3))
    n_s, n_t = 500, 1003
    ei = make_incidence(rng, n_s, n_t, int(n_t * mean_deg), long_row=300)
    nnz = ei.shape[1]
    inc = Incidence.from_edge_index(ei.to(device), n_src=n_s, n_dst=n_t)
    csr = inc.by_dst
    x = torch.from_numpy(rng.standard_normal((n_s, d)).astype(np.float32))
    w = torch.from_numpy(rng.uniform(0.5, 1.5, size=nnz).astype(np.float32))
    w_csr = w.to(device)[csr.perm.long()].contiguous()
    for reduce, aggr in ((0, "add"), (1, "mean")):
        for weights in (None, w_csr):
            ref = oracle.deepsets_aggregate(x, ei, w if weights is not None else torch.ones(nnz, dtype=torch.int64), aggr)
            ref = torch.cat([ref, ref.new_zeros(n_t - ref.shape[0], d)])
            a, _ = ops.segreduce(reduce, csr.rowptr, csr.col, weights, x.to(device), n_t, variant=1)
            b, _ = ops.segreduce(reduce, csr.rowptr, csr.col, weights, x.to(device), n_t, variant=2)
            torch.testing.assert_close(b.cpu(), ref, rtol=RTOL, atol=ATOL)
            torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-5)
    xb = x.bfloat16().to(device)
    if d % 8 == 0:
        a, _ = ops.segreduce(0, csr.rowptr, csr.col, None, xb, n_t, variant=1)
        b, _ = ops.segreduce(0, csr.rowptr, csr.col, None, xb, n_t, variant=2)
        torch.testing.assert_close(b.float(), a.float(), rtol=2e-2, atol=2e-2)
    auto, _ = ops.segreduce(0, csr.rowptr, csr.col, None, x.to(device), n_t)          # auto picks by nnz / n_t and by n_t:
    # the short-row kernel only above 16384 target rows (segreduce.hip kFlatMinRows; ops.CSR.variant)
    assert csr.variant("segreduce", n_t) == 1
    expect, _ = ops.segreduce(0, csr.rowptr, csr.col, None, x.to(device), n_t, variant=1)
    torch.testing.assert_close(auto, expect, rtol=0, atol=0)


def test_short_row_kernel_rejects_what_it_cannot_do(device):
    from allset_amd import Incidence, ops, _lib
    ei = torch.tensor([[0, 1, 2], [0, 0, 1]], dtype=torch.int64, device=device)
    inc = Incidence.from_edge_index(ei, n_src=3, n_dst=2)
    x = torch.randn(3, 516, device=device)
    with pytest.raises(_lib.AllSetHipError):
        ops.segreduce(0, inc.by_dst.rowptr, inc.by_dst.col, None, x, 2, variant=2)       # d > 256: more than one chunk
    with pytest.raises(_lib.AllSetHipError):
        ops.segreduce(2, inc.by_dst.rowptr, inc.by_dst.col, None, x[:, :128].contiguous(), 2, variant=2)   # max


@pytest.mark.parametrize("H,C", [(4, 32), (1, 64), (8, 16), (4, 8), (1, 256)])
@pytest.mark.parametrize("mean_deg", [0.7, 2.5, 12.0])
def test_pma_short_row_kernel(H, C, mean_deg, device):
    """allset_pma_fwd_ex variant 2 against variant 1 and the oracle (out, m, l), incl. empty rows and a long row."""
    from allset_amd import Incidence, ops
    rng = np.random.default_rng(H * 1000 + C + int(mean_deg * 10))
    n_s, n_t, d = 400, 900, H * C
    ei = make_incidence(rng, n_s, n_t, int(n_t * mean_deg), long_row=200, sort=True)
    inc = Incidence.from_edge_index(ei.to(device), n_src=n_s, n_dst=n_t)
    csr = inc.by_dst
    V = torch.from_numpy(rng.standard_normal((n_s, d)).astype(np.float32))
    alpha = torch.from_numpy((2.5 * rng.standard_normal((n_s, H))).astype(np.float32))
    ref, _ = oracle.pma_aggregate(V.view(n_s, H, C), alpha, ei, 0.2)
    ref = torch.cat([ref, ref.new_zeros(n_t - ref.shape[0], H, C)]).reshape(n_t, d)
    o1, m1, l1 = ops.pma_fwd(csr.rowptr, csr.col, alpha.to(device), V.to(device), H, 0.2, n_t, variant=1)
    o2, m2, l2 = ops.pma_fwd(csr.rowptr, csr.col, alpha.to(device), V.to(device), H, 0.2, n_t, variant=2)
    torch.testing.assert_close(o2.cpu(), ref, rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(o2, o1, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(m2, m1, rtol=0, atol=0)
    torch.testing.assert_close(l2, l1, rtol=1e-5, atol=1e-6)
    ob, _, _ = ops.pma_fwd(csr.rowptr, csr.col, alpha.to(device), V.bfloat16().to(device), H, 0.2, n_t, variant=2) if C % 8 == 0 else (None, None, None)
    if ob is not None:
        torch.testing.assert_close(ob.float().cpu(), ref, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("H,C", [(4, 32), (1, 64), (8, 16), (4, 8), (3, 20)])
@pytest.mark.parametrize("mean_deg", [0.7, 2.5, 12.0])
def test_pma_bwd_short_row_kernel(H, C, mean_deg, device):
    """allset_pma_bwd_src_ex variant 2 against variant 1 and against autograd of the oracle."""
    from allset_amd import Incidence, ops
    rng = np.random.default_rng(H * 77 + C + int(mean_deg * 10))
    n_s, n_t, d = 700, 300, H * C
    ei = make_incidence(rng, n_s, n_t, int(n_s * mean_deg), long_row=0, sort=True)
    inc = Incidence.from_edge_index(ei.to(device), n_src=n_s, n_dst=n_t)
    csr, T = inc.by_dst, inc.by_src
    V = torch.from_numpy(rng.standard_normal((n_s, d)).astype(np.float32))
    alpha = torch.from_numpy((2.0 * rng.standard_normal((n_s, H))).astype(np.float32))
    alpha[5, 0] = 0.0
    G = torch.from_numpy(rng.standard_normal((n_t, d)).astype(np.float32))
    Vo, ao = V.view(n_s, H, C).clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    ref, _ = oracle.pma_aggregate(Vo, ao, ei, 0.2)
    ref = torch.cat([ref, ref.new_zeros(n_t - ref.shape[0], H, C)])
    (ref * G.view(n_t, H, C)).sum().backward()
    out, m, l = ops.pma_fwd(csr.rowptr, csr.col, alpha.to(device), V.to(device), H, 0.2, n_t)
    stats = ops.pma_bwd_stats(out, G.to(device), m, l)
    res = {}
    for variant in (1, 2) if C % 4 == 0 else (1,):
        res[variant] = ops.pma_bwd_src(T.rowptr, T.col, alpha.to(device), V.to(device), G.to(device), stats, 0.2, variant=variant)
        torch.testing.assert_close(res[variant][0].cpu().view(n_s, H, C), Vo.grad, rtol=RTOL, atol=ATOL)
        torch.testing.assert_close(res[variant][1].cpu(), ao.grad, rtol=RTOL, atol=ATOL)
    if 2 in res:
        torch.testing.assert_close(res[2][0], res[1][0], rtol=1e-5, atol=1e-5)


def test_long_rows_first_order_is_a_permutation_and_changes_nothing(device):
    """Skewed degree distribution: the CSR carries a processing order (long rows first per XCD range); results of the
    row-per-wave kernels must be bitwise identical with and without it."""
    from allset_amd import Incidence, ops
    from allset_amd.synthetic import random_hypergraph
    hg = random_hypergraph(60000, 40000, 16, seed=2, device=device, dist="zipf", max_degree=4096)
    inc = Incidence.from_edge_index(hg.edge_index, n_src=hg.n_v, n_dst=hg.n_e)
    csr = inc.by_dst
    assert csr.max_deg > 1000 and csr.row_order is not None
    order = csr.row_order.long()
    assert torch.equal(torch.sort(order).values, torch.arange(hg.n_e, device=device))
    deg = csr.rowptr[1:] - csr.rowptr[:-1]
    nb = (hg.n_e + 3) // 4
    assert int(deg[order[0]]) >= int(deg[order[4 * (nb // 8 + (1 if nb % 8 else 0)) - 1]])       # long rows lead a range
    assert inc.by_src.row_order is None                                                          # vertex side: not skewed
    x = torch.randn(hg.n_v, 64, device=device)
    a, _ = ops.segreduce(0, csr.rowptr, csr.col, None, x, hg.n_e, variant=1)
    b, _ = ops.segreduce(0, csr.rowptr, csr.col, None, x, hg.n_e, variant=1, row_order=csr.row_order)
    assert torch.equal(a, b)
    alpha = torch.randn(hg.n_v, 4, device=device)
    o1 = ops.pma_fwd(csr.rowptr, csr.col, alpha, x, 4, 0.2, hg.n_e, variant=1)
    o2 = ops.pma_fwd(csr.rowptr, csr.col, alpha, x, 4, 0.2, hg.n_e, variant=1, row_order=csr.row_order)
    assert all(torch.equal(u, v) for u, v in zip(o1, o2))
    rev = inc.reversed(n_dst=hg.n_v)                   # E->V: its backward walks the hyperedge rows (long) as sources
    g = torch.randn(hg.n_v, 64, device=device)
    ev = torch.randn(hg.n_e, 64, device=device)
    ae = torch.randn(hg.n_e, 4, device=device)
    out, m, l = ops.pma_fwd(rev.by_dst.rowptr, rev.by_dst.col, ae, ev, 4, 0.2, hg.n_v)
    st = ops.pma_bwd_stats(out, g, m, l)
    T = rev.by_src
    assert T.row_order is not None
    r1 = ops.pma_bwd_src(T.rowptr, T.col, ae, ev, g, st, 0.2, variant=1)
    r2 = ops.pma_bwd_src(T.rowptr, T.col, ae, ev, g, st, 0.2, variant=1, row_order=T.row_order)
    assert all(torch.equal(u, v) for u, v in zip(r1, r2))


@pytest.mark.parametrize("mode", ["add", "mean", "pma"])
def test_self_loop_tail_split_dispatch(mode, device):
    """Add_Self_Loops' layout -- regular hyperedges followed by a block of singleton hyperedges -- takes the two-launch
    dispatch (CSR.short_tail: one wave per row for the regular rows, short-row kernel for the tail).  Results must equal
    the oracle's, forward and backward, in both directions."""
    from allset_amd import Incidence, deepsets_aggregate, pma_aggregate
    rng = np.random.default_rng(11)
    n_v, n_e, d, H = 900, 500, 64, 4
    v = np.concatenate([rng.integers(0, n_v, size=n_e * 12), np.arange(n_v)])
    e = np.concatenate([np.repeat(np.arange(n_e), 12), n_e + np.arange(n_v)])
    ei = torch.from_numpy(np.stack([v, e]).astype(np.int64))
    n_t = n_e + n_v
    inc = Incidence.from_edge_index(ei.to(device), n_src=n_v, n_dst=n_t)
    assert inc.by_dst.short_tail == n_e and inc.by_src.short_tail == -1        # hyperedge-major CSR has the tail
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n_v, d, generator=g)
    G = torch.randn(n_t, d, generator=g)
    xr = x.clone().requires_grad_(True)
    xd = x.to(device).requires_grad_(True)
    if mode == "pma":
        alpha = torch.randn(n_v, H, generator=g)
        ar, ad = alpha.clone().requires_grad_(True), alpha.to(device).requires_grad_(True)
        ref, _ = oracle.pma_aggregate(xr.view(-1, H, d // H), ar, ei, 0.2)
        ref = ref.reshape(-1, d)
        out, _, _ = pma_aggregate(xd, ad, inc, H, 0.2)
    else:
        norm = torch.ones(ei.shape[1], dtype=torch.int64)
        ref = oracle.deepsets_aggregate(xr, ei, norm, mode)
        out = deepsets_aggregate(xd, inc, norm.to(device), mode)
    (ref * G).sum().backward()
    (out * G.to(device)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=RTOL, atol=ATOL)
    # the reverse direction (E -> V): its transposed CSR (rows = hyperedges) carries the tail in the backward pass
    rev = inc.reversed()
    y = torch.randn(n_t, d, generator=g)
    Gv = torch.randn(n_v, d, generator=g)
    yr, yd = y.clone().requires_grad_(True), y.to(device).requires_grad_(True)
    rei = torch.stack([ei[1], ei[0]])
    if mode == "pma":
        al2 = torch.randn(n_t, H, generator=g)
        a2r, a2d = al2.clone().requires_grad_(True), al2.to(device).requires_grad_(True)
        ref2, _ = oracle.pma_aggregate(yr.view(-1, H, d // H), a2r, rei, 0.2)
        ref2 = ref2.reshape(-1, d)
        out2, _, _ = pma_aggregate(yd, a2d, rev, H, 0.2)
    else:
        ref2 = oracle.deepsets_aggregate(yr, rei, torch.ones(ei.shape[1], dtype=torch.int64), mode)
        out2 = deepsets_aggregate(yd, rev, torch.ones(ei.shape[1], dtype=torch.int64, device=device), mode)
    (ref2 * Gv).sum().backward()
    (out2 * Gv.to(device)).sum().backward()
    torch.testing.assert_close(out2.detach().cpu(), ref2.detach(), rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(yd.grad.cpu(), yr.grad, rtol=RTOL, atol=ATOL)
    if mode == "pma":
        torch.testing.assert_close(a2d.grad.cpu(), a2r.grad, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("rows,P,dc,dtype", [(1000, 8, 16, torch.float32), (777, 4, 32, torch.float32), (5, 2, 4, torch.float32),
                                             (1234, 8, 32, torch.bfloat16), (64, 1, 128, torch.float32), (0, 4, 8, torch.float32)])
def test_block_transpose_is_the_permute_copy(rows, P, dc, dtype, device):
    """The all-to-all's pack / unpack kernel: bit-exact copies of torch's permute, both directions, strided source."""
    from allset_amd import ops
    wide = torch.randn(rows, P * dc + 8, device=device).to(dtype)
    x = wide[:, :P * dc]                                                   # leading dimension > width
    assert ops.block_transpose_supported(x, P, True)
    packed = ops.block_transpose(x, P, True)
    assert packed.shape == (P, rows, dc)
    assert torch.equal(packed, x.reshape(rows, P, dc).permute(1, 0, 2).contiguous())
    back = ops.block_transpose(packed, P, False)
    assert torch.equal(back, x)
    assert not ops.block_transpose_supported(torch.zeros(4, 6, device=device), 2, True)       # 12-byte blocks


@pytest.mark.parametrize("d,H,dtype", [(16, 1, torch.float32), (8, 2, torch.float32), (16, 4, torch.float32), (32, 1, torch.bfloat16),
                                       (4, 1, torch.float32)])
def test_pma_colocated_operands_are_bitwise_the_plain_path(d, H, dtype, device, monkeypatch):
    """Narrow feature rows (the column-sharded layer): logits / backward statistics interleaved with the rows in one
    128-byte-pitched buffer (row-strided operands through the *_ld entry points) -- same kernels, same arithmetic: the
    results must be bit-identical to the separate-table path, forward and backward."""
    import numpy as np
    from allset_amd import Incidence, functional as AF
    rng = np.random.default_rng(d * 10 + H)
    n_s, n_t, nnz = 6000, 5000, 60_000
    ei = torch.unique(torch.stack([torch.from_numpy(rng.integers(0, n_s, nnz)), torch.from_numpy(rng.integers(0, n_t, nnz))]), dim=1).to(device)
    inc = Incidence.from_edge_index(ei, n_src=n_s, n_dst=n_t)
    V = torch.randn(n_s, d, device=device).to(dtype)
    alpha = torch.randn(n_s, H, device=device)
    G = torch.randn(n_t, d, device=device).to(dtype)
    res = []
    for colocate in (True, False):
        if not colocate:
            monkeypatch.setattr(AF, "_colocate", lambda V, heads: False)
        else:
            assert AF._colocate(V, H)
        v, a = V.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
        out, m, l = AF.pma_aggregate(v, a, inc, H, 0.2)
        (out.float() * G.float()).sum().backward()
        res.append((out.detach(), m, l, v.grad, a.grad))
    for x, y in zip(*res):
        assert torch.equal(x, y)


@pytest.mark.parametrize("aggr", ["add", "mean", "max", "min"])
@pytest.mark.parametrize("weighted", [False, True])
def test_aggregate_without_any_incidence(aggr, weighted, device):
    """An incidence with no entries at all (found by the randomised shapes with fresh seeds): every target is empty,
    outputs and every gradient are zero -- forward and backward, including the arg-extreme backward of max / min and the
    per-incidence weight gradient."""
    from allset_amd import Incidence, functional as AF
    ei = torch.zeros((2, 0), dtype=torch.int64, device=device)
    inc = Incidence.from_edge_index(ei, n_src=3, n_dst=2)
    x = torch.randn(3, 5, device=device, requires_grad=True)
    w = torch.zeros(0, device=device, requires_grad=True) if weighted else None
    out = AF.deepsets_aggregate(x, inc, w, aggr)
    assert out.shape == (2, 5) and float(out.abs().max()) == 0.0
    out.sum().backward()
    assert x.grad.shape == x.shape and float(x.grad.abs().max()) == 0.0
    if weighted:
        assert w.grad is not None and w.grad.numel() == 0
    V = torch.randn(3, 8, device=device, requires_grad=True)
    alpha = torch.randn(3, 2, device=device, requires_grad=True)
    o, m, l = AF.pma_aggregate(V, alpha, inc, 2, 0.2)
    o.sum().backward()
    assert float(o.abs().max()) == 0.0 and float(V.grad.abs().max()) == 0.0 and float(alpha.grad.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("heads,d", [(1, 16), (4, 32), (1, 8)])
def test_pma_kernels_with_size_split_dispatch(heads, d, dtype, device, monkeypatch):
    """Skewed row lengths and rows of at most one cache line (the column-sharded PMA layer at P = 8): ``CSR.sizes`` sends the long
    rows to the one-wave-per-row kernel and the compacted short rows to the short-row kernel (ops.SizeSplit).  Same results as the
    single-kernel dispatch, forward and backward, including rows without incidences."""
    from allset_amd import Incidence, ops
    from allset_amd.functional import _variant, _sizes
    from allset_amd.synthetic import random_hypergraph
    if dtype == torch.bfloat16 and d * 2 % 16:
        pytest.skip("bf16 rows must be 16-byte packets")
    hg = random_hypergraph(3000, 2500, 12, seed=5, device=device, dist="zipf", max_degree=700)
    ei = hg.edge_index.clone()
    ei = ei[:, ei[1] % 17 != 3]                                  # some hyperedges lose all members
    inc = Incidence.from_edge_index(ei, n_src=3000, n_dst=2500)
    g = torch.Generator(device="cpu").manual_seed(3)
    V = torch.randn(3000, d, generator=g).to(device).to(dtype)
    alpha = torch.randn(3000, heads, generator=g).to(device)
    gout = torch.randn(2500, d, generator=g).to(device).to(dtype)
    for csr, n_rows in ((inc.by_dst, 2500),):
        assert csr.sizes is not None and csr.sizes.long_ids.numel() > 0 and _sizes(csr, V, heads) is csr.sizes
        ref = ops.pma_fwd(csr.rowptr, csr.col, alpha, V, heads, 0.2, n_rows, variant=1, row_order=csr.row_order)
        got = ops.pma_fwd(csr.rowptr, csr.col, alpha, V, heads, 0.2, n_rows, variant=_variant(csr, "pma_fwd", n_rows, V, heads),
                          row_order=csr.row_order, sizes=csr.sizes)
        tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
        for a, r in zip(got, ref):
            torch.testing.assert_close(a.float(), r.float(), **tol)
    # backward: rows of the TRANSPOSED incidence are the sources; use the hyperedges as sources so that its rows are the skewed ones
    rev = inc.reversed(n_dst=3000)
    T = rev.by_src                                               # rows = hyperedges (Zipf sizes)
    assert T.sizes is not None
    Ve = torch.randn(2500, d, generator=g).to(device).to(dtype)
    ae = torch.randn(2500, heads, generator=g).to(device)
    gv = torch.randn(3000, d, generator=g).to(device).to(dtype)
    out, m, l = ops.pma_fwd(rev.by_dst.rowptr, rev.by_dst.col, ae, Ve, heads, 0.2, 3000, variant=1)
    stats = ops.pma_bwd_stats(out, gv, m, l)
    ref = ops.pma_bwd_src(T.rowptr, T.col, ae, Ve, gv, stats, 0.2, variant=1, row_order=T.row_order)
    got = ops.pma_bwd_src(T.rowptr, T.col, ae, Ve, gv, stats, 0.2, variant=1, row_order=T.row_order, sizes=T.sizes)
    tol = dict(rtol=3e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-5)
    for a, r in zip(got, ref):
        torch.testing.assert_close(a.float(), r.float(), **tol)
