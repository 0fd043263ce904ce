"""GPU: randomised shapes (hypothesis, fixed seed): CSR build is bit-exact against numpy's stable sort, and the
aggregation kernels agree with the oracle for arbitrary (n_src, n_dst, nnz, d, heads) incl. empty inputs, empty
segments, duplicated incidences, single rows holding everything and widths that are not multiples of 4."""
import numpy as np
import pytest
import torch
import os

from hypothesis import HealthCheck, assume, given, seed, settings, strategies as st

from oracle import allset_oracle as oracle

pytestmark = pytest.mark.gpu
# ALLSET_HYPOTHESIS_EXAMPLES / ALLSET_HYPOTHESIS_RANDOM=1: bug-hunting runs (more examples, fresh seeds); the default is a fixed,
# derandomized sample so that the suite is reproducible
_N = int(os.environ.get("ALLSET_HYPOTHESIS_EXAMPLES", "24"))      # (40 until round 4; the sweeps: ALLSET_HYPOTHESIS_EXAMPLES=700 ALLSET_HYPOTHESIS_RANDOM=1)
_DERAND = os.environ.get("ALLSET_HYPOTHESIS_RANDOM", "0") != "1"
COMMON = dict(deadline=None, max_examples=_N, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.filter_too_much], derandomize=_DERAND)


@st.composite
def incidences(draw, max_n=400, max_nnz=3000):
    n_s = draw(st.integers(1, max_n))
    n_t = draw(st.integers(1, max_n))
    nnz = draw(st.integers(0, max_nnz))
    shape = draw(st.sampled_from(["uniform", "one_row", "few_cols", "dups"]))
    rs = np.random.default_rng(draw(st.integers(0, 2 ** 31 - 1)))
    src = rs.integers(0, n_s, size=nnz)
    dst = rs.integers(0, n_t, size=nnz)
    if shape == "one_row":
        dst[:] = rs.integers(0, n_t)
    elif shape == "few_cols":
        src = src % max(1, min(3, n_s))
    elif shape == "dups" and nnz:
        k = max(1, nnz // 4)
        src, dst = np.tile(src[:k], 4)[:nnz], np.tile(dst[:k], 4)[:nnz]
    return n_s, n_t, torch.from_numpy(np.stack([src, dst]).astype(np.int64))


@settings(**COMMON)
@given(inc=incidences())
def test_csr_build_bit_exact(inc, device):
    from allset_amd import Incidence
    n_s, n_t, ei = inc
    I = Incidence.from_edge_index(ei.to(device), n_src=n_s, n_dst=n_t)
    for csr, keys, vals, nr in ((I.by_dst, ei[1].numpy(), ei[0].numpy(), n_t), (I.by_src, ei[0].numpy(), ei[1].numpy(), n_s)):
        order = np.argsort(keys, kind="stable")
        np.testing.assert_array_equal(csr.perm.cpu().numpy(), order)
        np.testing.assert_array_equal(csr.col.cpu().numpy(), vals[order])
        np.testing.assert_array_equal(csr.rowptr.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(keys, minlength=nr))]))


@settings(**COMMON)
@given(inc=incidences(), d=st.sampled_from([1, 2, 4, 7, 16, 33, 64, 128, 200, 256]), aggr=st.sampled_from(["add", "mean", "max", "min"]),
       weighted=st.booleans())
def test_deepsets_aggregate_random(inc, d, aggr, weighted, device):
    from allset_amd import Incidence, deepsets_aggregate
    n_s, n_t, ei = inc
    nnz = ei.shape[1]
    g = torch.Generator().manual_seed(nnz * 131 + d)
    x = torch.randn(n_s, d, generator=g)
    # (an element that is exactly 0.0 ties with the zero the oracle's scatter_reduce starts a max / min from, and torch then splits the
    #  gradient between the element and that start value -- torch_scatter, like this library, gives it to the arg-extremum: SURVEY
    #  A.1; randn does produce exact zeros, found by a fresh-seed sweep at nnz = 146, d = 200)
    x[x == 0] = 0.5
    norm = (0.5 + torch.rand(nnz, generator=g)) if weighted else torch.ones(nnz, dtype=torch.int64)
    G = torch.randn(n_t, d, generator=g)
    xr = x.clone().requires_grad_(True)
    if nnz == 0:                                               # (the reference cannot even size its output here)
        ref = xr.new_zeros(n_t, d) + 0.0 * xr.sum()
    else:
        ref = oracle.deepsets_aggregate(xr, ei, norm, aggr)
    if ref.shape[0] < n_t:                                     # the reference sizes by index.max()+1 (Q1)
        ref = torch.cat([ref, ref.new_zeros(n_t - ref.shape[0], d)])
    (ref * G).sum().backward()
    xd = x.to(device).requires_grad_(True)
    I = Incidence.from_edge_index(ei.to(device), n_src=n_s, n_dst=n_t)
    out = deepsets_aggregate(xd, I, norm.to(device), aggr)
    (out * G.to(device)).sum().backward()
    # (both sides are fp32 sums in different orders: a row of thousands of incidences -- the one_row shape -- carries 1e-7 of its sum of
    #  |terms| as rounding, more than a flat 1e-4 when the sum itself cancels; found by a fresh-seed sweep at n_t = 1, d = 128)
    wabs = norm.abs().double() if weighted else torch.ones(nnz, dtype=torch.float64)
    row_abs = torch.zeros(n_t, dtype=torch.float64).index_add_(0, ei[1], x.abs().double()[ei[0]].amax(1) * wabs) if nnz else torch.zeros(1, dtype=torch.float64)
    col_abs = torch.zeros(n_s, dtype=torch.float64).index_add_(0, ei[0], G.abs().double()[ei[1]].amax(1) * wabs) if nnz else torch.zeros(1, dtype=torch.float64)
    atol_o = max(1e-4, 5e-7 * float(row_abs.max())) if aggr in ("add", "mean") else 1e-4
    atol_g = max(1e-4, 5e-7 * float(col_abs.max()))
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=atol_o)
    if aggr in ("add", "mean"):                               # max/min ties may pick another (equal) argument
        torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=atol_g)
    else:
        torch.testing.assert_close(xd.grad.sum(0).cpu(), xr.grad.sum(0), rtol=1e-3, atol=1e-3)


@settings(**COMMON)
@given(inc=incidences(max_n=300, max_nnz=2000), heads=st.sampled_from([1, 2, 4, 8]), c=st.sampled_from([1, 4, 8, 16, 32]))
def test_pma_aggregate_random(inc, heads, c, device):
    from allset_amd import Incidence, pma_aggregate
    n_s, n_t, ei = inc
    d = heads * c
    g = torch.Generator().manual_seed(ei.shape[1] * 17 + d)
    V = torch.randn(n_s, d, generator=g)
    alpha = 2.0 * torch.randn(n_s, heads, generator=g)
    G = torch.randn(n_t, d, generator=g)
    Vr, ar = V.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    if ei.shape[1] == 0:
        ref = Vr.new_zeros(n_t, d) + 0.0 * (Vr.sum() + ar.sum())
    else:
        ref, _ = oracle.pma_aggregate(Vr.view(-1, heads, c), ar, ei, 0.2)
        ref = ref.reshape(-1, d)
    if ref.shape[0] < n_t:
        ref = torch.cat([ref, ref.new_zeros(n_t - ref.shape[0], d)])
    (ref * G).sum().backward()
    Vd, ad = V.to(device).requires_grad_(True), alpha.to(device).requires_grad_(True)
    I = Incidence.from_edge_index(ei.to(device), n_src=n_s, n_dst=n_t)
    out, m, l = pma_aggregate(Vd, ad, I, heads, 0.2)
    (out * G.to(device)).sum().backward()
    # a target that holds thousands of incidences sums thousands of fp32 terms (and its logit gradient cancels almost
    # completely): the rounding scale grows with the longest row -- found by fresh-seed runs with 2000 duplicates of one pair
    longest = int(max(torch.bincount(ei[1]).max(), torch.bincount(ei[0]).max())) if ei.shape[1] else 1      # target or source row
    slack = max(1.0, longest / 128.0)           # (a lone source under ~800 incidences: its alpha-gradient is an exact 0 reached as an fp32 sum of that many terms)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4 * slack)
    if ei.shape[1]:
        torch.testing.assert_close(Vd.grad.cpu(), Vr.grad, rtol=1e-4, atol=1e-4 * slack)
        torch.testing.assert_close(ad.grad.cpu(), ar.grad, rtol=1e-3, atol=1e-4 * slack)


@settings(**COMMON)
@given(n=st.integers(1, 700), K=st.sampled_from([64, 128, 256, 512]), N=st.sampled_from([64, 128, 256, 512]), has_ln=st.booleans(),
       relu_in=st.booleans(), relu_out=st.booleans(), sd=st.integers(0, 10 ** 6))
def test_fused_norm_linear_random_rows(n, K, N, has_ln, relu_in, relu_out, sd, device):
    """Any row count (tails of the 16-row chunks, n = 1) and every prologue / epilogue combination without dropout:
    output and all gradients of the fused Linear against float64 torch."""
    import torch.nn.functional as F
    from allset_amd import dense
    g = torch.Generator(device=device).manual_seed(sd)
    x = torch.randn(n, K, device=device, generator=g)
    W = torch.randn(N, K, device=device, generator=g) / K ** 0.5
    b = torch.randn(N, device=device, generator=g)
    gamma = 1 + 0.2 * torch.randn(K, device=device, generator=g)
    beta = 0.3 * torch.randn(K, device=device, generator=g)
    G = torch.randn(n, N, device=device, generator=g)
    ref_in = [t.double().requires_grad_(True) for t in (x, gamma, beta, W, b)]
    h = F.relu(ref_in[0]) if relu_in else ref_in[0]
    if has_ln:
        h = F.layer_norm(h, (K,), ref_in[1], ref_in[2], 1e-5)
    ref = F.linear(h, ref_in[3], ref_in[4])
    if relu_out:
        # a pre-activation within rounding of the relu kink takes the other branch in fp32 (a fresh-seed run found
        # -8.8e-8 in float64 against +5.2e-8 in the kernel): the gradient of that row then legitimately differs --
        # such rows get a zero cotangent, so they contribute nothing to any gradient on either side
        G = torch.where((ref.detach().abs() < 1e-5).any(1, keepdim=True).to(G.device), torch.zeros_like(G), G)
        ref = F.relu(ref)
    (ref * G.double()).sum().backward()
    dev_in = [t.clone().requires_grad_(True) for t in (x, gamma, beta, W, b)]
    y = dense.fused_norm_linear(dev_in[0], dev_in[1] if has_ln else None, dev_in[2] if has_ln else None, dev_in[3], dev_in[4],
                                1e-5, relu_in, 0.0, relu_out, 0.0)
    (y * G).sum().backward()
    torch.testing.assert_close(y.detach().double(), ref.detach(), rtol=1e-4, atol=1e-4)
    for nm, a, r in zip(["x", "gamma", "beta", "W", "b"], dev_in, ref_in):
        if nm in ("gamma", "beta") and not has_ln:
            continue
        scale = max(1.0, float(r.grad.abs().max()))
        torch.testing.assert_close(a.grad.double(), r.grad, rtol=2e-4, atol=2e-4 * scale, msg=lambda m: f"{nm}: {m}")


@settings(deadline=None, max_examples=max(16, _N * 3 // 4), suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.filter_too_much],
          derandomize=_DERAND)
@given(pma=st.booleans(), layers=st.integers(1, 3), mlp_layers=st.integers(1, 3), hidden=st.sampled_from([16, 64, 128, 256]),
       heads=st.sampled_from([1, 2, 4]), aggr=st.sampled_from(["add", "mean", "max"]), norm=st.sampled_from(["ln", "bn", "None"]),
       input_norm=st.booleans(), mask=st.booleans(), gpr=st.booleans(), wnorm=st.booleans(), sd=st.integers(0, 10 ** 6))
def test_setgnn_random_configurations_match_oracle(pma, layers, mlp_layers, hidden, heads, aggr, norm, input_norm, mask, gpr,
                                                   wnorm, sd, device):
    """Whole-model parity on random SetGNN configurations (depths, widths on and off the fused paths, heads,
    aggregations, normalisations, LearnMask, GPR, weighted incidences): logits and d(loss)/dx against the oracle,
    the same parameters on both sides."""
    from types import SimpleNamespace
    import cases
    from oracle import allset_oracle as oracle
    from allset_amd import SetGNN
    rng = np.random.default_rng(sd)
    n_v, n_e, f, k = 40, 17, 12, 5
    ei = cases.random_hypergraph(rng, n_v, n_e, 150, True)
    x = rng.standard_normal((n_v, f)).astype(np.float32)
    args = cases.make_args("pma_h1" if pma else "ds_add", f, hidden, k, All_num_layers=layers, MLP_num_layers=mlp_layers,
                           heads=heads if pma else 1, aggregate=aggr if not pma else "add", normalization=norm,
                           deepset_input_norm=input_norm, LearnMask=mask, GPR=gpr, Classifier_num_layers=2)
    nrm = cases._norm_deg_half_sym(ei) if (wnorm or mask) else np.ones(ei.shape[1], dtype=np.int64)
    norm_t = torch.from_numpy(nrm)
    torch.manual_seed(sd)
    model = SetGNN(args, norm=norm_t.to(torch.float32) if mask else None)
    model.reset_parameters()
    model.eval()
    sdict = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    for v in sdict.values():
        if v.is_floating_point():
            v.requires_grad_(False)
    xr = torch.from_numpy(x).clone().requires_grad_(True)
    ref = oracle.setgnn_forward(sdict, args, xr, torch.from_numpy(ei), norm_t)
    G = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
    (ref * G).sum().backward()
    model.to(device)
    xd = torch.from_numpy(x).to(device).requires_grad_(True)
    out = model(SimpleNamespace(x=xd, edge_index=torch.from_numpy(ei).to(device), norm=norm_t.to(device)))
    (out * G.to(device)).sum().backward()
    # The yardstick is the oracle in float64; the fp32 oracle's own distance from it sets the scale of what fp32 can
    # deliver on this configuration (stacked LayerNorms over near-constant rows can be ill-conditioned: fresh-seed runs
    # found cases where the fp32 ORACLE is 2e-3 of the gradient scale away from float64 and the product 5e-6).
    sd64 = {kk: (v.double() if v.is_floating_point() else v) for kk, v in sdict.items()}
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    ref64 = oracle.setgnn_forward(sd64, args, x64, torch.from_numpy(ei), norm_t.double() if norm_t.is_floating_point() else norm_t)
    (ref64 * G.double()).sum().backward()
    # parity is only meaningful where the model is stable: max / min aggregation routes gradients through arg-extremes, and a
    # near-tie makes logits and gradients jump under a 1e-6 change of the input (fresh-seed runs found a 3-layer `max` model
    # whose float64 gradient moves by 7 % of its scale) -- such examples are rejected, not asserted
    # (1e-5: a relu or arg-extreme kink closer than that can be crossed by fp32 rounding inside a 9-Linear stack -- a
    # fresh-seed run found one that the bf16x6 arithmetic crossed and the native fp32 arithmetic did not.  Both signs of the
    # perturbation: a kink right beside the evaluation point is only crossed by one of them)
    nrm64 = norm_t.double() if norm_t.is_floating_point() else norm_t
    dirn = torch.from_numpy(rng.standard_normal(x.shape))
    gs0 = max(1.0, float(x64.grad.abs().max()))
    os0 = max(1.0, float(ref64.detach().abs().max()))
    for sgn in (1.0, -1.0):
        xp = (torch.from_numpy(x).double() + sgn * 1e-5 * dirn).requires_grad_(True)
        refp = oracle.setgnn_forward(sd64, args, xp, torch.from_numpy(ei), nrm64)
        (refp * G.double()).sum().backward()
        assume(float((xp.grad - x64.grad).abs().max()) <= 3e-4 * gs0 and
               float((refp.detach() - ref64.detach()).abs().max()) <= 1e-4 * os0)
    scale = max(1.0, float(ref64.detach().abs().max()))
    o_tol = max(2e-4 * scale, 3.0 * float((ref.detach().double() - ref64.detach()).abs().max()))
    assert float((out.detach().cpu().double() - ref64.detach()).abs().max()) <= o_tol
    gs = max(1.0, float(x64.grad.abs().max()))
    depth = max(1.0, layers * mlp_layers / 2.0)             # rounding (and relu kinks inside it) accumulates with the stack's depth
    g_tol = max(1e-3 * gs * depth, 3.0 * float((xr.grad.double() - x64.grad).abs().max()))
    assert float((xd.grad.cpu().double() - x64.grad).abs().max()) <= g_tol


@settings(deadline=None, max_examples=max(12, _N // 2), suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.filter_too_much],
          derandomize=_DERAND)
@given(pma=st.booleans(), layers=st.integers(1, 2), mlp_layers=st.integers(1, 3), hidden=st.sampled_from([64, 128, 256, 512]),
       heads=st.sampled_from([1, 4, 8]), norm=st.sampled_from(["ln", "None"]), input_norm=st.booleans(), bow=st.booleans(),
       deferred=st.booleans(), twice=st.booleans(), sd=st.integers(0, 10 ** 6))
def test_setgnn_random_parameter_gradients_match_oracle(pma, layers, mlp_layers, hidden, heads, norm, input_norm, bow, deferred, twice, sd, device):
    """EVERY parameter gradient of random SetGNN configurations against the float64 oracle, at the widths with their own kernel
    families (64 / 128 one-pass, 256 / 512 tiled), up to 8 heads, over dense features (their gradient compared as well) or
    bag-of-words rows (3 % non-zeros, 300 columns: the first conv's projection runs from the non-zeros -- ``dense.sparse_pma_project``,
    the sparse LayerNorm + Linear of the Deep Sets encoder), eagerly or with all partial sums in one launch
    (``dense.deferred_param_grads``), optionally twice (the second backward ACCUMULATES into ``.grad``).  Eval mode: no masks to
    share.  A draw is accepted a priori by the oracle's own stability (``util.oracle_is_smooth_here``), never by the comparison."""
    from types import SimpleNamespace
    import cases
    import util
    from allset_amd import SetGNN, dense
    rng = np.random.default_rng(sd)
    n_v, n_e, k = 48, 19, 5
    f = 300 if bow else 12
    ei = cases.random_hypergraph(rng, n_v, n_e, 170, True)
    if bow:
        x = (rng.random((n_v, f)) < 0.03).astype(np.float32)
        x[np.arange(n_v), rng.integers(0, f, n_v)] = 1.0                 # no empty row (LayerNorm of a zero row is all kink)
    else:
        x = rng.standard_normal((n_v, f)).astype(np.float32)
    args = cases.make_args("pma_h1" if pma else "ds_add", f, hidden, k, All_num_layers=layers, MLP_num_layers=mlp_layers,
                           heads=heads if pma else 1, normalization=norm, deepset_input_norm=input_norm, Classifier_num_layers=2)
    nrm = np.ones(ei.shape[1], dtype=np.int64)
    norm_t = torch.from_numpy(nrm)
    torch.manual_seed(sd)
    model = SetGNN(args)
    model.reset_parameters()
    # (every relu firmly on or off by column -- cases.kinkfree_biases: with ~10^5 relu inputs per evaluation at these widths one in
    #  a few draws sits within rounding of a kink otherwise, and what this sweep is after is the gradient kernels of every shape)
    cases.kinkfree_biases(model.state_dict())
    model.eval()
    sdict = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    with torch.no_grad():
        shape = tuple(oracle.setgnn_forward(sdict, args, torch.from_numpy(x), torch.from_numpy(ei), norm_t).shape)
    G = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    # Two a-priori criteria, both of the float64 ORACLE alone.  The perturbation probe at 1e-5 (absolute on dense features, relative on
    # bag-of-words rows): an order above the rounding of the DEFAULT arithmetic at these widths (two fp16 planes carry 2^-22 of a row's
    # scale per product, 512 products per sum; a fresh-seed run found a 512-wide rFF unit within 5e-6 of its relu kink -- strict
    # arithmetic and the fp32 oracle stayed on one side, fp16 planes crossed).  And the direct one: no relu input closer to zero than
    # 2e-5 of its row's scale -- these stacks are bias-dominated, a perturbation of x barely reaches their deep layers, and another
    # fresh-seed run found a pre-activation 9e-8 from its kink that the probe could not move (util.oracle_relu_margin).
    assume(util.oracle_relu_margin(sdict, args, x, ei, nrm) > 2e-5)
    assume(util.oracle_is_smooth_here(sdict, args, x, ei, nrm, G, seed=sd, eps=1e-5, eps_bow=1e-5))

    def oracle_run(dtype):
        sdd = {kk: (v.clone().to(dtype).requires_grad_(True) if v.is_floating_point() and "running" not in kk else v.clone())
               for kk, v in sdict.items()}
        xo = torch.from_numpy(x).to(dtype).requires_grad_(True)
        lo = oracle.setgnn_forward(sdd, args, xo, torch.from_numpy(ei), norm_t)
        (lo * G.to(dtype)).sum().backward()
        return lo.detach(), xo.grad, {kk: v.grad for kk, v in sdd.items() if v.is_floating_point() and v.requires_grad}
    l32, gx32, g32 = oracle_run(torch.float32)
    l64, gx64, g64 = oracle_run(torch.float64)

    model.to(device)
    xd = torch.from_numpy(x).to(device)
    if not bow:
        xd.requires_grad_(True)
    data = SimpleNamespace(x=xd, edge_index=torch.from_numpy(ei).to(device), norm=norm_t.to(device))
    for _ in range(2 if twice else 1):
        out = model(data)
        loss = (out * G.to(device)).sum()
        if deferred:
            with dense.deferred_param_grads():
                loss.backward()
        else:
            loss.backward()
    reps = 2.0 if twice else 1.0
    scale = max(1.0, float(l64.abs().max()))
    assert float((out.detach().cpu().double() - l64).abs().max()) <= max(2e-4 * scale, 3.0 * float((l32.double() - l64).abs().max()))
    depth = max(1.0, layers * mlp_layers / 2.0)
    gscale = max(float(g.abs().max()) for g in g64.values() if g is not None)
    named = dict(model.named_parameters())
    for kk, r in g64.items():
        if r is None:
            assert named[kk].grad is None or float(named[kk].grad.abs().max()) == 0.0, kk
            continue
        got = named[kk].grad
        assert got is not None, kk
        s_k = max(float(r.abs().max()), 1e-2 * gscale, 1e-30)
        tol = max(1e-3 * s_k * depth, 3.0 * float((g32[kk].double() - r).abs().max()))
        err = float((got.detach().cpu().double() / reps - r).abs().max())
        assert err <= tol, f"{kk}: {err:.3e} > {tol:.3e} (scale {s_k:.3e})"
    if not bow:
        gs = max(1.0, float(gx64.abs().max()))
        tol = max(1e-3 * gs * depth, 3.0 * float((gx32.double() - gx64).abs().max()))
        assert float((xd.grad.cpu().double() / reps - gx64).abs().max()) <= tol


@settings(**COMMON)
@given(n=st.integers(1, 900), d=st.sampled_from([1, 3, 4, 7, 8, 32, 60, 64, 100, 128, 200, 256, 260, 512, 1000]), relu_in=st.booleans(),
       bf16=st.booleans(), sd=st.integers(0, 10 ** 6))
def test_layer_norm_random(n, d, relu_in, bf16, sd, device):
    """LayerNorm kernels (fp32: specialised widths and the generic path; bf16: d % 8 == 0, d <= 512) at random sizes against
    float64 on the same values."""
    import torch.nn.functional as F
    from allset_amd import dense
    if bf16:
        assume(d % 8 == 0 and d <= 512)
    dt = torch.bfloat16 if bf16 else torch.float32
    g = torch.Generator(device=device).manual_seed(sd)
    x = torch.randn(n, d, device=device, generator=g).to(dt)
    gamma = (1 + 0.2 * torch.randn(d, device=device, generator=g)).to(dt)
    beta = (0.3 * torch.randn(d, device=device, generator=g)).to(dt)
    G = torch.randn(n, d, device=device, generator=g).to(dt)
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, gamma, beta))
    hin = F.relu(xr) if relu_in else xr
    ref = F.layer_norm(hin, (d,), gr, br, 1e-5)
    (ref * G.double()).sum().backward()
    xd, gd, bd = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    out = dense.layer_norm(xd, gd, bd, 1e-5, relu_in, 0.0)
    (out * G).sum().backward()
    tol = 1.5e-2 if bf16 else 1e-4
    torch.testing.assert_close(out.detach().double(), ref.detach(), rtol=tol, atol=tol)
    # rows whose variance is tiny amplify rounding by rstd (up to 1/sqrt(eps)): compare gradients at the scale of the largest
    gsx = max(1.0, float(xr.grad.abs().max()))
    torch.testing.assert_close(xd.grad.double(), xr.grad, rtol=tol * 2, atol=tol * 2 * gsx)
    for a, r in ((gd.grad, gr.grad), (bd.grad, br.grad)):
        torch.testing.assert_close(a.double(), r, rtol=tol * 2, atol=tol * 2 * max(1.0, float(r.abs().max())))


@settings(**COMMON)
@given(n=st.integers(1, 3000), O=st.sampled_from([4, 8, 64, 100, 128, 132, 256, 260]), I=st.sampled_from([4, 12, 64, 128, 200, 256]),
       bf16=st.booleans(), sd=st.integers(0, 10 ** 6))
def test_weight_gradient_random(n, O, I, bf16, sd, device):
    """Split-K weight gradient (fp32 bf16x6 / bf16) at random sizes: ga^T u and the column sums against float64."""
    from allset_amd import dense
    dt = torch.bfloat16 if bf16 else torch.float32
    g = torch.Generator(device=device).manual_seed(sd)
    ga = torch.randn(n, O, device=device, generator=g).to(dt)
    u = torch.randn(n, I, device=device, generator=g).to(dt)
    gw, gb = dense.wgrad(ga, u, True)
    ref = ga.double().t() @ u.double()
    scale = float((ga.double().abs().t() @ u.double().abs()).max()) + 1e-30
    tol = 1e-2 if bf16 else 2e-6                 # bf16: the result is rounded to bf16 once
    assert float((gw.double() - ref).abs().max()) <= tol * scale
    refb = ga.double().sum(0)
    assert float((gb.double() - refb).abs().max()) <= tol * (float(ga.double().abs().sum(0).max()) + 1e-30)


@settings(**COMMON)
@given(n=st.integers(1, 700), K=st.sampled_from([32, 96, 160, 256, 512]), N=st.sampled_from([4, 12, 100, 256, 260, 512]),
       use_mask=st.booleans(), relu_in=st.booleans(), has_ln=st.booleans(), relu_out=st.booleans(), bias=st.booleans(),
       f16=st.booleans(), sd=st.integers(0, 10 ** 6))
def test_gemm_x6_random(n, K, N, use_mask, relu_in, has_ln, relu_out, bias, f16, sd, device):
    """The tiled GEMM at random sizes (row tails of the 128-row tile, column tails of the 256-column tile, 1..16 K
    steps) with every prologue / epilogue switch, in both arithmetics (``f16``: two fp16 planes, the A window per launch behind a
    LayerNorm, per row -- read one tile ahead -- otherwise), against float64; the error is measured against sum |terms|."""
    import torch.nn.functional as F
    from allset_amd import dense
    g = torch.Generator(device=device).manual_seed(sd)
    x = torch.randn(n, K, device=device, generator=g)
    W = torch.randn(N, K, device=device, generator=g) / K ** 0.5
    b = torch.randn(N, device=device, generator=g) if bias else None
    gamma = 1 + 0.2 * torch.randn(K, device=device, generator=g)
    beta = 0.3 * torch.randn(K, device=device, generator=g)
    ymask = torch.randn(n, K, device=device, generator=g) if use_mask else None
    a = x.double()
    if use_mask:
        a = torch.where(ymask.double() > 0, a / (1 - 0.25), torch.zeros_like(a))
    if relu_in:
        a = F.relu(a)
    stats = None
    if has_ln:
        src = x if not use_mask else torch.where(ymask > 0, x / (1 - 0.25), torch.zeros_like(x))
        stats = dense.row_stats(src.contiguous(), relu_in, 1e-5)
        a = F.layer_norm(a, (K,), gamma.double(), beta.double(), 1e-5)
    pre = a @ W.double().t() + (b.double() if bias else 0.0)
    ref = F.relu(pre) if relu_out else pre
    y = dense.gemm_x6(x, dense.gemm_x6_planes(W, False, f16=f16), N, b, mask_y=ymask, p_mask=0.25 if use_mask else 0.0, relu_in=relu_in,
                      stats=stats, gamma=gamma if has_ln else None, beta=beta if has_ln else None, relu_out=relu_out)
    scale = a.abs() @ W.double().abs().t() + (b.double().abs() if bias else 0.0) + 1e-30
    assert float(((y.double() - ref).abs() / scale).max()) < (2e-5 if has_ln else 2e-6)


@settings(**COMMON)
@given(n=st.integers(1, 900), d=st.sampled_from([4, 8, 32, 64, 100, 128, 200, 256]), with_colb=st.booleans(), with_res=st.booleans(),
       relu_out=st.booleans(), bf16=st.booleans(), sd=st.integers(0, 10 ** 6))
def test_layer_norm_res_random(n, d, with_colb, with_res, relu_out, bf16, sd, device):
    """``relu_out(LN(x + colb + res))`` (PMA tail) at random sizes, fp32 and bf16, against float64 on the same values."""
    import torch.nn.functional as F
    from allset_amd import dense
    if bf16:
        assume(d % 8 == 0)
    dt = torch.bfloat16 if bf16 else torch.float32
    g = torch.Generator(device=device).manual_seed(sd)
    mk = lambda *shape, s=1.0: (s * torch.randn(*shape, device=device, generator=g)).to(dt)
    x, colb, res = mk(n, d), (mk(d, s=0.5) if with_colb else None), (mk(n, d) if with_res else None)
    gamma, beta, G = (1 + mk(d, s=0.2)).to(dt), mk(d, s=0.3), mk(n, d)
    dd = lambda t: None if t is None else t.double().requires_grad_(True)
    xr, cr, rr, gr, br = dd(x), dd(colb), dd(res), dd(gamma), dd(beta)
    pre = F.layer_norm(xr + (cr if cr is not None else 0.0) + (rr if rr is not None else 0.0), (d,), gr, br, 1e-5)
    if relu_out:                                    # rows on the relu kink: zero cotangent (see test_fused_norm_linear_random_rows)
        G = torch.where((pre.detach().abs() < (1e-4 if bf16 else 1e-5)).any(1, keepdim=True), torch.zeros_like(G), G)
    ref = F.relu(pre) if relu_out else pre
    (ref * G.double()).sum().backward()
    dv = lambda t: None if t is None else t.clone().requires_grad_(True)
    xd, cd, rd, gd, bd = dv(x), dv(colb), dv(res), dv(gamma), dv(beta)
    out = dense.layer_norm_res(xd, cd, rd, gd, bd, 1e-5, relu_out, 0.0)
    (out * G).sum().backward()
    tol = 1.5e-2 if bf16 else 1e-4
    torch.testing.assert_close(out.detach().double(), ref.detach(), rtol=tol, atol=tol)
    gsx = max(1.0, float(xr.grad.abs().max()))
    torch.testing.assert_close(xd.grad.double(), xr.grad, rtol=2 * tol, atol=2 * tol * gsx)
    if with_res:
        torch.testing.assert_close(rd.grad.double(), rr.grad, rtol=2 * tol, atol=2 * tol * gsx)
    pairs = [(gd.grad, gr.grad), (bd.grad, br.grad)] + ([(cd.grad, cr.grad)] if with_colb else [])
    for a_, r_ in pairs:
        torch.testing.assert_close(a_.double(), r_, rtol=2 * tol, atol=2 * tol * max(1.0, float(r_.abs().max())))


@settings(**COMMON)
@given(n=st.integers(1, 900), K=st.sampled_from([64, 128, 256, 512]), N=st.sampled_from([64, 128, 256, 512]),
       p_in=st.sampled_from([0.0, 0.2, 0.5]), p_out=st.sampled_from([0.0, 0.3, 0.5]), sd=st.integers(0, 10 ** 6))
def test_dropout_masks_agree_between_fused_and_unfused_chains(n, K, N, p_in, p_out, sd, device):
    """With explicit seeds the one-kernel Linear (LDS-resident weights for widths <= 128, tiled GEMM beyond) and the
    unfused HIP chain LayerNorm -> library GEMM -> relu/dropout draw the SAME masks at any row count: forward output, the
    gradient of the Linear's input, weight and bias gradients."""
    import torch.nn.functional as F
    from allset_amd import _lib, dense
    g = torch.Generator(device=device).manual_seed(sd)
    x = torch.randn(n, K, device=device, generator=g)
    W = torch.randn(N, K, device=device, generator=g) / K ** 0.5
    b = torch.randn(N, device=device, generator=g)
    gamma = 1 + 0.2 * torch.randn(K, device=device, generator=g)
    beta = 0.3 * torch.randn(K, device=device, generator=g)
    G = torch.randn(n, N, device=device, generator=g)
    s_in, s_out = sd + 11, sd + 29
    u, st_u = dense.ln_fwd(x, gamma, beta, 1e-5, True, p_in, s_in)
    a = F.linear(u, W, b)
    G = torch.where((a.abs() < 1e-5).any(1, keepdim=True), torch.zeros_like(G), G)     # relu kink rows: zero cotangent
    y_ref = torch.empty_like(a)
    _lib.check(_lib.load().allset_relu_dropout_fwd(a.data_ptr(), p_out, s_out, y_ref.data_ptr(), a.numel(), None,
                                                   torch.cuda.current_stream().cuda_stream), "relu_dropout_fwd")
    ga = torch.where(y_ref > 0, G / (1 - p_out), torch.zeros_like(G))
    gu_ref = ga @ W
    if dense.fused_linear_supported(K, N):
        y, st = dense.fused_linear_fwd(x, W, b, gamma, beta, 1e-5, True, p_in, s_in, True, p_out, s_out)
        gx, dg, db = dense.fused_linear_bwd(G, y, p_out, W, x, st, gamma, True, p_in, s_in)
        gx_ref, dg_ref, db_ref = dense.ln_bwd(gu_ref, x, st_u, gamma, True, p_in, s_in)
        sc = max(1.0, float(gx_ref.abs().max()))
        torch.testing.assert_close(gx, gx_ref, rtol=1e-4, atol=1e-4 * sc)
        torch.testing.assert_close(dg, dg_ref, rtol=1e-4, atol=2e-3 * max(1.0, float(dg_ref.abs().max())))
    else:
        st = dense.row_stats(x, True, 1e-5)
        y = dense.gemm_x6(x, dense.gemm_x6_planes(W, False), N, b, relu_in=True, stats=st, gamma=gamma, beta=beta, p_in=p_in,
                          seed_in=s_in, relu_out=True, p_out=p_out, seed_out=s_out)
        gu = dense.gemm_x6(G, dense.gemm_x6_planes(W, True), K, None, mask_y=y, p_mask=p_out)
        torch.testing.assert_close(gu, gu_ref, rtol=1e-4, atol=1e-4 * max(1.0, float(gu_ref.abs().max())))
    torch.testing.assert_close(st, st_u, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(y, y_ref, rtol=1e-4, atol=1e-4)
    gw, gb = dense.wgrad_fused(G, y, p_out, x, st, gamma, beta, True, p_in, s_in)
    gw_ref = ga.t() @ u
    torch.testing.assert_close(gw, gw_ref, rtol=1e-4, atol=1e-4 * max(1.0, float(gw_ref.abs().max())))
    torch.testing.assert_close(gb, ga.sum(0), rtol=1e-4, atol=1e-4 * max(1.0, float(ga.sum(0).abs().max())))


@settings(**COMMON)
@given(inc=incidences(max_n=300, max_nnz=2000), d=st.sampled_from([8, 16, 32, 64, 128, 256]), aggr=st.sampled_from(["add", "mean"]),
       weighted=st.booleans(), pma_heads=st.sampled_from([0, 1, 2, 4]))
def test_bf16_storage_aggregate_random(inc, d, aggr, weighted, pma_heads, device):
    """bf16 feature storage (BASELINE configs[4] regime) at random shapes: the sum / mean aggregation and the attention pooling
    read bf16 rows, accumulate in fp32 and round once -- against the oracle in float64 on the same bf16 values."""
    from allset_amd import Incidence, deepsets_aggregate, pma_aggregate
    n_s, n_t, ei = inc
    nnz = ei.shape[1]
    assume(nnz > 0)
    g = torch.Generator().manual_seed(nnz * 7 + d)
    x = torch.randn(n_s, d, generator=g).to(torch.bfloat16)
    I = Incidence.from_edge_index(ei.to(device), n_src=n_s, n_dst=n_t)
    longest = int(max(torch.bincount(ei[1]).max(), torch.bincount(ei[0]).max()))
    if pma_heads == 0:
        norm = (0.5 + torch.rand(nnz, generator=g)) if weighted else torch.ones(nnz, dtype=torch.int64)
        ref = oracle.deepsets_aggregate(x.double(), ei, norm.double() if weighted else norm, aggr)
        out = deepsets_aggregate(x.to(device), I, norm.to(device), aggr)
    else:
        assume(d % pma_heads == 0 and (d // pma_heads) % 8 == 0)
        alpha = 2.0 * torch.randn(n_s, pma_heads, generator=g)
        ref, _ = oracle.pma_aggregate(x.double().view(n_s, pma_heads, -1), alpha.double(), ei, 0.2)
        ref = ref.reshape(-1, d)
        out, _, _ = pma_aggregate(x.to(device), alpha.to(device), I, pma_heads, 0.2)
    if ref.shape[0] < n_t:
        ref = torch.cat([ref, ref.new_zeros(n_t - ref.shape[0], d)])
    assert out.dtype == torch.bfloat16
    # one bf16 rounding of the result (2^-9 relative) on top of fp32 accumulation
    tol = 2.0 ** -8
    torch.testing.assert_close(out.float().cpu().double(), ref, rtol=tol, atol=tol * max(1.0, longest / 64.0))


@settings(**COMMON)
@given(n=st.integers(1, 900), K=st.sampled_from([64, 128, 256, 512]), N=st.sampled_from([4, 64, 100, 128, 256]), relu_in=st.booleans(),
       p=st.sampled_from([0.0, 0.3]), use_mask=st.booleans(), f16=st.booleans(), sd=st.integers(0, 10 ** 6))
def test_gemm_with_layer_norm_backward_epilogue_random(n, K, N, relu_in, p, use_mask, f16, sd, device):
    """allset_gemm_x6_lnb (backward-data of a wide Linear with the LayerNorm backward as the GEMM's epilogue) against the two
    kernels it replaces, allset_gemm_x6 + allset_ln_bwd, with the same seeds: identical masks, results to rounding."""
    from allset_amd import dense
    g = torch.Generator(device=device).manual_seed(sd)
    G = torch.randn(n, K, device=device, generator=g)                 # gradient of the Linear's output [n, out = K of this GEMM]
    W = torch.randn(K, N, device=device, generator=g) / K ** 0.5      # weight [out, in = N]
    x = torch.randn(n, N, device=device, generator=g)                 # the Linear's (pre-LayerNorm) input
    gamma = 1 + 0.2 * torch.randn(N, device=device, generator=g)
    ymask = torch.randn(n, K, device=device, generator=g) if use_mask else None
    stats = dense.row_stats(x, relu_in, 1e-5) if N <= 512 else None
    planes_t = dense.gemm_x6_planes(W, True, f16=False)
    gu = dense.gemm_x6(G, planes_t, N, None, mask_y=ymask, p_mask=0.25 if use_mask else 0.0)
    gx_ref, dg_ref, db_ref = dense.ln_bwd(gu, x, stats, gamma, relu_in, p, sd + 5)
    gx, dg, db = dense.gemm_x6_lnb(G, dense.gemm_x6_planes(W, True, f16=f16), x, stats, gamma, relu_in, p, sd + 5, mask_y=ymask,
                                   p_mask=0.25 if use_mask else 0.0)
    sc = max(1.0, float(gx_ref.abs().max()))
    torch.testing.assert_close(gx, gx_ref, rtol=1e-4, atol=1e-4 * sc)
    torch.testing.assert_close(dg, dg_ref, rtol=1e-4, atol=1e-3 * max(1.0, float(dg_ref.abs().max())))
    torch.testing.assert_close(db, db_ref, rtol=1e-4, atol=1e-3 * max(1.0, float(db_ref.abs().max())))


@settings(**COMMON)
@given(n=st.integers(1, 5000), K=st.sampled_from([128, 256]), N=st.sampled_from([128, 256]), relu=st.booleans(), aux=st.booleans(),
       bias=st.booleans(), pad=st.sampled_from([0, 8, 64]), sd=st.integers(0, 2 ** 31 - 1))
def test_linear_bf16_forward_random(n, K, N, relu, aux, bias, pad, sd, device):
    """csrc/fused_bf16.hip forward: random row counts, row-strided inputs, every fold; bound = one bf16 rounding of the fp32
    reference on the same bf16 inputs."""
    from allset_amd import dense
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(sd)
    big = torch.randn(n, K + pad, generator=g).to(torch.bfloat16).to(device)
    x = big[:, pad:] if pad else big
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(device)
    b = torch.randn(N, generator=g).to(torch.bfloat16).to(device) if bias else None
    aw = torch.randn(4, K, generator=g).to(torch.bfloat16).to(device) if aux else None
    y, a = dense.linear_bf16_fwd(x, W, b, relu, aw, None)
    ref = x.float() @ W.float().t() + (b.float() if bias else 0.0)
    ref = F.relu(ref) if relu else ref
    tol = ref.abs() * 2.0 ** -8 + 1e-3 * float(ref.abs().max()) * 2.0 ** -8 + 1e-30
    assert bool(((y.float() - ref).abs() <= tol).all())
    if aux:
        torch.testing.assert_close(a, x.float() @ aw.float().t(), rtol=1e-5, atol=1e-5)


@settings(**COMMON)
@given(n=st.integers(1, 5000), O=st.sampled_from([128, 256]), I=st.sampled_from([128, 256]), mask=st.booleans(), acc=st.booleans(),
       aux=st.booleans(), want_ga=st.booleans(), sd=st.integers(0, 2 ** 31 - 1))
def test_linear_bf16_backward_random(n, O, I, mask, acc, aux, want_ga, sd, device):
    from allset_amd import dense
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(sd)
    gy = torch.randn(n, O, generator=g).to(torch.bfloat16).to(device)
    W = (torch.randn(O, I, generator=g) / O ** 0.5).to(torch.bfloat16).to(device)
    y = F.relu(torch.randn(n, O, generator=g)).to(torch.bfloat16).to(device) if mask else None
    a = torch.randn(n, I, generator=g).to(torch.bfloat16).to(device) if acc else None
    ga4 = torch.randn(n, 4, generator=g).to(device) if aux else None
    aw = torch.randn(4, I, generator=g).to(torch.bfloat16).to(device) if aux else None
    gx, ga = dense.linear_bf16_bwd(gy, W, y, want_ga=want_ga and mask, acc_in=a, galpha=ga4, aux_w=aw)
    ga_ref = torch.where(y.float() > 0, gy.float(), torch.zeros((), device=device)) if mask else gy.float()
    if (want_ga and mask) or not mask:
        assert torch.equal(ga.float(), ga_ref)
    ref = ga_ref @ W.float() + (a.float() if acc else 0.0) + ((ga4 @ aw.float()) if aux else 0.0)
    tol = ref.abs() * 2.0 ** -8 + 1e-3 * float(ref.abs().max()) * 2.0 ** -8 + 1e-30
    assert bool(((gx.float() - ref).abs() <= tol).all())


@settings(**COMMON)
@given(n=st.integers(1, 4000), C=st.integers(1, 80), frac=st.floats(0.0, 1.0), sd=st.integers(0, 2 ** 31 - 1))
def test_fused_loss_random(n, C, frac, sd, device):
    from allset_amd.losses import nll_log_softmax
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(sd)
    logits = (4 * torch.randn(n, C, generator=g)).to(device)
    yv = torch.randint(0, C, (n,), generator=g).to(device)
    w = (torch.rand(n, generator=g) < frac).float().to(device)
    cnt = max(1.0, float(w.sum()))
    a = logits.clone().requires_grad_(True)
    loss = nll_log_softmax(a, yv, w, cnt)
    loss.backward()
    b = logits.double().requires_grad_(True)
    ref = -(F.log_softmax(b, dim=1).gather(1, yv.view(-1, 1)).squeeze(1) * w.double()).sum() / cnt
    ref.backward()
    torch.testing.assert_close(loss.double(), ref, rtol=1e-5, atol=1e-6)
    # (softmax - onehot cancels in fp32 where the label's probability is 1 - 1e-5: an absolute 3e-7 before the division by cnt; found by
    #  a fresh-seed sweep at n = 1, C = 2)
    torch.testing.assert_close(a.grad.double(), b.grad, rtol=1e-5, atol=max(1e-7, 5e-7 / cnt))
