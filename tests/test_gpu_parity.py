"""GPU: allset_amd.SetGNN (HIP kernels through the C ABI) against the golden vectors produced by the real
reference, and against the oracle run live on the same inputs.  fp32; tolerance rtol = atol = 1e-4
(BASELINE.json configs[1]: "fp32 vs reference within 1e-4"), atol relative to the tensor's max-abs."""
import pytest
import torch

import cases
import util

pytestmark = pytest.mark.gpu
RTOL = ATOL = 1e-4


@pytest.mark.parametrize("name", cases.ALL_CASES)
def test_setgnn_matches_golden_and_oracle(name, device):
    case, g = cases.build_case(name), util.load_golden(name)
    sd = util.state_dict_for(case, g)
    res = util.run_product(case, sd, device)
    util.assert_matches_golden(res, g, case["big"], rtol=RTOL, atol=ATOL)
    orc = util.run_oracle(case, sd)
    for k in ("logits", "v2e0", "e2v0", "grad_x"):
        scale = max(float(orc[k].abs().max()), 1e-3)
        torch.testing.assert_close(res[k], orc[k], rtol=RTOL, atol=ATOL * scale, msg=lambda m: f"{name}/{k}: {m}")
    gscale = max(float(v.abs().max()) for v in orc["grads"].values())
    for k, gexp in orc["grads"].items():
        if k in res["grads"]:
            scale = max(float(gexp.abs().max()), 1e-2 * gscale, 1e-3)   # see util.assert_matches_golden
            torch.testing.assert_close(res["grads"][k], gexp.detach(), rtol=RTOL, atol=ATOL * scale,
                                       msg=lambda m: f"{name}/grad {k}: {m}")


def test_attention_weights_match_reference(device):
    """PMA.forward(..., return_attention_weights=True) -> (out, (edge_index, alpha[nnz,H])) in the caller's
    edge-list order (pins the perm routing)."""
    name = "rand50_pma_h4"
    case, g = cases.build_case(name), util.load_golden(name)
    res = util.run_product(case, util.state_dict_for(case, g), device)
    model, data = res["model"], res["data"]
    with torch.no_grad():
        out, (ei, alpha) = model.V2EConvs[0].prop(data.x.detach(), data.edge_index, return_attention_weights=True)
    assert ei is data.edge_index and tuple(alpha.shape) == tuple(g["attn_v2e0"].shape)
    torch.testing.assert_close(alpha.cpu(), torch.from_numpy(g["attn_v2e0"]), rtol=RTOL, atol=1e-6)
    torch.testing.assert_close(out.cpu(), res["v2e0"], rtol=1e-5, atol=1e-5)


def test_edge_index_is_rebased_in_place_like_the_reference(device):
    """SetGNN.forward shifts data.edge_index[1] to start at 0, in place, once."""
    case, g = cases.build_case("rand50_ds_add"), util.load_golden("rand50_ds_add")
    res = util.run_product(case, util.state_dict_for(case, g), device)
    ei = res["data"].edge_index
    assert int(ei[1].min()) == 0 and int(torch.from_numpy(case["edge_index"])[1].min()) == 50
    v = ei._version
    with torch.no_grad():
        again = res["model"](res["data"])
        third = res["model"](res["data"])
    assert ei._version == v                                   # second call: untouched, cache hit
    torch.testing.assert_close(third, again, rtol=0, atol=0)  # deterministic kernels
    # (run_product's forward differentiates with respect to x; a no-grad forward takes the raw features through
    #  dense.input_norm_linear -- LayerNorm folded into one GEMM -- a different fp32 summation order)
    torch.testing.assert_close(again.cpu(), res["logits"], rtol=1e-5, atol=1e-6)


def test_learnmask_eval_follows_importance_between_no_grad_forwards(device):
    """`norm = Importance * norm` is a fresh temporary on every forward (reference models.py:451-452); under no_grad
    the caching allocator hands the next one the same address with _version 0, so a weight cache keyed on the address
    alone served the FIRST evaluation's mask for ever.  Three evaluations with different Importance, each against the
    oracle; the first one with Importance all ones (the cached '(None, None) = unweighted' case)."""
    from oracle import allset_oracle as oracle
    name = "rand50_ds_add_wnorm_mask"
    case, g = cases.build_case(name), util.load_golden(name)
    sd = util.state_dict_for(case, g)
    res = util.run_product(case, sd, device)
    model, data = res["model"], res["data"]
    nnz = case["norm"].shape[0]
    ones_norm = torch.ones(nnz, device=device)
    data_ones = type(data)(x=data.x.detach(), edge_index=data.edge_index, norm=ones_norm)
    gen = torch.Generator().manual_seed(5)
    for step in range(3):
        imp = torch.ones(nnz) if step == 0 else torch.rand(nnz, generator=gen) + 0.5
        with torch.no_grad():
            model.Importance.copy_(imp.to(device))
            got = model(data_ones)
        sd_o = dict(sd)
        sd_o["Importance"] = imp
        exp = oracle.setgnn_forward(sd_o, case["args"], torch.from_numpy(case["x"]), torch.from_numpy(case["edge_index"]),
                                    torch.ones(nnz))
        torch.testing.assert_close(got.cpu(), exp.detach(), rtol=RTOL, atol=ATOL * max(float(exp.abs().max()), 1e-3),
                                   msg=lambda m: f"evaluation {step}: {m}")


def test_incidence_cache_rebuilds_for_a_new_graph_at_a_recycled_address(device):
    """A second hypergraph with the same number of incidences whose edge_index lands at the freed address of the first
    must not be aggregated over the first one's CSR (SetGNN._inc_cache / cached_incidence keep a weakref)."""
    from types import SimpleNamespace
    from allset_amd import SetGNN
    from oracle import allset_oracle as oracle
    case = cases.build_case("rand50_ds_add")
    g = util.load_golden("rand50_ds_add")
    sd = util.state_dict_for(case, g)
    model = SetGNN(case["args"])
    model.load_state_dict(sd)
    model.eval().to(device)
    x = torch.from_numpy(case["x"]).to(device)
    ei0 = torch.from_numpy(case["edge_index"])
    perm = torch.randperm(50, generator=torch.Generator().manual_seed(1))
    ei1 = torch.stack([perm[ei0[0]], ei0[1]])                  # same nnz, same hyperedge ids, vertices relabelled
    ei1 = ei1[:, torch.argsort(ei1[0], stable=True)].contiguous()
    norm = torch.from_numpy(case["norm"])
    outs, ptrs = [], []
    for ei in (ei0, ei1):
        dev_ei = ei.clone().to(device)
        ptrs.append(dev_ei.data_ptr())
        with torch.no_grad():
            outs.append(model(SimpleNamespace(x=x, edge_index=dev_ei, norm=norm.to(device))).cpu())
        del dev_ei                                               # freed: the next clone usually reuses the block
    exp1 = oracle.setgnn_forward(sd, case["args"], torch.from_numpy(case["x"]), ei1.clone(), norm)
    torch.testing.assert_close(outs[1], exp1.detach(), rtol=RTOL, atol=ATOL * max(float(exp1.abs().max()), 1e-3))
    assert not torch.allclose(outs[0], outs[1])


def test_training_mode_runs_and_is_finite(device):
    """train() mode (dropout active, BatchNorm batch statistics) exercises the same kernels; only
    finiteness/shape can be asserted because dropout RNG streams differ between devices."""
    from allset_amd import SetGNN
    from types import SimpleNamespace
    for name in ("rand50_pma_h4", "rand50_ds_add_bn"):
        case = cases.build_case(name)
        model = SetGNN(case["args"]).to(device).train()
        model.reset_parameters()
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(device), edge_index=torch.from_numpy(case["edge_index"]).to(device),
                               norm=torch.from_numpy(case["norm"]).to(device))
        y = torch.randint(0, 7, (50,), device=device)
        losses = []
        for _ in range(5):
            opt.zero_grad()
            loss = torch.nn.functional.nll_loss(torch.log_softmax(model(data), dim=1), y)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        assert all(map(lambda v: v == v and abs(v) < 1e6, losses))


@pytest.mark.parametrize("attention", [False, True])
def test_sharded_layer_world1_equals_module_composition(attention, device):
    """allset_amd.dist layers with the HIP kernels (world size 1: collectives are identities) must equal the
    plain HalfNLHconv composition SetGNN runs -- covers HipPmaKernels / the E->V merge Function on the GPU."""
    import numpy as np
    import torch.nn.functional as F
    from allset_amd import HalfNLHconv, Incidence
    from allset_amd import dist as adist
    rng = np.random.default_rng(3)
    n_v, n_e, d, H = 301, 157, 64, 4
    pairs = sorted({(int(rng.integers(n_v)), int(rng.integers(n_e))) for _ in range(2500)})
    ei = torch.tensor(pairs, dtype=torch.int64).t().contiguous().to(device)
    torch.manual_seed(0)
    a = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=H, attention=attention).to(device).eval()
    b = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=H, attention=attention).to(device).eval()
    x = torch.randn(n_v, d, device=device)
    G = torch.randn(n_v, d, device=device)
    hg = adist.ShardedHypergraph(ei, n_v, n_e, 1, 0).build_incidences()
    xs = x.clone().requires_grad_(True)
    if attention:
        out = adist.sharded_pma_layer(a, b, xs, hg)
    else:
        out = adist.sharded_deepsets_layer(a, b, xs, hg, aggr="add")
    (out * G).sum().backward()
    gs = [p.grad.clone() for p in list(a.parameters()) + list(b.parameters())]
    for p in list(a.parameters()) + list(b.parameters()):
        p.grad = None
    inc = Incidence.from_edge_index(ei, n_src=n_v, n_dst=n_e)
    xr = x.clone().requires_grad_(True)
    ref = F.relu(b(F.relu(a(xr, inc, None, "add")), inc.reversed(n_dst=n_v), None, "add"))
    (ref * G).sum().backward()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xs.grad, xr.grad, rtol=1e-4, atol=1e-4)
    for g, p in zip(gs, list(a.parameters()) + list(b.parameters())):
        torch.testing.assert_close(g, p.grad, rtol=1e-3, atol=1e-3 * max(1.0, float(p.grad.abs().max())))


def test_setgnn_bf16_storage_tracks_fp32(device):
    """BASELINE configs[4] regime at model level: an AllSetTransformer in bfloat16 (bf16 feature storage through the
    bf16 instantiations of the PMA kernels, fp32 softmax statistics / accumulation) stays within bf16 rounding of the
    fp32 model.  The whole model is bf16 here (weights, LayerNorm, GEMMs through torch), so the bound is statistical:
    mean abs error below 3 % of the mean magnitude, worst element below 20 % of the max magnitude (the kernel-level bf16
    bound, 1e-2, is in test_gpu_ops.py::test_bf16_storage_fp32_accumulate)."""
    from types import SimpleNamespace
    from allset_amd import SetGNN
    case = cases.build_case("edge_pma_h4")                      # includes the 4096-member hyperedge
    g = util.load_golden("edge_pma_h4")
    sd = util.state_dict_for(case, g)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        model = SetGNN(case["args"])
        model.load_state_dict(sd)
        model = model.eval().to(device).to(dt)
        x = torch.from_numpy(case["x"]).to(device).to(dt).requires_grad_(True)
        data = SimpleNamespace(x=x, edge_index=torch.from_numpy(case["edge_index"]).to(device),
                               norm=torch.from_numpy(case["norm"]).to(device))
        logits = model(data)
        assert logits.dtype == dt
        G = torch.from_numpy(cases.cotangent(case["name"], logits.shape)).to(device).to(dt)
        (logits * G).sum().backward()
        outs[dt] = (logits.detach().float(), x.grad.detach().float())
    for a, b in zip(outs[torch.bfloat16], outs[torch.float32]):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().mean()) <= 3e-2 * float(b.abs().mean()) + 1e-4
        assert float((a - b).abs().max()) <= 0.2 * float(b.abs().max()) + 1e-3


def _zipf256_case():
    """BASELINE configs[4]'s regime in miniature: MLP_hidden 256, PMA with 4 heads, power-law hyperedge sizes (truncated Zipf, exponent
    1.6, sizes 1 .. 300) plus ONE hyperedge of 4096 members, self-loop hyperedges; live oracle only (no fixture)."""
    import zlib
    import numpy as np
    seed = zlib.crc32(b"zipf256_pma_h4") & 0x7FFFFFFF
    rng = np.random.default_rng(seed)
    n_v = 6000
    pairs = [(int(v), 0) for v in rng.choice(n_v, size=4096, replace=False)]
    for e in range(1, 320):
        k = int(min(max(rng.zipf(1.6), 1), 300))
        pairs += [(int(v), e) for v in rng.choice(n_v, size=k, replace=False)]
    ei = cases._finish(sorted(set(pairs)), n_v, True)
    x = rng.standard_normal((n_v, 48)).astype(np.float32)
    args = cases.make_args("pma_h4", 48, 256, 6)
    return dict(name="zipf256_pma_h4", args=args, x=x, edge_index=ei, norm=np.ones(ei.shape[1], dtype=np.int64), seed=seed, big=True,
                kinkfree=False)


def _rel_l2(got, exp):
    return float((got.double() - exp.double()).norm()) / max(float(exp.double().norm()), 1e-30)


@pytest.mark.parametrize("name", ["rand50_pma_h4", "edge_pma_h4", "wide256_pma_h4", "zipf256_pma_h4", "rand50_ds_add", "wide256_ds_add"])
def test_setgnn_bf16_matches_the_oracle_on_bf16_rounded_inputs(name, device):
    """BASELINE configs[4] regime at model level, AGAINST THE ORACLE, on logits, the input gradient AND EVERY PARAMETER GRADIENT:
    parameters and features are rounded to bf16 once, the product runs end to end in bfloat16 (bf16 storage, fp32 accumulation /
    softmax statistics), the oracle runs the same rounded numbers in fp32 on the CPU (= the expected values).  What bf16 can hold
    is measured, not guessed: the YARDSTICK is the same oracle evaluated in bfloat16 on the CPU (every stored activation rounded to
    2^-9 relative, torch's bf16 kernels) -- its relative L2 distance from the fp32 evaluation is 0.4 % on logits, 2 - 10 % on
    AllSetTransformer gradients, 11 - 17 % on AllDeepSets gradients (un-normalised segment sums in front of a LayerNorm amplify the
    rounding).  The product must be within 1.5 x the yardstick's distance (+ 5e-3 of the tensor's norm, + 3e-4 of the case's largest
    gradient norm: a gradient tensor a hundred times smaller than its neighbours -- lin_K.bias, the signal through the attention
    logits -- carries their absolute rounding noise) on every tensor.  A wrong weight gradient (an operand transposed, a missing
    mask) is at relative distance ~1 and fails by an order of magnitude."""
    from types import SimpleNamespace
    from allset_amd import SetGNN
    from oracle import allset_oracle as oracle
    if name == "zipf256_pma_h4":
        case = _zipf256_case()
        spec = [(k, tuple(v.shape)) for k, v in SetGNN(case["args"]).state_dict().items()]
        sd32 = {k: torch.from_numpy(v) for k, v in cases.make_state_dict(spec, case["seed"], False).items()}
    else:
        case = cases.build_case(name)
        sd32 = util.state_dict_for(case, util.load_golden(name))
    sd = {k: (v.to(torch.bfloat16).float() if v.is_floating_point() else v) for k, v in sd32.items()}
    xb = torch.from_numpy(case["x"]).to(torch.bfloat16)
    model = SetGNN(case["args"])
    model.load_state_dict(sd)
    model = model.eval().to(device).to(torch.bfloat16)
    x = xb.to(device).requires_grad_(True)
    data = SimpleNamespace(x=x, edge_index=torch.from_numpy(case["edge_index"]).to(device),
                           norm=torch.from_numpy(case["norm"]).to(device))
    logits = model(data)
    assert logits.dtype == torch.bfloat16
    G = torch.from_numpy(cases.cotangent(case["name"], logits.shape)).to(torch.bfloat16)
    (logits * G.to(device)).sum().backward()
    got = {"logits": logits.detach().float().cpu(), "grad_x": x.grad.float().cpu()}
    for k, p in model.named_parameters():
        if p.grad is not None:
            got["grad " + k] = p.grad.float().cpu()

    def run_oracle(dt):
        s = {k: (v.to(dt).detach().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
        xo = xb.to(dt).detach().requires_grad_(True)
        lo = oracle.setgnn_forward(s, case["args"], xo, torch.from_numpy(case["edge_index"]).clone(), torch.from_numpy(case["norm"]))
        (lo * G.to(dt)).sum().backward()
        out = {"logits": lo.detach().float(), "grad_x": xo.grad.float()}
        out.update({"grad " + k: v.grad.float() for k, v in s.items() if v.is_floating_point() and v.requires_grad and v.grad is not None})
        return out
    exp, yard = run_oracle(torch.float32), run_oracle(torch.bfloat16)
    assert set(k for k in exp if k.startswith("grad ")) <= set(got), sorted(set(exp) - set(got))      # every parameter gradient is compared
    gnorm = max(float(v.norm()) for k, v in exp.items() if k.startswith("grad "))
    worst = []
    for k, e in exp.items():
        assert torch.isfinite(got[k]).all(), k
        if k.startswith("grad ") and float(e.norm()) < 1e-3 * gnorm:
            continue                                      # analytically (near-)zero gradients: rounding noise on every side
        dp, dy = _rel_l2(got[k], e), _rel_l2(yard[k], e)
        floor = 3e-4 * gnorm / max(float(e.norm()), 1e-30) if k.startswith("grad ") else 0.0
        worst.append((dp / (1.5 * dy + 5e-3 + floor), k, dp, dy))
    bad = [w for w in worst if w[0] > 1.0]
    assert not bad, f"{name}: product further from the fp32 oracle than 1.5 x the bf16 oracle + floors on " + \
        "; ".join(f"{k} (product {dp:.3e}, bf16 oracle {dy:.3e})" for _, k, dp, dy in sorted(bad, reverse=True))


@pytest.mark.parametrize("name", ["wide256_pma_h4", "zipf256_pma_h4"])
def test_bf16_parity_bound_catches_a_wrong_weight_gradient(name, device, monkeypatch):
    """The bound above is tight enough to notice a wrong ``wgrad_bf16``: with the bf16 weight-gradient wrapper returning the
    TRANSPOSE of every square weight gradient (a swapped operand), the same test must fail -- on a weight gradient."""
    from allset_amd import dense
    real = dense.wgrad

    def transposed(ga, u, want_bias=True):
        gw, gb = real(ga, u, want_bias)
        if ga.dtype == torch.bfloat16 and gw.shape[0] == gw.shape[1]:
            gw = gw.t().contiguous()
        return gw, gb
    monkeypatch.setattr(dense, "wgrad", transposed)
    real2 = dense.wgrad_bf16_ex2

    def transposed2(ga, u, bits=None, g4=None, want_bias=True, defer_to=None):    # (round 6: the masked / auxiliary-row form of the same kernel)
        res = list(real2(ga, u, bits=bits, g4=g4, want_bias=want_bias))            # (reduced at once: the test transposes the result)
        if res[0].shape[0] == res[0].shape[1]:
            res[0] = res[0].t().contiguous()
        return tuple(res)
    monkeypatch.setattr(dense, "wgrad_bf16_ex2", transposed2)
    with pytest.raises(AssertionError, match=r"grad .*weight"):
        test_setgnn_bf16_matches_the_oracle_on_bf16_rounded_inputs(name, device)


@pytest.mark.parametrize("layers", [1, 2, 3])
@pytest.mark.parametrize("kind,inorm", [("ln", True), ("ln", False), ("None", False), ("bn", True)])
@pytest.mark.parametrize("width", [64, 72])
def test_mlp_depths_and_norms_match_oracle(layers, kind, inorm, width, device):
    """MLP (reference layers.py:496-579) at depths 1-3 with every normalisation, on the fused (width 64) and unfused
    (width 72) device paths, eval mode, against the oracle's functional MLP on the CPU."""
    from allset_amd import MLP
    from oracle import allset_oracle as oracle
    torch.manual_seed(layers * 10 + width)
    m = MLP(width, width, width, layers, dropout=0.5, Normalization=kind, InputNorm=inorm).eval()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
    x = torch.randn(517, width)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    xo = x.clone().requires_grad_(True)
    ref = oracle.mlp_forward(sd, "", xo, kind)
    ref.square().sum().backward()
    md = m.to(device)
    xg = x.to(device).requires_grad_(True)
    out = md(xg)
    out.square().sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(xg.grad.cpu(), xo.grad, rtol=1e-3, atol=1e-3 * max(1.0, float(xo.grad.abs().max())))


@pytest.mark.parametrize("aggr", ["add", "mean", "max"])
def test_halfnlhconv_identity_mlps(aggr, device):
    """num_layers = 0 -> f_enc / f_dec are identities: relu -> aggregate -> relu only."""
    import numpy as np
    from allset_amd import HalfNLHconv
    from oracle import allset_oracle as oracle
    rng = np.random.default_rng(4)
    ei = torch.from_numpy(np.stack([rng.integers(0, 40, 300), rng.integers(0, 25, 300)]).astype(np.int64))
    ei[1, -1] = 24
    x = torch.from_numpy(rng.standard_normal((40, 32)).astype(np.float32))
    conv = HalfNLHconv(32, 32, 32, 0, 0.0, "ln", True, attention=False).to(device).eval()
    out = conv(x.to(device), ei.to(device), torch.ones(300, dtype=torch.int64, device=device), aggr)
    ref = oracle.halfnlhconv_forward({}, "", x, ei, torch.ones(300, dtype=torch.int64), aggr, False, 1, "ln")
    torch.testing.assert_close(out.cpu(), ref, rtol=RTOL, atol=ATOL)


def test_pma_conv_training_path_fuses_relu_into_tail(device):
    """HalfNLHconv(attention) in training mode hands SetGNN's relu -> dropout to PMA.tail (ln1's pass).  With p = 0
    the fused training path must equal relu(eval-path raw output), values and gradients."""
    from allset_amd import HalfNLHconv, Incidence
    torch.manual_seed(5)
    n_v, n_e, d = 300, 120, 128
    ei = torch.stack([torch.randint(0, n_v, (2000,)), torch.randint(0, n_e, (2000,))]).to(device)
    ei[1, 0], ei[0, 1] = n_e - 1, n_v - 1
    conv = HalfNLHconv(d, d, d, 2, dropout=0.0, Normalization='ln', InputNorm=True, heads=4, attention=True).to(device)
    inc = Incidence.from_edge_index(ei, n_src=n_v)
    x = torch.randn(n_v, d, device=device)
    G = torch.randn(n_e, d, device=device)
    outs = []
    for training in (False, True):
        conv.train(training)
        conv.zero_grad()
        xx = x.clone().requires_grad_(True)
        y = conv(xx, inc, None, 'add', _post_dropout=0.0)
        (y * G).sum().backward()
        outs.append((y.detach(), xx.grad.clone(), {k: p.grad.clone() for k, p in conv.named_parameters() if p.grad is not None}))
    (y0, gx0, gp0), (y1, gx1, gp1) = outs
    torch.testing.assert_close(y1, y0, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gx1, gx0, rtol=1e-4, atol=1e-5)
    assert gp0.keys() == gp1.keys()
    for k in gp0:
        torch.testing.assert_close(gp1[k], gp0[k], rtol=1e-4, atol=1e-4 * max(1.0, float(gp0[k].abs().max())), msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("heads,hidden", [(8, 128), (2, 64), (1, 128), (16, 128)])
def test_pma_layer_head_counts_match_oracle(heads, hidden, device):
    """PMA with head counts either side of the 4-column auxiliary path (H <= 4: logits ride in the projection kernel,
    H > 4: library GEMM + in-kernel accumulation of the two gradient branches) against the CPU oracle."""
    from oracle import allset_oracle as oracle
    from allset_amd import PMA, Incidence
    torch.manual_seed(heads)
    n_s, n_t, nnz = 700, 300, 5000
    ei = torch.stack([torch.randint(0, n_s, (nnz,)), torch.randint(0, n_t, (nnz,))])
    ei[0, 0], ei[1, 1] = n_s - 1, n_t - 1
    pma = PMA(hidden, hidden, hidden, 2, heads=heads)
    sd = {f"p.{k}": v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in pma.state_dict().items()}
    x = torch.randn(n_s, hidden)
    G = torch.randn(n_t, hidden)
    xr = x.clone().requires_grad_(True)
    ref = oracle.pma_forward(sd, "p.", xr, ei, heads)
    (ref * G).sum().backward()
    pd = pma.to(device)
    xd = x.to(device).requires_grad_(True)
    out = pd(xd, Incidence.from_edge_index(ei.to(device), n_src=n_s, n_dst=n_t))
    (out * G.to(device)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(xr.grad.abs().max())))
    for k, p in pd.named_parameters():
        r = sd[f"p.{k}"].grad
        if r is None:
            continue
        scale = max(1.0, float(r.abs().max()))
        torch.testing.assert_close(p.grad.cpu(), r, rtol=2e-4, atol=2e-4 * scale, msg=lambda m: f"{k}: {m}")


def test_sharded_pma_merge_path_equals_single_rank_path(device, monkeypatch):
    """The cross-rank (m,l,o) merge of sharded PMA -- merge_pack kernel, strided gradient views, global statistics in the
    backward -- run on the GPU with the collectives stubbed to a 1-rank identity must equal the single-rank fast path."""
    import numpy as np
    from allset_amd import HalfNLHconv
    from allset_amd import dist as adist
    rng = np.random.default_rng(9)
    n_v, n_e, d, H = 400, 211, 128, 4
    pairs = sorted({(int(rng.integers(n_v)), int(rng.integers(n_e))) for _ in range(3000)} | {(n_v - 1, 0)})
    pairs = [p for p in pairs if p[0] != 5]                      # vertex 5 has no incidence at all
    ei = torch.tensor(pairs, dtype=torch.int64).t().contiguous().to(device)
    torch.manual_seed(1)
    a = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=H, attention=True).to(device).eval()
    b = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=H, attention=True).to(device).eval()
    x = torch.randn(n_v, d, device=device)
    G = torch.randn(n_v, d, device=device)
    hg = adist.ShardedHypergraph(ei, n_v, n_e, 1, 0).build_incidences()
    res = []
    for merge in (False, True):
        if merge:
            monkeypatch.setattr(adist, "_skip_collective", lambda group=None: False)
            monkeypatch.setattr(adist, "_all_gather_rows", lambda t, group=None: t)
            monkeypatch.setattr(adist, "_reduce_scatter_rows", lambda t, group=None: t)
            monkeypatch.setattr(adist.dist, "all_reduce", lambda *args, **kw: None)
        for p in list(a.parameters()) + list(b.parameters()):
            p.grad = None
        xs = x.clone().requires_grad_(True)
        out = adist.sharded_pma_layer(a, b, xs, hg)
        (out * G).sum().backward()
        res.append((out.detach().clone(), xs.grad.clone(), [p.grad.clone() for p in list(a.parameters()) + list(b.parameters())]))
    (o0, g0, p0), (o1, g1, p1) = res
    torch.testing.assert_close(o1, o0, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g1, g0, rtol=1e-4, atol=1e-5)
    for u, v in zip(p1, p0):
        torch.testing.assert_close(u, v, rtol=1e-4, atol=1e-4 * max(1.0, float(v.abs().max())))


@pytest.fixture
def one_rank_rccl(monkeypatch):
    """A 1-rank nccl (= RCCL) process group with the collectives forced on: the only way to push the real all-to-all /
    all-gather / reduce-scatter calls -- and the asynchronous, chunked exchange -- through RCCL on a 1-GPU box."""
    import socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    monkeypatch.setenv("ALLSET_FORCE_COLLECTIVES", "1")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("attention", [False, True])
@pytest.mark.parametrize("chunks", [1, 4])
def test_colsharded_layer_through_rccl_equals_module_composition(attention, chunks, device, one_rank_rccl):
    """Column-sharded layers on the HIP kernels, exchanges through RCCL (1 rank), plain and with the overlapped chunked
    exchange (asynchronous all-to-alls on RCCL's stream, waits on the compute stream): must equal the plain module
    composition.  A missing stream dependency in the chunked path would show up here as stale rows."""
    import numpy as np
    import torch.nn.functional as F
    from allset_amd import HalfNLHconv, Incidence
    from allset_amd import dist as adist
    rng = np.random.default_rng(13)
    n_v, n_e, d, H = 40_000, 30_001, 128, 4
    v = torch.from_numpy(rng.integers(0, n_v, size=400_000)); e = torch.from_numpy(rng.integers(0, n_e, size=400_000))
    ei = torch.unique(torch.stack([v, e]), dim=1).to(device)
    torch.manual_seed(0)
    a = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=H, attention=attention).to(device).eval()
    b = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=H, attention=attention).to(device).eval()
    hg = adist.ColumnShardedHypergraph(ei, n_v, n_e, 1, 0, chunks=chunks).build_incidences()
    assert hg.n_e_pad % chunks == 0 and hg.n_e_pad >= n_e
    x = torch.randn(hg.n_v_pad, d, device=device)
    G = torch.randn(hg.n_v_pad, d, device=device)
    for rep in range(3):                                  # repeated: buffers are recycled by the caching allocator
        for p in list(a.parameters()) + list(b.parameters()):
            p.grad = None
        xs = x.clone().requires_grad_(True)
        layer = adist.colsharded_pma_layer if attention else adist.colsharded_deepsets_layer
        kw = {} if attention else {"aggr": "add"}
        out = layer(a, b, xs, hg, chunks=chunks, **kw)
        (out * G).sum().backward()
    gs = [p.grad.clone() for p in list(a.parameters()) + list(b.parameters())]
    for p in list(a.parameters()) + list(b.parameters()):
        p.grad = None
    inc = Incidence.from_edge_index(ei, n_src=hg.n_v_pad, n_dst=hg.n_e_pad)
    xr = x.clone().requires_grad_(True)
    ref = F.relu(b(F.relu(a(xr, inc, None, "add")), inc.reversed(n_dst=hg.n_v_pad), None, "add"))
    (ref * G).sum().backward()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xs.grad, xr.grad, rtol=1e-4, atol=1e-4)
    for g, p in zip(gs, list(a.parameters()) + list(b.parameters())):
        torch.testing.assert_close(g, p.grad, rtol=1e-3, atol=1e-3 * max(1.0, float(p.grad.abs().max())))


def test_sharded_pma_merge_path_bf16(device, monkeypatch):
    """The row scheme's cross-rank (m,l,o) merge with bf16 storage (BASELINE configs[4] regime): logits and the merge
    arithmetic are fp32 inside, results return in bf16 -- must track the single-rank bf16 path within bf16 rounding."""
    import numpy as np
    from allset_amd import HalfNLHconv
    from allset_amd import dist as adist
    rng = np.random.default_rng(21)
    n_v, n_e, d, H = 400, 211, 256, 4
    pairs = sorted({(int(rng.integers(n_v)), int(rng.integers(n_e))) for _ in range(3000)} | {(n_v - 1, 0)})
    ei = torch.tensor(pairs, dtype=torch.int64).t().contiguous().to(device)
    torch.manual_seed(1)
    a = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=H, attention=True).to(device).to(torch.bfloat16).eval()
    b = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=H, attention=True).to(device).to(torch.bfloat16).eval()
    x = torch.randn(n_v, d, device=device).to(torch.bfloat16)
    G = torch.randn(n_v, d, device=device).to(torch.bfloat16)
    hg = adist.ShardedHypergraph(ei, n_v, n_e, 1, 0).build_incidences()
    res = []
    for merge in (False, True):
        if merge:
            monkeypatch.setattr(adist, "_skip_collective", lambda group=None: False)
            monkeypatch.setattr(adist, "_all_gather_rows", lambda t, group=None: t)
            monkeypatch.setattr(adist, "_reduce_scatter_rows", lambda t, group=None: t)
            monkeypatch.setattr(adist.dist, "all_reduce", lambda *args, **kw: None)
        xs = x.clone().requires_grad_(True)
        out = adist.sharded_pma_layer(a, b, xs, hg)
        assert out.dtype == torch.bfloat16
        (out.float() * G.float()).sum().backward()
        res.append((out.detach().float(), xs.grad.float()))
    (o0, g0), (o1, g1) = res
    assert float((o1 - o0).abs().mean()) < 0.03 * float(o0.abs().mean()) + 1e-3
    assert float((g1 - g0).abs().mean()) < 0.05 * float(g0.abs().mean()) + 1e-3


@pytest.mark.parametrize("heads,hidden", [(4, 256), (1, 64), (4, 128)])
def test_pma_pooling_and_ln0_as_one_node_equals_the_separate_nodes_bf16(heads, hidden, device, monkeypatch):
    """The same in the bf16 regime (allset_ln_res_bwd_pma_bf16): the statistics are computed from the gradient rows as stored,
    so outputs are identical and gradients agree to bf16 rounding of the parameter sums."""
    from allset_amd import PMA, Incidence, ops
    from allset_amd import functional as AF
    torch.manual_seed(3 + heads)
    n_s, n_t, nnz = 900, 400, 6000
    ei = torch.stack([torch.randint(0, n_s, (nnz,)), torch.randint(0, n_t - 20, (nnz,))]).to(device)
    inc = Incidence.from_edge_index(ei, n_src=n_s, n_dst=n_t)
    pma = PMA(hidden, hidden, hidden, 2, heads=heads).to(device).to(torch.bfloat16)
    x = torch.randn(n_s, hidden, device=device).to(torch.bfloat16)
    G = torch.randn(n_t, hidden, device=device).to(torch.bfloat16)

    def run():
        for p in pma.parameters():
            p.grad = None
        xd = x.clone().requires_grad_(True)
        out = pma(xd, inc)
        out.backward(G)
        return out.detach(), xd.grad, {k: p.grad.clone() for k, p in pma.named_parameters() if p.grad is not None}

    calls = []
    real = ops.pma_bwd_stats
    monkeypatch.setattr(ops, "pma_bwd_stats", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    assert AF.pma_pool_ln0_supported(x, heads)
    o1, g1, p1 = run()
    assert not calls, "the joint node must not launch the separate statistics pass"
    monkeypatch.setattr(AF, "pma_pool_ln0_supported", lambda *a, **k: False)
    o0, g0, p0 = run()
    assert calls
    torch.testing.assert_close(o1, o0, rtol=0, atol=0)
    torch.testing.assert_close(g1.float(), g0.float(), rtol=2e-2, atol=2e-2 * max(1e-3, float(g0.float().abs().max())))
    assert p0.keys() == p1.keys()
    for k in p0:
        torch.testing.assert_close(p1[k].float(), p0[k].float(), rtol=2e-2, atol=2e-2 * max(1e-3, float(p0[k].float().abs().max())),
                                   msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("heads,hidden", [(4, 128), (1, 64), (8, 256), (8, 512), (2, 512), (16, 512)])
def test_pma_pooling_and_ln0_as_one_node_equals_the_separate_nodes(heads, hidden, device, monkeypatch):
    """The joint pooling + ln0 node (backward statistics written by ln0's backward kernel, allset_ln_res_bwd_pma) against
    the two separate nodes (allset_pma_bwd_stats): same outputs, same gradients, targets without incidences included."""
    from allset_amd import PMA, Incidence, ops
    from allset_amd import functional as AF
    torch.manual_seed(7 + heads)
    n_s, n_t, nnz = 900, 400, 6000
    ei = torch.stack([torch.randint(0, n_s, (nnz,)), torch.randint(0, n_t - 20, (nnz,))]).to(device)   # last 20 targets are empty
    inc = Incidence.from_edge_index(ei, n_src=n_s, n_dst=n_t)
    pma = PMA(hidden, hidden, hidden, 2, heads=heads).to(device)
    x = torch.randn(n_s, hidden, device=device)
    G = torch.randn(n_t, hidden, device=device)

    def run():
        for p in pma.parameters():
            p.grad = None
        xd = x.clone().requires_grad_(True)
        out = pma(xd, inc)
        (out * G).sum().backward()
        return out.detach(), xd.grad, {k: p.grad.clone() for k, p in pma.named_parameters() if p.grad is not None}

    calls = []
    real = ops.pma_bwd_stats
    monkeypatch.setattr(ops, "pma_bwd_stats", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    assert AF.pma_pool_ln0_supported(x, heads)
    o1, g1, p1 = run()
    assert not calls, "the joint node must not launch the separate statistics pass"
    monkeypatch.setattr(AF, "pma_pool_ln0_supported", lambda *a, **k: False)
    o0, g0, p0 = run()
    assert calls
    torch.testing.assert_close(o1, o0, rtol=0, atol=0)
    torch.testing.assert_close(g1, g0, rtol=1e-5, atol=1e-5 * max(1.0, float(g0.abs().max())))
    assert p0.keys() == p1.keys()
    for k in p0:
        torch.testing.assert_close(p1[k], p0[k], rtol=1e-5, atol=1e-5 * max(1.0, float(p0[k].abs().max())), msg=lambda m: f"{k}: {m}")
