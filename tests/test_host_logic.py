"""CPU: host-side logic that needs no kernel -- module surface / state_dict layout / init / alpha folding."""
import numpy as np
import pytest
import torch

import cases
import util


@pytest.mark.parametrize("name", ["rand50_ds_add", "rand50_pma_h4", "rand50_ds_add_bn", "rand50_pma_h4_L2",
                                  "rand50_ds_add_wnorm_mask", "citeseer_pma_h4"])
def test_state_dict_layout_equals_reference(name):
    """Keys, order and shapes of allset_amd.SetGNN.state_dict() == the reference's (spec captured by
    oracle/gen_golden.py from the real models.SetGNN)."""
    from allset_amd import SetGNN
    case, g = cases.build_case(name), util.load_golden(name)
    norm = torch.from_numpy(case["norm"]).float() if case["args"].LearnMask else None
    model = SetGNN(case["args"], norm=norm)
    got = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert got == util.golden_spec(g)


def test_setgnn_zero_layers_is_a_classifier_only():
    from allset_amd import SetGNN
    m = SetGNN(cases.make_args("ds_add", 8, 16, 3, All_num_layers=0))
    assert len(m.V2EConvs) == 0 and m.classifier.lins[0].in_features == 8


def test_pma_reset_keeps_linear_biases():
    """reset_parameters() re-draws lin_K/lin_V weights but never their biases."""
    from allset_amd import PMA
    torch.manual_seed(0)
    p = PMA(16, 32, 32, 2, heads=4)
    b0, w0 = p.lin_K.bias.detach().clone(), p.lin_K.weight.detach().clone()
    p.reset_parameters()
    assert torch.equal(p.lin_K.bias, b0) and not torch.equal(p.lin_K.weight, w0)
    bound = (6.0 / (16 + 32)) ** 0.5
    assert float(p.lin_K.weight.abs().max()) <= bound
    assert p.hidden == 8 and tuple(p.att_r.shape) == (1, 4, 8) and p.dropout == 0.0 and p.bias is None


def test_alpha_fold_equals_unfolded_projection():
    """alpha = <lin_K(x), att_r> computed through the folded [in,H] mat-vec."""
    from allset_amd import PMA
    torch.manual_seed(1)
    p = PMA(24, 32, 32, 2, heads=4).double()
    x = torch.randn(40, 24, dtype=torch.float64, requires_grad=True)
    p.fold_alpha = True
    a1 = p._logits(x)
    g1 = torch.autograd.grad(a1.square().sum(), [x, p.att_r, p.lin_K.weight, p.lin_K.bias])
    p.fold_alpha = False
    a2 = p._logits(x)
    g2 = torch.autograd.grad(a2.square().sum(), [x, p.att_r, p.lin_K.weight, p.lin_K.bias])
    torch.testing.assert_close(a1, a2, rtol=1e-12, atol=1e-12)
    for u, v in zip(g1, g2):
        torch.testing.assert_close(u, v, rtol=1e-10, atol=1e-10)


def test_mlp_matches_oracle_on_cpu():
    """The dense tail is plain torch on both sides; check the module wiring (norm slots, ReLU order)."""
    from allset_amd import MLP
    from oracle import allset_oracle as oracle
    torch.manual_seed(2)
    for kind, inorm, layers in (("ln", True, 2), ("ln", False, 3), ("bn", True, 2), ("None", False, 1)):
        m = MLP(12, 20, 7, layers, dropout=0.5, Normalization=kind, InputNorm=inorm).eval()
        x = torch.randn(33, 12)
        sd = {k: v for k, v in m.state_dict().items()}
        torch.testing.assert_close(m(x), oracle.mlp_forward(sd, "", x, kind), rtol=1e-6, atol=1e-6)


def test_halfnlhconv_zero_layers_has_identity_mlps():
    from allset_amd import HalfNLHconv
    h = HalfNLHconv(8, 8, 8, 0, 0.0, "ln", True, attention=False)
    assert isinstance(h.f_enc, torch.nn.Identity) and len(list(h.parameters())) == 0


def test_unknown_aggr_is_rejected_before_touching_the_device():
    from allset_amd import functional as AF
    with pytest.raises(ValueError):
        AF.deepsets_aggregate(torch.zeros(2, 2), None, None, "median")


def test_case_generators_are_deterministic():
    a, b = cases.build_case("edge_pma_h4"), cases.build_case("edge_pma_h4")
    np.testing.assert_array_equal(a["edge_index"], b["edge_index"])
    ei = a["edge_index"]
    assert (np.diff(ei[0]) >= 0).all()              # sorted by vertex id, like the reference's preprocessing
    assert ei[1].min() == 5000                      # hyperedge ids start at n_V
    sizes = np.bincount(ei[1] - 5000)
    assert sizes.max() == 4096 and sizes[2] == 0 and sizes[0] == 1


def test_fused_adam_falls_back_to_torch_adam_for_tensors_it_does_not_take():
    """allset_amd.optim.FusedAdam on CPU parameters (and, on the device, bf16 ones) runs torch's functional Adam on the same state."""
    import torch
    from allset_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    pa = [torch.randn(5, 3, generator=g).requires_grad_(True), torch.randn(7, generator=g).requires_grad_(True)]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa, ob = FusedAdam(pa, lr=1e-2, weight_decay=0.01), torch.optim.Adam(pb, lr=1e-2, weight_decay=0.01)
    for _ in range(4):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)


def test_size_split_compacts_the_short_rows_and_lists_the_long_ones():
    """ops.size_split (host logic of the size-split dispatch): long-row list, short-row ids, compacted rowptr / col in row order;
    None when nothing is long or too many rows are."""
    import numpy as np
    import torch
    from allset_amd import ops
    rng = np.random.default_rng(0)
    deg = rng.integers(0, 6, size=200)
    deg[[5, 77, 150]] = [40, 33, 400]                          # three long rows (> 32)
    rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(deg)]).astype(np.int32))
    col = torch.arange(int(rowptr[-1]), dtype=torch.int32)     # col = incidence position: makes the compaction checkable
    sp = ops.size_split(rowptr, col, 200, int(deg.max()))
    assert sp is not None and sp.long_ids.tolist() == [5, 77, 150] and sp.threshold == 32
    short = [r for r in range(200) if r not in (5, 77, 150)]
    assert sp.short_ids.tolist() == short
    assert sp.rowptr_short.tolist() == np.concatenate([[0], np.cumsum(deg[short])]).tolist()
    expect = np.concatenate([np.arange(rowptr[r], rowptr[r + 1]) for r in short])
    assert sp.col_short.tolist() == expect.tolist()
    assert ops.size_split(rowptr, col, 200, 32) is None                                  # nothing longer than the threshold
    many = torch.from_numpy(np.concatenate([[0], np.cumsum(np.full(16, 40))]).astype(np.int32))
    assert ops.size_split(many, torch.zeros(640, dtype=torch.int32), 16, 40) is None     # every row is long: not a skewed CSR


def test_constant_features_scope_and_the_pre_dropout_contract_on_the_host():
    """``dense.constant_features()`` nests and unwinds on exceptions; on host tensors no conv claims the input dropout (the sparse /
    folded raw-feature kernels are device paths), ``SetGNN`` then applies it itself, and a conv handed ``_pre_dropout`` it did not
    claim refuses it (reference models.py:473 stays one dropout, never two, never none)."""
    import pytest
    from allset_amd import SetGNN, dense
    from allset_amd.layers import HalfNLHconv
    assert not dense.constant_features_active()
    with dense.constant_features():
        with dense.constant_features():
            assert dense.constant_features_active()
        assert dense.constant_features_active()
    assert not dense.constant_features_active()
    with pytest.raises(RuntimeError):
        with dense.constant_features():
            raise RuntimeError("boom")
    assert not dense.constant_features_active()
    x = torch.zeros(10, 300)
    x[:, 0] = 1.0
    for attention in (True, False):
        conv = HalfNLHconv(300, 64, 64, 2, 0.5, "ln", True, heads=4, attention=attention)
        assert not conv.takes_pre_dropout(x)
        ei = torch.stack([torch.arange(10), torch.arange(10) % 3])
        with pytest.raises(ValueError):
            conv(x, ei, torch.ones(10), "add", _pre_dropout=0.2)
    import cases
    model = SetGNN(cases.build_case("rand50_pma_h4")["args"])
    assert model.V2EConvs[0].prop._raw_input and not model.E2VConvs[0].prop._raw_input      # the conv that consumes data.x, marked by the model
    ds = SetGNN(cases.build_case("rand50_ds_add")["args"])
    assert ds.V2EConvs[0].f_enc._raw_input and not ds.V2EConvs[0].f_dec._raw_input


def test_oracle_relu_margin_sees_a_kink_that_the_perturbation_probe_cannot():
    """tests/util.py::oracle_relu_margin (the direct a-priori criterion of the randomized parameter-gradient sweep): the smallest
    |relu input| / row scale of the float64 oracle.  A bias moved so that ONE pre-activation of the classifier's hidden layer sits 1e-9
    from zero is reported as such, whatever else the network does; kink-free biases push the conv stacks' margin up by orders."""
    import cases
    import util
    from allset_amd import SetGNN
    from oracle import allset_oracle as oracle
    rng = np.random.default_rng(3)
    n_v, f, k = 30, 12, 4
    ei = cases.random_hypergraph(rng, n_v, 11, 90, True)
    x = rng.standard_normal((n_v, f)).astype(np.float32)
    nrm = np.ones(ei.shape[1], dtype=np.int64)
    args = cases.make_args("ds_add", f, 64, k, Classifier_num_layers=2)
    torch.manual_seed(3)
    model = SetGNN(args)
    model.reset_parameters()
    sd = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    m0 = util.oracle_relu_margin(sd, args, x, ei, nrm)
    assert 0.0 < m0 < 1e-2                                      # ~10^4 relu inputs: some within a percent of a row's scale of zero
    sd_free = cases.kinkfree_biases({kk: v.clone() for kk, v in sd.items()})
    # the classifier's hidden relu is the one site kinkfree_biases leaves alone: measure the conv stacks by moving it out of the way too
    sd_free["classifier.lins.0.bias"] += 50.0
    assert util.oracle_relu_margin(sd_free, args, x, ei, nrm) > 30.0 * m0
    # one classifier pre-activation placed 1e-9 above zero
    real, seen = torch.nn.functional.relu, []
    torch.nn.functional.relu = lambda t, *a, **kw: (seen.append(t.detach()), real(t, *a, **kw))[1]
    try:
        sd64 = {kk: (v.double() if v.is_floating_point() else v) for kk, v in sd_free.items()}
        oracle.setgnn_forward(sd64, args, torch.from_numpy(x).double(), torch.from_numpy(ei), torch.from_numpy(nrm))
    finally:
        torch.nn.functional.relu = real
    pre = [t for t in seen if t.shape[-1] == args.Classifier_hidden][-1]
    sd_kink = {kk: v.clone() for kk, v in sd_free.items()}
    sd_kink["classifier.lins.0.bias"] = sd_kink["classifier.lins.0.bias"].double()
    sd_kink["classifier.lins.0.bias"][5] -= pre[7, 5] - 1e-9
    m = util.oracle_relu_margin(sd_kink, args, x, ei, nrm)
    assert m < 1e-9
