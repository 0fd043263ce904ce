"""CPU: the oracle (oracle/allset_oracle.py) against the committed golden vectors, which were produced by
the real reference (oracle/gen_golden.py).  Where /root/reference exists (build container) the oracle
is additionally compared with the live reference."""
import numpy as np
import pytest
import torch

import cases
import util


@pytest.mark.parametrize("name", cases.ALL_CASES)
def test_case_inputs_are_reproducible(name):
    """The seeded numpy generators must recreate exactly the inputs the fixtures were made from."""
    case, g = cases.build_case(name), util.load_golden(name)
    assert cases.checksum(case["x"]) == int(g["chk_x"])
    assert cases.checksum(case["edge_index"]) == int(g["chk_edge_index"])
    assert cases.checksum(case["norm"]) == int(g["chk_norm"])
    if not case["big"]:
        np.testing.assert_array_equal(case["x"], g["in_x"])
        np.testing.assert_array_equal(case["edge_index"], g["in_edge_index"])


@pytest.mark.parametrize("name", cases.ALL_CASES)
def test_oracle_matches_golden(name):
    case, g = cases.build_case(name), util.load_golden(name)
    res = util.run_oracle(case, util.state_dict_for(case, g))
    # same ops as the reference; 2e-5 leaves room for a different host BLAS/threading on the GPU box
    util.assert_matches_golden(res, g, case["big"], rtol=2e-5, atol=2e-5)


def test_q1_trailing_isolated_vertex_disappears():
    """without self loops the isolated last vertex is absent from the E->V output."""
    g = util.load_golden("doc_noself_ds_add")
    assert int(g["n_rows_logits"]) == 4 and g["in_x"].shape[0] == 5
    assert int(util.load_golden("doc_self_ds_add")["n_rows_logits"]) == 5


def test_attention_weights_fixture_is_a_softmax():
    case, g = cases.build_case("rand50_pma_h4"), util.load_golden("rand50_pma_h4")
    p = torch.from_numpy(g["attn_v2e0"])
    dst = torch.from_numpy(case["edge_index"][1] - case["edge_index"][1].min())
    sums = torch.zeros(int(dst.max()) + 1, p.shape[1]).index_add_(0, dst, p)
    torch.testing.assert_close(sums, torch.ones_like(sums), rtol=1e-5, atol=1e-5)


def test_oracle_matches_live_reference_when_present():
    from oracle import ref_shim
    if not ref_shim.available():
        pytest.skip("/root/reference not present (GPU box)")
    _, ref_models = ref_shim.import_reference()
    from types import SimpleNamespace
    for name in ("rand50_pma_h4", "edge_ds_max", "rand50_ds_mean_wnorm"):
        case, g = cases.build_case(name), util.load_golden(name)
        sd = util.state_dict_for(case, g)
        model = ref_models.SetGNN(case["args"])
        model.load_state_dict(sd)
        model.eval()
        with torch.no_grad():
            ref = model(SimpleNamespace(x=torch.from_numpy(case["x"]), edge_index=torch.from_numpy(case["edge_index"]).clone(),
                                        norm=torch.from_numpy(case["norm"])))
        got = util.run_oracle(case, sd)["logits"]
        assert float((ref - got).abs().max()) <= 1e-6
