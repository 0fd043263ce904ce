"""allset_amd.preprocessing (vectorised ExtractV2E / Add_Self_Loops / norm_contruction / expand_edge_index)
against golden vectors produced by the reference's own preprocessing.py (oracle/gen_golden.py).  Index work:
bit-exact, compared as lexicographically sorted edge lists (the reference's own sorts are unstable).  Runs on the
host here and, under -m gpu, on the device."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import cases
import util


def canon(ei: torch.Tensor) -> np.ndarray:
    a = ei.cpu().numpy()
    return a[:, np.lexsort((a[1], a[0]))]


def run_all(name, dev):
    from allset_amd import preprocessing as P
    c, g = cases.build_preproc_case(name), util.load_golden(name)
    np.testing.assert_array_equal(c["edge_index"], g["in_edge_index"])
    data = SimpleNamespace(edge_index=torch.from_numpy(c["edge_index"]).to(dev), n_x=[c["n_v"]], num_hyperedges=[c["n_e"]])
    data = P.ExtractV2E(data)
    np.testing.assert_array_equal(canon(data.edge_index), g["extract"])
    assert bool((data.edge_index[0][1:] >= data.edge_index[0][:-1]).all())            # sorted by vertex id
    data = P.Add_Self_Loops(data)
    np.testing.assert_array_equal(canon(data.edge_index), g["selfloop"])
    assert int(data.totedges) == int(g["totedges"])
    assert bool((data.edge_index[0][1:] >= data.edge_index[0][:-1]).all())
    sl = torch.from_numpy(g["selfloop"]).to(dev)
    d1 = P.norm_contruction(SimpleNamespace(edge_index=sl), option="all_one")
    assert d1.norm.dtype == torch.int64                                                # the reference's int64 ones (Q3)
    np.testing.assert_array_equal(d1.norm.cpu().numpy(), g["norm_all_one"])
    d2 = P.norm_contruction(SimpleNamespace(edge_index=sl), option="deg_half_sym")
    np.testing.assert_allclose(d2.norm.cpu().numpy(), g["norm_deg_half_sym"], rtol=1e-6, atol=1e-7)
    for th in (0, 4):
        de = SimpleNamespace(edge_index=sl.clone(), n_x=torch.tensor([c["n_v"]]), num_hyperedges=[c["n_e"]],
                             totedges=int(g["totedges"]))
        de = P.expand_edge_index(de, edge_th=th)
        np.testing.assert_array_equal(canon(de.edge_index), g[f"expand_th{th}"])


@pytest.mark.parametrize("name", cases.PREPROC_CASES)
def test_preprocessing_host(name):
    run_all(name, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", cases.PREPROC_CASES)
def test_preprocessing_device(name, device):
    run_all(name, device)


@pytest.mark.gpu
def test_preprocessing_feeds_setgnn_at_scale(device):
    """1M-vertex edge list through ExtractV2E -> Add_Self_Loops -> norm_contruction -> SetGNN on the device
    (the reference's Python loops take minutes here); checks structural invariants."""
    from allset_amd import preprocessing as P, SetGNN
    from allset_amd.synthetic import random_hypergraph
    n = 1_000_000
    hg = random_hypergraph(n, n // 4, 8, seed=5, device=device, e_base=n)
    v, e = hg.edge_index[0], hg.edge_index[1]
    e[-1] = n + n // 4 - 1
    block = torch.cat([torch.stack([v, e]), torch.stack([e, v])], dim=1)
    data = SimpleNamespace(edge_index=block, n_x=[n], num_hyperedges=[n // 4])
    data = P.norm_contruction(P.Add_Self_Loops(P.ExtractV2E(data)), option="all_one")
    assert data.edge_index.shape[1] == hg.nnz + n                 # no size-1 hyperedges before: one loop per vertex
    assert int(data.totedges) == n // 4 + n
    args = cases.make_args("ds_add", 32, 32, 4)
    model = SetGNN(args).to(device).eval()
    data.x = torch.randn(n, 32, device=device)
    with torch.no_grad():
        out = model(data)
    assert tuple(out.shape) == (n, 4) and bool(torch.isfinite(out).all())
