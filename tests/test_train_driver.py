"""The train.py-compatible driver: CLI surface + on-disk format reader (CPU), and an end-to-end training run on the
device (-m gpu)."""
import os
import pickle

import numpy as np
import pytest
import scipy.sparse as sp
import torch


def test_cli_defaults_and_quirks_match_the_reference():
    """reference train.py:221-289 (SURVEY section 5 'Config / flags')."""
    from allset_amd.train import build_parser
    a = build_parser().parse_args([])
    assert (a.method, a.epochs, a.runs, a.lr, a.wd, a.dropout) == ('AllSetTransformer', 500, 20, 0.001, 0.0, 0.5)
    assert (a.All_num_layers, a.MLP_num_layers, a.MLP_hidden, a.Classifier_num_layers, a.Classifier_hidden) == (2, 2, 64, 2, 64)
    assert (a.aggregate, a.normtype, a.normalization, a.heads) == ('mean', 'all_one', 'ln', 1)
    assert a.PMA is True and a.add_self_loop is True and a.GPR is False and a.LearnMask is False and a.exclude_self is False
    b = build_parser().parse_args(['--add_self_loop', '--PMA', '--GPR'])
    assert b.add_self_loop is False        # store_false: passing the flag DISABLES self loops
    assert b.PMA is True                   # store_true with default True: cannot be disabled from the CLI
    assert b.GPR is False                  # store_false with default False: cannot be enabled from the CLI
    with pytest.raises(SystemExit):
        build_parser().parse_args(['--aggregate', 'max'])


def test_parse_method_sets_deepsets_flags():
    from allset_amd.train import build_parser, parse_method
    a = build_parser().parse_args(['--method', 'AllDeepSets'])
    a.num_features, a.num_classes = 8, 3
    m = parse_method(a, None)
    assert a.PMA is False and a.aggregate == 'add' and m.V2EConvs[0].attention is False
    a = build_parser().parse_args(['--method', 'HGNN'])
    with pytest.raises(ValueError):
        parse_method(a, None)


GOLDEN_DATAPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "datapt")


@pytest.mark.parametrize("name", ["pyg163_legacy.pt", "pyg163_zip.pt", "pyg2_store.pt", "pyg163_no_counts.pt"])
def test_processed_data_pt_reader(name):
    """The reference's processed dataset file (convert_datasets_to_pygDataset.py:170-175: (Data, slices) written by
    InMemoryDataset.collate) read WITHOUT torch_geometric, in both torch serialisation formats and both PyG attribute
    layouts; fixtures and their provenance: oracle/gen_datapt_fixture.py.  The file without n_x / num_hyperedges takes
    the fall-backs of reference train.py:333-339."""
    import importlib.machinery
    from allset_amd.train import build_parser, load_data, load_pyg_processed, preprocess
    assert importlib.machinery.PathFinder.find_spec("torch_geometric") is None      # the real package is not installed
    exp = np.load(os.path.join(GOLDEN_DATAPT, "expected.npz"))
    data = load_pyg_processed(os.path.join(GOLDEN_DATAPT, name))
    assert np.array_equal(data.x.numpy(), exp["x"]) and data.x.dtype == torch.float32
    assert np.array_equal(data.edge_index.numpy(), exp["edge_index"]) and data.edge_index.dtype == torch.int64
    assert np.array_equal(data.y.numpy(), exp["y"])
    assert data.n_x == [int(exp["n_x"])] and data.num_hyperedges == [int(exp["num_hyperedges"])]
    if name != "pyg163_no_counts.pt":
        assert abs(data.train_percent - 0.025) < 1e-9
    # and through the driver: same preprocessing chain as the pickle reader's output
    args = build_parser().parse_args(['--dname', 'cora', '--processed_data', os.path.join(GOLDEN_DATAPT, name)])
    d2 = preprocess(args, load_data(args))
    assert args.num_features == exp["x"].shape[1] and args.num_classes == len(np.unique(exp["y"]))
    n_v = int(exp["n_x"])
    v, e = d2.edge_index
    assert v.tolist() == sorted(v.tolist()) and int(v.max()) < n_v and int(e.min()) == n_v
    assert d2.edge_index.shape[1] == exp["edge_index"].shape[1] // 2 + n_v       # V->E half + one self loop per vertex


def test_processed_data_pt_reader_rejects_other_payloads(tmp_path):
    from allset_amd.train import load_pyg_processed
    torch.save({"x": torch.zeros(2, 2)}, tmp_path / "data.pt")
    with pytest.raises(ValueError):
        load_pyg_processed(str(tmp_path))                     # a directory resolves to <dir>/data.pt


def test_hypergcn_pickle_reader_and_preprocessing(tmp_path):
    """HyperGCN on-disk format (reference load_other_datasets.py:121-196): features scipy-sparse, labels list,
    hypergraph dict{he: [nodes]} -> [V|E;E|V] coalesced block list -> ExtractV2E/Add_Self_Loops/norm."""
    from allset_amd.train import build_parser, load_data, preprocess
    d = tmp_path / "toy"
    os.makedirs(d)
    feats = sp.csr_matrix(np.eye(5, 7, dtype=np.float32))
    pickle.dump(feats, open(d / "features.pickle", "wb"))
    pickle.dump([0, 1, 0, 1, 2], open(d / "labels.pickle", "wb"))
    pickle.dump({"a": [0, 1, 2], "b": [1, 2, 3], "c": [1, 1, 3]}, open(d / "hypergraph.pickle", "wb"))
    args = build_parser().parse_args(['--dname', 'toy', '--raw_data_dir', str(tmp_path)])
    data = load_data(args)
    assert args.num_features == 7 and args.num_classes == 3 and data.n_x == [5] and data.num_hyperedges == [3]
    ei = data.edge_index
    assert ei.shape[1] == 2 * 8                       # duplicate (1,c) coalesced away; both blocks present
    assert bool(((ei[0] < 5) == (ei[1] >= 5)).all())
    data = preprocess(args, data)
    v, e = data.edge_index
    assert v.tolist() == sorted(v.tolist()) and int(e.min()) == 5
    assert data.edge_index.shape[1] == 8 + 5 and int(data.totedges) == 3 + 5    # one self-loop hyperedge per vertex
    assert data.norm.dtype == torch.int64 and bool((data.norm == 1).all())


def test_synthetic_dataset_is_learnable_structure():
    from allset_amd.train import synthetic_dataset
    d = synthetic_dataset(n_v=500, n_e=300, seed=1)
    assert d.x.shape == (500, 64) and int(d.edge_index.max()) == 500 + 300 - 1
    v, e = d.edge_index[:, d.edge_index[0] < 500]
    same = (d.y[v] == torch.mode(d.y[v].view(-1, 1).expand(-1, 1), 0).values).float()   # smoke: tensors line up
    assert same.numel() == v.numel()


@pytest.mark.gpu
@pytest.mark.parametrize("hip_graph", [0, 1, -1])
@pytest.mark.parametrize("method", ["AllSetTransformer", "AllDeepSets"])
def test_training_run_improves_accuracy(method, hip_graph, device, tmp_path):
    from allset_amd.train import build_parser, run
    args = build_parser().parse_args(['--method', method, '--dname', 'synthetic', '--epochs', '40', '--runs', '2',
                                      '--All_num_layers', '1', '--MLP_hidden', '64', '--heads', '4', '--lr', '0.01',
                                      '--seed', '3', '--res_root', str(tmp_path), '--hip_graph', str(hip_graph)])
    res = run(args)
    assert float(res['best_test'].mean()) > 60.0          # 5 classes: chance = 20 %
    line = open(res['csv']).read().strip().split(',')
    assert line[0] == f'{method}_0.01_0.0_4' and '±' in line[1] and line[3] == str(res['num_params'])
