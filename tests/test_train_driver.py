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


GOLDEN_RAW = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raw")


def _same_dataset(data, exp, prefix, exact_x=True):
    assert data.x.dtype == torch.float32 and data.edge_index.dtype == torch.int64
    if exact_x:
        assert np.array_equal(data.x.numpy(), exp[f"{prefix}_x"])
    else:
        np.testing.assert_allclose(data.x.numpy(), exp[f"{prefix}_x"], rtol=0, atol=1e-6)
    assert np.array_equal(data.y.numpy(), exp[f"{prefix}_y"])
    assert np.array_equal(data.edge_index.numpy(), exp[f"{prefix}_edge_index"])
    assert data.n_x == [int(exp[f"{prefix}_n_x"])] and data.num_hyperedges == [int(exp[f"{prefix}_num_hyperedges"])]


def test_raw_readers_equal_the_reference_loaders_outputs():
    """The LE (.content / .edges), Cornell (node-labels / hyperedges txt) and yelp (csv set) readers against what the REFERENCE's
    own load_LE_dataset / load_cornell_dataset / load_yelp_dataset returned for the same files (tests/golden/raw/expected.npz,
    written by oracle/gen_raw_fixture.py from the imported reference; load_other_datasets.py:32,293,198)."""
    from allset_amd.train import load_cornell_dataset, load_le_dataset, load_yelp_dataset
    exp = np.load(os.path.join(GOLDEN_RAW, "expected.npz"))
    _same_dataset(load_le_dataset(GOLDEN_RAW, "toyLE"), exp, "le")
    seed = int(exp["cornell_seed"])
    np.random.seed(seed)                                   # the reference draws the noise from numpy's global generator
    _same_dataset(load_cornell_dataset(GOLDEN_RAW, "toy-trips", feature_noise=0.6), exp, "cornell")
    np.random.seed(seed)
    _same_dataset(load_cornell_dataset(GOLDEN_RAW, "toy-trips", feature_noise=1.0, feature_dim=100), exp, "cornell100")
    _same_dataset(load_yelp_dataset(os.path.join(GOLDEN_RAW, "yelp")), exp, "yelp")
    # an explicit generator gives other noise, same everything else
    d = load_cornell_dataset(GOLDEN_RAW, "toy-trips", feature_noise=0.6, rng=np.random.default_rng(0))
    assert np.array_equal(d.edge_index.numpy(), exp["cornell_edge_index"]) and not np.array_equal(d.x.numpy(), exp["cornell_x"])


def test_raw_readers_through_the_driver(tmp_path):
    """load_data picks the reader by dataset name as reference convert_datasets_to_pygDataset.py:122-163 does, shifts the labels
    of the Cornell / yelp sets to start at 0 (reference train.py:329-332) and feeds the same preprocessing chain."""
    import shutil
    from allset_amd.train import build_parser, load_data, preprocess
    exp = np.load(os.path.join(GOLDEN_RAW, "expected.npz"))
    shutil.copytree(os.path.join(GOLDEN_RAW, "toy-trips"), tmp_path / "walmart-trips")
    for f in os.listdir(tmp_path / "walmart-trips"):
        os.rename(tmp_path / "walmart-trips" / f, tmp_path / "walmart-trips" / f.replace("toy-trips", "walmart-trips"))
    args = build_parser().parse_args(['--dname', 'walmart-trips-100', '--raw_data_dir', str(tmp_path), '--feature_noise', '1'])
    np.random.seed(int(exp["cornell_seed"]))
    data = load_data(args)
    assert np.array_equal(data.x.numpy(), exp["cornell100_x"]) and int(data.y.min()) == 0
    assert np.array_equal(data.y.numpy(), exp["cornell100_y"] - exp["cornell100_y"].min())
    assert args.num_features == 100 and args.num_classes == len(np.unique(exp["cornell100_y"]))
    d2 = preprocess(args, data)
    n_v = int(exp["cornell100_n_x"])
    assert int(d2.edge_index[0].max()) < n_v and int(d2.edge_index[1].min()) == n_v
    args = build_parser().parse_args(['--dname', 'toyLE', '--raw_data_dir', GOLDEN_RAW])
    data = load_data(args)
    assert np.array_equal(data.edge_index.numpy(), exp["le_edge_index"]) and args.num_features == 6
    args = build_parser().parse_args(['--dname', 'yelp', '--raw_data_dir', GOLDEN_RAW])
    data = load_data(args)
    assert np.array_equal(data.edge_index.numpy(), exp["yelp_edge_index"]) and int(data.y.min()) == 0


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
