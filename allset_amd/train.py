#!/usr/bin/env python
"""``train.py``-compatible driver for AllSetTransformer / AllDeepSets on MI355X.

Mirrors the surface of reference ``src/train.py`` for the two AllSet methods: same command-line flags, defaults
and quirks (``--add_self_loop`` is ``store_false`` with default True, ``--PMA`` cannot be disabled except through
``--method AllDeepSets``, ... : train.py:221-289), same preprocessing branch (:344-353), ``parse_method`` (:28-42),
full-batch Adam loop with ``log_softmax`` + ``NLLLoss`` on the train split and an ``evaluate`` per epoch (:458-499),
``Logger`` statistics (:106-150) and the CSV line appended under ``hyperparameter_tunning/`` (:504-520).

Datasets: the reference's raw-data archive is not part of its repository.  This driver reads the
HyperGCN on-disk format the reference's ``load_citation_dataset`` consumes (``features.pickle`` scipy-sparse,
``labels.pickle`` list, ``hypergraph.pickle`` dict{hyperedge: [nodes]}; load_other_datasets.py:130-163) when
``--raw_data_dir`` points at it, and otherwise generates ``--dname synthetic`` (a planted-partition hypergraph with
noisy class-indicator features).  The baseline methods of the reference (HGNN, HCHA, HyperGCN, ...) are out of
scope and rejected.

    python -m allset_amd.train --method AllSetTransformer --dname synthetic --epochs 50 --runs 2 --heads 4 \\
        --MLP_hidden 128 --All_num_layers 1
"""
from __future__ import annotations

import argparse
import os
import os.path as osp
import pickle
import time
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dense
from ._lib import AllSetHipError
from .models import SetGNN
from .preprocessing import Add_Self_Loops, ExtractV2E, expand_edge_index, norm_contruction

ALLSET_METHODS = ('AllSetTransformer', 'AllDeepSets')


# --------------------------------------------------------------------------------------------------
# data
# --------------------------------------------------------------------------------------------------

class HypergraphData(SimpleNamespace):
    """The attributes train.py reads from its PyG ``Data`` object: x, edge_index, y, n_x, num_hyperedges, norm."""

    def to(self, device):
        for k, v in list(vars(self).items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self


def block_edge_list(node_ids: np.ndarray, he_ids: np.ndarray, num_nodes: int) -> torch.Tensor:
    """[[V | E], [E | V]] with hyperedge ids offset by ``num_nodes``, sorted and de-duplicated -- what the
    reference's loaders produce (load_other_datasets.py:165-181, ``coalesce``)."""
    v = np.asarray(node_ids, dtype=np.int64)
    e = np.asarray(he_ids, dtype=np.int64) + num_nodes
    ei = np.concatenate([np.stack([v, e]), np.stack([e, v])], axis=1)
    span = int(ei.max()) + 1
    key = np.unique(ei[0] * span + ei[1])
    return torch.from_numpy(np.stack([key // span, key % span]))


def load_hypergcn_dataset(path: str, dataset: str) -> HypergraphData:
    """Read ``<path>/<dataset>/{features,labels,hypergraph}.pickle`` (reference load_other_datasets.py:121-196)."""
    with open(osp.join(path, dataset, 'features.pickle'), 'rb') as f:
        features = pickle.load(f)
    features = np.asarray(features.todense() if hasattr(features, 'todense') else features, dtype=np.float32)
    with open(osp.join(path, dataset, 'labels.pickle'), 'rb') as f:
        labels = np.asarray(pickle.load(f), dtype=np.int64)
    with open(osp.join(path, dataset, 'hypergraph.pickle'), 'rb') as f:
        hypergraph = pickle.load(f)
    num_nodes = features.shape[0]
    assert num_nodes == len(labels)
    nodes, hes = [], []
    for idx, he in enumerate(hypergraph.keys()):
        members = list(hypergraph[he])
        nodes += members
        hes += [idx] * len(members)
    return HypergraphData(x=torch.from_numpy(features), y=torch.from_numpy(labels),
                          edge_index=block_edge_list(np.array(nodes), np.array(hes), num_nodes),
                          n_x=[num_nodes], num_hyperedges=[len(hypergraph)])


def load_le_dataset(path: str, dataset: str = "ModelNet40", train_percent: float = 0.025) -> HypergraphData:
    """``<path>/<dataset>/<dataset>.content`` (one line per id: id, features..., label) and ``<dataset>.edges`` (pairs
    ``node_id he_id``) -- the "LE" format of ModelNet40 / NTU2012 / Mushroom / zoo / 20newsW100 (reference
    load_other_datasets.py:32-119).  Same conventions: ids are remapped through the order of the .content file, hyperedge ids must
    follow the node ids without a gap, ``x`` / ``y`` keep the node rows only."""
    content = np.genfromtxt(osp.join(path, dataset, f'{dataset}.content'), dtype=np.dtype(str))
    if content.ndim == 1:
        content = content[None, :]
    features = content[:, 1:-1].astype(np.float32)
    labels = content[:, -1].astype(float).astype(np.int64)
    idx_map = {j: i for i, j in enumerate(content[:, 0].astype(np.int32))}
    raw = np.genfromtxt(osp.join(path, dataset, f'{dataset}.edges'), dtype=np.int32)
    if raw.ndim == 1:
        raw = raw[None, :]
    pairs = np.array([idx_map[int(v)] for v in raw.flatten()], dtype=np.int64).reshape(raw.shape).T      # [2, nnz]: nodes | hyperedges
    if int(pairs[0].max()) != int(pairs[1].min()) - 1:
        raise ValueError(f"{dataset}: hyperedge ids must start right behind the node ids (reference load_other_datasets.py:69)")
    if len(np.unique(pairs)) != int(pairs.max()) + 1:
        raise ValueError(f"{dataset}: node / hyperedge ids are not consecutive (reference load_other_datasets.py:72)")
    num_nodes = int(pairs[0].max()) + 1
    num_he = int(pairs[1].max()) - num_nodes + 1
    return HypergraphData(x=torch.from_numpy(features[:num_nodes].copy()), y=torch.from_numpy(labels[:num_nodes].copy()),
                          edge_index=block_edge_list(pairs[0], pairs[1] - num_nodes, num_nodes),
                          n_x=[num_nodes], num_hyperedges=[num_he], train_percent=train_percent)


def load_cornell_dataset(path: str, dataset: str = "amazon", feature_noise: float = 0.1, feature_dim: Optional[int] = None,
                         train_percent: float = 0.025, rng: Optional[np.random.Generator] = None) -> HypergraphData:
    """``<path>/<dataset>/node-labels-<dataset>.txt`` (one label per line, from 1) and ``hyperedges-<dataset>.txt`` (one hyperedge
    per line, comma-separated node ids) -- walmart-trips / house-committees / amazon-reviews (reference
    load_other_datasets.py:293-391).  Features are the one-hot label (zero-padded to ``feature_dim``) plus N(0, feature_noise)
    noise; the reference draws it from numpy's GLOBAL generator (unseeded: README.md:60), here ``rng`` (default: that same
    global state, so a caller that seeds numpy gets the reference's very numbers)."""
    labels = np.loadtxt(osp.join(path, dataset, f'node-labels-{dataset}.txt'), dtype=np.int64, ndmin=1)
    num_nodes = labels.shape[0]
    features = np.zeros((num_nodes, int(labels.max())))
    features[np.arange(num_nodes), labels - 1] = 1
    if feature_dim is not None:
        features = np.hstack((features, np.zeros((num_nodes, feature_dim - features.shape[1]), dtype=features.dtype)))
    features = (rng.normal(features, feature_noise, features.shape) if rng is not None
                else np.random.normal(features, feature_noise, features.shape))
    nodes, hes = [], []
    he_id = 0
    with open(osp.join(path, dataset, f'hyperedges-{dataset}.txt')) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            members = [int(v) for v in line.split(',')]
            nodes += members
            hes += [he_id] * len(members)
            he_id += 1
    nodes = np.asarray(nodes, dtype=np.int64)
    nodes = nodes - nodes.min()                                     # node ids shifted to start at 0 (reference :351-352)
    return HypergraphData(x=torch.from_numpy(features.astype(np.float32)), y=torch.from_numpy(labels),
                          edge_index=block_edge_list(nodes, np.asarray(hes, dtype=np.int64), num_nodes),
                          n_x=[num_nodes], num_hyperedges=[he_id], train_percent=train_percent)


def load_yelp_dataset(path: str, name_dictionary_size: int = 1000, train_percent: float = 0.025) -> HypergraphData:
    """The five csv files of the yelp restaurant hypergraph (reference load_other_datasets.py:198-291): features =
    [lat, long | one-hot state | one-hot city | bag-of-words of the name (sklearn CountVectorizer, english stop words, ascii
    accents, ``name_dictionary_size`` terms)], labels = binned stars, incidence csv with 1-based ``node`` / ``he`` columns."""
    import pandas as pd
    from sklearn.feature_extraction.text import CountVectorizer
    latlong = pd.read_csv(osp.join(path, 'yelp_restaurant_latlong.csv')).values
    loc = pd.read_csv(osp.join(path, 'yelp_restaurant_locations.csv'))
    state_int, city_int = loc.state_int.values, loc.city_int.values
    num_nodes = loc.shape[0]
    state_1hot = np.zeros((num_nodes, state_int.max()))
    state_1hot[np.arange(num_nodes), state_int - 1] = 1
    city_1hot = np.zeros((num_nodes, city_int.max()))
    city_1hot[np.arange(num_nodes), city_int - 1] = 1
    vectorizer = CountVectorizer(max_features=name_dictionary_size, stop_words='english', strip_accents='ascii')
    names = pd.read_csv(osp.join(path, 'yelp_restaurant_name.csv')).values.flatten()
    name_bow = np.asarray(vectorizer.fit_transform(names).todense())
    features = np.hstack([latlong, state_1hot, city_1hot, name_bow]).astype(np.float32)
    labels = pd.read_csv(osp.join(path, 'yelp_restaurant_business_stars.csv')).values.flatten().astype(np.int64)
    assert features.shape[0] == len(labels)
    H = pd.read_csv(osp.join(path, 'yelp_restaurant_incidence_H.csv'))
    return HypergraphData(x=torch.from_numpy(features), y=torch.from_numpy(labels),
                          edge_index=block_edge_list(H.node.values - 1, H.he.values - 1, num_nodes),
                          n_x=[num_nodes], num_hyperedges=[int(H.he.values.max())], train_percent=train_percent)


class _PygStandIn:
    """What a pickled ``torch_geometric`` object unpickles into when that package is absent: a bag of attributes.
    PyG 1.6.3 ``Data`` (the reference's pin, README.md:18-22) pickles as class reference + instance ``__dict__``
    (x, edge_index, edge_attr, y, pos, normal, face and the loaders' extras); PyG 2.x keeps them in ``_store._mapping``."""

    def __init__(self, *args, **kwargs):
        self.__dict__.update(kwargs)

    def __setstate__(self, state):
        if isinstance(state, tuple):                       # (dict, slots) protocol
            state = {**(state[0] or {}), **(state[1] or {})}
        self.__dict__.update(state)


class _PygUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split('.')[0] in ('torch_geometric', 'torch_sparse', 'torch_scatter'):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return _PygStandIn
        return super().find_class(module, name)


class _pyg_pickle:
    """``pickle_module`` for ``torch.load``: the stock pickle with torch_geometric classes resolved to stand-ins."""
    __name__ = 'pickle'
    Unpickler = _PygUnpickler
    load = staticmethod(lambda f, **kw: _PygUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump, dumps, Pickler = staticmethod(pickle.dump), staticmethod(pickle.dumps), pickle.Pickler


def _pyg_attrs(obj) -> dict:
    d = dict(vars(obj))
    store = d.get('_store')                                 # PyG 2.x: Data.__dict__ = {'_store': GlobalStorage{_mapping}}
    if store is not None:
        d = dict(getattr(store, '_mapping', None) or vars(store))
    return {k: v for k, v in d.items() if v is not None and not k.startswith('__')}


def load_pyg_processed(path: str) -> HypergraphData:
    """Read the reference's processed dataset file ``<root>/<name>/processed/data.pt`` (or ``data_noise_<f>.pt``):
    ``torch.save(self.collate([data]), ...)`` -- a ``(Data, slices)`` tuple (convert_datasets_to_pygDataset.py:170-175)
    which ``dataset_Hypergraph.__init__`` reads back with ``torch.load`` (:78) and ``train.py`` uses as
    ``dataset.data`` (train.py:327-339).  ``collate`` of a one-graph list keeps tensors as they are and turns the loaders'
    scalar attributes (``n_x``, ``num_hyperedges``, ``train_percent``: load_other_datasets.py:112-117,190-194) into
    1-element tensors.  torch_geometric is not needed: its classes unpickle into attribute bags.

    Returns the attributes ``train.py`` consumes, with its two fall-backs applied (train.py:333-339): ``n_x`` defaults
    to the feature row count, ``num_hyperedges`` to ``edge_index[0].max() - n_x + 1`` (consecutive hyperedge ids)."""
    if osp.isdir(path):
        path = osp.join(path, 'data.pt')
    obj = torch.load(path, map_location='cpu', pickle_module=_pyg_pickle, weights_only=False)
    if not (isinstance(obj, tuple) and len(obj) == 2 and isinstance(obj[1], dict)):
        raise ValueError(f"{path}: expected the (data, slices) tuple InMemoryDataset.collate writes, got {type(obj).__name__}")
    attrs, slices = _pyg_attrs(obj[0]), obj[1]
    for key in ('x', 'edge_index', 'y'):
        if not torch.is_tensor(attrs.get(key)):
            raise ValueError(f"{path}: the stored Data has no tensor attribute {key!r}")
    for key, sl in slices.items():                          # one graph per file: every slice is [0, size]
        if torch.is_tensor(sl) and sl.numel() != 2:
            raise ValueError(f"{path}: {sl.numel() - 1} graphs collated under {key!r}; the reference stores exactly one")
    x, ei, y = attrs['x'], attrs['edge_index'], attrs['y']
    if ei.dim() != 2 or ei.shape[0] != 2:
        raise ValueError(f"{path}: edge_index has shape {tuple(ei.shape)}")

    def scalar(v):
        return int(v.reshape(-1)[0]) if torch.is_tensor(v) else int(v[0] if isinstance(v, (list, tuple)) else v)
    n_x = scalar(attrs['n_x']) if 'n_x' in attrs else int(x.shape[0])
    n_he = scalar(attrs['num_hyperedges']) if 'num_hyperedges' in attrs else int(ei[0].max()) - n_x + 1
    out = HypergraphData(x=x.to(torch.float32), y=y.to(torch.int64).reshape(-1), edge_index=ei.to(torch.int64).contiguous(),
                         n_x=[n_x], num_hyperedges=[n_he])
    if 'train_percent' in attrs:
        tp = attrs['train_percent']
        out.train_percent = float(tp.reshape(-1)[0]) if torch.is_tensor(tp) else float(tp)
    return out


def synthetic_dataset(n_v: int = 4000, n_e: int = 2000, num_classes: int = 5, num_features: int = 64,
                      he_size: int = 6, purity: float = 0.8, feature_noise: float = 1.0, seed: int = 0) -> HypergraphData:
    """Planted-partition hypergraph: each hyperedge draws a class and fills ``purity`` of its members from it;
    features are a class indicator pattern plus N(0, feature_noise^2)."""
    rng = np.random.default_rng(seed)
    y = rng.integers(num_classes, size=n_v)
    by_class = [np.flatnonzero(y == c) for c in range(num_classes)]
    nodes, hes = [], []
    for e in range(n_e):
        c = int(rng.integers(num_classes))
        k_in = max(1, int(round(purity * he_size)))
        members = set(rng.choice(by_class[c], size=min(k_in, by_class[c].size), replace=False).tolist())
        while len(members) < he_size:
            members.add(int(rng.integers(n_v)))
        nodes += sorted(members)
        hes += [e] * len(members)
    proto = rng.standard_normal((num_classes, num_features)).astype(np.float32)
    x = proto[y] + feature_noise * rng.standard_normal((n_v, num_features)).astype(np.float32)
    return HypergraphData(x=torch.from_numpy(x), y=torch.from_numpy(y.astype(np.int64)),
                          edge_index=block_edge_list(np.array(nodes), np.array(hes), n_v),
                          n_x=[n_v], num_hyperedges=[n_e])


def rand_train_test_idx(label, train_prop=.5, valid_prop=.25, ignore_negative=True):
    """Random train/valid/test split over labelled nodes (reference preprocessing.py:472-499, non-balanced)."""
    labeled = torch.where(label != -1)[0] if ignore_negative else torch.arange(label.shape[0])
    n = labeled.shape[0]
    train_num, valid_num = int(n * train_prop), int(n * valid_prop)
    perm = torch.as_tensor(np.random.permutation(n))
    return {'train': labeled[perm[:train_num]], 'valid': labeled[perm[train_num:train_num + valid_num]],
            'test': labeled[perm[train_num + valid_num:]]}


# --------------------------------------------------------------------------------------------------
# model / evaluation (reference train.py:28-42, 106-210)
# --------------------------------------------------------------------------------------------------

def parse_method(args, data):
    if args.method == 'AllSetTransformer':
        if args.LearnMask:
            return SetGNN(args, data.norm)
        return SetGNN(args)
    if args.method == 'AllDeepSets':
        args.PMA = False
        args.aggregate = 'add'
        if args.LearnMask:
            return SetGNN(args, data.norm)
        return SetGNN(args)
    raise ValueError(f"method {args.method!r}: only {ALLSET_METHODS} are built (the reference's baselines are out of scope)")


class Logger:
    def __init__(self, runs, info=None):
        self.info = info
        self.results = [[] for _ in range(runs)]

    def add_result(self, run, result):
        assert len(result) == 3 and 0 <= run < len(self.results)
        self.results[run].append(result)

    def print_statistics(self, run=None):
        if run is not None:
            r = 100 * torch.tensor(self.results[run])
            best = r[:, 1].argmax().item()
            print(f'Run {run + 1:02d}:')
            print(f'Highest Train: {r[:, 0].max():.2f}')
            print(f'Highest Valid: {r[:, 1].max():.2f}')
            print(f'  Final Train: {r[best, 0]:.2f}')
            print(f'   Final Test: {r[best, 2]:.2f}')
            return None
        rows = []
        for r in 100 * torch.tensor(self.results):
            best = r[:, 1].argmax()
            rows.append((r[:, 0].max().item(), r[:, 1].max().item(), r[best, 0].item(), r[best, 2].item()))
        t = torch.tensor(rows)
        print('All runs:')
        for title, col in (('Highest Train', 0), ('Highest Valid', 1), ('  Final Train', 2), ('   Final Test', 3)):
            print(f'{title}: {t[:, col].mean():.2f} ± {t[:, col].std():.2f}')
        return t[:, 1], t[:, 3]


def eval_acc(y_true, y_pred):
    y_true = y_true.detach().cpu().numpy()
    y_hat = y_pred.argmax(dim=-1).detach().cpu().numpy()
    labeled = y_true == y_true
    return float(np.sum(y_true[labeled] == y_hat[labeled])) / max(int(labeled.sum()), 1)


@torch.no_grad()
def evaluate(model, data, split_idx, eval_func, result=None):
    if result is not None:
        out = result
    else:
        model.eval()
        out = F.log_softmax(model(data), dim=1)
    accs = [eval_func(data.y[split_idx[k]], out[split_idx[k]]) for k in ('train', 'valid', 'test')]
    losses = [F.nll_loss(out[split_idx[k]], data.y[split_idx[k]]) for k in ('train', 'valid', 'test')]
    return (*accs, *losses, out)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


# --------------------------------------------------------------------------------------------------
# command line (reference train.py:221-289, flags of the AllSet methods; baseline-only flags are accepted and ignored)
# --------------------------------------------------------------------------------------------------

def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument('--train_prop', type=float, default=0.5)
    p.add_argument('--valid_prop', type=float, default=0.25)
    p.add_argument('--dname', default='synthetic')
    p.add_argument('--method', default='AllSetTransformer')
    p.add_argument('--epochs', default=500, type=int)
    p.add_argument('--runs', default=20, type=int)
    p.add_argument('--cuda', default=0, choices=[-1, 0, 1], type=int)
    p.add_argument('--dropout', default=0.5, type=float)
    p.add_argument('--lr', default=0.001, type=float)
    p.add_argument('--wd', default=0.0, type=float)
    p.add_argument('--All_num_layers', default=2, type=int)
    p.add_argument('--MLP_num_layers', default=2, type=int)
    p.add_argument('--MLP_hidden', default=64, type=int)
    p.add_argument('--Classifier_num_layers', default=2, type=int)
    p.add_argument('--Classifier_hidden', default=64, type=int)
    p.add_argument('--display_step', type=int, default=-1)
    p.add_argument('--aggregate', default='mean', choices=['sum', 'mean'])
    p.add_argument('--normtype', default='all_one')
    p.add_argument('--add_self_loop', action='store_false')
    p.add_argument('--normalization', default='ln')
    p.add_argument('--deepset_input_norm', default=True)
    p.add_argument('--GPR', action='store_false')
    p.add_argument('--LearnMask', action='store_false')
    p.add_argument('--num_features', default=0, type=int)
    p.add_argument('--num_classes', default=0, type=int)
    p.add_argument('--feature_noise', default='1', type=str)
    p.add_argument('--exclude_self', action='store_true')
    p.add_argument('--PMA', action='store_true')
    p.add_argument('--heads', default=1, type=int)
    p.add_argument('--output_heads', default=1, type=int)
    for flag in ('--HyperGCN_mediators', '--HyperGCN_fast', '--HCHA_symdegnorm', '--UniGNN_use-norm'):
        p.add_argument(flag, action='store_true')
    p.add_argument('--HNHN_alpha', default=-1.5, type=float)
    p.add_argument('--HNHN_beta', default=-0.5, type=float)
    p.add_argument('--HNHN_nonlinear_inbetween', default=True, type=bool)
    p.add_argument('--UniGNN_degV', default=0)
    p.add_argument('--UniGNN_degE', default=0)
    # additions of this driver (absent from the reference)
    p.add_argument('--raw_data_dir', default=None, help='directory holding <dname>/{features,labels,hypergraph}.pickle')
    p.add_argument('--processed_data', default=None,
                   help="the reference's processed file <root>/<dname>/processed/data.pt (or its directory)")
    p.add_argument('--seed', default=None, type=int, help='seed numpy/torch (the reference fixes no seeds, README.md:60)')
    p.add_argument('--res_root', default='hyperparameter_tunning')
    p.add_argument('--hip_graph', default=-1, type=int, choices=[-1, 0, 1],
                   help='1: capture the training step, the eval forward and the metrics as hipGraphs (allset_amd/graphs.py); '
                        '0: eager launches; -1 (default): graphs where the loop is launch-bound (at most 200k vertices), eager '
                        'above; a failed capture falls back to eager launches with a warning')
    p.set_defaults(PMA=True, add_self_loop=True, exclude_self=False, GPR=False, LearnMask=False)
    return p


CORNELL_DATASETS = ('amazon-reviews', 'walmart-trips', 'house-committees', 'walmart-trips-100', 'house-committees-100')
HYPERGCN_DATASETS = ('cora', 'citeseer', 'pubmed', 'coauthor_cora', 'coauthor_dblp')


def load_data(args) -> HypergraphData:
    if getattr(args, 'processed_data', None) is not None:
        data = load_pyg_processed(args.processed_data)
        if args.dname in ('yelp', 'walmart-trips', 'house-committees', 'walmart-trips-100', 'house-committees-100'):
            data.y = data.y - data.y.min()                  # labels shifted to start at 0 (reference train.py:329-332)
    elif args.raw_data_dir is not None:
        # the reader per dataset name: reference convert_datasets_to_pygDataset.py:122-163
        if args.dname in CORNELL_DATASETS:
            base = args.dname[:-4] if args.dname.endswith('-100') else args.dname
            data = load_cornell_dataset(args.raw_data_dir, base, feature_noise=float(args.feature_noise),
                                        feature_dim=100 if args.dname.endswith('-100') else None)
            data.y = data.y - data.y.min()                  # labels shifted to start at 0 (reference train.py:329-332)
        elif args.dname == 'yelp':
            data = load_yelp_dataset(osp.join(args.raw_data_dir, 'yelp') if osp.isdir(osp.join(args.raw_data_dir, 'yelp')) else args.raw_data_dir)
            data.y = data.y - data.y.min()
        elif args.dname in HYPERGCN_DATASETS:
            data = load_hypergcn_dataset(args.raw_data_dir, args.dname)
        elif osp.exists(osp.join(args.raw_data_dir, args.dname, f'{args.dname}.content')):
            data = load_le_dataset(args.raw_data_dir, args.dname)
        else:
            data = load_hypergcn_dataset(args.raw_data_dir, args.dname)
    elif args.dname == 'synthetic':
        data = synthetic_dataset(feature_noise=float(args.feature_noise), seed=0 if args.seed is None else args.seed)
    else:
        raise FileNotFoundError(f"dataset {args.dname!r}: pass --raw_data_dir (the reference's raw-data archive is not "
                                "distributed with its repository) or use --dname synthetic")
    args.num_features = data.x.shape[1]
    args.num_classes = len(data.y.unique())
    return data


def preprocess(args, data: HypergraphData) -> HypergraphData:
    """The AllSet branch of reference train.py:344-353."""
    data = ExtractV2E(data)
    if args.add_self_loop:
        data = Add_Self_Loops(data)
    if args.exclude_self:
        data = expand_edge_index(data)
    return norm_contruction(data, option=args.normtype)


def run(args) -> dict:
    if args.method not in ALLSET_METHODS:
        raise ValueError(f"method {args.method!r}: only {ALLSET_METHODS} are built")
    if args.seed is not None:
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
    data = preprocess(args, load_data(args))
    splits = [rand_train_test_idx(data.y, args.train_prop, args.valid_prop) for _ in range(args.runs)]
    model = parse_method(args, data)
    if args.cuda not in (0, 1) or not torch.cuda.is_available():
        raise RuntimeError("allset_amd has no CPU path for the aggregation kernels: run with --cuda 0 on an MI355X")
    device = torch.device(f'cuda:{args.cuda}')
    model, data = model.to(device), data.to(device)
    num_params = count_parameters(model)
    logger = Logger(args.runs, args)
    runtimes = []
    # Python's cyclic collector walks every object of the process (torch alone holds ~10^6) on a generation-2 pass, and the
    # module-tree walks of model.train() / model.eval() in every epoch keep triggering them: measured 4x on an eager dataset-scale
    # step (0.47 -> 1.9 ms per forward).  Everything that exists now lives as long as the run: exempt it from the passes.
    import gc
    gc.collect()
    gc.freeze()
    try:
        return _run_loop(args, model, data, splits, device, logger, runtimes, num_params)
    finally:
        gc.unfreeze()                          # (a process-global switch: callers that import run() get their collector back)


def _run_loop(args, model, data, splits, device, logger, runtimes, num_params):
    for r in range(args.runs):
        t0 = time.time()
        split_idx = {k: v.to(device) for k, v in splits[r].items()}
        model.reset_parameters()
        # same Adam as the reference (train.py:469) as one kernel launch over all parameters (allset_amd/optim.py: torch's fused
        # capturable Adam takes ~40 us for the ~20 small tensors of a Cora-sized model, a tenth of a replayed step)
        from .optim import FusedAdam
        optimizer = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
        # the train-split loss as one forward and one backward kernel (allset_amd/losses.py): same value as
        # criterion(F.log_softmax(out, dim=1)[train_idx], y[train_idx]), the split as a 0/1 row mask made once per run
        from .losses import nll_log_softmax, split_mask
        train_mask, n_train = split_mask(split_idx['train'], data.y.shape[0]), int(split_idx['train'].numel())
        y_all = data.y.long()
        if r == 0:      # once per run() call: the loss kernels turn an out-of-range label into NaN, torch's nll_loss raises -- raise too
            nc = int(getattr(args, "num_classes", 0) or 0)
            lo, hi = int(y_all.min()), int(y_all.max())
            if lo < 0 or (nc > 0 and hi >= nc):
                raise ValueError(f"labels must lie in [0, {nc}): found [{lo}, {hi}]")
        use_graph = args.hip_graph == 1 or (args.hip_graph == -1 and data.y.shape[0] <= 200_000)
        if use_graph:                          # same loop, three graph launches per epoch instead of ~400 kernel launches
            from .graphs import GraphedForward, GraphedTrainStep
            graphed_step = graphed_eval = None
            try:
                graphed_step = GraphedTrainStep(model, data, lambda logits: nll_log_softmax(logits, y_all, train_mask, n_train), optimizer)
                graphed_eval = GraphedForward(model, data, constant_features=True)    # (the same data.x every epoch)
            except (RuntimeError, AllSetHipError) as exc:     # capture-time failures only (an op that synchronises, an unsupported
                if args.hip_graph == 1:                       # launch); shape / dtype bugs in the model are TypeError / ValueError
                    raise                                     # and IndexError and are not swallowed
                print(f"[allset_amd.train] hipGraph capture failed ({type(exc).__name__}: {exc}); running eager launches")
                use_graph = False
                del graphed_step, graphed_eval                # a captured step keeps its memory pool alive
                graphed_step = graphed_eval = None
                model.reset_parameters()
                optimizer = FusedAdam(model.parameters(), lr=args.lr, weight_decay=args.wd)
        # evaluate() of the reference (train.py:483: three accuracies + three losses per epoch, each through .cpu()) as one kernel
        # whose six numbers stay on the device: they are read back once per run (and every display_step epochs for the progress
        # line) -- at dataset scale the per-epoch host round trips cost several times the 0.4 ms training step
        from .losses import split_ids, split_metrics
        sp = split_ids(split_idx, data.y.shape[0], device)
        counts = torch.tensor([float(split_idx[k].numel()) for k in ('train', 'valid', 'test')], device=device)
        hist = torch.zeros((args.epochs, 7), dtype=torch.float32, device=device)       # [acc x 3, loss x 3, train-step loss]
        if use_graph:                          # the metrics of the replayed forward and the step's loss as a third small graph
            from .graphs import GraphedCallable
            graphed_metrics = GraphedCallable(
                lambda: torch.cat([split_metrics(graphed_eval.out, y_all, sp, counts), graphed_step.loss.detach().reshape(1)]), device)
        for epoch in range(args.epochs):
            if use_graph:
                graphed_step()
                graphed_eval()
                hist[epoch] = graphed_metrics()
                if args.display_step > 0 and epoch % args.display_step == 0:
                    m = hist[epoch].tolist()
                    print(f'Epoch: {epoch:02d}, Train Loss: {m[6]:.4f}, Valid Loss: {m[4]:.4f}, Test  Loss: {m[5]:.4f}, '
                          f'Train Acc: {100 * m[0]:.2f}%, Valid Acc: {100 * m[1]:.2f}%, Test  Acc: {100 * m[2]:.2f}%')
                continue
            else:
                model.train()
                optimizer.zero_grad()
                loss = nll_log_softmax(model(data), y_all, train_mask, n_train)
                # every parameter-gradient partial of the backward pass reduced by ONE launch (dense.deferred_param_grads)
                with dense.deferred_param_grads():
                    loss.backward()
                optimizer.step()
                model.eval()
                with torch.no_grad(), dense.constant_features():      # (the same data.x every epoch: raw sparse features from their non-zeros)
                    logits = model(data)
                loss = loss.detach()
            with torch.no_grad():
                hist[epoch, :6] = split_metrics(logits, y_all, sp, counts)
                hist[epoch, 6] = loss
            if args.display_step > 0 and epoch % args.display_step == 0:
                m = hist[epoch].tolist()
                print(f'Epoch: {epoch:02d}, Train Loss: {m[6]:.4f}, Valid Loss: {m[4]:.4f}, Test  Loss: {m[5]:.4f}, '
                      f'Train Acc: {100 * m[0]:.2f}%, Valid Acc: {100 * m[1]:.2f}%, Test  Acc: {100 * m[2]:.2f}%')
        for row in hist[:, :3].tolist():                   # one read-back per run
            logger.add_result(r, tuple(row))
        runtimes.append(time.time() - t0)
    avg_time, std_time = float(np.mean(runtimes)), float(np.std(runtimes))
    best_val, best_test = logger.print_statistics()
    os.makedirs(args.res_root, exist_ok=True)
    filename = f'{args.res_root}/{args.dname}_noise_{args.feature_noise}.csv'
    print(f"Saving results to {filename}")
    with open(filename, 'a+') as f:                                   # line format of reference train.py:512-518
        f.write(f'{args.method}_{args.lr}_{args.wd}_{args.heads}'
                f',{best_val.mean():.3f} ± {best_val.std():.3f}'
                f',{best_test.mean():.3f} ± {best_test.std():.3f}'
                f',{num_params}, {avg_time:.2f}s, {std_time:.2f}s'
                f',{avg_time // 60}min{(avg_time % 60):.2f}s\n')
    with open(f'{args.res_root}/all_args_{args.dname}_noise_{args.feature_noise}.csv', 'a+') as f:
        f.write(str(args) + '\n')
    return dict(best_val=best_val, best_test=best_test, num_params=num_params, avg_time=avg_time, csv=filename)


def main(argv=None):
    run(build_parser().parse_args(argv))
    print('All done! Exit python code')


if __name__ == '__main__':
    main()
