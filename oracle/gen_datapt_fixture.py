"""Writes tests/golden/datapt/*.pt: small processed-dataset files in the layout the reference's
``dataset_Hypergraph.process`` produces (convert_datasets_to_pygDataset.py:170-175):

    torch.save(self.collate([data]), self.processed_paths[0])          # -> (Data, slices)

torch_geometric (pinned 1.6.3, reference README.md:18-22) is NOT installed in the build container and cannot be, so the
``Data`` object is pickled through a class registered under the same qualified name, ``torch_geometric.data.data.Data``,
holding the instance attributes PyG 1.6.3's ``Data.__init__`` sets (x, edge_index, edge_attr, y, pos, normal, face, then
the loaders' extras) and nothing else -- 1.6.3 defines no ``__getstate__``, so the pickle is class reference + ``__dict__``.
The content is what ``load_citation_dataset`` (load_other_datasets.py:121-196) builds and ``collate`` of a one-element list
leaves: tensors unchanged, Python scalars (n_x, num_hyperedges, train_percent) wrapped into 1-element tensors, ``slices``
= {key: tensor([0, size along the cat dim])}.  [external: PyG 1.6.3 InMemoryDataset.collate]

Variants: legacy (torch <= 1.5) and zip serialisation; a PyG-2.x-style file (attributes under ``_store._mapping``); a file
without n_x / num_hyperedges (exercises the two fall-backs of reference train.py:333-339).
Test infrastructure only; run here once:  python oracle/gen_datapt_fixture.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "datapt")


def _register():
    class Data:                                   # attribute layout of torch_geometric 1.6.3 data/data.py
        def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None, normal=None, face=None, **kw):
            self.x, self.edge_index, self.edge_attr, self.y = x, edge_index, edge_attr, y
            self.pos, self.normal, self.face = pos, normal, face
            for k, v in kw.items():
                setattr(self, k, v)
    Data.__module__, Data.__qualname__ = "torch_geometric.data.data", "Data"

    class GlobalStorage:                          # PyG 2.x: attributes live in a mapping
        def __init__(self, mapping):
            self._mapping = mapping
    GlobalStorage.__module__, GlobalStorage.__qualname__ = "torch_geometric.data.storage", "GlobalStorage"
    for name in ("torch_geometric", "torch_geometric.data", "torch_geometric.data.data", "torch_geometric.data.storage"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torch_geometric.data.data"].Data = Data
    sys.modules["torch_geometric.data.storage"].GlobalStorage = GlobalStorage
    return Data, GlobalStorage


def content(seed=7, n_v=12, n_he=5, n_feat=6, n_cls=3):
    """A citation-style hypergraph: star-expansion edge list [[V|E],[E|V]] with hyperedge ids offset by n_v, sorted
    and de-duplicated (load_other_datasets.py:165-181)."""
    rng = np.random.default_rng(seed)
    pairs = set()
    for e in range(n_he):
        for v in rng.choice(n_v, size=int(rng.integers(2, 5)), replace=False):
            pairs.add((int(v), n_v + e))
    both = sorted(pairs | {(b, a) for a, b in pairs})
    ei = torch.tensor(both, dtype=torch.int64).t().contiguous()
    x = torch.from_numpy(rng.standard_normal((n_v, n_feat)).astype(np.float32))
    y = torch.from_numpy(rng.integers(n_cls, size=n_v).astype(np.int64))
    return x, ei, y, n_v, n_he


def main():
    os.makedirs(OUT, exist_ok=True)
    Data, GlobalStorage = _register()
    x, ei, y, n_v, n_he = content()
    extras = dict(n_x=torch.tensor([n_v]), train_percent=torch.tensor([0.025]), num_hyperedges=torch.tensor([n_he]))
    slices = {"x": torch.tensor([0, n_v]), "edge_index": torch.tensor([0, ei.shape[1]]), "y": torch.tensor([0, n_v]),
              "n_x": torch.tensor([0, 1]), "train_percent": torch.tensor([0, 1]), "num_hyperedges": torch.tensor([0, 1])}
    torch.save((Data(x=x, edge_index=ei, y=y, **extras), slices), os.path.join(OUT, "pyg163_legacy.pt"),
               _use_new_zipfile_serialization=False)
    torch.save((Data(x=x, edge_index=ei, y=y, **extras), slices), os.path.join(OUT, "pyg163_zip.pt"))
    bare = {k: v for k, v in slices.items() if k in ("x", "edge_index", "y")}
    torch.save((Data(x=x, edge_index=ei, y=y), bare), os.path.join(OUT, "pyg163_no_counts.pt"))
    d2 = Data.__new__(Data)
    d2.__dict__ = {"_store": GlobalStorage({"x": x, "edge_index": ei, "y": y, **extras})}
    torch.save((d2, slices), os.path.join(OUT, "pyg2_store.pt"))
    np.savez(os.path.join(OUT, "expected.npz"), x=x.numpy(), edge_index=ei.numpy(), y=y.numpy(), n_x=n_v, num_hyperedges=n_he)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
