"""Writes tests/golden/raw/: tiny RAW dataset files in the three on-disk formats the reference's loaders read besides the
HyperGCN pickles (load_other_datasets.py:32 LE ``.content`` / ``.edges``, :293 Cornell ``node-labels-*.txt`` /
``hyperedges-*.txt``, :198 the yelp csv set), and ``expected.npz`` = what the REFERENCE's own loaders return for them (imported
from /root/reference through oracle/ref_shim.py: ``Data`` as an attribute bag, ``torch_sparse.coalesce`` restated).
tests/test_train_driver.py reads the same files with allset_amd.train's readers and compares.

The data are made up here (seeded); only the formats are the reference's.  Test infrastructure only; run here once:
    python oracle/gen_raw_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "raw")
CORNELL_SEED = 1234


def write_le(rng):
    name, n_v, n_e, F = "toyLE", 23, 9, 6
    d = os.path.join(OUT, name)
    os.makedirs(d, exist_ok=True)
    ids = rng.permutation(np.arange(100, 100 + n_v))                     # node ids as they appear in the file: arbitrary ints
    he_ids = np.arange(500, 500 + n_e)
    with open(os.path.join(d, f"{name}.content"), "w") as f:
        for i in ids:                                                    # node lines first, then hyperedge lines (features unused)
            f.write(" ".join([str(i)] + [f"{v:.4f}" for v in rng.standard_normal(F)] + [str(int(rng.integers(0, 4)))]) + "\n")
        for h in he_ids:
            f.write(" ".join([str(h)] + ["0.0"] * F + ["0"]) + "\n")
    pairs = set()
    for k, h in enumerate(he_ids):
        for v in rng.choice(ids, size=int(rng.integers(2, 6)), replace=False):
            pairs.add((int(v), int(h)))
    for v in ids:                                                        # every node in some hyperedge (consecutive-id assertion)
        pairs.add((int(v), int(he_ids[int(rng.integers(n_e))])))
    pairs = sorted(pairs, key=lambda t: (rng.random(), t))               # unordered, as in the real files
    pairs += pairs[:5]                                                   # duplicates: coalesce removes them
    with open(os.path.join(d, f"{name}.edges"), "w") as f:
        for v, h in pairs:
            f.write(f"{v} {h}\n")
    return name


def write_cornell(rng):
    name, n_v, n_e, C = "toy-trips", 31, 12, 5
    d = os.path.join(OUT, name)
    os.makedirs(d, exist_ok=True)
    labels = np.concatenate([np.arange(1, C + 1), rng.integers(1, C + 1, size=n_v - C)])
    np.savetxt(os.path.join(d, f"node-labels-{name}.txt"), labels, fmt="%d")
    with open(os.path.join(d, f"hyperedges-{name}.txt"), "w") as f:
        for k in range(n_e):
            members = rng.choice(np.arange(1, n_v + 1), size=int(rng.integers(1, 7)), replace=False)      # ids from 1
            if k == 0:
                members = np.concatenate([[1], members[members != 1]])    # the minimum id occurs: ids shift to 0
            f.write(",".join(str(int(v)) for v in members) + "\n")
    return name


def write_yelp(rng):
    import pandas as pd
    d = os.path.join(OUT, "yelp")
    os.makedirs(d, exist_ok=True)
    n_v, n_e = 17, 8
    pd.DataFrame({"latitude": rng.uniform(30, 45, n_v).round(4), "longitude": rng.uniform(-120, -70, n_v).round(4)}).to_csv(
        os.path.join(d, "yelp_restaurant_latlong.csv"), index=False)
    pd.DataFrame({"state_int": np.concatenate([[1, 2, 3], rng.integers(1, 4, n_v - 3)]),
                  "city_int": np.concatenate([[1, 2, 3, 4, 5], rng.integers(1, 6, n_v - 5)])}).to_csv(
        os.path.join(d, "yelp_restaurant_locations.csv"), index=False)
    words = ["taco", "burger", "palace", "noodle", "house", "the", "golden", "cafe", "grill", "pizza", "and", "bistro"]
    names = [" ".join(rng.choice(words, size=int(rng.integers(1, 4)))) for _ in range(n_v)]
    pd.DataFrame({"name": names}).to_csv(os.path.join(d, "yelp_restaurant_name.csv"), index=False)
    pd.DataFrame({"stars": rng.integers(2, 11, n_v)}).to_csv(os.path.join(d, "yelp_restaurant_business_stars.csv"), index=False)
    node, he = [], []
    for h in range(1, n_e + 1):
        for v in rng.choice(np.arange(1, n_v + 1), size=int(rng.integers(2, 6)), replace=False):
            node.append(int(v)); he.append(h)
    pd.DataFrame({"node": node, "he": he}).to_csv(os.path.join(d, "yelp_restaurant_incidence_H.csv"), index=False)
    return d


def pack(prefix, data, out):
    out[f"{prefix}_x"] = data.x.numpy()
    out[f"{prefix}_y"] = data.y.numpy()
    out[f"{prefix}_edge_index"] = data.edge_index.numpy()
    out[f"{prefix}_n_x"] = np.int64(data.n_x)
    out[f"{prefix}_num_hyperedges"] = np.int64(data.num_hyperedges)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20260929)
    le, cornell, yelp_dir = write_le(rng), write_cornell(rng), write_yelp(rng)
    ref = ref_shim.import_reference_loaders()
    out = {}
    pack("le", ref.load_LE_dataset(path=OUT, dataset=le), out)
    np.random.seed(CORNELL_SEED)                                         # the reference draws the feature noise from numpy's global state
    pack("cornell", ref.load_cornell_dataset(path=OUT, dataset=cornell, feature_noise=0.6), out)
    np.random.seed(CORNELL_SEED)
    pack("cornell100", ref.load_cornell_dataset(path=OUT, dataset=cornell, feature_noise=1.0, feature_dim=100), out)
    pack("yelp", ref.load_yelp_dataset(path=yelp_dir, name_dictionary_size=1000), out)
    out["cornell_seed"] = np.int64(CORNELL_SEED)
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
