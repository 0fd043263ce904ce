"""CPU, build container only (skipped where /root/reference is absent): allset_amd.preprocessing against the LIVE
reference functions (reference preprocessing.py:394-469, 22-144, imported through oracle/ref_shim.py) on randomised
block edge lists -- sizes, duplicate-free incidences, size-1 hyperedges, vertices without hyperedges.  Index work:
bit-exact, compared as lexicographically sorted edge lists (the reference's own sorts are unstable)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")


def canon(ei) -> np.ndarray:
    a = ei.cpu().numpy() if torch.is_tensor(ei) else np.asarray(ei)
    return a[:, np.lexsort((a[1], a[0]))]


@settings(deadline=None, max_examples=int(os.environ.get("ALLSET_HYPOTHESIS_EXAMPLES", "40")),
          derandomize=os.environ.get("ALLSET_HYPOTHESIS_RANDOM", "0") != "1")
@given(n_v=st.integers(3, 60), n_e=st.integers(1, 30), extra=st.integers(0, 150), singles=st.integers(0, 8), sd=st.integers(0, 10 ** 6))
def test_preprocessing_equals_live_reference(n_v, n_e, extra, singles, sd):
    from allset_amd import preprocessing as P
    ref_pre = ref_shim.import_reference_preprocessing()
    rng = np.random.default_rng(sd)
    # every hyperedge gets a member; `singles` of them stay size-1, each owned by a distinct vertex (the reference's
    # Add_Self_Loops indexes out of bounds otherwise, preprocessing.py:428-440); the last vertex id is present
    singles = min(singles, n_e, n_v - 1)
    owners = rng.permutation(n_v - 1)[:singles]
    pairs = {(int(owners[e]), e) for e in range(singles)}
    for e in range(singles, n_e):                      # every other hyperedge has at least two distinct members
        a = int(rng.integers(n_v))
        pairs.add((a, e))
        pairs.add(((a + 1 + int(rng.integers(n_v - 1))) % n_v, e))
    if singles < n_e:
        pairs.add((n_v - 1, n_e - 1))
    for _ in range(extra):
        if singles < n_e:
            pairs.add((int(rng.integers(n_v)), int(rng.integers(singles, n_e))))
    pairs = sorted(pairs)
    v = np.array([p[0] for p in pairs], dtype=np.int64)
    e = np.array([p[1] for p in pairs], dtype=np.int64) + n_v
    block = np.concatenate([np.stack([v, e]), np.stack([e, v])], axis=1)
    block = block[:, rng.permutation(block.shape[1])]

    def fresh():
        return SimpleNamespace(edge_index=torch.from_numpy(block).clone(), n_x=[n_v], num_hyperedges=[n_e])
    r, p = ref_pre.ExtractV2E(fresh()), P.ExtractV2E(fresh())
    np.testing.assert_array_equal(canon(p.edge_index), canon(r.edge_index))
    r, p = ref_pre.Add_Self_Loops(r), P.Add_Self_Loops(p)
    np.testing.assert_array_equal(canon(p.edge_index), canon(r.edge_index))
    assert int(p.totedges) == int(r.totedges)
    sl = canon(r.edge_index)
    for opt in ("all_one", "deg_half_sym"):
        rn = ref_pre.norm_contruction(SimpleNamespace(edge_index=torch.from_numpy(sl).clone()), option=opt)
        pn = P.norm_contruction(SimpleNamespace(edge_index=torch.from_numpy(sl).clone()), option=opt)
        assert pn.norm.dtype == rn.norm.dtype
        np.testing.assert_allclose(pn.norm.numpy(), rn.norm.numpy(), rtol=1e-6, atol=1e-7)
    for th in (0, 3):
        def de():
            return SimpleNamespace(edge_index=torch.from_numpy(sl).clone(), n_x=torch.tensor([n_v]), num_hyperedges=[n_e],
                                   totedges=int(r.totedges))
        re_, pe = ref_pre.expand_edge_index(de(), edge_th=th), P.expand_edge_index(de(), edge_th=th)
        np.testing.assert_array_equal(canon(pe.edge_index), canon(re_.edge_index))
