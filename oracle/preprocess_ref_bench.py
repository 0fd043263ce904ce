#!/usr/bin/env python
"""Wall time of the REFERENCE's own host preprocessing (``/root/reference/src/preprocessing.py:394-469``: ExtractV2E ->
Add_Self_Loops -> norm_contruction('deg_half_sym')) on synthetic hypergraphs of growing size.  TEST / MEASUREMENT INFRASTRUCTURE,
container-only (needs /root/reference; runs the reference's Python loops on the host CPU, ``oracle/ref_shim.py`` for the absent wheels).
Companion of ``tools/preprocess_bench.py`` (the product's device preprocessing at |V| = |E| = 1M on the GPU box): the reason SURVEY 8(f1)
exists.  Output committed as profiles/r05_preprocess_reference_host.txt.

usage: python oracle/preprocess_ref_bench.py [n ...]      (default 2000 8000 32000; hyperedge size 16, |E| = |V|)"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

ref_pre = ref_shim.import_reference_preprocessing()
sizes = [int(a) for a in sys.argv[1:] if not a.startswith("-")] or [2000, 8000, 32000]
SINGLE = 0.05 if "--singletons" in sys.argv else 0.0     # that share of the hyperedges has ONE member (they exist in every real dataset:
print(f"reference host preprocessing (torch {torch.__version__}, {os.cpu_count()} host threads; Add_Self_Loops is Python loops)"      # their owners go on the skip list)
      + (f"; {SINGLE:.0%} of the hyperedges are singletons" if SINGLE else ""))
prev = None
for n in sizes:
    rng = np.random.default_rng(3)
    n_single = int(SINGLE * n)
    sizes_e = np.full(n, 16)
    sizes_e[:n_single] = 1
    v = np.concatenate([rng.choice(n, size=int(k), replace=False) if k > 1 else np.array([i]) for i, k in enumerate(sizes_e)])
    e = np.repeat(np.arange(n), sizes_e) + n
    miss = np.setdiff1d(np.arange(n), v)                    # every vertex id present: overwrite members of the LAST hyperedges' tails
    v[len(v) - len(miss):] = miss
    block = torch.from_numpy(np.concatenate([np.stack([v, e]), np.stack([e, v])], axis=1))
    block = block[:, torch.randperm(block.shape[1], generator=torch.Generator().manual_seed(1))]
    data = SimpleNamespace(edge_index=block.clone(), n_x=[n], num_hyperedges=[n])
    t0 = time.perf_counter()
    data = ref_pre.ExtractV2E(data)
    t1 = time.perf_counter()
    data = ref_pre.Add_Self_Loops(data)
    t2 = time.perf_counter()
    data = ref_pre.norm_contruction(data, option="deg_half_sym")
    t3 = time.perf_counter()
    growth = "" if prev is None else f"   Add_Self_Loops x{(t2 - t1) / prev[1]:.1f} for x{n / prev[0]:.0f} vertices"
    print(f"|V| = |E| = {n:7d}, nnz {len(v):9d}: ExtractV2E {1e3 * (t1 - t0):9.1f} ms  Add_Self_Loops {1e3 * (t2 - t1):10.1f} ms  "
          f"norm {1e3 * (t3 - t2):8.1f} ms{growth}", flush=True)
    prev = (n, t2 - t1)
n0, t0_ = prev
print(f"extrapolated to |V| = 1M (the loop is O(|V| * |skip list| + |E| * nnz): at least linear): Add_Self_Loops >= {t0_ * 1e6 / n0:.0f} s")
