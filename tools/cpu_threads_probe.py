"""Which host thread count is fairest (fastest) for the CPU baseline?  Times the oracle aggregation."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import allset_oracle as oracle
from allset_amd.synthetic import random_hypergraph
n, d = 100_000, 128
hg = random_hypergraph(n, n, 16, seed=1, device="cpu")
x = torch.randn(n, d)
print("cpu_count", os.cpu_count())
for th in (8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    torch.set_num_threads(th)
    oracle.v2e2v_aggregation_fwd_bwd(x, hg.edge_index, hg.norm, "add")
    t0 = time.perf_counter(); oracle.v2e2v_aggregation_fwd_bwd(x, hg.edge_index, hg.norm, "add"); dt = time.perf_counter() - t0
    print(f"threads {th:4d}: {dt:.2f} s  {hg.nnz*d/dt:.3e} edges*d/s", flush=True)
