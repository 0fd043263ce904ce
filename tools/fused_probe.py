"""Workload for rocprofv3 --pmc passes over the dense-tail kernels (forward, backward-data, weight gradient at the bench
shape, the K2 variants: LayerNorm + dropout prologue, relu + dropout epilogue, activation mask)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allset_amd import dense
dev = torch.device("cuda:0")
n, d, p = 1_000_000, 128, 0.5
x = torch.randn(n, d, device=dev); W = torch.randn(d, d, device=dev) / d ** 0.5; b = torch.randn(d, device=dev)
g, bt = torch.ones(d, device=dev), torch.zeros(d, device=dev); G = torch.randn(n, d, device=dev)
mask = torch.empty(dense.activation_mask_words(n, d), dtype=torch.int32, device=dev)
for _ in range(3):
    y, st = dense.fused_linear_fwd(x, W, b, g, bt, 1e-5, True, p, 1, True, p, 2, None, mask)
    dense.fused_linear_bwd(G, None, p, W, x, st, g, True, p, 1, None, mask)
    dense.wgrad_fused(G, None, p, x, st, g, bt, True, p, 1, mask=mask)
    dense.fused_linear_bwd_all(G, mask, p, W, x, st, g, bt, True, p, 1)         # round 2: the pair above as one pass
torch.cuda.synchronize()
