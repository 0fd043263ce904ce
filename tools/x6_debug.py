import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from allset_amd import dense
dev = torch.device("cuda:0")
torch.manual_seed(0)
for K, N in ((128, 128), (64, 64)):
    for n in (16, 48, 1000):
        x = torch.randn(n, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.zeros(N, device=dev)
        y, _ = dense.fused_linear_fwd(x, W, b)
        ref = x @ W.t()
        err = (y - ref).abs()
        print(K, N, n, "max err", float(err.max()), "nan", int(torch.isnan(y).sum()))
        if n == 16:
            bad = (err > 1e-3) | torch.isnan(y)
            print(" bad rows:", bad.any(1).nonzero().flatten().tolist()[:20], " bad cols:", bad.any(0).nonzero().flatten().tolist()[:40])
            # unit tests: x = e_k rows
            x2 = torch.zeros(16, K, device=dev); x2[torch.arange(16), torch.arange(16) * (K // 16)] = 1.0
            y2, _ = dense.fused_linear_fwd(x2, W, b)
            ref2 = x2 @ W.t()
            print(" unit err", float((y2 - ref2).abs().max()))
