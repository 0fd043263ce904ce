import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allset_amd import dense
dev = torch.device("cuda:0")
n, d = 1_000_000, 128
x = torch.randn(n, d, device=dev); W = torch.randn(d, d, device=dev) / d ** 0.5; b = torch.randn(d, device=dev)
for _ in range(3):
    dense.fused_linear_fwd(x, W, b)
torch.cuda.synchronize()
