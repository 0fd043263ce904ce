#!/usr/bin/env python
"""allset_gemm_x6 (csrc/wide_mlp.hip): accuracy against float64 and time against the library fp32 GEMM.
usage: gemm_x6_probe.py [rows] [K] [N]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from allset_amd import dense
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(rows, K, device=dev) * torch.exp(2 * torch.randn(rows, 1, device=dev))
W = torch.randn(N, K, device=dev) / K ** 0.5
b = torch.randn(N, device=dev)


def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return statistics.median(ts)


planes = dense.gemm_x6_planes(W, False)
y = dense.gemm_x6(x, planes, N, b)
sub = slice(0, 4096)
ref = x[sub].double() @ W.double().t() + b.double()
scale = (x[sub].double().abs() @ W.double().abs().t()) + b.double().abs()
print("gemm_x6   max err / sum|terms|:", float(((y[sub].double() - ref).abs() / scale).max()))
yl = F.linear(x, W, b)
print("library   max err / sum|terms|:", float(((yl[sub].double() - ref).abs() / scale).max()))
ms = t(lambda: dense.gemm_x6(x, planes, N, b))
ml = t(lambda: F.linear(x, W, b))
fl = 2 * rows * N * K
print(f"rows={rows} K={K} N={N}:  gemm_x6 {ms*1e3:.0f} us ({fl/ms/1e9:.0f} TFLOP/s fp32-equivalent, {rows*(K+N)*4/ms/1e6:.0f} GB/s)   "
      f"library fp32 {ml*1e3:.0f} us ({fl/ml/1e9:.0f} TFLOP/s)   planes {t(lambda: dense.gemm_x6_planes(W, False))*1e3:.0f} us")
# with the LayerNorm prologue and relu/dropout epilogue
g, be = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
st = dense.row_stats(x, True, 1e-5)
ms2 = t(lambda: dense.gemm_x6(x, planes, N, b, relu_in=True, stats=st, gamma=g, beta=be, p_in=0.5, seed_in=3, relu_out=True, p_out=0.5, seed_out=4))
print(f"  with relu->LN->dropout prologue and relu->dropout epilogue: {ms2*1e3:.0f} us  (+ row_stats {t(lambda: dense.row_stats(x, True, 1e-5))*1e3:.0f} us)")
