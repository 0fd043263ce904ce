#!/usr/bin/env python
"""Accuracy and speed of the fused Linear on the bf16 matrix pipe (bf16x6) vs the native fp32 MFMA kernels vs a float64
reference.  Run twice: ALLSET_DENSE_MFMA=f32 python tools/x6_accuracy.py ; python tools/x6_accuracy.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from allset_amd import dense
dev = torch.device("cuda:0")
mode = os.environ.get("ALLSET_DENSE_MFMA", "bf16x6")
torch.manual_seed(0)
for scale_name, xs in (("N(0,1)", 1.0), ("wide dynamic range", None)):
    n, K, N = 200_000, 128, 128
    x = torch.randn(n, K, device=dev)
    if xs is None:
        x = x * torch.exp(4 * torch.randn(n, K, device=dev))
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    ref = (x.double() @ W.double().t() + b.double())
    y, _ = dense.fused_linear_fwd(x, W, b)
    lib = torch.nn.functional.linear(x, W, b)
    denom = (x.double().abs() @ W.double().abs().t() + b.double().abs())          # sum |terms|: the natural error scale
    for name, got in ((f"fused[{mode}]", y), ("hipBLASLt fp32", lib)):
        err = (got.double() - ref).abs()
        print(f"{scale_name:20s} {name:16s} max|err|/sum|terms| = {float((err / denom).max()):.3e}   "
              f"rms = {float((err / denom).pow(2).mean().sqrt()):.3e}   max rel-to-max = {float(err.max() / ref.abs().max()):.3e}")
n = 1_000_000
x = torch.randn(n, 128, device=dev); W = torch.randn(128, 128, device=dev) / 11.3; b = torch.zeros(128, device=dev)
g, bt = torch.ones(128, device=dev), torch.zeros(128, device=dev)
def timeit(fn, it=20):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(it):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return statistics.median(ts)
print(f"[{mode}] plain fwd            {timeit(lambda: dense.fused_linear_fwd(x, W, b)):.3f} ms")
print(f"[{mode}] LN + fwd             {timeit(lambda: dense.fused_linear_fwd(x, W, b, g, bt)):.3f} ms")
print(f"[{mode}] relu LN drop fwd relu drop {timeit(lambda: dense.fused_linear_fwd(x, W, b, g, bt, 1e-5, True, 0.5, 1, True, 0.5, 2)):.3f} ms")
G = torch.randn(n, 128, device=dev)
y0, st0 = dense.fused_linear_fwd(x, W, b, g, bt)
y1, st1 = dense.fused_linear_fwd(x, W, b, g, bt, 1e-5, True, 0.5, 1, True, 0.5, 2)
print(f"[{mode}] bwd  LN                    {timeit(lambda: dense.fused_linear_bwd(G, None, 0.0, W, x, st0, g, False, 0.0, 0)):.3f} ms")
print(f"[{mode}] bwd  relu LN drop | relu drop {timeit(lambda: dense.fused_linear_bwd(G, y1, 0.5, W, x, st1, g, True, 0.5, 1)):.3f} ms")
print(f"[{mode}] wgrad LN                   {timeit(lambda: dense.wgrad_fused(G, None, 0.0, x, st0, g, bt, False, 0.0, 0)):.3f} ms")
print(f"[{mode}] wgrad relu LN drop | relu drop {timeit(lambda: dense.wgrad_fused(G, y1, 0.5, x, st1, g, bt, True, 0.5, 1)):.3f} ms")
