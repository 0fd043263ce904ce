#!/usr/bin/env python
"""Dense-tail micro-benchmark at the bench shape: unfused chain vs fused kernels (HIP events, median)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from allset_amd import dense
dev = torch.device("cuda:0")
n, d = 1_000_000, 128
x = torch.randn(n, d, device=dev); W = torch.randn(d, d, device=dev) / d ** 0.5; b = torch.randn(d, device=dev)
g = torch.ones(d, device=dev); bt = torch.zeros(d, device=dev)

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return statistics.median(ts)

flop = 2 * n * d * d
t = timeit(lambda: F.linear(x, W, b)); print(f"hipBLASLt linear              {t:.3f} ms  {flop/t/1e9:.0f} TF")
t = timeit(lambda: dense.fused_linear_fwd(x, W, b)); print(f"fused plain                   {t:.3f} ms  {flop/t/1e9:.0f} TF")
t = timeit(lambda: F.linear(dense.ln_fwd(x, g, bt, 1e-5, False, 0.0, 0)[0], W, b)); print(f"ln_fwd + linear               {t:.3f} ms")
t = timeit(lambda: dense.fused_linear_fwd(x, W, b, g, bt)); print(f"fused LN+linear               {t:.3f} ms")
t = timeit(lambda: dense.relu_dropout(F.linear(dense.ln_fwd(x, g, bt, 1e-5, True, 0.5, 1)[0], W, b), 0.5)); print(f"relu+ln+drop, linear, relu+drop {t:.3f} ms")
t = timeit(lambda: dense.fused_linear_fwd(x, W, b, g, bt, 1e-5, True, 0.5, 1, True, 0.5, 2)); print(f"fused (all pro/epilogue)      {t:.3f} ms")
