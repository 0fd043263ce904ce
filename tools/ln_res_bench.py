#!/usr/bin/env python
"""ln_res_fwd / ln_res_bwd variants at [1M, 128] fp32 (the PMA tail's two LayerNorms): time and achieved bytes per second."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allset_amd import dense
dev = torch.device("cuda:0"); n, d, H = 1_000_000, 128, 4
g = torch.Generator().manual_seed(0)
x, res, gy = (torch.randn(n, d, generator=g).to(dev) for _ in range(3))
colb, gamma, beta = (torch.randn(d, generator=g).to(dev) for _ in range(3))
m, l = torch.randn(n, H, generator=g).to(dev), torch.rand(n, H, generator=g).to(dev) + 0.5


def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


y0, st0 = dense.ln_res_fwd(x, colb, None, gamma, beta, 1e-5, False, 0.0, 0, None)
y1, st1 = dense.ln_res_fwd(x, None, res, gamma, beta, 1e-5, True, 0.0, 0, None)
B = n * d * 4
for name, fn, nb in (
        ("fwd  x + colb", lambda: dense.ln_res_fwd(x, colb, None, gamma, beta, 1e-5, False, 0.0, 0, None), 2 * B),
        ("fwd  x + res, relu", lambda: dense.ln_res_fwd(x, None, res, gamma, beta, 1e-5, True, 0.0, 0, None), 3 * B),
        ("bwd  x + colb", lambda: dense.ln_res_bwd(gy, x, colb, None, st0, gamma, beta, False, 0.0, 0), 3 * B),
        ("bwd  x + colb + pooling stats", lambda: dense.ln_res_bwd_pma(gy, x, colb, st0, gamma, beta, m, l), 3 * B),
        ("bwd  x + res, relu", lambda: dense.ln_res_bwd(gy, x, None, res, st1, gamma, beta, True, 0.0, 0), 4 * B)):
    ms = t(fn)
    print(f"{name:32s} {ms * 1e3:7.1f} us   {nb / ms / 1e9:6.2f} TB/s")
