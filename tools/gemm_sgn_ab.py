#!/usr/bin/env python
"""A/B of allset_gemm_wide_sgn (the relu backward of a wide Linear's input as the backward-data GEMM's epilogue, ABI 15) against the
pair it replaces (allset_gemm_wide + allset_relu_dropout_bwd) at benchmark scale and at dataset scale.  Run on the GPU box."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from allset_amd import dense, _lib
dev = torch.device("cuda:0")
lib = _lib.load()

def timed(fn, reps=20):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return statistics.median(ts)

for n, d in ((1_000_000, 256), (250_000, 512), (4391, 512), (3327, 256)):
    g = torch.Generator(device=dev).manual_seed(1)
    W = torch.randn(d, d, device=dev, generator=g) / d ** 0.5
    G = torch.randn(n, d, device=dev, generator=g)
    x = torch.randn(n, d, device=dev, generator=g)
    y = torch.relu(torch.randn(n, d, device=dev, generator=g))
    planes = dense.gemm_x6_planes(W, True, f16=True)
    gx = torch.empty_like(x)
    def pair():
        gu = dense.gemm_x6(G, planes, d, None, mask_y=y, p_mask=0.0)
        _lib.check(lib.allset_relu_dropout_bwd(gu.data_ptr(), x.data_ptr(), 0.0, gx.data_ptr(), x.numel(), torch.cuda.current_stream().cuda_stream), "relu_dropout_bwd")
        return gx
    def fused():
        return dense.gemm_x6(G, planes, d, None, mask_y=y, p_mask=0.0, sgn_x=x)
    assert torch.equal(pair(), fused())
    a, b = timed(pair), timed(fused)
    print(f"[{n}, {d}] x [{d}, {d}]: GEMM + relu backward {a * 1e3:8.1f} us   one launch {b * 1e3:8.1f} us   ({(a - b) * 1e3:+.1f})", flush=True)
