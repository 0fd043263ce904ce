#!/usr/bin/env python
"""Would a chunk pipeline that runs one chunk's dense kernels BESIDE another chunk's aggregation pay?  
Launches the headline's gather kernel and one of its dense kernels on two streams at once, on independent operands, and compares
the pair's wall time with the sum of the two alone.  The gather kernel is HBM-bound (0.93 of peak), the dense kernels vector /
matrix bound at ~0.5 of peak: if they overlapped perfectly the pair would take max(a, b).
Run on the GPU box: python tools/stream_overlap_probe.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allset_amd import Incidence, dense, ops
from allset_amd.synthetic import random_hypergraph

dev = torch.device("cuda:0")
n, d = 1_000_000, 128
hg = random_hypergraph(n, n, 16, seed=3, device=dev)
inc = Incidence.from_edge_index(hg.edge_index, n_src=n, n_dst=n)
x = torch.randn(n, d, device=dev)
xa = torch.relu(torch.randn(n, d, device=dev))
W = torch.randn(d, d, device=dev) / d ** 0.5
b = torch.zeros(d, device=dev); g = torch.ones(d, device=dev)
gy = torch.randn(n, d, device=dev)
y, st = dense.fused_linear_fwd(xa, W, b, g, b, 1e-5, True, 0.5, 7, True, 0.5, 8)
mask = torch.empty(dense.activation_mask_words(n, d), dtype=torch.int32, device=dev)
dense.fused_linear_fwd(xa, W, b, g, b, 1e-5, True, 0.5, 7, True, 0.5, 8, None, mask)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def agg():
    ops.segreduce(0, inc.by_dst.rowptr, inc.by_dst.col, None, x, n)


def fwd():
    dense.fused_linear_fwd(xa, W, b, g, b, 1e-5, True, 0.5, 7, True, 0.5, 8, None, mask)


def bwd():
    dense.fused_linear_bwd_all(gy, mask, 0.5, W, xa, st, g, b, True, 0.5, 7)


def timed(fns, iters=15):
    def once():
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for stream, fn in zip((sA, sB), fns):
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                fn()
        for stream in (sA, sB)[:len(fns)]:
            torch.cuda.current_stream().wait_stream(stream)
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e)
    once(); once()
    return statistics.median(once() for _ in range(iters))


ta, tf, tb = timed([agg]), timed([fwd]), timed([bwd])
print(f"alone: segreduce {ta:.3f} ms, fused_linear_fwd (heavy) {tf:.3f} ms, fused_linear_bwd_all (heavy) {tb:.3f} ms")
for name, fn, t in (("fwd", fwd, tf), ("bwd", bwd, tb)):
    both = timed([agg, fn])
    both2 = timed([fn, agg])
    print(f"segreduce || {name}: {both:.3f} ms (other launch order {both2:.3f}); serial sum {ta + t:.3f}, max {max(ta, t):.3f} "
          f"-> {100 * (ta + t - min(both, both2)) / (ta + t):.0f} % of the serial time saved")
both = timed([fwd, bwd])
print(f"fwd || bwd: {both:.3f} ms; serial sum {tf + tb:.3f}")
