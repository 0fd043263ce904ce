#!/usr/bin/env python
"""Does blocking the SOURCE table make narrow-row gathers faster?  (round 4, the N = 8 column layer: 128M gathers of 64-byte rows
from a 512 MB table run at ~48 G requests/s = 2.65 ms per pass, profiles/r01_colshard_kernels.txt.)

Hypothesis: the bound is DRAM row activations (one per random 64-byte or 128-byte access); a table slice that stays in the
256 MiB Infinity Cache takes no activations.  The probe times the short-row segreduce kernel over n_t = 8M target rows with
`members` uniformly drawn sources out of n_s, d = 16 fp32 (64-byte rows), for n_s in {8M (the layer's pass), 4M, 2M, 1M, 512k}
with members = 16 * n_s / 8M (= the share of a 16-member hyperedge that falls into a source block of that size): 8M/n_s such
passes make one full pass.  Also d = 32 (128-byte rows, the N = 4 layer) and the single-GPU shape with a strided table
(d = 132 of pitch 160: PMA's logits in a row tail).
Run on the GPU box: python tools/mall_block_probe.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allset_amd import ops

dev = torch.device("cuda:0")


def timed(fn, iters=8):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return statistics.median(ts)


def probe(n_t, n_s, members, d, variant, ld=None, label=""):
    g = torch.Generator(device=dev).manual_seed(n_s + members)
    col = torch.randint(0, n_s, (n_t * members,), device=dev, generator=g, dtype=torch.int32)
    rowptr = (torch.arange(n_t + 1, device=dev, dtype=torch.int64) * members).to(torch.int32)
    if ld is None:
        x = torch.randn(n_s, d, device=dev)
    else:
        x = torch.randn(n_s, ld, device=dev)[:, :d]
    ms = timed(lambda: ops.segreduce(0, rowptr, col, None, x, n_t, variant=variant))
    nnz = n_t * members
    print(f"{label:34s} n_t={n_t:>9d} n_s={n_s:>9d} members={members:>2d} d={d:>3d} table={n_s * (ld or d) * 4 / 2**20:7.0f} MiB "
          f"nnz={nnz / 1e6:6.1f}M  {ms:7.3f} ms  {nnz / ms / 1e6:6.1f} G gathers/s  {nnz * d * 4 / ms / 1e6:7.0f} GB/s gathered", flush=True)
    return ms


print("# 64-byte rows (N = 8 column layer); full pass = (8M / n_s) block passes + read-modify-write of the [8M,16] output per extra pass")
full = probe(8_000_000, 8_000_000, 16, 16, 2, label="one pass, whole table")
for n_s, m in ((4_000_000, 8), (2_000_000, 4), (1_000_000, 2), (500_000, 1)):
    t = probe(8_000_000, n_s, m, 16, 2, label=f"source block 1/{8_000_000 // n_s}")
    k = 8_000_000 // n_s
    print(f"      -> {k} block passes = {k * t:.3f} ms (+ {(k - 1) * 2 * 0.512 / 5.0:.3f} ms output read-modify-write at 5 TB/s) vs {full:.3f} ms")
print("# 128-byte rows (N = 4 column layer)")
full4 = probe(4_000_000, 4_000_000, 16, 32, 2, label="one pass, whole table")
for n_s, m in ((2_000_000, 8), (1_000_000, 4)):
    t = probe(4_000_000, n_s, m, 32, 2, label=f"source block 1/{4_000_000 // n_s}")
    print(f"      -> {4_000_000 // n_s} block passes = {4_000_000 // n_s * t:.3f} ms vs {full4:.3f} ms")
print("# single-GPU shape, 512-byte rows; then rows of 528 bytes at pitch 640 (PMA logits in a row tail: 5 lines per gather)")
probe(1_000_000, 1_000_000, 16, 128, 1, label="d=128 contiguous")
probe(1_000_000, 1_000_000, 16, 132, 1, ld=160, label="d=132 of pitch 160")
probe(1_000_000, 1_000_000, 16, 128, 1, ld=160, label="d=128 of pitch 160")
for n_s, m in ((500_000, 8), (250_000, 4)):
    t = probe(1_000_000, n_s, m, 128, 1, label=f"d=128, source block 1/{1_000_000 // n_s}")
