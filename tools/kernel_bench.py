#!/usr/bin/env python
"""Per-kernel achieved algorithmic GB/s (HIP events, median of N launches) across degree distributions.
usage: python tools/kernel_bench.py [--n 1000000] [--d 128] [--heads 4] [--dists fixed,poisson,zipf]"""
import argparse, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allset_amd import Incidence, ops
from allset_amd.synthetic import random_hypergraph

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--heads", type=int, default=4)
ap.add_argument("--degree", type=int, default=16)
ap.add_argument("--dists", default="fixed,poisson,zipf")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
args = ap.parse_args()
dev = torch.device("cuda:0")


def timeit(fn, iters=args.iters):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts)


def report(name, ms, nbytes):
    print(f"  {name:34s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.0f} GB/s  ({nbytes / ms / 1e6 / 8000:5.1%} of 8 TB/s)", flush=True)


n, d, H = args.n, args.d, args.heads
DT = torch.float32 if args.dtype == "f32" else torch.bfloat16
ES = 4 if args.dtype == "f32" else 2
for dist in args.dists.split(","):
    hg = random_hypergraph(n, n, args.degree, seed=11, device=dev, dist=dist)
    inc = Incidence.from_edge_index(hg.edge_index, n_src=n, n_dst=n)
    nnz = hg.nnz
    rp = inc.by_dst.rowptr
    deg = (rp[1:] - rp[:-1])
    print(f"[{dist}] nnz={nnz} max hyperedge size={int(deg.max())} max vertex degree={int((inc.by_src.rowptr[1:]-inc.by_src.rowptr[:-1]).max())}")
    x = torch.randn(n, d, device=dev).to(DT)
    w = torch.rand(nnz, device=dev) + 0.5
    alpha = torch.randn(n, H, device=dev)
    pass_bytes = nnz * (ES * d + 4) + (n + 1) * 4 + n * ES * d
    for label, csr in (("V->E (by hyperedge)", inc.by_dst), ("E->V (by vertex)", inc.by_src)):
        report(f"segreduce sum {label}", timeit(lambda: ops.segreduce(0, csr.rowptr, csr.col, None, x, n, variant=1)), pass_bytes)
        if csr.row_order is not None:
            report(f"  .. long rows first", timeit(lambda: ops.segreduce(0, csr.rowptr, csr.col, None, x, n, variant=1, row_order=csr.row_order)), pass_bytes)
        report(f"  .. short-row kernel", timeit(lambda: ops.segreduce(0, csr.rowptr, csr.col, None, x, n, variant=2)), pass_bytes)
    csr, T = inc.by_dst, inc.by_src
    report("segreduce sum weighted", timeit(lambda: ops.segreduce(0, csr.rowptr, csr.col, w, x, n)), pass_bytes + nnz * 4)
    report("segreduce mean", timeit(lambda: ops.segreduce(1, csr.rowptr, csr.col, None, x, n)), pass_bytes)
    report("segreduce max (+argext)", timeit(lambda: ops.segreduce(2, csr.rowptr, csr.col, None, x, n, want_arg=True)), pass_bytes + n * 4 * d)  # + int32 argext
    if args.dtype == "f32":
        out, arg = ops.segreduce(2, csr.rowptr, csr.col, None, x, n, want_arg=True)
        posT = inc.pos_dst_of_src()
        report("segmax_bwd", timeit(lambda: ops.segmax_bwd(T.rowptr, T.col, posT, None, arg, x, n)), nnz * (8 * d + 8) + (n + 1) * 4 + n * 4 * d)
        gO = torch.randn(n, d, device=dev)
        report("sddmm_rowdot sum", timeit(lambda: ops.sddmm_rowdot(0, csr.rowptr, csr.col, x, gO, None)), nnz * (4 * d + 8) + (n + 1) * 4 + n * 4 * d)
        report("sddmm_rowdot max", timeit(lambda: ops.sddmm_rowdot(2, csr.rowptr, csr.col, x, gO, arg)), nnz * (4 * d + 8) + (n + 1) * 4 + n * 8 * d)
        del out, arg, gO
    report("  .. pma_fwd short-row kernel", timeit(lambda: ops.pma_fwd(csr.rowptr, csr.col, alpha, x, H, 0.2, n, variant=2)), nnz * (ES * d + 4 + 4 * H) + (n + 1) * 4 + n * (ES * d + 8 * H))
    if csr.row_order is not None:
        report("  .. pma_fwd long rows first", timeit(lambda: ops.pma_fwd(csr.rowptr, csr.col, alpha, x, H, 0.2, n, variant=1, row_order=csr.row_order)), nnz * (ES * d + 4 + 4 * H) + (n + 1) * 4 + n * (ES * d + 8 * H))
    report("pma_fwd", timeit(lambda: ops.pma_fwd(csr.rowptr, csr.col, alpha, x, H, 0.2, n, variant=1)), nnz * (ES * d + 4 + 4 * H) + (n + 1) * 4 + n * (ES * d + 8 * H))
    o, m, l = ops.pma_fwd(csr.rowptr, csr.col, alpha, x, H, 0.2, n)
    g = torch.randn(n, d, device=dev).to(DT)
    report("pma_bwd_stats", timeit(lambda: ops.pma_bwd_stats(o, g, m, l)), n * (2 * ES * d + 16 * H))
    st = ops.pma_bwd_stats(o, g, m, l)
    report("  .. pma_bwd_src short-row kernel", timeit(lambda: ops.pma_bwd_src(T.rowptr, T.col, alpha, x, g, st, 0.2, variant=2)), nnz * (ES * d + 4 + 8 * H) + (n + 1) * 4 + n * (2 * ES * d + 8 * H))
    report("pma_bwd_src", timeit(lambda: ops.pma_bwd_src(T.rowptr, T.col, alpha, x, g, st, 0.2, variant=1)), nnz * (ES * d + 4 + 8 * H) + (n + 1) * 4 + n * (2 * ES * d + 8 * H))
    del o, m, l, g, st, x, w, inc, hg
    torch.cuda.empty_cache()
