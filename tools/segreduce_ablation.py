#!/usr/bin/env python
"""Why does segreduce_kernel reach 0.93 of 8 TB/s at d = 128 and 0.80-0.85 at d = 256 (1M rows, degree 16)?  Two candidates:
the kernel (at d = 256 a row takes the whole wave, NS = 1: a degree-16 row is two dependent batches of 8 gathers instead of one
batch of 16) or the machine (FETCH_SIZE counts what leaves L2, not what the 256-MB Infinity Cache behind it serves: a 512-MB
source table can be half resident there, a 1-GB table a quarter).  The tool separates them: the same launch over source tables
of 256 MB ... 2 GB at both widths, and the kernel rebuilt with 16 gathers per batch at NS = 1 / with half-wave rows (two column
passes) at d = 256.  Run on the GPU box: python tools/segreduce_ablation.py"""
import ctypes, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from allset_amd import synthetic
from allset_amd.incidence import Incidence

src = [os.path.join(ROOT, "allset_amd", "csrc", f) for f in ("segreduce.hip", "abi.hip")]
dev = torch.device("cuda:0")
P, I64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
variants = [("shipped (8 gathers per batch, whole-wave rows)", []),
            ("16 gathers per batch at NS = 1", ["-DALLSET_SEG_UNROLL_ONE_SLOT=16"]),
            ("half-wave rows, two column passes", ["-DALLSET_SEG_MAX_LPR=32"])]
# (source rows, target rows, d)
shapes = [(1_000_000, 1_000_000, 128), (2_000_000, 2_000_000, 128), (500_000, 500_000, 256), (1_000_000, 1_000_000, 256)]
if "--tables" in sys.argv:       # 1M target rows of degree 16 at d = 128 over source tables of 31 MB ... 2 GB: what the cache behind L2 is worth
    shapes = [(n_s, 1_000_000, 128) for n_s in (62_500, 125_000, 250_000, 375_000, 500_000, 750_000, 1_000_000, 2_000_000, 4_000_000)]
libs = []
procs = []
for k, (name, flags) in enumerate(variants):
    so = f"/tmp/segabl_{k}.so"
    procs.append(subprocess.Popen(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-shared", "-fPIC",
                                   "-I", os.path.join(ROOT, "include"), "-o", so] + flags + src))
    libs.append(so)
for p in procs:
    assert p.wait() == 0
for n, n_e, d in shapes:
    hg = synthetic.random_hypergraph(n, n_e, 16, seed=3, device=dev)
    ei = hg.edge_index.clone()
    ei[1] -= ei[1].min()
    inc = Incidence.from_edge_index(ei, n_src=n)
    csr = inc.by_dst
    x = torch.randn(n, d, device=dev)
    out = torch.empty(csr.rowptr.numel() - 1, d, device=dev)
    n_t = out.shape[0]
    alg = inc.nnz * (4 * d + 4) + (n_t + 1) * 4 + n_t * 4 * d
    ref = None
    print(f"n_s = {n}, n_t = {n_t}, d = {d}, nnz = {inc.nnz}: source table {x.numel() * 4 / 2**20:.0f} MB, algorithmic bytes {alg / 1e9:.2f} GB", flush=True)
    for (name, flags), so in zip(variants, libs):
        if d <= 128 and flags:
            continue                       # (neither switch changes the d = 128 kernel)
        lib = ctypes.CDLL(so)
        fn = lib.allset_segreduce_fwd
        fn.argtypes = [I, I, P, P, P, P, I64, P, I64, P, I64, I64, I64, P]
        lib.allset_last_error.restype = ctypes.c_char_p
        def run():
            rc = fn(0, 0, csr.rowptr.data_ptr(), csr.col.data_ptr(), None, x.data_ptr(), d, out.data_ptr(), d, None, n_t, n, d,
                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0, lib.allset_last_error()
        run(); torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        else:
            torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
        ts = []
        for _ in range(20):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); run(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        t = statistics.median(ts)
        print(f"   {name:50s} {t:.3f} ms  {alg / t / 1e9:.2f} TB/s = {alg / t / 1e9 / 8:.3f} of 8 TB/s", flush=True)
    del x, out, inc, csr, hg, ei
    torch.cuda.empty_cache()
