#!/usr/bin/env python
"""Ablation of fused_linear_bwd_all_kernel<128,128,LN,dropout,relu,mask> (cdna_hip_programming.md, 'ablate before
optimising'): builds single-kernel variants of csrc/fused_bwd.hip with the global loads / the gx stores / either MFMA phase
removed (values kept live) and times each at [1M,128] x [128,128].  Run on the GPU box: python tools/bwd_all_ablation.py"""
import ctypes, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ["ALLSET_BWD_ROLES"] = "0"
os.environ["ALLSET_BWD_STAGE"] = "0"
src = [os.path.join(ROOT, "allset_amd", "csrc", f) for f in ("fused_bwd.hip", "fused_bwd2.hip", "fused_bwd3.hip", "fused_bwd4.hip", "abi.hip")]
dev = torch.device("cuda:0")
n, d = 1_000_000, 128
x = torch.randn(n, d, device=dev); W = torch.randn(d, d, device=dev) / d ** 0.5
gy = torch.randn(n, d, device=dev); st = torch.rand(n, 2, device=dev) + 0.5
gam = torch.ones(d, device=dev); bet = torch.zeros(d, device=dev); gx = torch.empty(n, d, device=dev)
mask = torch.randint(-2**31, 2**31 - 1, ((n + 15) // 16 * 2 * 32,), dtype=torch.int32, device=dev)
P, I64, F, U64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64, ctypes.c_int
variants = [("full", []), ("no-load", ["-DALLSET_ABL_NOLOAD"]), ("no-store", ["-DALLSET_ABL_NOSTORE"]),
            ("no-wgrad-mfma", ["-DALLSET_ABL_NOWG"]), ("no-bwd-mfma", ["-DALLSET_ABL_NOBD"]),
            ("no-mfma", ["-DALLSET_ABL_NOWG", "-DALLSET_ABL_NOBD"]),
            ("no-mfma no-load no-store", ["-DALLSET_ABL_NOWG", "-DALLSET_ABL_NOBD", "-DALLSET_ABL_NOLOAD", "-DALLSET_ABL_NOSTORE"]),
            ("no-load no-store", ["-DALLSET_ABL_NOLOAD", "-DALLSET_ABL_NOSTORE"])]
light = "--light" in sys.argv
if light:
    sys.argv.remove("--light")
    variants = [(nm, fl + ["-DALLSET_ABL_LIGHT"]) for nm, fl in variants]
if len(sys.argv) > 1:
    variants = ([v for v in variants if v[0] in sys.argv[1:]] +
                [(a, a.split() + (["-DALLSET_ABL_LIGHT"] if light else [])) for a in sys.argv[1:] if a.startswith("-D")])
for name, flags in variants:
    so = f"/tmp/bwdall_{abs(hash(name))}.so"
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-shared", "-fPIC", "-DALLSET_ABL_SINGLE",
                    "-o", so] + flags + src, check=True)
    lib = ctypes.CDLL(so)
    fn = lib.allset_fused_linear_bwd_all
    fn.argtypes = [P, I64, P, F, P, P, I64, P, P, P, I, F, U64, P, I64, P, P, P, I64, I64, I64, I64, P, P, I64, I64, P]
    ns = ctypes.c_int64(0)
    lib.allset_fused_linear_bwd_all_slices.argtypes = [I64, ctypes.POINTER(I64)]
    lib.allset_fused_linear_bwd_all_slices(n, ctypes.byref(ns))
    pw = torch.empty(ns.value * d * d, device=dev); pb = torch.empty(ns.value * d, device=dev); pl = torch.empty(ns.value * 2 * d, device=dev)
    lib.allset_last_error.restype = ctypes.c_char_p
    def run():
        rc = fn(gy.data_ptr(), d, None if light else mask.data_ptr(), 0.0 if light else 0.5, W.data_ptr(), x.data_ptr(), d, st.data_ptr(), gam.data_ptr(), bet.data_ptr(),
                0 if light else 1, 0.0 if light else 0.5, 77,
                gx.data_ptr(), d, pl.data_ptr(), pw.data_ptr(), pb.data_ptr(), ns.value, n, d, d, None, None, 0, 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.allset_last_error()
    run(); torch.cuda.synchronize(); ts = []
    for _ in range(20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); run(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    print(f"{name:28s} {statistics.median(ts):.3f} ms", flush=True)
