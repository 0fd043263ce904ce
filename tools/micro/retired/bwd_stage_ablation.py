#!/usr/bin/env python
"""(stage kernel) Ablation of fused_linear_bwd_stage_kernel (csrc/fused_bwd3.hip): variants with L2-hot operands (every pair re-reads one
chunk: no HBM latency), without the workgroup barriers (results wrong, timing only), without the MFMAs, without the gx stores,
each timed at [1M,128] x [128,128] next to the one-wave kernel (ALLSET_BWD_PAIR=0).  Run on the GPU box:
python tools/bwd_pair_ablation.py [--light]"""
import ctypes, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ["ALLSET_BWD_ROLES"] = "0"
src = [os.path.join(ROOT, "allset_amd", "csrc", f) for f in ("fused_bwd.hip", "fused_bwd2.hip", "fused_bwd3.hip", "fused_bwd4.hip", "fused_bwd5.hip", "abi.hip")]
dev = torch.device("cuda:0")
n, d = 1_000_000, 128
x = torch.randn(n, d, device=dev); W = torch.randn(d, d, device=dev) / d ** 0.5
gy = torch.randn(n, d, device=dev); st = torch.rand(n, 2, device=dev) + 0.5
gam = torch.ones(d, device=dev); bet = torch.zeros(d, device=dev); gx = torch.empty(n, d, device=dev)
mask = torch.randint(-2**31, 2**31 - 1, ((n + 15) // 16 * 2 * 32,), dtype=torch.int32, device=dev)
P, I64, F, U64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64, ctypes.c_int
variants = [("stage: full", []), ("stage: no barriers", ["-DALLSET_ABL3_NOBAR"]), ("stage: no MFMA", ["-DALLSET_ABL3_NOMFMA"]),
            ("stage: no stores", ["-DALLSET_ABL3_NOSTORE"]), ("stage: no MFMA, no barriers", ["-DALLSET_ABL3_NOMFMA", "-DALLSET_ABL3_NOBAR"]),
            ("stage: phase timing", ["-DALLSET_ABL3_TIMING"]), ("stage: phase timing, no MFMA", ["-DALLSET_ABL3_TIMING", "-DALLSET_ABL3_NOMFMA"])]
variants += [(a, a.split()) for a in sys.argv[1:] if a.startswith("-D")]
light = "--light" in sys.argv
os.environ["ALLSET_BWD_STAGE"] = "1"
for name, flags in variants + [("one wave per SIMD (fused_bwd.hip)", None)]:
    if flags is None:
        os.environ["ALLSET_BWD_STAGE"] = "0"
        flags = []
    so = f"/tmp/bwdstage_{abs(hash(name))}.so"
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-shared", "-fPIC",
                    "-I", os.path.join(ROOT, "include"), "-o", so] + flags + src, check=True)
    lib = ctypes.CDLL(so)
    fn = lib.allset_fused_linear_bwd_all
    fn.argtypes = [P, I64, P, F, P, P, I64, P, P, P, I, F, U64, P, I64, P, P, P, I64, I64, I64, I64, P, P, I64, I64, P]
    ns = ctypes.c_int64(0)
    lib.allset_fused_linear_bwd_all_slices_for.argtypes = [I64, I64, I64, I, ctypes.POINTER(I64)]
    lib.allset_fused_linear_bwd_all_slices_for(n, d, d, 0, ctypes.byref(ns))
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
    print(f"{name:40s} {statistics.median(ts):.3f} ms", flush=True)
    if "-DALLSET_ABL3_TIMING" in flags:
        t = pw[:8].tolist()
        names = ["wait B0", "S0 ga split", "wait B1", "S1 bwd-data", "wait B2", "S2 epilogue+u", "wait B3", "S3 wgrad"]
        tot = sum(t)
        print("   cycles per stage, wave 0 of workgroup 0 (%d stages):" % ((n + 63) // 64 // 256 + 1), ", ".join(f"{nm} {v / ((n + 63) // 64 / 256):.0f}" for nm, v in zip(names, t)), f"| total {tot / ((n + 63) // 64 / 256):.0f}")
