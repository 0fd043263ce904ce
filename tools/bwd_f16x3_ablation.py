#!/usr/bin/env python
"""Ablation of fused_linear_bwd_f16x3_kernel (csrc/fused_bwd6.hip; tools/bwd_roles_ablation.py is the same for fused_bwd4.hip): variants without the workgroup barriers (results wrong,
timing only), without the MFMAs, without the gx stores, with per-segment cycle counters, each timed at [1M,128] x [128,128].
Run on the GPU box: python tools/bwd_f16x3_ablation.py [--light] [--only <substring>] [-DFLAG ...]
(the comparison arms of rounds 2-3 -- the pair / stage / three-waves kernels and their ablation scripts -- live in
the git history before round 5, tools/micro/retired/, as they were when they lost their A/B; the library no longer builds them.)"""
import ctypes, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
src = [os.path.join(ROOT, "allset_amd", "csrc", f) for f in ("fused_bwd.hip", "fused_bwd4.hip", "fused_bwd6.hip", "abi.hip")]
if os.environ.get("ALLSET_BWD6_SRC"):          # A/B against another version of the kernel file
    src[2] = os.environ["ALLSET_BWD6_SRC"]
if os.environ.get("ALLSET_BWD_SRC"):           # (and the matching dispatch file, when the launch signature changed in between)
    src[0] = os.environ["ALLSET_BWD_SRC"]
dev = torch.device("cuda:0")
n, d = 1_000_000, 128
x = torch.randn(n, d, device=dev); W = torch.randn(d, d, device=dev) / d ** 0.5
gy = torch.randn(n, d, device=dev); st = torch.rand(n, 2, device=dev) + 0.5
gam = torch.ones(d, device=dev); bet = torch.zeros(d, device=dev); gx = torch.empty(n, d, device=dev)
mask = torch.randint(-2**31, 2**31 - 1, ((n + 15) // 16 * 2 * 32,), dtype=torch.int32, device=dev)
P, I64, F, U64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64, ctypes.c_int
variants = [("f16x3: full", []), ("bf16x6 roles kernel (fused_bwd4)", ["-DALLSET_NO_F16X3"]), ("f16x3: no barriers", ["-DALLSET_ABL6_NOBAR"]), ("f16x3: no MFMA", ["-DALLSET_ABL6_NOMFMA"]),
            ("f16x3: no stores", ["-DALLSET_ABL6_NOSTORE"]), ("f16x3: no MFMA, no barriers", ["-DALLSET_ABL6_NOMFMA", "-DALLSET_ABL6_NOBAR"]),
            ("f16x3: segment timing", ["-DALLSET_ABL6_TIMING"]), ("f16x3: segment timing, no MFMA", ["-DALLSET_ABL6_TIMING", "-DALLSET_ABL6_NOMFMA"])]
variants += [(a, a.split()) for a in sys.argv[1:] if a.startswith("-D")]
light = "--light" in sys.argv
if "--only" in sys.argv:
    key = sys.argv[sys.argv.index("--only") + 1]
    variants = [v for v in variants if key in v[0]]
for name, flags in variants:
    so = f"/tmp/bwdf16x3_{abs(hash(name))}.so"
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950"] + ([] if "-DSLP" in flags else ["-fno-slp-vectorize"]) + ["-shared", "-fPIC",
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
    if "-DALLSET_ABL6_TIMING" in flags:
        torch.cuda.synchronize()
        t = pw[:8].tolist()
        stages = (n + 31) // 32 / 256
        names = ["V S0+S2a", "V wait", "V S2b", "V wait", "M S1", "M wait", "M S3", "M wait"]
        print("   cycles per 32-row stage (wave 0 = vector, wave 8 = matrix, workgroup 0):", ", ".join(f"{nm} {v / stages:.0f}" for nm, v in zip(names, t)),
              f"| V total {sum(t[:4]) / stages:.0f}, M total {sum(t[4:]) / stages:.0f}")
