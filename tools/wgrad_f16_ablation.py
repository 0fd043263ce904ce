#!/usr/bin/env python
"""Ablation of wgrad_f16_kernel (csrc/wgrad_f16.hip): the shipped kernel against builds without the MFMAs, without the loop's barriers,
and with the stage maximum posted from the register set that has had two half trips to land (results wrong, timing only), each timed at
[1M, 256] x [256, 256] with the activation mask and an input dropout of 0.5 -- the bench's d = 256 call.
Run on the GPU box: python tools/wgrad_f16_ablation.py [--only <name>] [-DFLAG ...]"""
import ctypes, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
src = [os.path.join(ROOT, "allset_amd", "csrc", f) for f in ("wgrad_f16.hip", "abi.hip")]
dev = torch.device("cuda:0")
n, O, I = 1_000_000, 256, 256
x = torch.randn(n, I, device=dev); gy = torch.randn(n, O, device=dev); st = torch.rand(n, 2, device=dev) + 0.5
gam = torch.ones(I, device=dev); bet = torch.zeros(I, device=dev)
mask = torch.randint(-2**31, 2**31 - 1, ((n + 15) // 16 * (O // 64) * 32,), dtype=torch.int32, device=dev)
P, I64, F, U64, Ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64, ctypes.c_int
variants = [("full", []), ("no MFMA", ["-DALLSET_ABL_WF_NOMFMA"]), ("no barriers", ["-DALLSET_ABL_WF_NOBAR"]),
            ("both waves of a SIMD in the same phase order", ["-DALLSET_ABL_WF_SAMEPHASE"]), ("post from the older set", ["-DALLSET_ABL_WF_POST"]), ("no MFMA, post from the older set", ["-DALLSET_ABL_WF_NOMFMA", "-DALLSET_ABL_WF_POST"])]
variants += [("segment timing", ["-DALLSET_ABL_WF_TIMING"]), ("segment timing, no MFMA", ["-DALLSET_ABL_WF_TIMING", "-DALLSET_ABL_WF_NOMFMA"])]
variants += [(a, a.split()) for a in sys.argv[1:] if a.startswith("-D")]
if "--only" in sys.argv:
    variants = [v for v in variants if sys.argv[sys.argv.index("--only") + 1] == v[0]]
for name, flags in variants:
    so = f"/tmp/wgradf16_{abs(hash(name))}.so"
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-shared", "-fPIC",
                    "-I", os.path.join(ROOT, "include"), "-o", so] + flags + src, check=True)
    lib = ctypes.CDLL(so)
    fn = lib.allset_wgrad_f16x3
    fn.argtypes = [P, I64, P, F, P, I64, P, P, P, Ci, F, U64, P, I64, Ci, I64, I64, I64, I64, P, P]
    ns = I64(0)
    lib.allset_wgrad_f16x3_slices.argtypes = [I64, I64, I64, ctypes.POINTER(I64)]
    lib.allset_wgrad_f16x3_slices(n, O, I, ctypes.byref(ns))
    M = O * I + O
    part = torch.empty(ns.value, M, device=dev)
    def run():
        rc = fn(gy.data_ptr(), O, mask.data_ptr(), 0.5, x.data_ptr(), I, st.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1, 0.5, 7,
                part.data_ptr(), M, 1, ns.value, n, O, I, None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = statistics.median(ts)
    print(f"{name:40s} {ms*1e3:7.0f} us  ({n * (O + I) * 4 / ms / 1e6:.0f} GB/s)  slices {ns.value}", flush=True)
    if "-DALLSET_ABL_WF_TIMING" in flags:
        t = part[0, :16].tolist()
        for w, o in (("wave 0 (MFMA first)", 0), ("wave 4 (split first)", 8)):
            tot = sum(t[o:o + 5]) or 1.0
            print(f"    {w}: MFMA {t[o]/tot:.2f}  split+store {t[o+1]/tot:.2f}  load issue {t[o+2]/tot:.2f}  post {t[o+3]/tot:.2f}  barrier {t[o+4]/tot:.2f}   ({tot/1e6:.2f} Mcycles)", flush=True)
