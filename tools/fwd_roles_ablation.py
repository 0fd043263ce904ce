#!/usr/bin/env python
"""Ablation of fused_linear_fwd_roles_kernel (csrc/fused_fwd2.hip) at [1M,128] x [128,128], LayerNorm + dropout in + relu/dropout
out + mask: without the barriers (timing only), without the MFMAs, without the y stores, with segment timing (the symmetric
kernel it replaced at K = N = 128 is no longer reachable there since ABI 9: no getenv dispatch).  Run on the GPU box: python tools/fwd_roles_ablation.py [--light]"""
import ctypes, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
src = [os.path.join(ROOT, "allset_amd", "csrc", f) for f in ("fused_mlp.hip", "fused_fwd2.hip", "abi.hip")]
dev = torch.device("cuda:0")
n, d = 1_000_000, 128
x = torch.relu(torch.randn(n, d, device=dev)) * (torch.rand(n, d, device=dev) > 0.5)        # bench-like activations: mostly zeros
W = torch.randn(d, d, device=dev) / d ** 0.5; b = torch.randn(d, device=dev)
gam = torch.ones(d, device=dev); bet = torch.zeros(d, device=dev)
y = torch.empty(n, d, device=dev); st = torch.empty(n, 2, device=dev)
mask = torch.empty((n + 15) // 16 * 2 * 32, dtype=torch.int32, device=dev)
P, I64, F, U64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64, ctypes.c_int
light = "--light" in sys.argv
# --combo ln,din,dout,mask (each 0/1): which prologue / epilogue pieces the timed call has (default 1,1,1,1; --light = 1,0,0,0)
combo = [int(v) for v in sys.argv[sys.argv.index("--combo") + 1].split(",")] if "--combo" in sys.argv else ([1, 0, 0, 0] if light else [1, 1, 1, 1])
variants = [("roles: full", [])] if "--combo" in sys.argv else [("roles: full", []), ("roles: no barriers", ["-DALLSET_ABL5_NOBAR"]), ("roles: no MFMA", ["-DALLSET_ABL5_NOMFMA"]),
            ("roles: no stores", ["-DALLSET_ABL5_NOSTORE"]), ("roles: no MFMA, no stores", ["-DALLSET_ABL5_NOMFMA", "-DALLSET_ABL5_NOSTORE"]),
            ("roles: segment timing", ["-DALLSET_ABL5_TIMING"])]
variants += [(a, a.split()) for a in sys.argv[1:] if a.startswith("-D")]
for name, flags in variants:
    so = f"/tmp/fwdroles_{abs(hash(name))}.so"
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-shared", "-fPIC",
                    "-I", os.path.join(ROOT, "include"), "-o", so] + flags + src, check=True)
    lib = ctypes.CDLL(so)
    fn = lib.allset_fused_linear_fwd
    fn.argtypes = [P, I64, P, P, F, I, F, U64, P, P, I, F, U64, P, I64, P, I64, I64, I64, P, P, P, P, P, P]
    lib.allset_last_error.restype = ctypes.c_char_p
    def run():
        ln, din, dout, mk = combo
        rc = fn(x.data_ptr(), d, gam.data_ptr() if ln else None, bet.data_ptr() if ln else None, 1e-5, 0, 0.5 if din else 0.0, 11,
                W.data_ptr(), b.data_ptr(), 1 if (dout or mk) else 0, 0.5 if dout else 0.0, 12, y.data_ptr(), d, st.data_ptr(), n, d, d, None,
                mask.data_ptr() if mk else None, None, None, None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.allset_last_error()
    run(); torch.cuda.synchronize(); ts = []
    for _ in range(20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); run(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    print(f"{name:40s} {statistics.median(ts):.3f} ms", flush=True)
    if "-DALLSET_ABL5_TIMING" in flags:
        t = y[0, :8].tolist()
        stages = (n + 31) // 32 / 256
        print("   cycles per 32-row stage (wave 0 = vector, wave 8 = matrix, workgroup 0): "
              f"V S0 {t[0] / stages:.0f}, V E {t[1] / stages:.0f}, V wait {t[2] / stages:.0f}, V other {t[3] / stages:.0f} | "
              f"M S1 {t[4] / stages:.0f}, M wait {t[6] / stages:.0f}, M other {t[7] / stages:.0f}")
