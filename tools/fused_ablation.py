#!/usr/bin/env python
"""Ablation of fused_linear_fwd_kernel (cdna_hip_programming.md, 'ablate before optimising'): builds variants of
csrc/fused_mlp.hip with the global loads / the stores / the MFMAs removed (values kept live) and times each."""
import ctypes, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
src = [os.path.join(ROOT, "allset_amd", "csrc", f) for f in ("fused_mlp.hip", "abi.hip")]
dev = torch.device("cuda:0")
n, d = 1_000_000, 128
x = torch.randn(n, d, device=dev); W = torch.randn(d, d, device=dev) / d ** 0.5; b = torch.randn(d, device=dev)
y = torch.empty(n, d, device=dev); st = torch.rand(n, 2, device=dev)
P, I64, F, U64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64, ctypes.c_int
for name, flags in (("full", []), ("no-load", ["-DALLSET_ABLATE_NOLOAD"]), ("no-store", ["-DALLSET_ABLATE_NOSTORE"]),
                    ("no-load no-store", ["-DALLSET_ABLATE_NOLOAD", "-DALLSET_ABLATE_NOSTORE"]),
                    ("no-mfma", ["-DALLSET_ABLATE_NOMFMA"]),
                    ("no-mfma no-store", ["-DALLSET_ABLATE_NOMFMA", "-DALLSET_ABLATE_NOSTORE"]),
                    ("no-mfma no-load", ["-DALLSET_ABLATE_NOMFMA", "-DALLSET_ABLATE_NOLOAD"])):
    so = f"/tmp/fused_{name.replace(' ', '_')}.so"
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + flags + src, check=True)
    lib = ctypes.CDLL(so)
    fn = lib.allset_fused_linear_fwd
    fn.argtypes = [P, I64, P, P, F, I, F, U64, P, P, I, F, U64, P, I64, P, I64, I64, I64, P, P, P, P, P, P]
    fb = lib.allset_fused_linear_bwd
    fb.argtypes = [P, I64, P, I64, F, P, P, I64, P, P, I, F, U64, P, I64, P, I64, I64, I64, I64, P, P, P, I64, P, P, P]
    npart = ctypes.c_int64(0)
    lib.allset_fused_linear_bwd_partials.argtypes = [I64, ctypes.POINTER(I64)]
    lib.allset_fused_linear_bwd_partials(n, ctypes.byref(npart))
    parts = torch.empty(npart.value * 2 * d, device=dev)
    gam = torch.ones(d, device=dev); gxo = torch.empty(n, d, device=dev)
    def run_fwd():
        rc = fn(x.data_ptr(), d, None, None, 1e-5, 0, 0.0, 0, W.data_ptr(), b.data_ptr(), 0, 0.0, 0, y.data_ptr(), d, None, n, d, d, None, None, None, None, None,
                torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    def run_bwd():      # LayerNorm backward epilogue, no dropout, no y mask (the K1 shape of the bench)
        rc = fb(x.data_ptr(), d, None, 0, 0.0, W.data_ptr(), y.data_ptr(), d, st.data_ptr(), gam.data_ptr(), 0, 0.0, 0,
                gxo.data_ptr(), d, parts.data_ptr(), npart.value, n, d, d, None, None, None, 0, None, None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.allset_last_error()
    lib.allset_last_error.restype = ctypes.c_char_p
    out = []
    for run in (run_fwd, run_bwd):
        run(); torch.cuda.synchronize(); ts = []
        for _ in range(20):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); run(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        out.append(statistics.median(ts))
    print(f"{name:18s} fwd {out[0]:.3f} ms   bwd(LN) {out[1]:.3f} ms", flush=True)
