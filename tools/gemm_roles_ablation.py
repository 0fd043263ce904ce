#!/usr/bin/env python
"""Ablation arms of the split-role forward on fp16 planes (csrc/wide_mlp.hip gemm_f16_roles_kernel) at [1M, 256] x [256, 256], the
bench's d = 256 forward (relu -> LayerNorm -> dropout | relu -> dropout, mask out): the plain build, without the MFMAs, without the
prologue's arithmetic, without the epilogue, without the barriers (the arms' results are wrong; only their times mean something).
python tools/gemm_roles_ablation.py --build-only  (build container: leaves the libraries under .abl/), then on the GPU box:
python tools/gemm_roles_ablation.py"""
import ctypes, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
build_only = "--build-only" in sys.argv
if not build_only:
    import torch
src = [os.path.join(ROOT, "allset_amd", "csrc", f) for f in ("wide_mlp.hip", "abi.hip")]
n, K, N = 1_000_000, 256, 256
ARMS = [("timed", ["-DALLSET_ABL_GR_TIMING"]), ("plain", []), ("no MFMA", ["-DALLSET_ABL_GR_NOMFMA"]), ("no prologue arithmetic", ["-DALLSET_ABL_GR_NOSTAGE"]),
        ("no epilogue", ["-DALLSET_ABL_GR_NOEPI"]), ("no barriers", ["-DALLSET_ABL_GR_NOBAR"]),
        ("no MFMA, no prologue, no epilogue", ["-DALLSET_ABL_GR_NOMFMA", "-DALLSET_ABL_GR_NOSTAGE", "-DALLSET_ABL_GR_NOEPI"])]
if build_only:
    procs = []
    for name, flags in ARMS:
        so = os.path.join(ROOT, ".abl", "grabl_" + name.replace(" ", "_").replace(",", "") + ".so")
        os.makedirs(os.path.dirname(so), exist_ok=True)
        procs.append(subprocess.Popen(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-shared", "-fPIC",
                                       "-I", os.path.join(ROOT, "include"), "-o", so] + flags + src))
    for p in procs:
        assert p.wait() == 0
    sys.exit(0)
dev = torch.device("cuda:0")
x = torch.randn(n, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
st = torch.stack([torch.zeros(n, device=dev), torch.ones(n, device=dev)], 1).contiguous()
gam = torch.ones(K, device=dev); bet = torch.zeros(K, device=dev)
P, I64, F, U64, Ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64, ctypes.c_int
for name, flags in ARMS:
    so = os.path.join(ROOT, ".abl", "grabl_" + name.replace(" ", "_").replace(",", "") + ".so")
    lib = ctypes.CDLL(so)
    lib.allset_gemm_f16x3_plane_bytes.restype = I64
    lib.allset_gemm_f16x3_plane_bytes.argtypes = [I64, I64]
    nb = lib.allset_gemm_f16x3_plane_bytes(N, K)
    planes = torch.empty(nb, dtype=torch.uint8, device=dev)
    lib.allset_gemm_f16x3_planes.argtypes = [P, I64, Ci, P, I64, I64, P]
    s = torch.cuda.current_stream().cuda_stream
    assert lib.allset_gemm_f16x3_planes(W.data_ptr(), K, 0, planes.data_ptr(), N, K, s) == 0
    words = (n + 15) // 16 * (N // 64) * 32
    mask = torch.zeros(words, dtype=torch.int32, device=dev)
    y = torch.empty(n, N, device=dev)
    fw = lib.allset_gemm_wide
    fw.argtypes = [Ci, P, I64, P, I64, P, F, Ci, P, P, P, F, U64, P, P, Ci, F, U64, P, P, F, Ci, P, I64, I64, I64, I64, P, P]
    def fwd(light=False):
        rc = fw(2, x.data_ptr(), K, None, 0, None, 0.0, 0 if light else 1, st.data_ptr(), gam.data_ptr(), bet.data_ptr(), 0.0 if light else 0.5, 3,
                planes.data_ptr(), b.data_ptr(), 1, 0.0 if light else 0.5, 4, None if light else mask.data_ptr(), None, 1e-5, 0, y.data_ptr(), N, n, N, K, None, s)
        assert rc == 0, rc
    for label, light in (("heavy", False), ("light (LN | relu)", True)):
        fwd(light); torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fwd(light); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
        print(f"{name:36s} {label:18s} {statistics.median(ts)*1e3:7.0f} us", flush=True)
        if name == "timed":
            t = y.view(-1)[:8].tolist()
            for w, o, nm in (("vector wave 0", 0, ("staging", "requests", "tick wait", "other")), ("matrix wave 8", 4, ("MFMA phase", "epilogue", "tick wait", "other"))):
                tot = sum(t[o:o + 4]) or 1.0
                print("    " + w + ": " + "  ".join(f"{nm[i]} {t[o + i] / tot:.2f}" for i in range(4)) + f"   ({tot / 1e6:.2f} Mcycles)", flush=True)
