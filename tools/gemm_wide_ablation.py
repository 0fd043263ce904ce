#!/usr/bin/env python
"""Per-segment cycle counters of gemm_x6_kernel (csrc/wide_mlp.hip) in its fp16x3 forms: the forward behind relu -> LayerNorm -> dropout
with a relu -> dropout epilogue and the 1-bit mask output, and the backward-data GEMM with the LayerNorm-backward epilogue, at
[1M, 256] x [256, 256] -- the bench's d = 256 calls.  Waves 0 and 4 of workgroup 0 share a SIMD and run a step's phases in opposite
order.  python tools/gemm_wide_ablation.py --build-only  (build container), then on the GPU box: python tools/gemm_wide_ablation.py"""
import ctypes, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--build-only" not in sys.argv:
    import torch
src = [os.path.join(ROOT, "allset_amd", "csrc", f) for f in ("wide_mlp.hip", "abi.hip")]
n, K, N = 1_000_000, 256, 256
if "--build-only" not in sys.argv:
    dev = torch.device("cuda:0")
    x = torch.randn(n, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    st = torch.stack([torch.zeros(n, device=dev), torch.ones(n, device=dev)], 1).contiguous()
    gam = torch.ones(K, device=dev); bet = torch.zeros(K, device=dev)
    G = torch.randn(n, N, device=dev)
P, I64, F, U64, Ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64, ctypes.c_int
names = ["MFMA", "prologue+split+store", "load issue", "K barriers", "epi row pass", "lookahead", "epi acc->LDS+requests", "epi barriers"]
for name, flags in [("timed", ["-DALLSET_ABL_GX_TIMING"]), ("plain build", [])]:
    # (wide_mlp.hip takes minutes to compile on the GPU box: `--build-only` here in the build container leaves the two libraries
    # under .abl/, which travels with the snapshot)
    so = os.path.join(ROOT, ".abl", "gxabl_" + name.replace(" ", "_") + ".so")
    if not os.path.exists(so) or "--build-only" in sys.argv:
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-shared", "-fPIC",
                        "-I", os.path.join(ROOT, "include"), "-o", so] + flags + src, check=True)
    if "--build-only" in sys.argv:
        continue
    lib = ctypes.CDLL(so)
    lib.allset_gemm_f16x3_plane_bytes.restype = I64
    lib.allset_gemm_f16x3_plane_bytes.argtypes = [I64, I64]
    nb = lib.allset_gemm_f16x3_plane_bytes(N, K)
    planes = torch.empty(nb, dtype=torch.uint8, device=dev); planes_t = torch.empty(nb, dtype=torch.uint8, device=dev)
    lib.allset_gemm_f16x3_planes.argtypes = [P, I64, Ci, P, I64, I64, P]
    s = torch.cuda.current_stream().cuda_stream
    assert lib.allset_gemm_f16x3_planes(W.data_ptr(), K, 0, planes.data_ptr(), N, K, s) == 0
    assert lib.allset_gemm_f16x3_planes(W.data_ptr(), K, 1, planes_t.data_ptr(), K, N, s) == 0
    words = (n + 15) // 16 * (N // 64) * 32
    mask = torch.zeros(words, dtype=torch.int32, device=dev)
    y = torch.empty(n, N, device=dev); gx = torch.empty(n, K, device=dev)
    fw = lib.allset_gemm_wide
    fw.argtypes = [Ci, P, I64, P, I64, P, F, Ci, P, P, P, F, U64, P, P, Ci, F, U64, P, P, F, Ci, P, I64, I64, I64, I64, P, P]
    bw = lib.allset_gemm_wide_lnb
    bw.argtypes = [Ci, P, I64, P, I64, P, F, P, P, I64, P, P, Ci, F, U64, P, I64, P, I64, I64, I64, I64, P, P]
    lib.allset_gemm_x6_lnb_partials.restype = I64
    lib.allset_gemm_x6_lnb_partials.argtypes = [I64]
    npart = lib.allset_gemm_x6_lnb_partials(n)
    part = torch.empty(npart, 2, K, device=dev)
    def fwd():
        rc = fw(2, x.data_ptr(), K, None, 0, None, 0.0, 1, st.data_ptr(), gam.data_ptr(), bet.data_ptr(), 0.5, 3, planes.data_ptr(), b.data_ptr(), 1, 0.5, 4,
                mask.data_ptr(), None, 1e-5, 0, y.data_ptr(), N, n, N, K, None, s)
        assert rc == 0, rc
    def bwd():
        rc = bw(2, G.data_ptr(), N, None, 0, mask.data_ptr(), 0.5, planes_t.data_ptr(), x.data_ptr(), K, st.data_ptr(), gam.data_ptr(), 1, 0.5, 3,
                gx.data_ptr(), K, part.data_ptr(), npart, n, K, N, None, s)
        assert rc == 0, rc
    for label, fn, outbuf in (("forward  (relu->LN->dropout | relu->dropout, mask out)", fwd, y), ("backward (mask in | LayerNorm-backward epilogue)", bwd, gx)):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
        print(f"{name:12s} {label:56s} {statistics.median(ts)*1e3:7.0f} us", flush=True)
        if flags:
            t = outbuf.view(-1)[:16].tolist()
            for w, o in (("wave 0 (MFMA first)", 0), ("wave 4 (stage first)", 8)):
                tot = sum(t[o:o + 8]) or 1.0
                print("    " + w + ": " + "  ".join(f"{nm} {t[o + i] / tot:.2f}" for i, nm in enumerate(names)) + f"   ({tot / 1e6:.2f} Mcycles)", flush=True)
