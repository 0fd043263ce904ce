#!/usr/bin/env python
"""Ablation of linear_bf16_kernel (csrc/fused_bf16.hip; cdna_hip_programming.md: 'ablate before optimising'): variants of the
kernel without its row loads / matrix phase / LDS weight-fragment reads / epilogue / global stores, timed at the configs[4]
per-GPU call ([250000, 256] x [256, 256], forward + the bit-masked backward-data).

  python tools/linear_bf16_ablation.py --build-only     in the build container: leaves .abl/bf16_<arm>.so (travels with gpurun)
  python tools/linear_bf16_ablation.py [rows] [--warm]  on the GPU box (default: COLD operands, 1 GiB written between two launches;
                                                        --warm: the same operands back to back, served by the Infinity Cache)
Extra arms: ALLSET_BF16_ABL_EXTRA="name:-DFLAG -DFLAG2;name2:-DFLAG3"."""
import ctypes, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABL = os.path.join(ROOT, ".abl")
SRC = [os.path.join(ROOT, "allset_amd", "csrc", f) for f in ("fused_bf16.hip", "abi.hip")]
ARMS = [("full", []), ("half-sector stores (round-6 first session)", ["-DALLSET_BF16_HALF_SECTOR_STORES"]), ("slab epilogue", ["-DALLSET_BF16_SLAB_EPILOGUE"]), ("three row buffers", ["-DALLSET_BF16_DEPTH3"]), ("early first request", ["-DALLSET_BF16_EARLY_REQUEST"]), ("no-load", ["-DALLSET_BF16_ABL_NOLOAD"]), ("no-store", ["-DALLSET_BF16_ABL_NOSTORE"]),
        ("no-epilogue", ["-DALLSET_BF16_ABL_NOEPI"]), ("no-mfma", ["-DALLSET_BF16_ABL_NOMFMA"]),
        ("no-mfma no-epilogue", ["-DALLSET_BF16_ABL_NOMFMA", "-DALLSET_BF16_ABL_NOEPI"]),
        ("no-load no-epilogue", ["-DALLSET_BF16_ABL_NOLOAD", "-DALLSET_BF16_ABL_NOEPI"]),
        ("no-load no-store", ["-DALLSET_BF16_ABL_NOLOAD", "-DALLSET_BF16_ABL_NOSTORE"]),
        ("piece map (wrong results)", ["-DALLSET_BF16_ABL_PIECEMAP"]),
        ("piece map no-mfma no-epilogue", ["-DALLSET_BF16_ABL_PIECEMAP", "-DALLSET_BF16_ABL_NOMFMA", "-DALLSET_BF16_ABL_NOEPI"]),
        ("launch + weight staging only", ["-DALLSET_BF16_ABL_NOLOAD", "-DALLSET_BF16_ABL_NOMFMA", "-DALLSET_BF16_ABL_NOEPI"])]
for spec in filter(None, os.environ.get("ALLSET_BF16_ABL_EXTRA", "").split(";")):
    nm, fl = spec.split(":", 1)
    ARMS.append((nm, fl.split()))


def so_of(name):
    return os.path.join(ABL, "bf16_" + name.replace(" ", "_") + ".so")


def build():
    os.makedirs(ABL, exist_ok=True)
    procs = []
    for name, flags in ARMS:
        cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
               "-o", so_of(name)] + flags + SRC
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(f"{name}: hipcc failed\n{out}")
        print("built", so_of(name))


if "--build-only" in sys.argv:
    build()
    raise SystemExit(0)

import torch
dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("-")]
WARM = "--warm" in sys.argv          # back-to-back launches on the same operands (they stay in the 256-MiB Infinity Cache)
n = int(args[0]) if args else 250000
K = N = 256
g = torch.Generator().manual_seed(0)
x = torch.randn(n, K, generator=g).to(torch.bfloat16).to(dev)
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
b = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
y = torch.empty(n, N, dtype=torch.bfloat16, device=dev)
bits = torch.randint(0, 256, (n, N // 8), dtype=torch.uint8, device=dev)
acc = torch.randn(n, K, generator=g).to(torch.bfloat16).to(dev)
P, I64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
flush = torch.zeros(256 * 1024 * 1024, dtype=torch.int32, device=dev)
print(f"[{n}, {K}] x [{K}, {N}] bf16, {'warm' if WARM else 'cold'} operands, one launch per HIP-event pair (~5 us of event overhead included); "
      f"one read + one write at 6.3 TB/s = {n * (K + N) * 2 / 6.3e12 * 1e6:.0f} us")
only = os.environ.get("ALLSET_ABL_ONLY")
for name, _ in ARMS:
    if only and only not in name:
        continue
    if not os.path.exists(so_of(name)):
        print(f"{name}: not built (run with --build-only in the build container)")
        continue
    lib = ctypes.CDLL(so_of(name))
    lib.allset_last_error.restype = ctypes.c_char_p
    fwd = lib.allset_linear_bf16_fwd
    fwd.argtypes = [P, I64, P, P, I, P, P, P, P, I64, I64, I64, I64, P]
    fm = lib.allset_linear_bf16_fwd_mask
    fm.argtypes = [P, I64, P, P, P, I64, P, I64, I64, I64, P]
    bb = lib.allset_linear_bf16_bwd_bits
    bb.argtypes = [P, I64, P, P, P, I64, P, I64, I64, I64, I64, P]
    st = lambda: torch.cuda.current_stream().cuda_stream

    def run_fwd():
        assert fwd(x.data_ptr(), K, W.data_ptr(), b.data_ptr(), 1, None, None, None, y.data_ptr(), N, n, K, N, st()) == 0, lib.allset_last_error()

    def run_fwd_mask():
        assert fm(x.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), N, bits.data_ptr(), n, K, N, st()) == 0, lib.allset_last_error()

    def run_bwd_bits():
        assert bb(x.data_ptr(), K, bits.data_ptr(), W.data_ptr(), None, 0, y.data_ptr(), N, n, K, N, st()) == 0, lib.allset_last_error()

    def run_bwd_bits_acc():
        assert bb(x.data_ptr(), K, bits.data_ptr(), W.data_ptr(), acc.data_ptr(), K, y.data_ptr(), N, n, K, N, st()) == 0, lib.allset_last_error()
    out = []
    for run in (run_fwd, run_fwd_mask, run_bwd_bits, run_bwd_bits_acc):
        for _ in range(3):
            run()
        torch.cuda.synchronize(); ts = []
        for _ in range(12):
            # COLD operands, as inside a training step: 1 GiB written between two launches empties the 256-MiB Infinity Cache
            # (back-to-back launches on the same 256 MB of operands run 10-20 us faster than the same kernel does in the step)
            if not WARM:
                flush.add_(1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); run(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        out.append(statistics.median(ts) * 1e3)
    print(f"{name:30s} fwd {out[0]:6.1f} us   fwd+mask {out[1]:6.1f}   bwd(bits) {out[2]:6.1f}   bwd(bits, acc_in) {out[3]:6.1f}", flush=True)
