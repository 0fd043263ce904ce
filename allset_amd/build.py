"""Compile ``allset_amd/csrc/*.hip`` for gfx950 into the in-tree ``allset_amd/liballset_hip.so``.

hipcc cross-compiles without a GPU, so this runs in the build container; the built ``.so`` is
git-ignored but travels to the GPU box with the gpurun snapshot.  ``python -m allset_amd.build``.
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys
from typing import List

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
BUILD_DIR = os.path.join(PKG_DIR, "csrc", "build")
LIB_PATH = os.path.join(PKG_DIR, "liballset_hip.so")
ARCH = "gfx950"
SOURCES = ["abi.hip", "csr_build.hip", "segreduce.hip", "pma.hip", "dense.hip", "fused_mlp.hip", "fused_fwd2.hip", "fused_bwd.hip", "fused_bwd4.hip", "fused_bwd6.hip", "fused_bf16.hip", "batchnorm.hip", "input_linear.hip", "sparse_input.hip", "narrow_linear.hip", "wide_mlp.hip", "wgrad_f16.hip", "loss.hip", "optim.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(INCLUDE, "allset_hip.h"), os.path.join(INCLUDE, "allset_hip_ext.h")]
CXXFLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC",
            "-Wall", "-Wno-unused-function",
            # No SLP vectorisation: packed-f32 VALU (v_pk_fma_f32 & co.) beside MFMAs is slower on gfx950
            # (MI355X_MICROARCH.md) and, in fused_linear_bwd_x6_kernel, hipcc 7.2's packed code produced wrong
            # low halves nondeterministically (tools/x6_debug.py history; scalar code is bit-stable).
            "-fno-slp-vectorize"] + os.environ.get("ALLSET_EXTRA_CXXFLAGS", "").split()


# The kernels were validated (bitwise run-to-run determinism of the LayerNorm-backward epilogue, tests/test_gpu_dense.py) with
# this compiler and the -fno-slp-vectorize workaround above.  Another hipcc may miscompile differently -- or not need the
# workaround: re-run the GPU suite, then extend this tuple (or set ALLSET_ALLOW_ANY_HIPCC=1 to build at your own risk).
VALIDATED_HIPCC = ("7.2.26015",)


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build liballset_hip.so")
    return exe


def hipcc_version() -> str:
    out = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        if line.startswith("HIP version:"):
            return line.split(":", 1)[1].strip()
    return "unknown"


def _check_compiler() -> None:
    ver = hipcc_version()
    if not any(ver.startswith(v) for v in VALIDATED_HIPCC) and os.environ.get("ALLSET_ALLOW_ANY_HIPCC", "0") != "1":
        raise RuntimeError(
            f"hipcc reports HIP version {ver}; liballset_hip.so was validated with {VALIDATED_HIPCC} only (the fused "
            "dense kernels depend on a compiler workaround, see CXXFLAGS in allset_amd/build.py).  Re-run "
            "`pytest -m gpu` with this compiler and add it to VALIDATED_HIPCC, or set ALLSET_ALLOW_ANY_HIPCC=1.")


def _stale(target: str, deps: List[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(BUILD_DIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if force or _stale(obj, [path] + HEADERS):
        cmd = [_hipcc()] + CXXFLAGS + ["-c", path, "-o", obj]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{res.stdout}\n{res.stderr}")
    return obj


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Build (if stale) and return the path of liballset_hip.so."""
    os.makedirs(BUILD_DIR, exist_ok=True)
    if force or any(_stale(os.path.join(BUILD_DIR, os.path.splitext(src)[0] + ".o"), [os.path.join(CSRC, src)] + HEADERS)
                    for src in SOURCES):
        _check_compiler()                      # only when something is actually compiled
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or _stale(LIB_PATH, objs):
        # libamdhip64 is resolved at load time against the copy torch has already mapped (same SONAME)
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(f"built {LIB_PATH} ({os.path.getsize(LIB_PATH)} bytes)")
    return LIB_PATH


EXAMPLE_SRC = os.path.join(os.path.dirname(PKG_DIR), "examples", "abi_example.cpp")
EXAMPLE_BIN = os.path.splitext(EXAMPLE_SRC)[0] + ".bin"


def build_example(force: bool = False, verbose: bool = False) -> str:
    """examples/abi_example.cpp (the C ABI from plain C++) -> examples/abi_example.bin, linked against the in-tree library through a
    relative rpath.  Built with the library so that tests/test_gpu_c_example.py only has to RUN it on the GPU box: the first hipcc
    start on a fresh box pages in the whole compiler (73 s measured on one box for a 0.9 s compile)."""
    lib = build_library()
    if force or _stale(EXAMPLE_BIN, [EXAMPLE_SRC, lib] + HEADERS):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-std=c++17", "-I" + INCLUDE, EXAMPLE_SRC, "-L" + os.path.dirname(lib), "-lallset_hip",
               "-Wl,-rpath,$ORIGIN/../allset_amd", "-o", EXAMPLE_BIN]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {EXAMPLE_SRC}:\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(f"built {EXAMPLE_BIN} ({os.path.getsize(EXAMPLE_BIN)} bytes)")
    return EXAMPLE_BIN


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
    build_example(force="--force" in sys.argv, verbose=True)
