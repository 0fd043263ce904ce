"""GPU: the C ABI used from plain C++ (examples/abi_example.cpp): no Python, no torch in the process -- hipMalloc'd
buffers, the caller's stream, status codes.  Built with hipcc against the in-tree liballset_hip.so (by build(), or here) and run."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_example_builds_and_runs(tmp_path, device):
    lib_dir = os.path.join(ROOT, "allset_amd")
    src = os.path.join(ROOT, "examples", "abi_example.cpp")
    assert os.path.exists(os.path.join(lib_dir, "liballset_hip.so")), "build the library first (python -m allset_amd.build)"
    # __graft_entry__.build() leaves examples/abi_example.bin beside the library (relative rpath); it is used when it is newer than its
    # source and the headers -- a fresh GPU box pays tens of seconds for its first hipcc start -- and rebuilt here otherwise
    pre = os.path.join(ROOT, "examples", "abi_example.bin")
    deps = [src] + [os.path.join(ROOT, "include", h) for h in ("allset_hip.h", "allset_hip_ext.h")]
    if os.path.exists(pre) and os.access(pre, os.X_OK) and all(os.path.getmtime(pre) >= os.path.getmtime(d) for d in deps):
        exe = pre
    else:
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        exe = str(tmp_path / "abi_example")
        cmd = [hipcc, "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-L" + lib_dir, "-lallset_hip",
               "-Wl,-rpath," + lib_dir, "-o", exe]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.strip().endswith("OK"), run.stdout + run.stderr
