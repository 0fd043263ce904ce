"""CPU: the C-ABI shared library loads, exports every symbol include/allset_hip.h declares, and rejects
bad arguments with a status code + message (validation runs before any device work, so no GPU needed)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "allset_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(allset_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("allset_version", "allset_last_error", "allset_csr_build", "allset_segreduce_fwd",
                 "allset_segmax_bwd", "allset_sddmm_rowdot", "allset_pma_fwd", "allset_pma_bwd_stats",
                 "allset_pma_bwd_src", "allset_pma_attention", "allset_csr_build_workspace_bytes"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from allset_amd import _lib
    lib = _lib.load()
    for sym in declared_symbols():
        assert hasattr(lib, sym), f"{sym} declared in allset_hip.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (allset_[a-z0-9_]+)", out)))
    assert exported == declared_symbols()


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "allset_hip.h"\nint main(void){return ALLSET_ABI_VERSION == 10 ? 0 : 1;}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                    str(tmp_path / "t")], check=True)
    subprocess.run([str(tmp_path / "t")], check=True)


def test_version_and_error_reporting():
    from allset_amd import _lib
    lib = _lib.load()
    assert lib.allset_version() == _lib.ABI_VERSION == 10
    rc = lib.allset_segreduce_fwd(99, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0)
    assert rc == -1 and b"bad reduce" in lib.allset_last_error()
    rc = lib.allset_segreduce_fwd(0, 7, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0)      # dtype must be F32 (0) or BF16 (1)
    assert rc == -1 and b"dtype" in lib.allset_last_error()
    rc = lib.allset_segreduce_fwd(0, 0, 0, 0, 0, 0, 4, 0, 4, 0, 5, 5, 4, 0)      # null pointers
    assert rc == -1 and b"null" in lib.allset_last_error()
    rc = lib.allset_segreduce_fwd(0, 0, 0, 0, 0, 0, 4, 0, 4, 0, 0, 5, 4, 0)      # n_t == 0: nothing to do
    assert rc == 0 and lib.allset_last_error() == b""
    rc = lib.allset_pma_fwd(0, 0, 0, 0, 0, 0, 0.2, 0, 0, 0, 0, 4, 4, 0, 8, 0)    # heads = 0
    assert rc == -1
    rc = lib.allset_pma_fwd(0, 0, 0, 0, 0, 0, 0.2, 0, 0, 0, 0, 4, 4, 1000, 8, 0)  # heads > built max
    assert rc == -3
    with pytest.raises(_lib.AllSetHipError):
        _lib.check(-1, "demo")


def test_no_fallback_cpu_tensors_are_refused():
    import torch
    from allset_amd import _lib, Incidence
    with pytest.raises(_lib.AllSetHipError):
        Incidence.from_edge_index(torch.zeros(2, 3, dtype=torch.int64))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "allset_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            text = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), fn
