"""CPU: the C-ABI shared library loads, exports every symbol include/allset_hip.h (core) and include/allset_hip_ext.h declare, and rejects
bad arguments with a status code + message (validation runs before any device work, so no GPU needed)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "allset_hip.h")              # the core: SURVEY 8(b2), frozen
HEADER_EXT = os.path.join(ROOT, "include", "allset_hip_ext.h")      # the package's plumbing, own version


def _code(path):
    return re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)


def declared_symbols(which="all"):
    text = {"core": _code(HEADER), "ext": _code(HEADER_EXT), "all": _code(HEADER) + _code(HEADER_EXT)}[which]
    return sorted(set(re.findall(r"\b(allset_[a-z0-9_]+)\s*\(", text)))


def core_prototypes():
    """name -> prototype text of the core header, whitespace-normalised."""
    out = {}
    for m in re.finditer(r"(?:^|\n)((?:int|const char\*)\s+(allset_[a-z0-9_]+)\s*\([^;]*\));", _code(HEADER)):
        out[m.group(2)] = " ".join(m.group(1).split())
    return out


# The core surface exactly as it stood in ALLSET_ABI_VERSION 10 of the former single header (+ allset_core_version, the one
# addition the split itself made).  A reference-side binding is written against THIS; it may only
# ever change together with ALLSET_CORE_ABI_VERSION.
CORE_V1 = {
    "allset_core_version": "int allset_core_version(void)",
    "allset_version": "int allset_version(void)",
    "allset_last_error": "const char* allset_last_error(void)",
    "allset_csr_build_workspace_bytes": "int allset_csr_build_workspace_bytes(int64_t nnz, int64_t n_rows, size_t* bytes)",
    "allset_csr_build": "int allset_csr_build(const int64_t* row_ids, const int64_t* col_ids, int64_t nnz, int64_t row_base, "
                        "int64_t col_base, int64_t n_rows, int32_t* rowptr, int32_t* col, int32_t* perm, void* workspace, "
                        "size_t workspace_bytes, void* stream)",
    "allset_segreduce_fwd": "int allset_segreduce_fwd(int reduce, int dtype, const int32_t* rowptr, const int32_t* col, const float* w, "
                            "const void* x, int64_t ldx, void* out, int64_t ldo, int32_t* argext, int64_t n_t, int64_t n_s, int64_t d, "
                            "void* stream)",
    "allset_segreduce_fwd_ex": "int allset_segreduce_fwd_ex(int reduce, int dtype, int variant, int64_t nnz, const int32_t* row_order, "
                               "const int32_t* rowptr, const int32_t* col, const float* w, const void* x, int64_t ldx, void* out, "
                               "int64_t ldo, int32_t* argext, int64_t n_t, int64_t n_s, int64_t d, void* stream)",
    "allset_segmax_bwd": "int allset_segmax_bwd(const int32_t* rowptrT, const int32_t* colT, const int32_t* posT, const float* wT, "
                         "const int32_t* argext, const float* gout, int64_t ldg, float* gx, int64_t ldx, int64_t n_s, int64_t n_t, "
                         "int64_t d, void* stream)",
    "allset_sddmm_rowdot": "int allset_sddmm_rowdot(int reduce, const int32_t* rowptr, const int32_t* col, const float* x, int64_t ldx, "
                           "const float* gout, int64_t ldg, const int32_t* argext, float* gw, int64_t n_t, int64_t n_s, int64_t d, "
                           "void* stream)",
    "allset_pma_fwd": "int allset_pma_fwd(int dtype, const int32_t* rowptr, const int32_t* col, const float* alpha, const void* V, "
                      "int64_t ldv, float slope, void* out, int64_t ldo, float* m, float* l, int64_t n_t, int64_t n_s, int64_t H, "
                      "int64_t C, void* stream)",
    "allset_pma_fwd_ex": "int allset_pma_fwd_ex(int dtype, int variant, int64_t nnz, const int32_t* row_order, const int32_t* rowptr, "
                         "const int32_t* col, const float* alpha, const void* V, int64_t ldv, float slope, void* out, int64_t ldo, "
                         "float* m, float* l, int64_t n_t, int64_t n_s, int64_t H, int64_t C, void* stream)",
    "allset_pma_attention": "int allset_pma_attention(const int32_t* rowptr, const int32_t* col, const float* alpha, const float* m, "
                            "const float* l, float slope, float* p, int64_t n_t, int64_t H, void* stream)",
    "allset_pma_bwd_stats": "int allset_pma_bwd_stats(int dtype, const void* out, int64_t ldo, const void* gout, int64_t ldg, "
                            "const float* m, const float* l, float* stats, int64_t n_t, int64_t H, int64_t C, void* stream)",
    "allset_pma_bwd_src": "int allset_pma_bwd_src(int dtype, const int32_t* rowptrT, const int32_t* colT, const float* alpha, "
                          "const void* V, int64_t ldv, const void* gout, int64_t ldg, const float* stats, float slope, void* gV, "
                          "int64_t ldgv, float* galpha, int64_t n_s, int64_t n_t, int64_t H, int64_t C, void* stream)",
    "allset_pma_bwd_src_ex": "int allset_pma_bwd_src_ex(int dtype, int variant, int64_t nnz, const int32_t* row_order, "
                             "const int32_t* rowptrT, const int32_t* colT, const float* alpha, const void* V, int64_t ldv, "
                             "const void* gout, int64_t ldg, const float* stats, float slope, void* gV, int64_t ldgv, float* galpha, "
                             "int64_t n_s, int64_t n_t, int64_t H, int64_t C, void* stream)",
}


def test_core_surface_is_frozen():
    got = core_prototypes()
    assert sorted(got) == sorted(CORE_V1)
    for name, proto in CORE_V1.items():
        assert got[name] == " ".join(proto.split()), name
    assert re.search(r"#define\s+ALLSET_CORE_ABI_VERSION\s+1\b", open(HEADER).read())
    assert not set(declared_symbols("core")) & set(declared_symbols("ext"))       # one declaration per symbol
    assert "allset_hip_ext.h" not in _code(HEADER)                                # the core header stands alone


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
        assert hasattr(lib, sym), f"{sym} declared in include/*.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (allset_[a-z0-9_]+)", out)))
    assert exported == declared_symbols()


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    for body in ('#include "allset_hip.h"\nint main(void){return ALLSET_CORE_ABI_VERSION == 1 ? 0 : 1;}\n',        # core alone
                 '#include "allset_hip_ext.h"\nint main(void){return ALLSET_ABI_VERSION == 15 && ALLSET_CORE_ABI_VERSION == 1 ? 0 : 1;}\n'):
        src.write_text(body)
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                        str(tmp_path / "t")], check=True)
        subprocess.run([str(tmp_path / "t")], check=True)


def test_version_and_error_reporting():
    from allset_amd import _lib
    lib = _lib.load()
    assert lib.allset_version() == _lib.ABI_VERSION == 15
    assert lib.allset_core_version() == _lib.CORE_ABI_VERSION == 1
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
