"""GPU: the C ABI refuses bad arguments with a status code and a message -- it never throws, crashes or launches.
Called through ctypes with raw pointers, the way a foreign binding would (include/allset_hip.h: 0 ok, < 0 error,
allset_last_error() for the text)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _err(lib, rc):
    assert rc < 0, "a bad argument must not be accepted"
    msg = lib.allset_last_error()
    assert msg and len(msg) > 5
    return msg.decode()


def test_aggregation_entries_reject_bad_arguments(device):
    from allset_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    n, d, nnz = 64, 32, 200
    rowptr = torch.zeros(n + 1, dtype=torch.int32, device=device)
    col = torch.zeros(nnz, dtype=torch.int32, device=device)
    x = torch.randn(n, d, device=device)
    out = torch.empty(n, d, device=device)
    P = lambda t: t.data_ptr()
    # segreduce: bad reduce code, bad dtype, null x with rows, leading dimension below the width, negative sizes
    f = lib.allset_segreduce_fwd_ex
    assert "reduce" in _err(lib, f(9, 0, 0, nnz, None, P(rowptr), P(col), None, P(x), d, P(out), d, None, n, n, d, st))
    _err(lib, f(0, 7, 0, nnz, None, P(rowptr), P(col), None, P(x), d, P(out), d, None, n, n, d, st))
    _err(lib, f(0, 0, 0, nnz, None, P(rowptr), P(col), None, None, d, P(out), d, None, n, n, d, st))
    _err(lib, f(0, 0, 0, nnz, None, P(rowptr), P(col), None, P(x), d - 1, P(out), d, None, n, n, d, st))
    _err(lib, f(0, 0, 0, nnz, None, P(rowptr), P(col), None, P(x), d, P(out), d, None, -1, n, d, st))
    # the short-row variant refuses layouts it is not built for (rows not 16-byte aligned)
    _err(lib, f(0, 0, 2, nnz, None, P(rowptr), P(col), None, P(x) + 4, d, P(out), d, None, n, n - 1, d, st))
    # PMA: heads beyond the built maximum, C = 0, null logits
    g = lib.allset_pma_fwd
    alpha = torch.randn(n, 4, device=device)
    m = torch.empty(n, 4, device=device); l = torch.empty(n, 4, device=device)
    _err(lib, g(0, P(rowptr), P(col), P(alpha), P(x), d, 0.2, P(out), d, P(m), P(l), n, n, 4096, 1, st))
    _err(lib, g(0, P(rowptr), P(col), None, P(x), d, 0.2, P(out), d, P(m), P(l), n, n, 4, 8, st))
    _err(lib, g(0, P(rowptr), P(col), P(alpha), P(x), 4, 0.2, P(out), d, P(m), P(l), n, n, 4, 8, st))
    # CSR build: negative counts
    ws = ctypes.c_uint64(0)
    _err(lib, lib.allset_csr_build_workspace_bytes(-5, 10, ctypes.byref(ws)))
    torch.cuda.synchronize()                 # nothing was launched, nothing is pending


def test_dense_entries_reject_bad_arguments(device):
    from allset_amd import _lib, dense
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    n, d = 100, 128
    x = torch.randn(n, d + 4, device=device)[:, :d]                       # 16-byte aligned rows, ld = d + 4
    y = torch.empty(n, d, device=device)
    stats = torch.empty(n, 2, device=device)
    g, b = torch.ones(d, device=device), torch.zeros(d, device=device)
    P = lambda t: t.data_ptr()
    # dropout probability outside [0, 1)
    _err(lib, lib.allset_ln_fwd(P(x), d + 4, P(g), P(b), 1e-5, 0, 1.0, 0, P(y), d, P(stats), n, d, None, st))
    _err(lib, lib.allset_ln_fwd(P(x), d + 4, P(g), P(b), 1e-5, 0, -0.1, 0, P(y), d, P(stats), n, d, None, st))
    # null output, negative row count
    _err(lib, lib.allset_ln_fwd(P(x), d + 4, P(g), P(b), 1e-5, 0, 0.0, 0, None, d, P(stats), n, d, None, st))
    _err(lib, lib.allset_ln_fwd(P(x), d + 4, P(g), P(b), 1e-5, 0, 0.0, 0, P(y), d, P(stats), -3, d, None, st))
    # widths the fused / tiled kernels are not built for report "unsupported", they do not guess
    assert lib.allset_fused_linear_supported(100, 128) == 0 and lib.allset_gemm_x6_supported(256, 100) == 0
    W = torch.randn(128, 100, device=device)
    xb = torch.randn(n, 100, device=device)
    rc = lib.allset_fused_linear_fwd(P(xb), 100, P(W), None, 1e-5, 0, 0.0, 0, None, None, 0, 0.0, 0, P(y), d, None, n, 100, 128,
                                     None, None, None, None, None, st)
    assert rc == -3 and b"fused_linear" in lib.allset_last_error()       # ALLSET_ERR_UNSUPPORTED
    planes = torch.empty(int(lib.allset_gemm_x6_plane_bytes(256, 256)), dtype=torch.uint8, device=device)
    xa = torch.randn(n, 256, device=device)
    o = torch.empty(n, 256, device=device)
    # misaligned A rows, LayerNorm-apply without gamma, dropout p = 1
    _err(lib, lib.allset_gemm_x6(P(xa) + 4, 256, None, 0, 0.0, 0, None, None, None, 0.0, 0, P(planes), None, 0, 0.0, 0, P(o), 256, n - 1, 256,
                                 256, None, st))
    _err(lib, lib.allset_gemm_x6(P(xa), 256, None, 0, 0.0, 0, P(stats), None, None, 0.0, 0, P(planes), None, 0, 0.0, 0, P(o), 256, n, 256, 256,
                                 None, st))
    _err(lib, lib.allset_gemm_x6(P(xa), 256, None, 0, 0.0, 0, None, None, None, 1.0, 0, P(planes), None, 0, 0.0, 0, P(o), 256, n, 256, 256,
                                 None, st))
    assert lib.allset_gemm_x6_plane_bytes(256, 100) == -1
    # the Python layer turns every one of these into AllSetHipError
    with pytest.raises(_lib.AllSetHipError):
        dense.gemm_x6_planes(torch.randn(256, 100, device=device), False)
    torch.cuda.synchronize()


def test_blocked_and_aux_entries_reject_bad_arguments(device):
    """ABI 7 / 8 entry points: a block width that is not a power of two in [4, C / 2], a leading dimension that is not the block
    width, widths the split-role kernels are not built for, a partial buffer too small for the aux sections."""
    from allset_amd import _lib
    lib = _lib.load()
    if not lib.allset_fused_linear_blocked_supported(128, 128):
        pytest.skip("default kernel family not selected")
    st = torch.cuda.current_stream().cuda_stream
    n, d = 64, 128
    x = torch.randn(n, d, device=device); y = torch.empty(n, d, device=device); W = torch.randn(d, d, device=device)
    P = lambda t: t.data_ptr()
    f = lib.allset_fused_linear_fwd_blocked
    ok = f(P(x), 16, 16, None, None, 1e-5, 0, 0.0, 0, P(W), None, 0, 0.0, 0, P(y), 32, 32, None, n, d, d, None, None, st)
    assert ok == 0
    _err(lib, f(P(x), 16, 24, None, None, 1e-5, 0, 0.0, 0, P(W), None, 0, 0.0, 0, P(y), d, 0, None, n, d, d, None, None, st))     # 24: not a power of two
    _err(lib, f(P(x), d, 16, None, None, 1e-5, 0, 0.0, 0, P(W), None, 0, 0.0, 0, P(y), d, 0, None, n, d, d, None, None, st))      # ldx != block width
    _err(lib, f(P(x), 128, 128, None, None, 1e-5, 0, 0.0, 0, P(W), None, 0, 0.0, 0, P(y), d, 0, None, n, d, d, None, None, st))   # block = whole row
    assert lib.allset_fused_linear_blocked_supported(64, 64) == 0
    W64 = torch.randn(64, 64, device=device)
    _err(lib, f(P(x), 16, 16, None, None, 1e-5, 0, 0.0, 0, P(W64), None, 0, 0.0, 0, P(y), 64, 0, None, n, 64, 64, None, None, st))
    # the aux one-pass backward: partial stride below O*I + O + 4*I + 4, wrong slice count
    ns = ctypes.c_int64(0)
    assert lib.allset_fused_linear_bwd_all_slices_for(n, d, d, 0, ctypes.byref(ns)) == 0
    M = d * d + d + 4 * d + 4
    part = torch.empty(ns.value * M, device=device); g4 = torch.randn(n, 4, device=device); w4 = torch.randn(4, d, device=device)
    gx = torch.empty(n, d, device=device)
    h = lib.allset_fused_linear_bwd_all_aux
    assert h(P(y), d, P(W), P(x), d, P(g4), P(w4), P(gx), d, P(part), M, ns.value, n, d, d, st) == 0
    _err(lib, h(P(y), d, P(W), P(x), d, P(g4), P(w4), P(gx), d, P(part), M - 8, ns.value, n, d, d, st))
    _err(lib, h(P(y), d, P(W), P(x), d, P(g4), P(w4), P(gx), d, P(part), M, ns.value + 1, n, d, d, st))
    _err(lib, h(P(y), d, P(W), P(x), d, None, P(w4), P(gx), d, P(part), M, ns.value, n, d, d, st))
    assert lib.allset_fused_linear_bwd_all_aux_supported(64, 128) == 0
    torch.cuda.synchronize()


def test_batchnorm_entries_reject_bad_arguments(device):
    """ABI 9 additions (csrc/batchnorm.hip, norm_mode of the fused Linear): status codes, not launches."""
    from allset_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    n, d = 100, 128
    x = torch.randn(n, d, device=device)
    P = lambda t: t.data_ptr()
    ns = ctypes.c_int64(0)
    assert lib.allset_col_moments_slices(n, ctypes.byref(ns)) == 0 and ns.value >= 1
    part = torch.empty(ns.value, d, device=device)
    f = lib.allset_col_moments
    assert f(P(x), d, n, d, 0, None, P(part), ns.value, st) == 0
    assert "width" in _err(lib, f(P(x), d, n, 6, 0, None, P(part), ns.value, st))            # width not a multiple of 4
    _err(lib, f(P(x), d, n, d, 0, None, P(part), ns.value + 1, st))                            # wrong slice count
    _err(lib, f(P(x), d - 4, n, d, 0, None, P(part), ns.value, st))                            # leading dimension below the width
    _err(lib, f(P(x) + 4, d, n, d, 0, None, P(part), ns.value, st))                            # rows not 16-byte aligned
    _err(lib, f(None, d, n, d, 0, None, P(part), ns.value, st))
    _err(lib, lib.allset_col_moments_slices(-1, ctypes.byref(ns)))
    gx, s = torch.zeros(n, d, device=device), torch.ones(d, device=device)
    h = lib.allset_col_affine_add
    assert h(P(gx), d, P(x), d, P(s), P(s), 1, n, d, st) == 0
    _err(lib, h(P(gx), d, P(x), d, None, P(s), 1, n, d, st))
    _err(lib, h(P(gx), d, P(x), d, P(s), P(s), 1, -3, d, st))
    _err(lib, h(P(gx), d, P(x), d, P(s), P(s), 1, n, 2048, st))
    # norm_mode: unknown value, column affine without scale / shift, column affine backward without the row statistics
    W, b = torch.randn(d, d, device=device), torch.zeros(d, device=device)
    y, stats = torch.empty(n, d, device=device), torch.empty(n, 2, device=device)
    fw = lib.allset_fused_linear_fwd_nm
    assert fw(P(x), d, P(s), P(b), 1e-5, 1, 0, 0.0, 0, P(W), P(b), 0, 0.0, 0, P(y), d, P(stats), n, d, d, None, None, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(stats.cpu(), torch.tensor([0.0, 1.0]).repeat(n, 1))                     # what the backward entries expect
    assert "norm_mode" in _err(lib, fw(P(x), d, P(s), P(b), 1e-5, 7, 0, 0.0, 0, P(W), P(b), 0, 0.0, 0, P(y), d, P(stats), n, d, d, None, None, st))
    _err(lib, fw(P(x), d, None, None, 1e-5, 1, 0, 0.0, 0, P(W), P(b), 0, 0.0, 0, P(y), d, P(stats), n, d, d, None, None, st))
    sl = ctypes.c_int64(0)
    lib.allset_fused_linear_bwd_all_slices_for(n, d, d, 0, ctypes.byref(sl))
    M = d * d + d + 2 * d
    partb = torch.empty(sl.value, M, device=device)
    gy, gxb = torch.randn(n, d, device=device), torch.empty(n, d, device=device)
    bw = lib.allset_fused_linear_bwd_all_nm
    flat = partb.view(-1)
    args = lambda stats_p, mode: (P(gy), d, None, 0.0, P(W), P(x), d, stats_p, P(s), P(b), mode, 0, 0.0, 0, P(gxb), d,
                                  flat[d * d + d:].data_ptr(), P(flat), flat[d * d:].data_ptr(), sl.value, n, d, d, None, M, st)
    assert bw(*args(P(stats), 1)) == 0
    _err(lib, bw(*args(None, 1)))
    _err(lib, bw(*args(P(stats), 5)))
    torch.cuda.synchronize()


def test_dataset_scale_entries_reject_bad_arguments(device):
    """ABI 10 additions (csrc/input_linear.hip, sparse_input.hip, narrow_linear.hip, the batched reduction, the loss total)."""
    from allset_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()
    n, d, O = 50, 300, 64
    K = lib.allset_input_linear_k(d)
    assert K >= d + 1 and K % 16 == 0
    x = torch.randn(n, d, device=device)
    xh = torch.empty(n, K, device=device)
    f = lib.allset_xhat_rows
    assert f(P(x), d, n, d, 1e-5, 0.0, 0, None, P(xh), K, st) == 0
    _err(lib, f(P(x), d, n, d, 1e-5, 0.0, 0, None, P(xh), d, st))               # no room for the ones column
    _err(lib, f(P(x), d, n, 5000, 1e-5, 0.0, 0, None, P(xh), K, st))            # wider than 4096
    _err(lib, f(P(x), d, n, d, 1e-5, 1.0, 0, None, P(xh), K, st))               # p = 1
    _err(lib, f(None, d, n, d, 1e-5, 0.0, 0, None, P(xh), K, st))
    W, g, b = torch.randn(O, d, device=device), torch.ones(d, device=device), torch.zeros(d, device=device)
    Wp = torch.empty(O, K, device=device)
    assert lib.allset_fold_ln_linear(P(W), d, P(g), P(b), None, O, d, P(Wp), K, st) == 0
    _err(lib, lib.allset_fold_ln_linear(P(W), d, P(g), P(b), None, O, d, P(Wp), d, st))
    _err(lib, lib.allset_fold_ln_linear(P(W), d, None, P(b), None, O, d, P(Wp), K, st))
    M = torch.zeros(O, K, device=device)
    gW, gg = torch.empty(O, d, device=device), torch.empty(2, d, device=device)
    u = lib.allset_unfold_ln_linear
    assert u(P(M), K, P(W), d, P(g), P(b), O, d, P(gW), d, None, P(gg), P(gg[1]), st) == 0
    _err(lib, u(P(M), d, P(W), d, P(g), P(b), O, d, P(gW), d, None, P(gg), P(gg[1]), st))
    _err(lib, u(P(M), K, P(W), d, P(g), P(b), O, d, None, d, None, P(gg), P(gg[1]), st))
    su = torch.zeros(lib.allset_sparse_ln_linear_slices(), 2, O, device=device)
    ux = lib.allset_unfold_ln_linear_ex
    assert ux(P(M), K, P(W), d, P(g), P(b), O, d, P(gW), d, None, P(gg), P(gg[1]), P(su), su.shape[0], st) == 0
    _err(lib, ux(P(M), K, P(W), d, P(g), P(b), O, d, P(gW), d, None, P(gg), P(gg[1]), P(su), 0, st))
    # sparse first layer
    assert lib.allset_sparse_ln_linear_supported(64) == 1 and lib.allset_sparse_ln_linear_supported(7) == 0
    WT = torch.empty(d + 2, O, device=device)
    assert lib.allset_fold_ln_linear_t(P(W), d, P(g), P(b), None, O, d, P(WT), st) == 0
    _err(lib, lib.allset_fold_ln_linear_t(P(W), d - 1, P(g), P(b), None, O, d, P(WT), st))
    rowptr = torch.zeros(n + 1, dtype=torch.int32, device=device)
    y, rm = torch.empty(n, O, device=device), torch.empty(n, device=device)
    sf = lib.allset_sparse_ln_linear_fwd
    assert sf(P(rowptr), None, None, n, d, P(WT), O, 1e-5, 0.0, 0, None, P(y), O, None, P(rm), st) == 0      # no non-zeros at all
    torch.cuda.synchronize()
    torch.testing.assert_close(y, WT[d + 1].expand(n, O))                       # LayerNorm of a zero row is beta: y = b + W beta
    assert _err(lib, sf(P(rowptr), None, None, n, d, P(WT), 7, 1e-5, 0.0, 0, None, P(y), O, None, P(rm), st))
    _err(lib, sf(P(rowptr), None, None, n, d, P(WT), O, 1e-5, 0.0, 0, None, P(y), O - 4, None, P(rm), st))
    _err(lib, sf(None, None, None, n, d, P(WT), O, 1e-5, 0.0, 0, None, P(y), O, None, P(rm), st))
    colptr = torch.zeros(d + 1, dtype=torch.int32, device=device)
    gy, Ms = torch.randn(n, O, device=device), torch.empty(O, d, device=device)
    sb = lib.allset_sparse_ln_linear_bwd
    assert sb(P(colptr), None, None, None, P(rm), P(gy), O, n, d, O, P(Ms), d, P(su), st) == 0
    torch.cuda.synchronize()
    assert float(Ms.abs().max()) == 0.0
    torch.testing.assert_close(su[:, 0].sum(0), gy.sum(0), rtol=1e-5, atol=1e-5)
    _err(lib, sb(P(colptr), None, None, None, P(rm), P(gy), O, n, d, O, P(Ms), d - 1, P(su), st))
    _err(lib, sb(P(colptr), None, None, None, P(rm), P(gy), O, n, d, O, P(Ms), d, None, st))
    # classifier-head backward
    assert lib.allset_linear_narrow_supported(7, 64) == 1 and lib.allset_linear_narrow_supported(17, 64) == 0
    assert lib.allset_linear_narrow_supported(7, 66) == 0 and lib.allset_linear_narrow_supported(7, 512) == 0
    ns = ctypes.c_int64(0)
    assert lib.allset_linear_narrow_slices(n, ctypes.byref(ns)) == 0 and ns.value >= 1
    h, Wc, gl = torch.randn(n, 64, device=device), torch.randn(7, 64, device=device), torch.randn(n, 7, device=device)
    gh, part = torch.empty(n, 64, device=device), torch.empty(ns.value, 7 * 64 + 8, device=device)
    nb = lib.allset_linear_narrow_bwd
    assert nb(P(gl), 7, P(h), 64, P(Wc), n, 7, 64, P(gh), 64, P(part), part.shape[1], ns.value, st) == 0
    assert "N <=" in _err(lib, nb(P(gl), 7, P(h), 64, P(Wc), n, 17, 64, P(gh), 64, P(part), part.shape[1], ns.value, st))
    _err(lib, nb(P(gl), 7, P(h), 64, P(Wc), n, 7, 64, P(gh), 64, P(part), part.shape[1], ns.value + 1, st))
    _err(lib, nb(P(gl), 7, P(h), 64, P(Wc), n, 7, 64, P(gh), 64, P(part), 7 * 64, ns.value, st))
    _err(lib, nb(P(gl), 7, P(h) + 4, 64, P(Wc), n, 7, 64, P(gh), 64, P(part), part.shape[1], ns.value, st))
    # batched reduction
    assert lib.allset_reduce_partials_batchable(64, 4288) == 1 and lib.allset_reduce_partials_batchable(64, 6) == 0
    assert lib.allset_reduce_partials_batchable(4096, 4096) == 0
    parts = [torch.randn(9, 16, device=device), torch.randn(70, 8, device=device)]
    outs = [torch.empty(16, device=device), torch.empty(8, device=device)]
    arr = lambda vals: (ctypes.c_void_p * len(vals))(*vals)
    i64 = lambda vals: (ctypes.c_int64 * len(vals))(*vals)
    rb = lib.allset_reduce_partials_batched
    assert rb(arr([P(t) for t in parts]), i64([9, 70]), i64([16, 8]), i64([16, 8]), arr([P(t) for t in outs]), 2, st) == 0
    torch.cuda.synchronize()
    for t, o in zip(parts, outs):
        torch.testing.assert_close(o, t.sum(0), rtol=1e-5, atol=1e-5)
    _err(lib, rb(arr([P(t) for t in parts]), i64([9, 70]), i64([16, 8]), i64([16, 6]), arr([P(t) for t in outs]), 2, st))
    _err(lib, rb(arr([P(t) for t in parts]), i64([9, 70]), i64([12, 8]), i64([16, 8]), arr([P(t) for t in outs]), 2, st))
    _err(lib, rb(arr([P(parts[0]), 0]), i64([9, 70]), i64([16, 8]), i64([16, 8]), arr([P(t) for t in outs]), 2, st))
    _err(lib, rb(None, None, None, None, None, lib.allset_reduce_partials_batch_max() + 1, st))
    c64, cf = torch.zeros(1, dtype=torch.int64, device=device), [torch.zeros((), device=device) for _ in range(3)]
    rx = lib.allset_reduce_partials_batched_ex
    assert rx(None, None, None, None, None, 0, P(c64), arr([P(t) for t in cf]), 3, st) == 0             # counters only
    torch.cuda.synchronize()
    assert int(c64) == 1 and [float(t) for t in cf] == [1.0, 1.0, 1.0]
    _err(lib, rx(None, None, None, None, None, 0, P(c64), None, 3, st))
    _err(lib, rx(None, None, None, None, None, 0, None, arr([0]), lib.allset_reduce_partials_batch_max_counters() + 1, st))
    # loss total
    logits, yl = torch.randn(n, 7, device=device), torch.randint(0, 7, (n,), device=device)
    npart = ctypes.c_int64(0)
    lib.allset_nll_partials(n, ctypes.byref(npart))
    partials, ticket = torch.empty(npart.value + 1, device=device), torch.zeros(1, dtype=torch.int32, device=device)
    lt = lib.allset_nll_logsoftmax_fwd_total
    assert lt(P(logits), 7, P(yl), None, 1.0 / n, P(partials), npart.value, P(ticket), P(partials[npart.value:]), n, 7, st) == 0
    torch.cuda.synchronize()
    ref = torch.nn.functional.nll_loss(torch.log_softmax(logits, 1), yl)
    torch.testing.assert_close(partials[npart.value], ref, rtol=1e-5, atol=1e-6)
    assert int(ticket) == 0                                                      # re-armed
    _err(lib, lt(P(logits), 7, P(yl), None, 1.0 / n, P(partials), npart.value, None, P(partials[npart.value:]), n, 7, st))
    _err(lib, lt(P(logits), 7, P(yl), None, 1.0 / n, P(partials), npart.value + 1, P(ticket), P(partials), n, 7, st))
    torch.cuda.synchronize()
