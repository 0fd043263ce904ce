"""One reduction launch per backward pass at the benchmark's scale (ABI 14): ``dense.deferred_param_grads`` takes the large partial
buffers of the one-pass backward kernels -- 256 slices x 16.8k floats at [n, 128] x [128, 128], 256 x 65.8k in the bf16 regime, which
``allset_reduce_partials`` sums as a two-launch tree -- and bf16 parameters.  Everything here is a bit-for-bit comparison with the
per-kernel reductions: the batched kernel walks the tree with the tree's association (csrc/dense.hip reduce_partials_tree_body)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _P(t):
    return ctypes.c_void_p(t.data_ptr())


def test_batched_reduction_walks_the_two_launch_tree_bit_for_bit(device):
    from allset_amd import _lib, dense
    lib = _lib.load()
    g = torch.Generator(device=device).manual_seed(5)
    # (P, stride, M, bf16 out): tree / tree with a row stride and a bf16 result / one-launch forms / the largest tree (8 slabs)
    shapes = [(256, 16768, 16768, False), (256, 66820, 65792, True), (9, 16, 16, False), (70, 8, 8, True), (512, 4100, 4100, False),
              (300, 1028, 1028, True), (65, 131072, 131072, False)]
    assert all(lib.allset_reduce_partials_batchable(P, M) == 1 for P, _, M, _ in shapes)
    assert lib.allset_reduce_partials_batchable(513, 64) == 0 and lib.allset_reduce_partials_batchable(2048, 768) == 0
    # magnitudes spread over 2^+-12: any other association of the sums shows in the last bits
    parts = [torch.randn(P, s, device=device, generator=g) * torch.exp2(torch.randint(-12, 13, (P, 1), device=device, generator=g).float())
             for P, s, _, _ in shapes]
    ref = [dense.reduce_partials_to(t, M, torch.bfloat16 if b16 else torch.float32) if (b16 or s != M) else dense.reduce_partials(t)
           for t, (P, s, M, b16) in zip(parts, shapes)]
    outs = [torch.empty(M, dtype=torch.bfloat16 if b16 else torch.float32, device=device) for _, _, M, b16 in shapes]
    arr = lambda vals: (ctypes.c_void_p * len(vals))(*vals)
    i64 = lambda vals: (ctypes.c_int64 * len(vals))(*vals)
    dts = (ctypes.c_int32 * len(shapes))(*[_lib.BF16 if b16 else _lib.F32 for *_, b16 in shapes])
    st = dense.stream_of(device)
    rc = lib.allset_reduce_partials_batched_ex2(arr([t.data_ptr() for t in parts]), i64([s[0] for s in shapes]), i64([s[1] for s in shapes]),
                                                i64([s[2] for s in shapes]), arr([t.data_ptr() for t in outs]), None, dts, len(shapes), None, None, 0, st)
    assert rc == 0, lib.allset_last_error()
    torch.cuda.synchronize()
    for o, r, s in zip(outs, ref, shapes):
        assert torch.equal(o, r.reshape(-1)), s
    # the fp32-only entries take the tree as well
    o32 = torch.empty(16768, device=device)
    assert lib.allset_reduce_partials_batched(arr([parts[0].data_ptr()]), i64([256]), i64([16768]), i64([16768]), arr([o32.data_ptr()]), 1, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(o32, ref[0])
    # an accumuland: a COLUMN RANGE of the first buffer as an entry of its own, its sum added to an existing vector in the same launch
    old = torch.randn(1024, device=device, generator=g)
    o_acc = torch.empty(1024, device=device)
    f32 = (ctypes.c_int32 * 1)(_lib.F32 | _lib.REDUCE_AS_TREE)           # (the whole buffer is a two-launch tree: keep its association)
    assert lib.allset_reduce_partials_batched_ex2(arr([parts[0].data_ptr() + 4 * 4096]), i64([256]), i64([16768]), i64([1024]), arr([o_acc.data_ptr()]),
                                                  arr([old.data_ptr()]), f32, 1, None, None, 0, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(o_acc, ref[0][4096:5120] + old)
    assert lib.allset_reduce_partials_batched_ex2(arr([parts[3].data_ptr()]), i64([70]), i64([8]), i64([8]), arr([outs[3].data_ptr()]),
                                                  arr([old.data_ptr()]), (ctypes.c_int32 * 1)(_lib.BF16), 1, None, None, 0, st) != 0      # bf16 output + accumuland
    # argument checks of the new entry: an unknown output type, a bf16 output that is not 8-byte aligned
    bad = (ctypes.c_int32 * 1)(7)
    assert lib.allset_reduce_partials_batched_ex2(arr([parts[2].data_ptr()]), i64([9]), i64([16]), i64([16]), arr([outs[2].data_ptr()]), None, bad, 1,
                                                  None, None, 0, st) != 0
    b16 = (ctypes.c_int32 * 1)(_lib.BF16)
    assert lib.allset_reduce_partials_batched_ex2(arr([parts[3].data_ptr()]), i64([70]), i64([8]), i64([8]), arr([outs[3].data_ptr() + 2]), None, b16, 1,
                                                  None, None, 0, st) != 0
    torch.cuda.synchronize()


def _layer(model, d, heads, dtype, device, seed):
    from allset_amd import HalfNLHconv
    torch.manual_seed(seed)
    a = HalfNLHconv(d, d, d, 2, 0.5, "ln", True, heads=heads, attention=model == "pma").to(device).to(dtype).train()
    b = HalfNLHconv(d, d, d, 2, 0.5, "ln", True, heads=heads, attention=model == "pma").to(device).to(dtype).train()
    return a, b


@pytest.mark.parametrize("model,d,heads,dtype,dist", [("deepsets", 128, 1, torch.float32, "fixed"), ("pma", 128, 4, torch.float32, "fixed"),
                                                      ("pma", 128, 1, torch.float32, "poisson"), ("pma", 256, 4, torch.bfloat16, "zipf"),
                                                      ("deepsets", 64, 1, torch.float32, "fixed")])
def test_layer_step_with_one_reduction_launch_equals_the_eager_step(model, d, heads, dtype, dist, device):
    """A V->E->V layer step in training mode (dropout live, the same seeds) with and without ``deferred_param_grads``: every
    parameter gradient and the input gradient bit-identical; inside the scope the large one-pass kernels reduce nothing themselves
    (the two-launch reductions of the eager step disappear) and every parameter's ``.grad`` is filled on exit; a second backward
    accumulates."""
    from allset_amd import dense, synthetic
    from allset_amd import dist as adist
    n = 40_000
    hg = synthetic.random_hypergraph(n, n, 12, seed=9, device=device, dist=dist)
    a, b = _layer(model, d, heads, dtype, device, 21)
    params = list(a.parameters()) + list(b.parameters())
    sh = adist.ShardedHypergraph(hg.edge_index, n, n, 1, 0).build_incidences()
    g = torch.Generator(device=device).manual_seed(1)
    x = torch.randn(n, d, device=device, generator=g).to(dtype)
    G = torch.randn(n, d, device=device, generator=g).to(dtype)
    fn = adist.sharded_pma_layer if model == "pma" else adist.sharded_deepsets_layer

    def run(deferred):
        torch.manual_seed(77)                                  # the dropout seeds are draws of torch's CPU generator
        xs = x.clone().requires_grad_(True)
        kw = dict(dropout=0.5, training=True)
        if model != "pma":
            kw["aggr"] = "add"
        out = fn(a, b, xs, sh, **kw)
        if deferred:
            with dense.deferred_param_grads():
                out.backward(G)
        else:
            out.backward(G)
        return out.detach(), xs.grad

    calls = {"tree": 0, "all": 0}
    real, real_to = dense.reduce_partials, dense.reduce_partials_to

    def counted(part, *a_, **k_):
        calls["all"] += 1
        P, M = part.shape[0], part[0].numel()
        if P > 64 and P * M > (1 << 21):
            calls["tree"] += 1
        return real(part, *a_, **k_)

    def counted_to(part, M, dt):
        calls["all"] += 1
        if part.shape[0] > 64 and part.shape[0] * M > (1 << 21):
            calls["tree"] += 1
        return real_to(part, M, dt)

    dense.reduce_partials, dense.reduce_partials_to = counted, counted_to
    try:
        o0, gx0 = run(False)
        ref = [p.grad.clone() if p.grad is not None else None for p in params]
        assert sum(r is not None for r in ref) >= 16
        eager_trees = calls["tree"]
        for p in params:
            p.grad = None
        calls["tree"] = calls["all"] = 0
        o1, gx1 = run(True)
        deferred_trees = calls["tree"]
    finally:
        dense.reduce_partials, dense.reduce_partials_to = real, real_to
    assert not dense._Deferred.active and not dense._Deferred.pending
    assert torch.equal(o0, o1) and torch.equal(gx0, gx1)
    for p, r in zip(params, ref):
        if r is None:
            assert p.grad is None
            continue
        assert p.grad is not None and p.grad.dtype == p.dtype and torch.equal(p.grad, r)
    if d == 64:
        return                                                # (one partial row per wave there: too many rows for the batched launch)
    assert eager_trees >= (1 if dtype == torch.bfloat16 else 6 if model == "pma" else 8), eager_trees
    assert deferred_trees == 0, deferred_trees
    _, _ = run(True)                                          # .grad exists: accumulated into
    for p, r in zip(params, ref):
        if r is None:
            continue
        torch.testing.assert_close(p.grad.float(), 2 * r.float(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-6, atol=0)
