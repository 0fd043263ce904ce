"""Dense tail on the HIP kernels of ``csrc/dense.hip``: LayerNorm (fused with the ReLU before it and the dropout
after it), ReLU+dropout, and ``Linear`` whose weight gradient is the split-K fp32-MFMA kernel.  The two
well-shaped GEMMs of a Linear (``x W^T`` and ``gy W``) stay on hipBLASLt through torch.

These are ``torch.autograd.Function``s over ROCm fp32 tensors; ``layers.MLP`` routes here for device tensors.
Dropout masks are a counter-based hash of (seed, element index): the seed is drawn from torch's CPU generator
(so ``torch.manual_seed`` governs it, without a device sync) and the backward regenerates the mask.
"""
from __future__ import annotations

import contextlib
import os
from ctypes import byref, c_int64
from typing import Optional, Tuple

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import check, ptr, require_device, stream_of, on_device
from .ops import _ld, _rowmajor, _timed

Tensor = torch.Tensor


class _SeedSource:
    """Where dropout seeds come from.  Eager: a fresh 63-bit draw from torch's CPU generator per dropout site.
    Under ``device_seed_counter`` (hipGraph capture, ``graphs.py``): the host value is only a per-site salt
    (1, 2, 3, ...) and the randomness comes from a device-resident int64 counter the kernels read at start
    (``seed_base`` in include/allset_hip.h), so replays of a captured graph draw fresh masks."""
    base: Optional[Tensor] = None
    salt: int = 0


_SEED_SCRATCH = torch.empty((), dtype=torch.int64)      # (preallocated: the draw is 1 us instead of 3)


def _next_host_seed() -> int:
    """A fresh 63-bit draw from torch's CPU generator (reproducible under ``torch.manual_seed``)."""
    return int(_SEED_SCRATCH.random_().item())


def _draw_seed() -> int:
    if _SeedSource.base is not None:
        _SeedSource.salt += 1
        return _SeedSource.salt + (_rank() << 32)
    seed = _next_host_seed()
    # Ranks of a sharded job seed torch identically (replicated initial weights), and a mask is a hash of (seed,
    # LOCAL element index): without the rank in the seed every rank would drop the same positions of its row block.
    return (seed ^ (_rank() * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF


def _rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def _seed_base() -> Optional[Tensor]:
    return _SeedSource.base


class device_seed_counter:
    """Context manager: dropout sites inside draw their masks from ``counter`` (int64 device tensor, 1 element)."""

    def __init__(self, counter: Tensor):
        if counter.dtype != torch.int64 or counter.numel() != 1 or not counter.is_cuda:
            raise _lib.AllSetHipError("device_seed_counter needs a 1-element int64 device tensor")
        self.counter = counter

    def __enter__(self):
        self.prev = (_SeedSource.base, _SeedSource.salt)
        _SeedSource.base, _SeedSource.salt = self.counter, 0
        return self

    def __exit__(self, *exc):
        _SeedSource.base, _SeedSource.salt = self.prev
        return False


def _check_f32(*ts: Tensor) -> None:
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise _lib.AllSetHipError(f"dense tail kernels are fp32 (got {t.dtype})")


def _check_dtype(dtype: torch.dtype, *ts: Tensor) -> None:
    for t in ts:
        if t is not None and t.dtype != dtype:
            raise _lib.AllSetHipError(f"expected {dtype} throughout (got {t.dtype})")


# ---- raw wrappers --------------------------------------------------------------------------------

# ---- deferred parameter gradients: ONE reduction launch per backward pass ------------------------------------------------------------
# Every fused backward kernel leaves per-workgroup partial sums of its parameter gradients; `reduce_partials` finishes each with a
# launch of its own -- 8 of the 41 kernels of a graphed Cora-shaped step, ~5 us each because the step is a dependent chain.
# Inside `deferred_param_grads()` a backward node whose parameters are LEAF tensors does not reduce: it queues (partials,
# parameter, section) and returns None for those gradients; leaving the context reduces everything queued with one batched
# launch (`allset_reduce_partials_batched`: the same sums, bit for bit) and assigns / accumulates `.grad`.  Opt-in because the
# gradients bypass autograd's accumulation nodes (tensor hooks on parameters do not see them): `graphs.GraphedTrainStep` uses it.
class _Deferred:
    active = False
    pending: list = []          # (part [P, stride], M, [(param, offset, shape), ...])


def _deferrable(part: Tensor, *params, M: Optional[int] = None) -> bool:
    """``part`` [P, stride] fp32 partials whose first ``M`` columns (default: all) sum to the gradients of ``params`` -- all fp32, or
    all bf16 (ABI 14: the batched launch rounds each sum once, as ``reduce_partials_to`` does)."""
    if not _Deferred.active or part.dim() != 2 or part.dtype != torch.float32:
        return False
    P = part.shape[0]
    M = part.shape[1] if M is None else M
    if M % 4 != 0 or part.stride(1) != 1 or part.stride(0) % 4 != 0 or not _lib.load().allset_reduce_partials_batchable(P, M):
        return False
    # ONE batched launch on ONE device: partials (and parameters) on another device than the scope's first entry reduce the usual way
    if _Deferred.pending and part.device != _Deferred.pending[0][0].device:
        return False
    real = [p for p in params if p is not None]
    if not real or real[0].dtype not in (torch.float32, torch.bfloat16):
        return False
    # a parameter with tensor hooks / post-accumulate hooks (DDP-style gradient sync) must go through autograd's accumulation node
    return all(isinstance(p, torch.nn.Parameter) and p.is_leaf and p.requires_grad and p.dtype == real[0].dtype
               and p.device == part.device and not getattr(p, "_backward_hooks", None)
               and not getattr(p, "_post_accumulate_grad_hooks", None)
               for p in real)


def _defer(part: Tensor, sections, M: Optional[int] = None) -> None:
    """``sections``: (parameter or None, offset into the summed row, shape).  A parameter named twice in one scope (shared weights)
    accumulates."""
    secs = [(p, off, tuple(shape)) for p, off, shape in sections if p is not None]
    _Deferred.pending.append((part, part.shape[1] if M is None else M, secs))


def _defer_or_reduce(part: Tensor, sections, defer: bool):
    """``sections``: [(parameter or None, offset, shape)] of one partial buffer [P, M].  Inside ``deferred_param_grads()`` (and with
    ``defer``) the buffer is queued and every section comes back None; otherwise one reduction launch and the sections as views."""
    part2 = part.reshape(part.shape[0], -1)
    params = [p for p, _, _ in sections]
    if defer and any(p is not None for p in params) and _deferrable(part2, *params):
        _defer(part2, sections)
        return [None] * len(sections)
    red = reduce_partials(part2).reshape(-1)
    out = []
    for _, off, shape in sections:
        numel = 1
        for v in shape:
            numel *= v
        out.append(red[off:off + numel].view(shape))
    return out


def flush_param_grads(bump_i64: Optional[Tensor] = None, bump_f32=None) -> None:
    """Reduce everything queued with one batched launch and assign / accumulate the parameters' ``.grad``.  ``bump_i64`` (a
    1-element int64 device tensor) and ``bump_f32`` (a list of 0-dim float32 device tensors, or a callable returning one -- called
    after the gradients are in place) are advanced by one in the SAME launch: a training step's dropout-seed counter and its
    optimizer's step counters (``graphs.GraphedTrainStep``).  A parameter that already holds a gradient (autograd accumulated
    another use of it during the backward pass -- PMA's ``att_r`` through the folded logits -- or an earlier backward) gets its
    section as an entry of its own whose sum the kernel adds to the existing gradient: no ``add_`` launch per such parameter."""
    pend, _Deferred.pending = _Deferred.pending, []
    lib = _lib.load()
    import ctypes
    dev = pend[0][0].device if pend else (bump_i64.device if bump_i64 is not None else None)
    later = []
    # flat entry list: (partial pointer, P, row stride, M, dtype, base in that dtype's output buffer, accumuland pointer or 0)
    tot = {torch.float32: 0, torch.bfloat16: 0}
    entries, assign = [], []

    def entry(part, col0, M, dt, acc, M_whole=None):
        base = tot[dt]
        tot[dt] += M
        # a column range of a buffer keeps the association the WHOLE buffer's reduction has (the bits of a per-kernel reduction)
        tree = M_whole is not None and bool(lib.allset_reduce_partials_is_tree(part.shape[0], M_whole))
        entries.append((part.data_ptr() + 4 * col0, part.shape[0], part.stride(0), M, dt, base, acc, tree))
        return base

    seen = set()                                  # a parameter queued twice in one scope (shared weights): the second section accumulates
    for part, M, sections in pend:
        dt = sections[0][0].dtype if sections else torch.float32
        whole = None
        for p, off, shape in sections:
            numel = 1
            for v in shape:
                numel *= v
            g_old = p.grad
            if id(p) in seen:
                if whole is None:
                    whole = entry(part, 0, M, dt, 0)
                later.append((p, dt, whole + off, numel, shape))
                continue
            seen.add(id(p))
            if g_old is None:
                if whole is None:
                    whole = entry(part, 0, M, dt, 0)
                assign.append((p, dt, whole + off, numel, shape))
            elif (dt == torch.float32 and numel % 4 == 0 and off % 4 == 0 and g_old.dtype == torch.float32 and g_old.is_contiguous()
                  and g_old.data_ptr() % 16 == 0 and g_old.device == part.device and tuple(g_old.shape) == tuple(shape)
                  and lib.allset_reduce_partials_batchable(part.shape[0], numel)):
                assign.append((p, dt, entry(part, off, numel, dt, g_old.data_ptr(), M), numel, shape))   # sum + the existing gradient
            else:
                if whole is None:
                    whole = entry(part, 0, M, dt, 0)
                later.append((p, dt, whole + off, numel, shape))
    outs = {dt: (torch.empty(n, dtype=dt, device=dev) if n else None) for dt, n in tot.items()}
    keep = []                                     # (the accumulands stay alive until the launch has been issued)
    with torch.no_grad():
        for p, dt, b, numel, shape in assign:
            keep.append(p.grad)
            p.grad = outs[dt][b:b + numel].view(shape)
    counters = list(bump_f32() if callable(bump_f32) else (bump_f32 or []))
    if dev is None and counters:
        dev = counters[0].device
    if dev is None:
        return
    cap = int(lib.allset_reduce_partials_batch_max())
    ccap = int(lib.allset_reduce_partials_batch_max_counters())
    if len(counters) > ccap:                  # (more optimizer counters than one launch takes: the rest the usual way)
        torch._foreach_add_(counters[ccap:], 1.0)
        counters = counters[:ccap]
    first = True
    arr = lambda vals: (ctypes.c_void_p * max(len(vals), 1))(*vals)
    i64 = lambda vals: (ctypes.c_int64 * max(len(vals), 1))(*vals)
    for k0 in range(0, max(len(entries), 1), cap):
        chunk = entries[k0:k0 + cap]
        n = len(chunk)
        cs = counters if first else []
        optrs = [outs[e[4]].data_ptr() + outs[e[4]].element_size() * e[5] for e in chunk]
        odts = (ctypes.c_int32 * max(n, 1))(*[(_lib.BF16 if e[4] == torch.bfloat16 else _lib.F32) | (_lib.REDUCE_AS_TREE if e[7] else 0) for e in chunk])
        with on_device(dev):
            check(lib.allset_reduce_partials_batched_ex2(
                arr([e[0] for e in chunk]), i64([e[1] for e in chunk]), i64([e[2] for e in chunk]), i64([e[3] for e in chunk]),
                arr(optrs), arr([e[6] for e in chunk]) if any(e[6] for e in chunk) else None, odts, n,
                ptr(bump_i64) if first else None, arr([c.data_ptr() for c in cs]), len(cs), stream_of(dev)),
                "allset_reduce_partials_batched_ex2")
        first = False
    del keep
    with torch.no_grad():
        for p, dt, b, numel, shape in later:
            p.grad.add_(outs[dt][b:b + numel].view(shape))


class deferred_param_grads:
    """Context manager around ``loss.backward()``: parameter gradients of the fused backward kernels are reduced by ONE batched
    launch on exit (see above).  ``bump_i64`` / ``bump_f32``: counters advanced by one in that launch (:func:`flush_param_grads`).
    Not re-entrant."""

    def __init__(self, bump_i64: Optional[Tensor] = None, bump_f32=None):
        self.bump_i64, self.bump_f32 = bump_i64, bump_f32

    def __enter__(self):
        if _Deferred.active:
            raise _lib.AllSetHipError("deferred_param_grads is not re-entrant")
        _Deferred.active, _Deferred.pending = True, []
        return self

    def __exit__(self, exc_type, exc, tb):
        _Deferred.active = False
        if exc_type is None:
            flush_param_grads(self.bump_i64, self.bump_f32)
        else:
            _Deferred.pending = []
        return False


def reduce_partials(part: Tensor) -> Tensor:
    """Sum a partial buffer [P, ...] over its first axis with the dedicated kernel (the torch reduction runs at
    ~1 TB/s on these shapes); returns a tensor shaped like ``part[0]``."""
    P = part.shape[0]
    if P == 1:
        return part[0]
    M = part[0].numel()
    if M % 4 != 0 or P > 4096 or os.environ.get("ALLSET_TORCH_REDUCE", "0") == "1":
        return part.sum(dim=0)
    dev = part.device
    out = torch.empty(part.shape[1:], dtype=torch.float32, device=dev)
    scratch = torch.empty(((P + 63) // 64) * M, dtype=torch.float32, device=dev) if P > 64 else None
    with on_device(dev):
        check(_lib.load().allset_reduce_partials(ptr(part), P, M, ptr(out), ptr(scratch), stream_of(dev)),
              "allset_reduce_partials")
    return out


def reduce_partials_slice(part: Tensor, col0: int, M: int, dtype: torch.dtype) -> Tensor:
    """Sum columns ``[col0, col0 + M)`` of a partial buffer over its rows with the bits the reduction of the WHOLE buffer gives for
    them (a large buffer is summed as a two-launch tree, a small one in one launch with another association): what a backward node
    uses for the part of its partials it cannot queue in ``deferred_param_grads()``."""
    P, width = part.shape
    lib = _lib.load()
    view = part[:, col0:col0 + M]
    if not lib.allset_reduce_partials_is_tree(P, width) or lib.allset_reduce_partials_is_tree(P, M) \
            or not lib.allset_reduce_partials_batchable(P, M):
        return reduce_partials_to(view, M, dtype)
    import ctypes
    dev = part.device
    out = torch.empty(M, dtype=dtype, device=dev)
    flags = (ctypes.c_int32 * 1)((_lib.BF16 if dtype == torch.bfloat16 else _lib.F32) | _lib.REDUCE_AS_TREE)
    with on_device(dev):
        check(lib.allset_reduce_partials_batched_ex2((ctypes.c_void_p * 1)(view.data_ptr()), (ctypes.c_int64 * 1)(P),
                                                     (ctypes.c_int64 * 1)(part.stride(0)), (ctypes.c_int64 * 1)(M),
                                                     (ctypes.c_void_p * 1)(out.data_ptr()), None, flags, 1, None, None, 0, stream_of(dev)),
              "allset_reduce_partials_batched_ex2")
    return out


def reduce_partials_to(part: Tensor, M: int, dtype: torch.dtype) -> Tensor:
    """Sum the first ``M`` columns of a partial buffer [P, stride] over its rows; ``dtype`` float32 or bfloat16 (rounded once)."""
    P, stride = part.shape[0], part.stride(0)          # (a column slice of a wider partial buffer keeps the buffer's row stride)
    if part.stride(1) != 1:
        raise _lib.AllSetHipError("reduce_partials_to: rows must be contiguous")
    dev = part.device
    out = torch.empty(M, dtype=dtype, device=dev)
    scratch = torch.empty(((P + 63) // 64) * M, dtype=torch.float32, device=dev) if P > 64 else None
    with on_device(dev):
        check(_lib.load().allset_reduce_partials_ex(ptr(part), P, stride, M, ptr(out), 1 if dtype == torch.bfloat16 else 0,
                                                    ptr(scratch), stream_of(dev)), "allset_reduce_partials_ex")
    return out


def ln_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, relu_in: bool, p: float, seed: int,
           seed_base: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    dev = require_device(x, gamma, beta)
    x = _rowmajor(x)
    n, d = x.shape
    y = torch.empty((n, d), dtype=x.dtype, device=dev)
    stats = torch.empty((n, 2), dtype=torch.float32, device=dev)
    if x.dtype == torch.bfloat16:                 # bf16 activations and parameters, fp32 arithmetic (configs[4] regime)
        if gamma.dtype != torch.bfloat16 or beta.dtype != torch.bfloat16 or not ln_bf16_supported(d):
            raise _lib.AllSetHipError("bf16 LayerNorm: bf16 gamma / beta and d % 8 == 0, d <= 512 required")
        with on_device(dev), _timed("ln_fwd", dev, 2 * n * d * 2):
            check(_lib.load().allset_ln_fwd_bf16(ptr(x), _ld(x), ptr(gamma.contiguous()), ptr(beta.contiguous()), eps,
                                                 int(relu_in), p, seed, ptr(y), max(d, 1), ptr(stats), n, d, ptr(seed_base),
                                                 stream_of(dev)), "allset_ln_fwd_bf16")
        return y, stats
    _check_f32(x, gamma, beta)
    with on_device(dev), _timed("ln_fwd", dev, 2 * n * d * 4):
        check(_lib.load().allset_ln_fwd(ptr(x), _ld(x), ptr(gamma.contiguous()), ptr(beta.contiguous()), eps,
                                        int(relu_in), p, seed, ptr(y), max(d, 1), ptr(stats), n, d, ptr(seed_base),
                                        stream_of(dev)),
              "allset_ln_fwd")
    return y, stats


def ln_bf16_supported(d: int) -> bool:
    return bool(_lib.load().allset_ln_bf16_supported(d))


def ln_bwd(gy: Tensor, x: Tensor, stats: Tensor, gamma: Tensor, relu_in: bool, p: float, seed: int,
           seed_base: Optional[Tensor] = None, want_gx: bool = True, defer_to=None) -> Tuple[Optional[Tensor], Tensor, Tensor]:
    dev = require_device(gy, x, stats, gamma)
    gy, x = _rowmajor(gy), _rowmajor(x)
    n, d = x.shape
    lib = _lib.load()
    npart = c_int64(0)
    if x.dtype == torch.bfloat16:
        if gy.dtype != torch.bfloat16:
            gy = gy.to(torch.bfloat16)
        check(lib.allset_ln_bwd_bf16_partials(n, d, byref(npart)), "allset_ln_bwd_bf16_partials")
        partials = torch.empty((npart.value, 2, d), dtype=torch.float32, device=dev)
        gx = torch.empty((n, d), dtype=x.dtype, device=dev) if want_gx else None
        with on_device(dev), _timed("ln_bwd", dev, (3 if want_gx else 2) * n * d * 2):
            check(lib.allset_ln_bwd_bf16(ptr(gy), _ld(gy), ptr(x), _ld(x), ptr(stats), ptr(gamma.contiguous()), int(relu_in), p,
                                         seed, ptr(gx), max(d, 1), ptr(partials), npart.value, n, d, ptr(seed_base),
                                         stream_of(dev)), "allset_ln_bwd_bf16")
        if (2 * d) % 4 == 0 and npart.value <= 4096:
            red = reduce_partials_to(partials.view(npart.value, 2 * d), 2 * d, torch.bfloat16).view(2, d)
            return gx, red[0], red[1]
        red = reduce_partials(partials)
        return gx, red[0].to(torch.bfloat16), red[1].to(torch.bfloat16)
    check(lib.allset_ln_bwd_partials(n, d, byref(npart)), "allset_ln_bwd_partials")
    partials = torch.empty((npart.value, 2, d), dtype=torch.float32, device=dev)
    gx = torch.empty((n, d), dtype=x.dtype, device=dev) if want_gx else None
    with on_device(dev), _timed("ln_bwd", dev, (3 if want_gx else 2) * n * d * 4):
        check(lib.allset_ln_bwd(ptr(gy), _ld(gy), ptr(x), _ld(x), ptr(stats), ptr(gamma.contiguous()), int(relu_in), p,
                                seed, ptr(gx), max(d, 1), ptr(partials), npart.value, n, d, ptr(seed_base),
                                stream_of(dev)),
              "allset_ln_bwd")
    dg_, db_ = _defer_or_reduce(partials, [(defer_to[0] if defer_to else None, 0, (d,)), (defer_to[1] if defer_to else None, d, (d,))],
                                defer_to is not None)
    return gx, dg_, db_


def wgrad(ga: Tensor, u: Tensor, want_bias: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
    """gW [O, I] = ga^T @ u,  gb [O] = ga.sum(0)  (ga: [n, O], u: [n, I]); fp32, or bf16 activations (fp32 accumulation,
    result cast back to bf16)."""
    dev = require_device(ga, u)
    ga, u = _rowmajor(ga), _rowmajor(u)
    n, O = ga.shape
    I = u.shape[1]
    lib = _lib.load()
    ns = c_int64(0)
    bf16 = ga.dtype == torch.bfloat16
    if bf16:
        check(lib.allset_wgrad_bf16_slices(n, O, I, byref(ns)), "allset_wgrad_bf16_slices")
    else:
        check(lib.allset_wgrad_slices(n, O, I, byref(ns)), "allset_wgrad_slices")
    if bf16 and ns.value <= 4096:
        # one partial buffer [slices, gW | gb], one reduction launch that also rounds to bf16 (this regime is launch-bound);
        # O and I are multiples of 4 (checked by the kernel), so the row needs no padding
        M = O * I + (O if want_bias else 0)
        part = torch.empty((ns.value, M), dtype=torch.float32, device=dev)
        with on_device(dev), _timed("wgrad", dev, n * (O + I) * ga.element_size()):
            check(lib.allset_wgrad_bf16_ex(ptr(ga), _ld(ga), ptr(u), _ld(u), ptr(part), M, int(want_bias), ns.value, n, O, I,
                                           stream_of(dev)), "allset_wgrad_bf16_ex")
        red = reduce_partials_to(part, M, torch.bfloat16)
        return red[:O * I].view(O, I), (red[O * I:] if want_bias else None)
    if not bf16 and O % 4 == 0 and I % 4 == 0:
        M = O * I + (O if want_bias else 0)                # one partial buffer [slices, gW | gb], one reduction launch
        part = torch.empty((ns.value, M), dtype=torch.float32, device=dev)
        with on_device(dev), _timed("wgrad", dev, n * (O + I) * 4):
            check(lib.allset_wgrad_fused_ex(ptr(ga), _ld(ga), None, 0, 0.0, ptr(u), _ld(u), None, None, None, 0, 0.0, 0, ptr(part), M,
                                            int(want_bias), ns.value, n, O, I, None, None, stream_of(dev)), "allset_wgrad_fused_ex")
        red = reduce_partials(part)
        return red[:O * I].view(O, I), (red[O * I:] if want_bias else None)
    part_w = torch.empty((ns.value, O, I), dtype=torch.float32, device=dev)
    part_b = torch.empty((ns.value, O), dtype=torch.float32, device=dev) if want_bias else None
    with on_device(dev), _timed("wgrad", dev, n * (O + I) * ga.element_size()):
        if bf16:
            check(lib.allset_wgrad_bf16(ptr(ga), _ld(ga), ptr(u), _ld(u), ptr(part_w), ptr(part_b), ns.value, n, O, I,
                                        stream_of(dev)), "allset_wgrad_bf16")
        else:
            check(lib.allset_wgrad(ptr(ga), _ld(ga), ptr(u), _ld(u), ptr(part_w), ptr(part_b), ns.value, n, O, I,
                                   stream_of(dev)), "allset_wgrad")
    gw = reduce_partials(part_w)
    gb = reduce_partials(part_b) if want_bias else None
    if bf16:
        gw, gb = gw.to(torch.bfloat16), (gb.to(torch.bfloat16) if gb is not None else None)
    return gw, gb


# ---- wide Linear layers (csrc/wide_mlp.hip): tiled GEMM, fp32-accurate on the bf16 matrix pipe --------------------------

def gemm_x6_supported(N: int, K: int) -> bool:
    return bool(_lib.load().allset_gemm_x6_supported(N, K))


def wide_f16(ln_forward: bool = False) -> bool:
    """Which arithmetic a tiled wide GEMM (widths 256 / 512) runs in: the exact bf16x6 split under ``set_arithmetic("strict")``, two
    fp16 planes otherwise (auto / fp16x3).  Measured at [1M, 256] x [256, 256] (profiles/r05d_d256*_bench_line.json): forward behind a
    LayerNorm 0.81 -> 0.67 ms, backward-data with the LayerNorm-backward epilogue 1.16 -> 1.01 ms -- modest, because these kernels are
    bound by load latency at one workgroup per CU, not by the matrix pipe.  (``ln_forward`` kept for callers.)"""
    return _arith != _lib.ARITH_BF16X6


class _Planes:
    """Opaque pre-split weight planes of :func:`gemm_x6` + which arithmetic they were split for."""
    __slots__ = ("buf", "f16")

    def __init__(self, buf: Tensor, f16: bool):
        self.buf, self.f16 = buf, f16


# ---- the plane images of a whole step, built by ONE launch ------------------------------------------------------------------------
# A training step at the reference's tuned widths (MLP_hidden 256 / 512) rebuilds the fp16 plane images of W (forward) and W^T
# (backward-data) of every wide Linear -- weights change every step -- each a ~5-us launch in front of its GEMM.  All weights are known
# when the forward starts: `prefetch_wide_planes` builds every image a model will ask for in one batched launch and
# `gemm_x6_planes` serves them.  An entry is keyed by the weight's storage, shape, version and the weight epoch (bumped by every
# optimizer step that writes parameters behind torch's back: allset_amd.optim.FusedAdam), consumed ONCE, and dropped by the next prefetch.
class _PlaneStore:
    entries: dict = {}
    epoch = 0


def weights_changed() -> None:
    """Parameters were updated in place by something torch's version counters do not see (a raw-pointer optimizer kernel)."""
    _PlaneStore.epoch += 1
    _PlaneStore.entries = {}


def _plane_key(W: Tensor, transpose: bool):
    return (W.data_ptr(), tuple(W.shape), W.stride(0), W._version, _PlaneStore.epoch, bool(transpose))


def prefetch_wide_planes(weights, with_transposed: bool) -> None:
    """``weights``: fp32 [out, in] weight tensors of Linears on the tiled wide path; builds planes(W) -- and planes(W^T) when a backward
    will follow -- of all of them with one launch (fp16-plane arithmetic only; anything it does not take is built on demand as before)."""
    _PlaneStore.entries = {}
    if not weights or not wide_f16():
        return
    import ctypes
    lib = _lib.load()
    cap = int(lib.allset_gemm_f16x3_planes_batch_max())
    todo = []
    for W in weights:
        if not (W.is_cuda and W.dtype == torch.float32 and W.dim() == 2 and W.stride(1) == 1 and W.data_ptr() % 16 == 0):
            continue
        for tr in ((False, True) if with_transposed else (False,)):
            N, K = (W.shape[1], W.shape[0]) if tr else (W.shape[0], W.shape[1])
            nbytes = int(lib.allset_gemm_f16x3_plane_bytes(N, K))
            if nbytes > 0:
                todo.append((W, tr, N, K, nbytes))
    if not todo:
        return
    dev = todo[0][0].device
    todo = [t for t in todo if t[0].device == dev]
    for k0 in range(0, len(todo), cap):
        chunk = todo[k0:k0 + cap]
        bufs = [torch.empty(t[4], dtype=torch.uint8, device=dev) for t in chunk]
        n = len(chunk)
        with on_device(dev):
            check(lib.allset_gemm_f16x3_planes_batched(
                (ctypes.c_void_p * n)(*[t[0].data_ptr() for t in chunk]), (ctypes.c_int64 * n)(*[t[0].stride(0) for t in chunk]),
                (ctypes.c_int32 * n)(*[int(t[1]) for t in chunk]), (ctypes.c_void_p * n)(*[b.data_ptr() for b in bufs]),
                (ctypes.c_int64 * n)(*[t[2] for t in chunk]), (ctypes.c_int64 * n)(*[t[3] for t in chunk]), n, stream_of(dev)),
                "allset_gemm_f16x3_planes_batched")
        for t, b in zip(chunk, bufs):
            # the entry keeps the weight tensor ALIVE: while an image waits here its key (an address) cannot come to mean another tensor
            # (a dead model's weight freed, a new model's weight of the same shape allocated in its place, version 0 again)
            _PlaneStore.entries[_plane_key(t[0], t[1])] = (_Planes(b, True), t[0])


def gemm_x6_planes(W: Tensor, transpose: bool, f16: Optional[bool] = None) -> "_Planes":
    """Pre-split planes of ``B`` for :func:`gemm_x6`: ``B = W`` ([N, K]) or, with ``transpose``, ``B = W^T`` (``W`` [K, N]) --
    three bf16 planes (exact split), or two fp16 planes + per-column scales (``f16``; default: the process-wide arithmetic)."""
    dev = require_device(W)
    _check_f32(W)
    W = _rowmajor(W)
    f16 = wide_f16() if f16 is None else bool(f16)            # (callers that know their prologue pass f16 themselves)
    if f16 and _PlaneStore.entries:
        hit = _PlaneStore.entries.pop(_plane_key(W, transpose), None)       # built by this step's prefetch launch; used once
        if hit is not None:
            return hit[0]
    N, K = (W.shape[1], W.shape[0]) if transpose else (W.shape[0], W.shape[1])
    lib = _lib.load()
    nbytes = int((lib.allset_gemm_f16x3_plane_bytes if f16 else lib.allset_gemm_x6_plane_bytes)(N, K))
    if nbytes < 0:
        raise _lib.AllSetHipError(f"gemm_x6: N={N}, K={K} not supported (K % 32 == 0, N % 4 == 0)")
    planes = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with on_device(dev):
        fn, name = (lib.allset_gemm_f16x3_planes, "allset_gemm_f16x3_planes") if f16 else (lib.allset_gemm_x6_planes, "allset_gemm_x6_planes")
        check(fn(ptr(W), _ld(W), int(transpose), ptr(planes), N, K, stream_of(dev)), name)
    return _Planes(planes, f16)


def row_stats(x: Tensor, relu_in: bool, eps: float) -> Tensor:
    """[n, 2] {mean, rstd} of the rows of ``relu_in ? relu(x) : x`` -- the LayerNorm-apply prologue's input."""
    dev = require_device(x)
    _check_f32(x)
    x = _rowmajor(x)
    n, d = x.shape
    stats = torch.empty((n, 2), dtype=torch.float32, device=dev)
    with on_device(dev), _timed("row_stats", dev, n * d * 4):
        check(_lib.load().allset_row_stats(ptr(x), _ld(x), int(relu_in), eps, ptr(stats), n, d, stream_of(dev)), "allset_row_stats")
    return stats


def gemm_x6(A: Tensor, planes: Tensor, N: int, bias: Optional[Tensor] = None, *, mask_y: Optional[Tensor] = None,
            p_mask: float = 0.0, relu_in: bool = False, stats: Optional[Tensor] = None, gamma: Optional[Tensor] = None,
            beta: Optional[Tensor] = None, p_in: float = 0.0, seed_in: int = 0, relu_out: bool = False, p_out: float = 0.0,
            seed_out: int = 0, seed_base: Optional[Tensor] = None, mask_bits: Optional[Tensor] = None,
            mask_out: Optional[Tensor] = None, stats_out: Optional[Tensor] = None, stats_eps: float = 1e-5,
            stats_relu: bool = False, sgn_x: Optional[Tensor] = None) -> Tensor:
    """``out = epi(pro(A) @ B^T + bias)`` with ``B`` given as :func:`gemm_x6_planes` (include/allset_hip_ext.h allset_gemm_x6 /
    allset_gemm_f16x3: the planes say which).  ``stats_out`` [n, 2] (N == 256): the row statistics of ``relu?(out)`` for the next
    Linear's LayerNorm prologue, written by the epilogue instead of a :func:`row_stats` pass over the output.  ``sgn_x`` [n, N]
    (backward-data of a Linear behind a bare relu: allset_gemm_wide_sgn): the result where ``sgn_x > 0``, zero elsewhere."""
    planes, f16 = planes.buf, planes.f16
    dev = require_device(A, planes, bias, mask_y, stats, gamma, beta, sgn_x)
    _check_f32(A, bias, mask_y, stats, gamma, beta, sgn_x)
    A = _rowmajor(A)
    n, K = A.shape
    if mask_y is not None:
        mask_y = _rowmajor(mask_y)
    out = torch.empty((n, N), dtype=torch.float32, device=dev)
    lib = _lib.load()
    if sgn_x is not None:
        if (bias is not None or relu_in or stats is not None or p_in or relu_out or p_out or mask_out is not None or stats_out is not None
                or tuple(sgn_x.shape) != (n, N)):
            raise _lib.AllSetHipError("gemm_x6(sgn_x=...): a backward-data GEMM takes no prologue / epilogue options besides the masks")
        sgn_x = _rowmajor(sgn_x)
        with on_device(dev), _timed("gemm_x6", dev, n * (K * (2 if mask_y is not None else 1) + 2 * N) * 4):
            check(lib.allset_gemm_wide_sgn(
                _lib.ARITH_FP16X3 if f16 else _lib.ARITH_BF16X6, ptr(A), _ld(A), ptr(mask_y), _ld(mask_y) if mask_y is not None else 0,
                ptr(mask_bits), p_mask, ptr(planes), ptr(sgn_x), _ld(sgn_x), ptr(out), max(N, 1), n, N, K, ptr(seed_base), stream_of(dev)),
                "allset_gemm_wide_sgn")
        return out
    with on_device(dev), _timed("gemm_x6", dev, n * (K * (2 if mask_y is not None else 1) + N) * 4):
        check(lib.allset_gemm_wide(
            _lib.ARITH_FP16X3 if f16 else _lib.ARITH_BF16X6, ptr(A), _ld(A), ptr(mask_y), _ld(mask_y) if mask_y is not None else 0,
            ptr(mask_bits), p_mask, int(relu_in), ptr(stats),
            ptr(gamma.contiguous() if gamma is not None else None), ptr(beta.contiguous() if beta is not None else None), p_in,
            seed_in, ptr(planes), ptr(bias.contiguous() if bias is not None else None), int(relu_out), p_out, seed_out,
            ptr(mask_out), ptr(stats_out), stats_eps, int(stats_relu), ptr(out), max(N, 1), n, N, K, ptr(seed_base), stream_of(dev)),
            "allset_gemm_wide")
    return out


def gemm_x6_lnb(G: Tensor, planes_t: Tensor, x: Tensor, stats: Tensor, gamma: Tensor, relu_in: bool, p: float, seed: int,
                mask_y: Optional[Tensor] = None, p_mask: float = 0.0, seed_base: Optional[Tensor] = None,
                mask_bits: Optional[Tensor] = None, defer_to=None) -> Tuple[Tensor, Tensor, Tensor]:
    """Backward-data of a wide Linear fused with the LayerNorm backward of its input (include/allset_hip.h
    allset_gemm_x6_lnb): returns (gx, dgamma, dbeta).  ``planes_t``: ``gemm_x6_planes(weight, True)``; N = x.shape[1] <= 256."""
    planes_t, f16 = planes_t.buf, planes_t.f16
    dev = require_device(G, planes_t, x, stats, gamma, mask_y)
    _check_f32(G, x, stats, gamma, mask_y)
    G, x = _rowmajor(G), _rowmajor(x)
    n, K = G.shape
    N = x.shape[1]
    if mask_y is not None:
        mask_y = _rowmajor(mask_y)
    lib = _lib.load()
    npart = int(lib.allset_gemm_x6_lnb_partials(n))
    partials = torch.empty((npart, 2, N), dtype=torch.float32, device=dev)
    gx = torch.empty((n, N), dtype=torch.float32, device=dev)
    with on_device(dev), _timed("gemm_x6_lnb", dev, n * (K * (2 if mask_y is not None else 1) + 2 * N) * 4):
        check(lib.allset_gemm_wide_lnb(_lib.ARITH_FP16X3 if f16 else _lib.ARITH_BF16X6, ptr(G), _ld(G), ptr(mask_y),
                                       _ld(mask_y) if mask_y is not None else 0, ptr(mask_bits), p_mask, ptr(planes_t),
                                       ptr(x), _ld(x), ptr(stats), ptr(gamma.contiguous()), int(relu_in), p, seed, ptr(gx), max(N, 1),
                                       ptr(partials), npart, n, N, K, ptr(seed_base), stream_of(dev)), "allset_gemm_wide_lnb")
    dg_, db_ = _defer_or_reduce(partials, [(defer_to[0] if defer_to else None, 0, (N,)), (defer_to[1] if defer_to else None, N, (N,))],
                                defer_to is not None)
    return gx, dg_, db_


# ---- arithmetic of the fused Linear kernels (include/allset_hip_ext.h ALLSET_ARITH_*) ---------------------------------------------
_ARITH_NAMES = {"auto": _lib.ARITH_AUTO, "bf16x6": _lib.ARITH_BF16X6, "strict": _lib.ARITH_BF16X6, "exact": _lib.ARITH_BF16X6,
                "fp16x3": _lib.ARITH_FP16X3}
_arith = _lib.ARITH_AUTO


def set_arithmetic(mode) -> str:
    """How the fused Linear kernels (widths 64 / 128) emulate the reference's fp32 products on the 16-bit matrix pipe:
    ``"auto"`` (default: fp16x3 where built -- K = N = 128 --, bf16x6 elsewhere; a function of the shapes only), ``"bf16x6"`` (alias
    ``"strict"`` / ``"exact"``: the exact three-plane split everywhere, accurate for ANY dynamic range), ``"fp16x3"`` (as auto where
    it is built; shapes without an fp16x3 kernel keep bf16x6).  Process-wide; returns the previous mode's name.  The choice travels
    as an argument of every call (the library keeps no state).  A backward runs in the mode that is set when IT runs."""
    global _arith
    if isinstance(mode, str):
        if mode not in _ARITH_NAMES:
            raise ValueError(f"set_arithmetic: unknown mode {mode!r} (auto, bf16x6 / strict / exact, fp16x3)")
        code = _ARITH_NAMES[mode]
    else:
        code = int(mode)
        if code not in (_lib.ARITH_AUTO, _lib.ARITH_BF16X6, _lib.ARITH_FP16X3):
            raise ValueError(f"set_arithmetic: unknown mode {mode!r}")
    prev, _arith = _arith, code
    return get_arithmetic(prev)


def get_arithmetic(code: Optional[int] = None) -> str:
    return {_lib.ARITH_AUTO: "auto", _lib.ARITH_BF16X6: "bf16x6", _lib.ARITH_FP16X3: "fp16x3"}[_arith if code is None else code]


class arithmetic:
    """``with dense.arithmetic("strict"): ...`` -- :func:`set_arithmetic` for a block."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = set_arithmetic(self.mode)
        return self

    def __exit__(self, *exc):
        set_arithmetic(self.prev)
        return False


def _arith_for(direction: int, K: int, N: int, has_ln: bool, norm_mode: int, has_aux: bool = False) -> int:
    """The ``arith`` argument of a call: the process-wide mode; an explicit fp16x3 request degrades to AUTO for shapes the fp16x3
    kernels are not built for (the library itself would refuse it) -- and for the forward with auxiliary output columns (PMA's value
    projection with the folded logit columns), which the C predicate cannot see: ``set_arithmetic("fp16x3")`` documents "shapes
    without an fp16x3 kernel keep bf16x6"."""
    if _arith == _lib.ARITH_FP16X3 and (has_aux or not _lib.load().allset_fused_linear_arith_supported(direction, int(K), int(N), int(has_ln),
                                                                                          int(norm_mode), _lib.ARITH_FP16X3)):
        return _lib.ARITH_AUTO
    return _arith


def fused_linear_supported(K: int, N: int) -> bool:
    return bool(_lib.load().allset_fused_linear_supported(K, N))


def x6_active() -> bool:
    """Always true since ABI 9 (one kernel family: fp32-accurate arithmetic on the bf16 matrix pipe); kept for callers."""
    return int(_lib.load().allset_fused_linear_mask_words(16, 64)) > 0


def activation_mask_words(n: int, N: int) -> int:
    """dwords of the 1-bit activation mask of an [n, N] output (0: not supported in this mode / for this width)."""
    return int(_lib.load().allset_fused_linear_mask_words(n, N))


def fused_linear_fwd(x: Tensor, weight: Tensor, bias: Optional[Tensor], gamma: Optional[Tensor] = None,
                     beta: Optional[Tensor] = None, eps: float = 1e-5, relu_in: bool = False, p_in: float = 0.0,
                     seed_in: int = 0, relu_out: bool = False, p_out: float = 0.0, seed_out: int = 0,
                     seed_base: Optional[Tensor] = None, mask_out: Optional[Tensor] = None,
                     aux_w: Optional[Tensor] = None, aux_b: Optional[Tensor] = None, aux_out: Optional[Tensor] = None,
                     norm_mode: int = 0) -> Tuple[Tensor, Optional[Tensor]]:
    """y = epi(pro(x) @ W^T + b) in one pass (csrc/fused_mlp.hip).  Returns (y, stats or None).  ``mask_out`` (int32
    tensor of ``activation_mask_words(n, N)`` elements) receives the 1-bit ``y > 0`` mask for the backward kernels.
    ``norm_mode`` 1 (ALLSET_NORM_COLUMN_AFFINE): (gamma, beta) are a per-column scale / shift, no row statistics."""
    dev = require_device(x, weight, bias, gamma, beta)
    _check_f32(x, weight, bias, gamma, beta)
    x = _rowmajor(x)
    n, K = x.shape
    N = weight.shape[0]
    weight = weight.contiguous()
    y = torch.empty((n, N), dtype=x.dtype, device=dev)
    stats = torch.empty((n, 2), dtype=torch.float32, device=dev) if gamma is not None else None
    if norm_mode and (aux_out is not None or gamma is None):
        raise _lib.AllSetHipError("fused_linear_fwd: the column-affine prologue takes gamma / beta and no auxiliary columns")
    with on_device(dev), _timed("fused_linear_fwd", dev, n * (K + N) * 4):
        check(_lib.load().allset_fused_linear_fwd_ex(
            ptr(x), _ld(x), 0, ptr(gamma.contiguous() if gamma is not None else None),
            ptr(beta.contiguous() if beta is not None else None), eps, int(norm_mode), int(relu_in), p_in, seed_in, ptr(weight),
            ptr(bias.contiguous() if bias is not None else None), int(relu_out), p_out, seed_out, ptr(y), max(N, 1), 0,
            ptr(stats), n, K, N, ptr(seed_base), ptr(mask_out), ptr(aux_w), ptr(aux_b), ptr(aux_out),
            _arith_for(0, K, N, gamma is not None, norm_mode, aux_out is not None), stream_of(dev)), "allset_fused_linear_fwd_ex")
    return y, stats


def blocked_linear_supported(K: int, N: int) -> bool:
    """The fused Linear can read / write COLUMN-BLOCKED operands at these widths (include/allset_hip.h, ABI 8)."""
    return bool(_lib.load().allset_fused_linear_blocked_supported(int(K), int(N)))


def fused_linear_fwd_blocked(x: Tensor, x_cb: int, weight: Tensor, bias: Optional[Tensor], gamma: Optional[Tensor],
                             beta: Optional[Tensor], eps: float, relu_in: bool, p_in: float, seed_in: int, relu_out: bool,
                             p_out: float, seed_out: int, seed_base: Optional[Tensor], mask_out: Optional[Tensor], y_cb: int
                             ) -> Tuple[Tensor, Optional[Tensor]]:
    """:func:`fused_linear_fwd` with ``x`` and / or ``y`` column-blocked: a blocked [n, C] operand with block width cb is the 2-D
    tensor [(C / cb) * n, cb] (``x_cb`` / ``y_cb`` = 0: plain [n, C])."""
    dev = require_device(x, weight, bias, gamma, beta)
    _check_f32(x, weight, bias, gamma, beta)
    N, K = weight.shape
    x = x.contiguous() if x_cb else _rowmajor(x)
    n = x.shape[0] // (K // x_cb) if x_cb else x.shape[0]
    if x_cb and (x.shape[1] != x_cb or x.shape[0] != (K // x_cb) * n):
        raise _lib.AllSetHipError(f"fused_linear_fwd_blocked: x of shape {tuple(x.shape)} is not [{K} / {x_cb} blocks x n, {x_cb}]")
    weight = weight.contiguous()
    y = torch.empty(((N // y_cb) * n, y_cb) if y_cb else (n, N), dtype=x.dtype, device=dev)
    stats = torch.empty((n, 2), dtype=torch.float32, device=dev) if gamma is not None else None
    with on_device(dev), _timed("fused_linear_fwd", dev, n * (K + N) * 4):
        check(_lib.load().allset_fused_linear_fwd_ex(
            ptr(x), x_cb if x_cb else _ld(x), x_cb, ptr(gamma.contiguous() if gamma is not None else None),
            ptr(beta.contiguous() if beta is not None else None), eps, 0, int(relu_in), p_in, seed_in, ptr(weight),
            ptr(bias.contiguous() if bias is not None else None), int(relu_out), p_out, seed_out, ptr(y), y_cb if y_cb else max(N, 1),
            y_cb, ptr(stats), n, K, N, ptr(seed_base), ptr(mask_out), ptr(None), ptr(None), ptr(None),
            _arith_for(0, K, N, gamma is not None, 0), stream_of(dev)), "allset_fused_linear_fwd_ex")
    return y, stats


def fused_linear_bwd_all_blocked(gy: Tensor, gy_cb: int, mask: Optional[Tensor], p_out: float, weight: Tensor, x: Tensor, x_cb: int,
                                 stats: Optional[Tensor], gamma: Optional[Tensor], beta: Optional[Tensor], relu_in: bool, p_in: float,
                                 seed_in: int, seed_base: Optional[Tensor], want_bias: bool = True
                                 ) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor], Tensor, Optional[Tensor]]:
    """:func:`fused_linear_bwd_all` with ``gy`` blocked by ``gy_cb`` and ``x`` (hence ``gx``) by ``x_cb`` (0 = plain)."""
    dev = require_device(gy, mask, weight, x, stats, gamma, beta)
    _check_f32(gy, weight, x, stats, gamma, beta)
    O, I = weight.shape
    gy = gy.contiguous() if gy_cb else _rowmajor(gy)
    x = x.contiguous() if x_cb else _rowmajor(x)
    n = x.shape[0] // (I // x_cb) if x_cb else x.shape[0]
    weight = weight.contiguous()
    lib = _lib.load()
    ns = c_int64(0)
    check(lib.allset_fused_linear_bwd_all_slices_for(n, O, I, 0, byref(ns)), "allset_fused_linear_bwd_all_slices_for")
    P = ns.value
    gx = torch.empty_like(x)
    M = O * I + O + (2 * I if stats is not None else 0)
    M = (M + 3) // 4 * 4
    part = torch.empty((P, M), dtype=torch.float32, device=dev)
    flat = part.view(-1)
    part_w, part_b = flat, flat[O * I:]
    part_ln = flat[O * I + O:] if stats is not None else None
    with on_device(dev), _timed("fused_linear_bwd_all", dev, n * (O + 2 * I) * 4):
        check(lib.allset_fused_linear_bwd_all_ex(
            ptr(gy), gy_cb if gy_cb else _ld(gy), gy_cb, ptr(mask), p_out, ptr(weight), ptr(x), x_cb if x_cb else _ld(x), x_cb, ptr(stats),
            ptr(gamma.contiguous() if gamma is not None else None), ptr(beta.contiguous() if beta is not None else None),
            0, int(relu_in), p_in, seed_in, ptr(gx), x_cb if x_cb else max(I, 1), x_cb, ptr(part_ln), ptr(part_w),
            ptr(part_b if want_bias else None), P, n, O, I, ptr(seed_base), ptr(None), 0, M,
            _arith_for(1, I, O, stats is not None, 0), stream_of(dev)), "allset_fused_linear_bwd_all_ex")
    red = reduce_partials(part)
    gw = red[:O * I].view(O, I)
    gb = red[O * I:O * I + O] if want_bias else None
    if stats is None:
        return gx, None, None, gw, gb
    return gx, red[O * I + O:O * I + O + I], red[O * I + O + I:O * I + O + 2 * I], gw, gb


def wgrad_fused(gy: Tensor, y: Optional[Tensor], p_out: float, x: Tensor, stats: Optional[Tensor],
                gamma: Optional[Tensor], beta: Optional[Tensor], relu_in: bool, p_in: float, seed_in: int,
                want_bias: bool = True, seed_base: Optional[Tensor] = None, mask: Optional[Tensor] = None, defer_to=None
                ) -> Tuple[Tensor, Optional[Tensor]]:
    """Weight/bias gradient of the fused Linear with both operands recomputed on the fly (csrc/dense.hip).
    ``mask`` (from ``fused_linear_fwd(mask_out=...)``) replaces ``y`` as the source of the epilogue mask."""
    dev = require_device(gy, y, x, stats, gamma, beta, mask)
    gy, x = _rowmajor(gy), _rowmajor(x)
    if y is not None:
        y = _rowmajor(y)
    n, O = gy.shape
    I = x.shape[1]
    lib = _lib.load()
    ns = c_int64(0)
    M = O * I + (O if want_bias else 0)
    if stats is not None and y is None and wide_f16() and lib.allset_wgrad_f16x3_supported(O, I):
        # widths 256 / 512 behind a LayerNorm, auto / fp16x3 arithmetic: two fp16 planes per operand, 256 x 128 tiles (csrc/wgrad_f16.hip)
        check(lib.allset_wgrad_f16x3_slices(n, O, I, byref(ns)), "allset_wgrad_f16x3_slices")
        part = torch.empty((ns.value, M), dtype=torch.float32, device=dev)
        with on_device(dev), _timed("wgrad_fused", dev, n * (O + I) * 4):
            check(lib.allset_wgrad_f16x3(ptr(gy), _ld(gy), ptr(mask), p_out, ptr(x), _ld(x), ptr(stats), ptr(gamma.contiguous()),
                                         ptr(beta.contiguous()), int(relu_in), p_in, seed_in, ptr(part), M, int(want_bias), ns.value,
                                         n, O, I, ptr(seed_base), stream_of(dev)), "allset_wgrad_f16x3")
        gw_, gb_ = _defer_or_reduce(part, [(defer_to[0] if defer_to else None, 0, (O, I)),
                                           ((defer_to[1] if defer_to else None) if want_bias else None, O * I, (O,))], defer_to is not None)
        return gw_, (gb_ if want_bias else None)
    check(lib.allset_wgrad_slices(n, O, I, byref(ns)), "allset_wgrad_slices")
    # one partial buffer [slices, gW | gb] and one reduction launch (O and I are multiples of 4, checked by the kernel)
    part = torch.empty((ns.value, M), dtype=torch.float32, device=dev)
    with on_device(dev), _timed("wgrad_fused", dev, n * (O * (2 if (y is not None and mask is None) else 1) + I) * 4):
        check(lib.allset_wgrad_fused_ex(ptr(gy), _ld(gy), ptr(y), _ld(y) if y is not None else 0, p_out, ptr(x), _ld(x),
                                        ptr(stats), ptr(gamma.contiguous() if gamma is not None else None),
                                        ptr(beta.contiguous() if beta is not None else None), int(relu_in), p_in, seed_in,
                                        ptr(part), M, int(want_bias), ns.value, n, O, I, ptr(seed_base), ptr(mask),
                                        stream_of(dev)), "allset_wgrad_fused_ex")
    gw_, gb_ = _defer_or_reduce(part, [(defer_to[0] if defer_to else None, 0, (O, I)),
                                       ((defer_to[1] if defer_to else None) if want_bias else None, O * I, (O,))], defer_to is not None)
    return gw_, (gb_ if want_bias else None)


def fused_linear_bwd(gy: Tensor, y: Optional[Tensor], p_out: float, weight: Tensor, x: Tensor, stats: Optional[Tensor],
                     gamma: Optional[Tensor], relu_in: bool, p_in: float, seed_in: int,
                     seed_base: Optional[Tensor] = None, mask: Optional[Tensor] = None, acc_in: Optional[Tensor] = None,
                     aux_g: Optional[Tensor] = None, aux_w: Optional[Tensor] = None, want_gx: bool = True
                     ) -> Tuple[Optional[Tensor], Optional[Tensor], Optional[Tensor]]:
    """(gx, dgamma, dbeta) of the fused Linear w.r.t. its input and LayerNorm parameters (csrc/fused_mlp.hip).
    ``mask`` replaces ``y`` as the source of the relu/dropout epilogue mask.  ``acc_in`` [n, I]: another gradient
    branch of the same input, summed in the kernel (``gx = acc_in + ...``; the buffer is reused for the result)."""
    dev = require_device(gy, y, weight, x, stats, gamma, mask, acc_in)
    gy, x = _rowmajor(gy), _rowmajor(x)
    if y is not None:
        y = _rowmajor(y)
    weight = weight.contiguous()
    n, O = gy.shape
    I = x.shape[1]
    lib = _lib.load()
    if acc_in is not None:
        acc_in = _rowmajor(acc_in)
        gx = acc_in if acc_in.is_contiguous() else torch.empty((n, I), dtype=torch.float32, device=dev)
    elif want_gx or stats is None or not x6_active():
        gx = torch.empty((n, I), dtype=torch.float32, device=dev)
    else:
        gx = None                                  # LayerNorm partials only (the input needs no gradient)
    partials, npart = None, c_int64(0)
    if stats is not None:
        check(lib.allset_fused_linear_bwd_partials(n, byref(npart)), "allset_fused_linear_bwd_partials")
        partials = torch.empty((npart.value, 2, I), dtype=torch.float32, device=dev)
    with on_device(dev), _timed("fused_linear_bwd", dev, n * (O * (2 if (y is not None and mask is None) else 1) + 2 * I) * 4):
        check(lib.allset_fused_linear_bwd(ptr(gy), _ld(gy), ptr(y), _ld(y) if y is not None else 0, p_out, ptr(weight),
                                          ptr(x), _ld(x), ptr(stats), ptr(gamma.contiguous() if gamma is not None else None),
                                          int(relu_in), p_in, seed_in, ptr(gx), max(I, 1), ptr(partials), npart.value,
                                          n, O, I, ptr(seed_base), ptr(mask), ptr(acc_in), _ld(acc_in) if acc_in is not None else 0,
                                          ptr(aux_g), ptr(aux_w), stream_of(dev)), "allset_fused_linear_bwd")
    if partials is None:
        return gx, None, None
    red = reduce_partials(partials)
    return gx, red[0], red[1]


def fused_linear_bwd_all_supported(O: int, I: int, has_ln: bool, drop_in: bool, relu_in: bool, has_mask: bool,
                                   has_acc: bool = False) -> bool:
    """The one-pass backward kernel (csrc/fused_bwd.hip) is built for this Linear."""
    if os.environ.get("ALLSET_BWD_SPLIT", "0") == "1":             # comparison arm: the two-kernel pair of round 1
        return False
    return bool(_lib.load().allset_fused_linear_bwd_all_supported(O, I, int(has_ln), int(drop_in), int(relu_in), int(has_mask),
                                                                  int(has_acc)))


_ONE_PASS_PREFERRED = {}


def one_pass_preferred(O: int, I: int) -> bool:
    """True when ``allset_fused_linear_bwd_all`` launches one of the two-waves-per-SIMD kernels for these widths (round 3:
    O = I = 128; one partial slice per workgroup instead of one per wave).  Those beat the backward-data + weight-gradient pair
    also for a Linear WITHOUT a LayerNorm prologue (0.35-0.4 ms against 0.21 + 0.32); the one-wave kernel did not."""
    key = (int(O), int(I))                                          # the dispatch is a pure function of the widths (ABI 9)
    hit = _ONE_PASS_PREFERRED.get(key)
    if hit is None:
        lib = _lib.load()
        a, b = c_int64(0), c_int64(0)
        check(lib.allset_fused_linear_bwd_all_slices_for(1 << 20, int(O), int(I), 0, byref(a)), "allset_fused_linear_bwd_all_slices_for")
        check(lib.allset_fused_linear_bwd_all_slices(1 << 20, byref(b)), "allset_fused_linear_bwd_all_slices")
        hit = _ONE_PASS_PREFERRED[key] = bool(lib.allset_fused_linear_bwd_all_supported(int(O), int(I), 0, 0, 0, 0, 0)) and a.value < b.value
    return hit


def fused_linear_bwd_all(gy: Tensor, mask: Optional[Tensor], p_out: float, weight: Tensor, x: Tensor, stats: Optional[Tensor],
                         gamma: Optional[Tensor], beta: Optional[Tensor], relu_in: bool, p_in: float, seed_in: int,
                         seed_base: Optional[Tensor] = None, acc_in: Optional[Tensor] = None, want_bias: bool = True,
                         norm_mode: int = 0, defer_to=None) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor], Tensor, Optional[Tensor]]:
    """(gx, dgamma, dbeta, gW, gb) of the fused Linear from ONE pass over gy and x (include/allset_hip.h
    allset_fused_linear_bwd_all).  ``defer_to`` = the (gamma, beta, weight, bias) PARAMETERS: inside ``deferred_param_grads()`` their
    gradients are queued for the batched reduction and come back as None."""
    dev = require_device(gy, mask, weight, x, stats, gamma, beta, acc_in)
    _check_f32(gy, weight, x, stats, gamma, beta, acc_in)
    gy, x = _rowmajor(gy), _rowmajor(x)
    weight = weight.contiguous()
    n, O = gy.shape
    I = x.shape[1]
    lib = _lib.load()
    ns = c_int64(0)
    check(lib.allset_fused_linear_bwd_all_slices_for(n, O, I, int(acc_in is not None), byref(ns)), "allset_fused_linear_bwd_all_slices_for")
    P = ns.value
    if acc_in is not None:
        acc_in = _rowmajor(acc_in)
        gx = acc_in if acc_in.is_contiguous() else torch.empty((n, I), dtype=torch.float32, device=dev)
    else:
        gx = torch.empty((n, I), dtype=torch.float32, device=dev)
    # ONE partial buffer [P, O*I + O + 2*I]: gW, gb and the LayerNorm partials of a slice side by side, one reduction launch
    M = O * I + O + (2 * I if stats is not None else 0)
    M = (M + 3) // 4 * 4
    part = torch.empty((P, M), dtype=torch.float32, device=dev)
    flat = part.view(-1)
    part_w, part_b = flat, flat[O * I:]
    part_ln = flat[O * I + O:] if stats is not None else None
    if norm_mode:
        if acc_in is not None or stats is None:
            raise _lib.AllSetHipError("fused_linear_bwd_all: the column-affine prologue takes stats / gamma / beta and no acc_in")
        with on_device(dev), _timed("fused_linear_bwd_all", dev, n * (O + 2 * I) * 4):
            check(lib.allset_fused_linear_bwd_all_ex(
                ptr(gy), _ld(gy), 0, ptr(mask), p_out, ptr(weight), ptr(x), _ld(x), 0, ptr(stats), ptr(gamma.contiguous()),
                ptr(beta.contiguous()), int(norm_mode), int(relu_in), p_in, seed_in, ptr(gx), max(I, 1), 0, ptr(part_ln), ptr(part_w),
                ptr(part_b if want_bias else None), P, n, O, I, ptr(seed_base), ptr(None), 0, M,
                _arith_for(1, I, O, True, norm_mode), stream_of(dev)), "allset_fused_linear_bwd_all_ex")
        red = reduce_partials(part)
        return (gx, red[O * I + O:O * I + O + I], red[O * I + O + I:O * I + O + 2 * I], red[:O * I].view(O, I),
                red[O * I:O * I + O] if want_bias else None)
    with on_device(dev), _timed("fused_linear_bwd_all", dev, n * (O + 2 * I) * 4):
        check(lib.allset_fused_linear_bwd_all_ex(
            ptr(gy), _ld(gy), 0, ptr(mask), p_out, ptr(weight), ptr(x), _ld(x), 0, ptr(stats),
            ptr(gamma.contiguous() if gamma is not None else None), ptr(beta.contiguous() if beta is not None else None),
            0, int(relu_in), p_in, seed_in, ptr(gx), max(I, 1), 0, ptr(part_ln), ptr(part_w), ptr(part_b if want_bias else None), P, n, O, I,
            ptr(seed_base), ptr(acc_in), _ld(acc_in) if acc_in is not None else 0, M,
            _arith_for(1, I, O, stats is not None, 0), stream_of(dev)), "allset_fused_linear_bwd_all_ex")
    if defer_to is not None and _deferrable(part, *defer_to):
        g_p, b_p, w_p, bias_p = defer_to
        _defer(part, [(w_p, 0, (O, I)), (bias_p if want_bias else None, O * I, (O,)),
                      (g_p if stats is not None else None, O * I + O, (I,)), (b_p if stats is not None else None, O * I + O + I, (I,))])
        return gx, None, None, None, None
    red = reduce_partials(part)
    gw = red[:O * I].view(O, I)
    gb = red[O * I:O * I + O] if want_bias else None
    if stats is None:
        return gx, None, None, gw, gb
    return gx, red[O * I + O:O * I + O + I], red[O * I + O + I:O * I + O + 2 * I], gw, gb


def fused_linear_bwd_all_aux_supported(O: int, I: int) -> bool:
    return bool(_lib.load().allset_fused_linear_bwd_all_aux_supported(int(O), int(I)))


def fused_linear_bwd_all_aux(gy: Tensor, weight: Tensor, x: Tensor, aux_g: Tensor, aux_w: Tensor, defer_to=None
                             ) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """(gx, gW, gb, gaux_w [4, I], gaux_b [4]) of the plain Linear with four auxiliary output columns, from ONE pass over gy and x
    (include/allset_hip.h allset_fused_linear_bwd_all_aux)."""
    dev = require_device(gy, weight, x, aux_g, aux_w)
    _check_f32(gy, weight, x, aux_g, aux_w)
    gy, x = _rowmajor(gy), _rowmajor(x)
    weight, aux_g, aux_w = weight.contiguous(), aux_g.contiguous(), aux_w.contiguous()
    n, O = gy.shape
    I = x.shape[1]
    lib = _lib.load()
    ns = c_int64(0)
    check(lib.allset_fused_linear_bwd_all_slices_for(n, O, I, 0, byref(ns)), "allset_fused_linear_bwd_all_slices_for")
    P = ns.value
    gx = torch.empty((n, I), dtype=torch.float32, device=dev)
    M = (O * I + O + 4 * I + 4 + 3) // 4 * 4
    part = torch.empty((P, M), dtype=torch.float32, device=dev)
    with on_device(dev), _timed("fused_linear_bwd_all", dev, n * (O + 2 * I + 4) * 4):
        check(lib.allset_fused_linear_bwd_all_aux(ptr(gy), _ld(gy), ptr(weight), ptr(x), _ld(x), ptr(aux_g), ptr(aux_w), ptr(gx),
                                                  max(I, 1), ptr(part), M, P, n, O, I, stream_of(dev)),
              "allset_fused_linear_bwd_all_aux")
    o = O * I
    if defer_to is not None and (o + O) % 4 == 0 and defer_to[1] is not None and _deferrable(part, *defer_to, M=o + O):
        # (weight, bias) PARAMETERS: queued; the four auxiliary rows (their weight is a folded tensor, not a parameter) reduce at once
        _defer(part, [(defer_to[0], 0, (O, I)), (defer_to[1], o, (O,))], M=o + O)
        redx = reduce_partials_slice(part, o + O, 4 * I + 4, torch.float32)
        return gx, None, None, redx[:4 * I].view(4, I), redx[4 * I:4 * I + 4]
    red = reduce_partials(part)
    return gx, red[:o].view(O, I), red[o:o + O], red[o + O:o + O + 4 * I].view(4, I), red[o + O + 4 * I:o + O + 4 * I + 4]


def wgrad_supported(ga: Tensor, u: Tensor) -> bool:
    return (ga.is_cuda and ga.dtype == u.dtype and ga.dtype in (torch.float32, torch.bfloat16) and ga.shape[1] % 4 == 0
            and u.shape[1] % 4 == 0)


# ---- autograd functions ------------------------------------------------------------------------------

def wgrad_padded(ga: Tensor, u: Tensor, want_bias: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
    """:func:`wgrad` for widths that are not multiples of 4 (a 10-class classifier head, 1433 raw input features): the operand
    with the odd width is copied once into a zero-padded buffer and the result sliced.  One more pass over that operand -- against
    the library's choice for a [O x n] x [n x I] product with a tiny output, which streams n = 1M rows at 1.4 ms for a
    [10 x 64] result and needs a separate reduction for the bias gradient."""
    n, O = ga.shape
    I = u.shape[1]
    Op, Ip = (O + 3) // 4 * 4, (I + 3) // 4 * 4
    if Op != O:
        gp = ga.new_zeros((n, Op))
        gp[:, :O] = ga
        ga = gp
    if Ip != I:
        up = u.new_zeros((n, Ip))
        up[:, :I] = u
        u = up
    gw, gb = wgrad(ga, u, want_bias)
    return gw[:O, :I], (gb[:O] if gb is not None else None)


def dropout_scale(shape, p: float, seed: int, device) -> Tensor:
    """The hash dropout's per-element factor (0 or 1 / (1 - p)) for a row-major tensor of ``shape`` and host seed ``seed``: the
    relu-dropout kernel on ones -- the same hash and element indexing every fused site uses."""
    ones = torch.ones(shape, dtype=torch.float32, device=device)
    y = torch.empty_like(ones)
    with on_device(device):
        check(_lib.load().allset_relu_dropout_fwd(ptr(ones), float(p), int(seed), ptr(y), ones.numel(), ptr(_seed_base()), stream_of(device)),
              "allset_relu_dropout_fwd")
    return y


class _HashDropout(torch.autograd.Function):
    """``x * keep / (1 - p)`` with the library's counter-hash mask (a dropout site that is not the prologue / epilogue of a fused
    kernel -- behind a torch BatchNorm, say -- still draws its mask the way every other site does: reproducible from the seed)."""

    @staticmethod
    def forward(ctx, x, p):
        scale = dropout_scale(x.shape, p, _draw_seed(), x.device)
        ctx.save_for_backward(scale)
        return x * scale

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return g * scale, None


def hash_dropout(x: Tensor, p: float, training: bool) -> Tensor:
    """``F.dropout`` for device fp32 tensors through the hash mask; anything else through torch's generator."""
    if not training or p <= 0.0:
        return x
    if x.is_cuda and x.dtype == torch.float32:
        return _HashDropout.apply(x.contiguous(), float(p))
    return torch.nn.functional.dropout(x, p=p, training=True)


class _LayerNormFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu_in, p):
        seed = _draw_seed() if p > 0.0 else 0
        base = _seed_base() if p > 0.0 else None
        y, stats = ln_fwd(x, gamma, beta, eps, relu_in, p, seed, base)
        ctx.save_for_backward(x, stats, gamma)
        ctx.cfg = (bool(relu_in), float(p), seed, base)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, stats, gamma = ctx.saved_tensors
        relu_in, p, seed, base = ctx.cfg
        gx, dg, db = ln_bwd(gy.contiguous(), x, stats, gamma, relu_in, p, seed, base, want_gx=ctx.needs_input_grad[0])
        return gx, dg, db, None, None, None


class _ReluDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        dev = require_device(x)
        _check_f32(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        seed = _draw_seed() if p > 0.0 else 0
        n = x.numel()
        with on_device(dev), _timed("relu_dropout_fwd", dev, 2 * n * 4):
            check(_lib.load().allset_relu_dropout_fwd(ptr(x), p, seed, ptr(y), n, ptr(_seed_base() if p > 0.0 else None),
                                                      stream_of(dev)), "allset_relu_dropout_fwd")
        ctx.save_for_backward(y)
        ctx.p = float(p)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        dev = y.device
        gy = gy.contiguous()
        gx = torch.empty_like(y)
        n = y.numel()
        with on_device(dev), _timed("relu_dropout_bwd", dev, 3 * n * 4):
            check(_lib.load().allset_relu_dropout_bwd(ptr(gy), ptr(y), ctx.p, ptr(gx), n, stream_of(dev)),
                  "allset_relu_dropout_bwd")
        return gx, None


class _Linear(torch.autograd.Function):
    """y = x W^T + b.  Forward and grad-input are library GEMMs (hipBLASLt); grad-weight / grad-bias are the
    split-K MFMA kernel (hipBLASLt's choice for the [O x n] x [n x I] shape is ~5x slower at n = 1M)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None
        need_w, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if need_w and linear_narrow_supported(gy, x, weight):          # a classifier head: one kernel for all three gradients
            defer = ctx.params if (not ctx.has_bias or need_b) else None
            gx, gw, gb = linear_narrow_bwd(gy, x, weight, ctx.needs_input_grad[0], defer_to=defer)
            return gx, gw, gb if need_b else None
        if ctx.needs_input_grad[0]:
            gx = gy @ weight
        if need_w or need_b:
            if wgrad_supported(gy, x):
                gw, gb = wgrad(gy, x, want_bias=need_b)
            elif (gy.is_cuda and gy.dtype == x.dtype and gy.dtype in (torch.float32, torch.bfloat16) and x.dim() == 2
                  and gy.shape[0] >= 8192):
                # odd widths with many rows: pad to a multiple of 4, same kernel (with few rows the library's GEMM is fine:
                # measured on the Citeseer-shaped step, 3327 x 3703 features, the padded path is 0.03 ms slower)
                gw, gb = wgrad_padded(gy, x, want_bias=need_b)
            else:
                gw = gy.t() @ x
                gb = gy.sum(dim=0) if need_b else None
        return gx, gw, gb


class _FusedNormLinear(torch.autograd.Function):
    """``y = epi( pro(x) @ W^T + b )`` with ``pro = [relu] -> [LayerNorm] -> [dropout p_in]`` and
    ``epi = [relu] -> [dropout p_out]`` -- one kernel forward (fused_mlp.hip / fused_fwd2.hip), one kernel backward
    (the one-pass ``fused_linear_bwd_all``; the two-kernel pair where that is not built); nothing but x, the row statistics and
    the 1-bit epilogue mask is kept.  ``in_cb`` / ``out_cb`` > 0: x / y (and their gradients) are COLUMN-BLOCKED 2-D tensors
    [(C / cb) * n, cb] -- the exchange layout of the column-sharded layer (dist.py)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, weight, bias, eps, relu_in, p_in, relu_out, p_out, in_cb=0, out_cb=0):
        seed_in = _draw_seed() if p_in > 0.0 else 0
        seed_out = _draw_seed() if p_out > 0.0 else 0
        base = _seed_base() if (p_in > 0.0 or p_out > 0.0) else None
        keep_y = relu_out or p_out > 0.0
        # the backward needs only the sign pattern of y: a 1-bit mask written by the forward kernel (1/32 of y's bytes)
        # (no mask in inference: nothing will run backward)
        words = activation_mask_words(x.shape[0], weight.shape[0]) if (keep_y and any(ctx.needs_input_grad)) else 0
        mask = torch.empty(words, dtype=torch.int32, device=x.device) if words > 0 else None
        if in_cb or out_cb:
            n_rows = x.shape[0] // (weight.shape[1] // in_cb) if in_cb else x.shape[0]
            words = activation_mask_words(n_rows, weight.shape[0]) if (keep_y and any(ctx.needs_input_grad)) else 0
            mask = torch.empty(words, dtype=torch.int32, device=x.device) if words > 0 else None
            y, stats = fused_linear_fwd_blocked(x, in_cb, weight, bias, gamma, beta, eps, relu_in, p_in, seed_in, relu_out, p_out,
                                                seed_out, base, mask, out_cb)
        else:
            y, stats = fused_linear_fwd(x, weight, bias, gamma, beta, eps, relu_in, p_in, seed_in, relu_out, p_out, seed_out,
                                        base, mask)
        ctx.save_for_backward(x, stats, gamma, beta, weight, y if (keep_y and mask is None) else None, mask)
        ctx.cfg = (bool(relu_in), float(p_in), seed_in, float(p_out), bias is not None, base)
        ctx.layout = (int(in_cb), int(out_cb))
        ctx.params = (gamma, beta, weight, bias)     # (the objects themselves: deferred_param_grads assigns their .grad)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, stats, gamma, beta, weight, y, mask = ctx.saved_tensors
        relu_in, p_in, seed_in, p_out, has_bias, base = ctx.cfg
        if x.shape[0] == 0:                     # no rows (a rank without hyperedges): every gradient is zero; empty tensors carry
            z = lambda t, need: torch.zeros_like(t) if (t is not None and need) else None      # no row statistics to hand the kernels
            return (z(x, ctx.needs_input_grad[0]), z(gamma, ctx.needs_input_grad[1]), z(beta, ctx.needs_input_grad[2]),
                    z(weight, ctx.needs_input_grad[3]),
                    weight.new_zeros(weight.shape[0]) if (has_bias and ctx.needs_input_grad[4]) else None,
                    None, None, None, None, None, None, None)
        gy = gy.contiguous()
        gx = dg = db = gw = gb = None
        need_b = has_bias and ctx.needs_input_grad[4]
        in_cb, out_cb = ctx.layout
        if in_cb or out_cb:                       # blocked operands: the one-pass kernel reads / writes them in place
            gx, dg, db, gw, gb = fused_linear_bwd_all_blocked(gy, out_cb, mask, p_out, weight, x, in_cb, stats, gamma, beta, relu_in,
                                                              p_in, seed_in, base, want_bias=need_b)
            return gx, dg, db, gw, gb, None, None, None, None, None, None, None
        # one-pass kernel where a LayerNorm prologue makes both halves of the pair recompute the same operand (0.49-0.55 ms
        # against 0.62-0.63); without one the plain weight-gradient kernel is cheap and the pair wins (0.48 vs 0.58 ms)
        if (ctx.needs_input_grad[0] and ctx.needs_input_grad[3] and y is None and x.shape[0] > 0 and
                (gamma is not None or one_pass_preferred(weight.shape[0], weight.shape[1])) and
                fused_linear_bwd_all_supported(weight.shape[0], weight.shape[1], gamma is not None, p_in > 0.0, relu_in,
                                               mask is not None)):
            # everything from one read of gy and x
            need = ctx.needs_input_grad
            defer = ctx.params if (_Deferred.active and (gamma is None or (need[1] and need[2])) and (not has_bias or need[4])) else None
            gx, dg, db, gw, gb = fused_linear_bwd_all(gy, mask, p_out, weight, x, stats, gamma, beta, relu_in, p_in, seed_in,
                                                      base, want_bias=need_b, defer_to=defer)
            return gx, dg, db, gw, gb, None, None, None, None, None, None, None
        if ctx.needs_input_grad[3] or need_b:
            gw, gb = wgrad_fused(gy, y, p_out, x, stats, gamma, beta, relu_in, p_in, seed_in, want_bias=need_b,
                                 seed_base=base, mask=mask)
        if ctx.needs_input_grad[0] or (gamma is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])):
            gx, dg, db = fused_linear_bwd(gy, y, p_out, weight, x, stats, gamma, relu_in, p_in, seed_in, base, mask,
                                          want_gx=ctx.needs_input_grad[0])
        return gx, dg, db, gw, gb, None, None, None, None, None, None, None


class _WideNormLinear(torch.autograd.Function):
    """The same layer as :class:`_FusedNormLinear` for widths beyond the LDS-resident-weight kernels (256, 512):
    forward = row statistics + one tiled bf16x6 GEMM with the prologue applied as the A operand is staged and the epilogue
    on the output tile (csrc/wide_mlp.hip); backward = the same GEMM against W^T with the epilogue mask applied to the
    incoming gradient, the LayerNorm-backward kernel, and the split-K weight gradient with both operands recomputed."""

    @staticmethod
    def forward(ctx, x, gamma, beta, weight, bias, eps, relu_in, p_in, relu_out, p_out, stats_in=None, emit=None):
        # stats_in: this input's row statistics as the PREVIOUS wide Linear's epilogue wrote them (emit = (eps, relu) of the consumer's
        # prologue asks this one to write its output's): one allset_row_stats pass over [n, 256] less per pair of Linears
        seed_in = _draw_seed() if p_in > 0.0 else 0
        seed_out = _draw_seed() if p_out > 0.0 else 0
        base = _seed_base() if (p_in > 0.0 or p_out > 0.0) else None
        stats = (stats_in if stats_in is not None else row_stats(x, relu_in, eps)) if gamma is not None else None
        stats_y = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device) if emit is not None else None
        keep_y = relu_out or p_out > 0.0
        # the backward's "y > 0" test from a 1-bit mask the forward's epilogue writes (round 5: the backward kernels used to re-read
        # the fp32 output twice -- 2 GB per Linear at [1M, 256]); widths that are not multiples of 64 keep y
        words = activation_mask_words(x.shape[0], weight.shape[0]) if (keep_y and x.shape[0] > 0) else 0
        mask = torch.empty(words, dtype=torch.int32, device=x.device) if words > 0 else None
        y = gemm_x6(x, gemm_x6_planes(weight, False, f16=wide_f16(gamma is not None)), weight.shape[0], bias, relu_in=relu_in, stats=stats, gamma=gamma,
                    beta=beta, p_in=p_in, seed_in=seed_in, relu_out=relu_out, p_out=p_out, seed_out=seed_out, seed_base=base,
                    mask_out=mask, stats_out=stats_y, stats_eps=emit[0] if emit is not None else 1e-5,
                    stats_relu=bool(emit[1]) if emit is not None else False)
        ctx.save_for_backward(x, stats, gamma, beta, weight, y if (keep_y and mask is None) else None, mask)
        ctx.cfg = (bool(relu_in), float(p_in), seed_in, float(p_out), bias is not None, base)
        ctx.params = (gamma, beta, weight, bias)     # (the objects themselves: deferred_param_grads assigns their .grad)
        if emit is None:
            return y
        ctx.mark_non_differentiable(stats_y)
        return y, stats_y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _g_stats=None):
        x, stats, gamma, beta, weight, y, mask = ctx.saved_tensors
        relu_in, p_in, seed_in, p_out, has_bias, base = ctx.cfg
        if x.shape[0] == 0:                     # no rows: every gradient is zero (empty tensors carry no row statistics)
            z = lambda t, need: torch.zeros_like(t) if (t is not None and need) else None
            return (z(x, ctx.needs_input_grad[0]), z(gamma, ctx.needs_input_grad[1]), z(beta, ctx.needs_input_grad[2]),
                    z(weight, ctx.needs_input_grad[3]),
                    weight.new_zeros(weight.shape[0]) if (has_bias and ctx.needs_input_grad[4]) else None,
                    None, None, None, None, None, None, None)
        gy = gy.contiguous()
        gx = dg = db = gw = gb = None
        need_b = has_bias and ctx.needs_input_grad[4]
        need = ctx.needs_input_grad
        p_g, p_bt, p_w, p_b = ctx.params
        dfr = _Deferred.active
        if ctx.needs_input_grad[3] or need_b:
            gw, gb = wgrad_fused(gy, y, p_out, x, stats, gamma, beta, relu_in, p_in, seed_in, want_bias=need_b, seed_base=base, mask=mask,
                                 defer_to=(p_w, p_b) if (dfr and need[3] and (not has_bias or need[4])) else None)
        need_x = ctx.needs_input_grad[0]
        if need_x or (gamma is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])):
            # gradient of the Linear's input u = dropout(LN(relu(x))): (gy * epilogue mask) @ W
            if gamma is not None and weight.shape[1] <= 256:
                # one kernel: the Linear's input gradient never leaves the chip, the LayerNorm backward is the GEMM's epilogue
                gx, dg, db = gemm_x6_lnb(gy, gemm_x6_planes(weight, True), x, stats, gamma, relu_in, p_in, seed_in, mask_y=y,
                                         p_mask=p_out, seed_base=base, mask_bits=mask,
                                         defer_to=(p_g, p_bt) if (dfr and need[1] and need[2]) else None)
                return gx, dg, db, gw, gb, None, None, None, None, None, None, None
            # (behind a bare relu -- p_in == 0 there, see wide_linear_supported -- the mask by the sign of x is the GEMM's epilogue)
            planes_t = gemm_x6_planes(weight, True)
            sgn = x if (gamma is None and relu_in and planes_t.f16 and x.stride(-1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
                        and _lib.load().allset_gemm_wide_sgn_supported(_lib.ARITH_FP16X3, weight.shape[1], weight.shape[0])) else None
            gu = gemm_x6(gy, planes_t, weight.shape[1], None, mask_y=y, p_mask=p_out, mask_bits=mask, sgn_x=sgn)
            if gamma is not None:
                gx, dg, db = ln_bwd(gu, x, stats, gamma, relu_in, p_in, seed_in, base, want_gx=need_x,
                                    defer_to=(p_g, p_bt) if (dfr and need[1] and need[2]) else None)
            elif relu_in and sgn is None:       # (rows of x that are not 16-byte aligned: the elementwise pass)
                gx = torch.empty_like(x)
                with torch.cuda.device(x.device), _timed("relu_dropout_bwd", x.device, 3 * x.numel() * 4):
                    check(_lib.load().allset_relu_dropout_bwd(ptr(gu), ptr(x.contiguous()), 0.0, ptr(gx), x.numel(),
                                                              stream_of(x.device)), "allset_relu_dropout_bwd")
            else:
                gx = gu
        return gx, dg, db, gw, gb, None, None, None, None, None, None, None


def wide_linear_supported(K: int, N: int, has_ln: bool, relu_in: bool = False, p_in: float = 0.0) -> bool:
    """The tiled bf16x6 GEMM path takes this layer: widths the LDS-resident kernels do not cover, K a multiple of 32,
    LayerNorm rows of at most 512, and no input dropout without a LayerNorm (its mask is regenerated by the LayerNorm
    backward kernel)."""
    if fused_linear_supported(K, N) or not x6_active() or not gemm_x6_supported(N, K) or not gemm_x6_supported(K, N):
        return False
    if has_ln:
        return K <= 512 and K % 4 == 0
    return p_in == 0.0


def wide_stats_chain_supported(K: int, N: int, has_ln: bool, relu_in: bool = False, p_in: float = 0.0) -> bool:
    """A [N, K] Linear on the tiled wide path whose epilogue can write the next Linear's row statistics (N == 256: a wave holds a whole
    output row) -- :func:`fused_norm_linear`'s ``emit_stats``."""
    return N == 256 and wide_linear_supported(K, N, has_ln, relu_in, p_in)


def fused_norm_linear(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], weight: Tensor, bias: Optional[Tensor],
                      eps: float = 1e-5, relu_in: bool = False, p_in: float = 0.0, relu_out: bool = False,
                      p_out: float = 0.0, in_cb: int = 0, out_cb: int = 0, stats_in: Optional[Tensor] = None,
                      emit_stats: Optional[Tuple[float, bool]] = None):
    """``emit_stats = (eps, relu)`` (only where :func:`wide_stats_chain_supported`): returns ``(y, stats_y)``, the row statistics the
    next Linear's LayerNorm prologue would otherwise compute with a pass of its own; ``stats_in`` hands them to that Linear."""
    if p_out > 0.0 and not relu_out:
        # the backward recovers the epilogue mask from the sign of y, which is only right behind a relu (MLP._post is
        # always relu -> dropout, reference layers.py:575-577)
        raise _lib.AllSetHipError("fused_norm_linear: an output dropout needs relu_out=True")
    N, K = weight.shape
    if in_cb or out_cb:
        if not blocked_linear_supported(K, N):
            raise _lib.AllSetHipError(f"fused_norm_linear: column-blocked operands are not built for a [{N}, {K}] weight")
        return _FusedNormLinear.apply(x, gamma, beta, weight, bias, float(eps), bool(relu_in), float(p_in), bool(relu_out),
                                      float(p_out), int(in_cb), int(out_cb))
    if fused_linear_supported(K, N):
        if stats_in is not None or emit_stats is not None:
            raise _lib.AllSetHipError("fused_norm_linear: stats_in / emit_stats belong to the tiled wide path")
        return _FusedNormLinear.apply(x, gamma, beta, weight, bias, float(eps), bool(relu_in), float(p_in), bool(relu_out), float(p_out))
    if emit_stats is not None and N != 256:
        raise _lib.AllSetHipError("fused_norm_linear: emit_stats needs 256 output features")
    if stats_in is None and emit_stats is None:
        return _WideNormLinear.apply(x, gamma, beta, weight, bias, float(eps), bool(relu_in), float(p_in), bool(relu_out), float(p_out))
    return _WideNormLinear.apply(x, gamma, beta, weight, bias, float(eps), bool(relu_in), float(p_in), bool(relu_out), float(p_out),
                                 stats_in, (float(emit_stats[0]), bool(emit_stats[1])) if emit_stats is not None else None)


def ln_res_supported(d: int, dtype: torch.dtype = torch.float32) -> bool:
    if dtype == torch.bfloat16:
        return bool(_lib.load().allset_ln_bf16_supported(d))
    return dtype == torch.float32 and bool(_lib.load().allset_ln_res_supported(d))


def ln_res_fwd(x: Tensor, colb: Optional[Tensor], res: Optional[Tensor], gamma: Tensor, beta: Tensor, eps: float,
               relu_out: bool, p: float, seed: int, seed_base: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """fp32, or bf16 activations and parameters (fp32 arithmetic and stats)."""
    dev = require_device(x, colb, res, gamma, beta)
    bf16 = x.dtype == torch.bfloat16
    if bf16:
        _check_dtype(torch.bfloat16, x, colb, res, gamma, beta)
    else:
        _check_f32(x, colb, res, gamma, beta)
    x = _rowmajor(x)
    res = _rowmajor(res) if res is not None else None
    n, d = x.shape
    y = torch.empty((n, d), dtype=x.dtype, device=dev)
    stats = torch.empty((n, 2), dtype=torch.float32, device=dev)
    lib = _lib.load()
    fn, name = (lib.allset_ln_res_fwd_bf16, "allset_ln_res_fwd_bf16") if bf16 else (lib.allset_ln_res_fwd, "allset_ln_res_fwd")
    with on_device(dev), _timed("ln_res_fwd", dev, (2 + (res is not None)) * n * d * x.element_size()):
        check(fn(ptr(x), _ld(x), ptr(colb.contiguous() if colb is not None else None), ptr(res), _ld(res) if res is not None else 0,
                 ptr(gamma.contiguous()), ptr(beta.contiguous()), eps, int(relu_out), p, seed, ptr(y), max(d, 1), ptr(stats), n, d,
                 ptr(seed_base), stream_of(dev)), name)
    return y, stats


def ln_res_bwd(gy: Tensor, x: Tensor, colb: Optional[Tensor], res: Optional[Tensor], stats: Tensor, gamma: Tensor,
               beta: Tensor, relu_out: bool, p: float, seed: int, seed_base: Optional[Tensor] = None, defer_to=None
               ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """(gs, dgamma, dbeta, dcolb); gs is the gradient of x and of res.  bf16 activations: the three parameter
    gradients are accumulated in fp32 and cast back.  ``defer_to`` = the (gamma, beta, colb-or-None) PARAMETERS (fp32): inside
    ``deferred_param_grads()`` their gradients are queued and come back as None."""
    dev = require_device(gy, x, colb, res, stats, gamma, beta)
    bf16 = x.dtype == torch.bfloat16
    if bf16:
        _check_dtype(torch.bfloat16, gy, x, colb, res, gamma, beta)
    gy, x = _rowmajor(gy), _rowmajor(x)
    res = _rowmajor(res) if res is not None else None
    n, d = x.shape
    lib = _lib.load()
    npart = c_int64(0)
    if bf16:
        check(lib.allset_ln_bwd_bf16_partials(n, d, byref(npart)), "allset_ln_bwd_bf16_partials")
    else:
        check(lib.allset_ln_res_bwd_partials(n, d, byref(npart)), "allset_ln_res_bwd_partials")
    partials = torch.empty((npart.value, 3, d), dtype=torch.float32, device=dev)
    gs = torch.empty((n, d), dtype=x.dtype, device=dev)
    fn, name = (lib.allset_ln_res_bwd_bf16, "allset_ln_res_bwd_bf16") if bf16 else (lib.allset_ln_res_bwd, "allset_ln_res_bwd")
    with on_device(dev), _timed("ln_res_bwd", dev, (3 + (res is not None)) * n * d * x.element_size()):
        check(fn(ptr(gy), _ld(gy), ptr(x), _ld(x), ptr(colb.contiguous() if colb is not None else None), ptr(res),
                 _ld(res) if res is not None else 0, ptr(stats), ptr(gamma.contiguous()), ptr(beta.contiguous()), int(relu_out), p,
                 seed, ptr(gs), max(d, 1), ptr(partials), npart.value, n, d, ptr(seed_base), stream_of(dev)), name)
    if not bf16 and defer_to is not None:
        dg_, db_, dc_ = _defer_or_reduce(partials, [(defer_to[0], 0, (d,)), (defer_to[1], d, (d,)),
                                                    (defer_to[2] if colb is not None else None, 2 * d,
                                                     tuple(defer_to[2].shape) if (colb is not None and defer_to[2] is not None) else (d,))], True)
        return gs, dg_, db_, dc_
    if bf16 and (3 * d) % 4 == 0 and npart.value <= 4096:
        red = reduce_partials_to(partials.view(npart.value, 3 * d), 3 * d, torch.bfloat16).view(3, d)   # summed and rounded in one launch
    else:
        red = reduce_partials(partials)
        if bf16:
            red = red.to(torch.bfloat16)
    return gs, red[0], red[1], red[2]


def ln_res_bwd_pma_supported(d: int, heads: int, dtype: torch.dtype = torch.float32) -> bool:
    if dtype == torch.bfloat16:
        return bool(_lib.load().allset_ln_res_bwd_pma_bf16_supported(d, heads))
    return dtype == torch.float32 and bool(_lib.load().allset_ln_res_bwd_pma_supported(d, heads))


def ln_res_bwd_pma(gy: Tensor, x: Tensor, colb: Tensor, stats: Tensor, gamma: Tensor, beta: Tensor, m: Tensor, l: Tensor,
                   defer_to=None) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Backward of ``LayerNorm(x + colb)`` where ``x`` is PMA's pooled output, with the attention-backward statistics written by
    the same pass: returns (gs, dgamma, dbeta, dcolb, pma_stats [n, H, 2]) -- include/allset_hip.h allset_ln_res_bwd_pma.
    fp32, or bf16 activations and parameters (fp32 statistics)."""
    dev = require_device(gy, x, colb, stats, gamma, beta, m, l)
    bf16 = x.dtype == torch.bfloat16
    if bf16:
        _check_dtype(torch.bfloat16, gy, x, colb, gamma, beta)
        _check_f32(stats, m, l)
    else:
        _check_f32(gy, x, colb, stats, gamma, beta, m, l)
    gy, x = _rowmajor(gy), _rowmajor(x)
    n, d = x.shape
    H = m.shape[1]
    lib = _lib.load()
    npart = c_int64(0)
    if bf16:
        check(lib.allset_ln_bwd_bf16_partials(n, d, byref(npart)), "allset_ln_bwd_bf16_partials")
    else:
        check(lib.allset_ln_res_bwd_partials(n, d, byref(npart)), "allset_ln_res_bwd_partials")
    partials = torch.empty((npart.value, 3, d), dtype=torch.float32, device=dev)
    gs = torch.empty((n, d), dtype=x.dtype, device=dev)
    pstats = torch.empty((n, H, 2), dtype=torch.float32, device=dev)
    fn, name = ((lib.allset_ln_res_bwd_pma_bf16, "allset_ln_res_bwd_pma_bf16") if bf16
                else (lib.allset_ln_res_bwd_pma, "allset_ln_res_bwd_pma"))
    with on_device(dev), _timed("ln_res_bwd", dev, 3 * n * d * x.element_size() + n * H * 16):
        check(fn(ptr(gy), _ld(gy), ptr(x), _ld(x), ptr(colb.contiguous()), ptr(stats), ptr(gamma.contiguous()),
                 ptr(beta.contiguous()), ptr(gs), max(d, 1), ptr(partials), npart.value, n, d,
                 ptr(m.contiguous()), ptr(l.contiguous()), ptr(pstats), H, stream_of(dev)), name)
    if not bf16 and defer_to is not None:        # (gamma, beta, att_r) PARAMETERS: queued inside deferred_param_grads()
        dg_, db_, dc_ = _defer_or_reduce(partials, [(defer_to[0], 0, (d,)), (defer_to[1], d, (d,)),
                                                    (defer_to[2], 2 * d, tuple(defer_to[2].shape) if defer_to[2] is not None else (d,))], True)
        return gs, dg_, db_, dc_, pstats
    if bf16 and (3 * d) % 4 == 0 and npart.value <= 4096:
        red = reduce_partials_to(partials.view(npart.value, 3 * d), 3 * d, torch.bfloat16).view(3, d)
    else:
        red = reduce_partials(partials)
        if bf16:
            red = red.to(torch.bfloat16)
    return gs, red[0], red[1], red[2], pstats


class _LayerNormRes(torch.autograd.Function):
    """``y = dropout_p(relu_out(LayerNorm(x + colb + res)))`` in one pass each way (csrc/dense.hip ``ln_res_*``)."""

    @staticmethod
    def forward(ctx, x, colb, res, gamma, beta, eps, relu_out, p):
        seed = _draw_seed() if p > 0.0 else 0
        base = _seed_base() if p > 0.0 else None
        cb = colb.reshape(-1) if colb is not None else None
        y, stats = ln_res_fwd(x, cb, res, gamma, beta, eps, relu_out, p, seed, base)
        ctx.save_for_backward(x, cb, res, stats, gamma, beta)
        ctx.cfg = (bool(relu_out), float(p), seed, base, colb.shape if colb is not None else None)
        ctx.params = (gamma, beta, colb)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, cb, res, stats, gamma, beta = ctx.saved_tensors
        relu_out, p, seed, base, cshape = ctx.cfg
        need = ctx.needs_input_grad
        dfr = _Deferred.active and need[3] and need[4] and (cshape is None or need[1]) and x.dtype == torch.float32
        gs, dg, db, dc = ln_res_bwd(gy.contiguous(), x, cb, res, stats, gamma, beta, relu_out, p, seed, base,
                                    defer_to=ctx.params if dfr else None)
        return (gs, (dc.reshape(cshape) if (cshape is not None and dc is not None) else None), (gs if res is not None else None), dg, db,
                None, None, None)


def layer_norm_res(x: Tensor, colb: Optional[Tensor], res: Optional[Tensor], gamma: Tensor, beta: Tensor,
                   eps: float = 1e-5, relu_out: bool = False, p: float = 0.0) -> Tensor:
    """``dropout_p(relu_out(LayerNorm(x + colb + res)))``; ``colb`` broadcasts over rows (any shape with d elements)."""
    return _LayerNormRes.apply(x, colb, res, gamma, beta, float(eps), bool(relu_out), float(p))


class _PmaProject(torch.autograd.Function):
    """``x -> (x_V = x W_V^T + b_V,  alpha = x w_a^T + b_a)``: PMA's value projection and its folded logit mat-vec
    (reference layers.py:126-131) as one autograd node.  With up to 4 heads the logits are four auxiliary output
    columns of the projection kernel and their input gradient a rank-4 update inside its backward-data kernel -- no
    skinny GEMMs, no add pass for the two gradient branches of ``x``; with more heads the logits use the library GEMM and
    the branches are summed through ``acc_in``."""

    @staticmethod
    def forward(ctx, x, w_v, b_v, w_a, b_a):
        H = w_a.shape[0]
        aux = H <= 4
        if aux:
            w4 = w_a if H == 4 else torch.cat([w_a, w_a.new_zeros(4 - H, w_a.shape[1])])
            b4 = None if b_a is None else (b_a if H == 4 else torch.cat([b_a, b_a.new_zeros(4 - H)]))
            a4 = torch.empty((x.shape[0], 4), dtype=torch.float32, device=x.device)
            x_v, _ = fused_linear_fwd(x, w_v, b_v, aux_w=w4.contiguous(), aux_b=b4, aux_out=a4)
            alpha = a4 if H == 4 else a4[:, :H].contiguous()
        else:
            w4 = w_a
            x_v, _ = fused_linear_fwd(x, w_v, b_v)
            alpha = torch.nn.functional.linear(x, w_a, b_a)
        ctx.save_for_backward(x, w_v, w4)
        ctx.cfg = (b_v is not None, b_a is not None, aux, H)
        ctx.params = (w_v, b_v)
        return x_v, alpha

    @staticmethod
    @once_differentiable
    def backward(ctx, g_v, g_alpha):
        x, w_v, w4 = ctx.saved_tensors
        has_bv, has_ba, aux, H = ctx.cfg
        g_v, g_alpha = g_v.contiguous(), g_alpha.contiguous()
        gx = gwv = gbv = gwa = gba = None
        if (aux and ctx.needs_input_grad[0] and g_v.is_cuda and g_v.dtype == torch.float32
                and fused_linear_bwd_all_aux_supported(w_v.shape[0], w_v.shape[1])):
            # one pass: both gradient branches of x, the projection's weight / bias gradient and the logit columns' own
            g4 = g_alpha if H == 4 else torch.cat([g_alpha, g_alpha.new_zeros(g_alpha.shape[0], 4 - H)], dim=1)
            defer = ctx.params if (_Deferred.active and has_bv and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]) else None
            gx, gwv, gbv, gwa4, gba4 = fused_linear_bwd_all_aux(g_v, w_v, x, g4, w4, defer_to=defer)
            return (gx, gwv, gbv if has_bv else None, gwa4[:H] if H < 4 else gwa4,
                    (gba4[:H] if H < 4 else gba4) if has_ba else None)
        if ctx.needs_input_grad[0]:
            if aux:
                g4 = g_alpha if H == 4 else torch.cat([g_alpha, g_alpha.new_zeros(g_alpha.shape[0], 4 - H)], dim=1)
                gx, _, _ = fused_linear_bwd(g_v, None, 0.0, w_v, x, None, None, False, 0.0, 0, aux_g=g4.contiguous(),
                                            aux_w=w4.contiguous())
            else:
                gx, _, _ = fused_linear_bwd(g_v, None, 0.0, w_v, x, None, None, False, 0.0, 0, acc_in=g_alpha @ w4)
        if ctx.needs_input_grad[1] or (has_bv and ctx.needs_input_grad[2]):
            gwv, gbv = wgrad(g_v, x, want_bias=has_bv)
        if ctx.needs_input_grad[3] or (has_ba and ctx.needs_input_grad[4]):
            if wgrad_supported(g_alpha, x):
                gwa, gba = wgrad(g_alpha, x, want_bias=has_ba)
            else:
                gwa, gba = g_alpha.t() @ x, (g_alpha.sum(0) if has_ba else None)
        return gx, gwv, gbv, gwa, gba


def pma_project(x: Tensor, w_v: Tensor, b_v: Optional[Tensor], w_a: Tensor, b_a: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    return _PmaProject.apply(x, w_v, b_v, w_a, b_a)


class _PmaResidualFF(torch.autograd.Function):
    """``y = dropout_p(relu_post(LN(out + relu(W2 relu(W1 out + b1) + b2))))`` -- the second half of the PMA tail
    (reference layers.py:156-157 with a 2-layer rFF) as one autograd node: three kernels forward, five backward, and the
    two gradient branches of ``out`` (through the LayerNorm and through rFF) are summed by the last backward-data
    kernel (``acc_in``) instead of an extra add pass."""

    @staticmethod
    def forward(ctx, out, w1, b1, w2, b2, gamma, beta, eps, relu_post, p):
        y1, _ = fused_linear_fwd(out, w1, b1)
        words = activation_mask_words(out.shape[0], w2.shape[0])
        mask = torch.empty(words, dtype=torch.int32, device=out.device) if words > 0 else None
        z, _ = fused_linear_fwd(y1, w2, b2, None, None, 1e-5, True, 0.0, 0, True, 0.0, 0, None, mask)
        seed = _draw_seed() if p > 0.0 else 0
        base = _seed_base() if p > 0.0 else None
        y, stats = ln_res_fwd(out, None, z, gamma, beta, eps, relu_post, p, seed, base)
        ctx.save_for_backward(out, y1, z, mask, stats, w1, w2, gamma, beta)
        ctx.cfg = (bool(relu_post), float(p), seed, base, b1 is not None, b2 is not None)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        out, y1, z, mask, stats, w1, w2, gamma, beta = ctx.saved_tensors
        relu_post, p, seed, base, has_b1, has_b2 = ctx.cfg
        need = ctx.needs_input_grad
        prm = getattr(ctx, "params", None)
        dfr = _Deferred.active and prm is not None and need[5] and need[6] and out.dtype == torch.float32
        gs, dg, db, _ = ln_res_bwd(gy.contiguous(), out, None, z, stats, gamma, beta, relu_post, p, seed, base,
                                   defer_to=(prm[0], prm[1], None) if dfr else None)
        zy = None if mask is not None else z
        if mask is not None and gs.shape[0] > 0 and one_pass_preferred(w2.shape[0], w2.shape[1]) and one_pass_preferred(w1.shape[0], w1.shape[1]):
            # round 3: the split-role one-pass kernel covers both Linears (relu prologue + mask; plain + acc_in): two launches
            # instead of four, each reading its gradient and its input once
            g1, _, _, gw2, gb2 = fused_linear_bwd_all(gs, mask, 0.0, w2, y1, None, None, None, True, 0.0, 0, want_bias=has_b2)
            gout, _, _, gw1, gb1 = fused_linear_bwd_all(g1, None, 0.0, w1, out, None, None, None, False, 0.0, 0, acc_in=gs, want_bias=has_b1)
            return gout, gw1, gb1, gw2, gb2, dg, db, None, None, None
        # (the one-wave one-pass kernel is not used here: without a LayerNorm to recompute, the plain weight-gradient kernel
        # costs 0.19 ms and the pair 0.48-0.59 ms per Linear against 0.58 ms for that kernel -- measured, round 2)
        gw2, gb2 = wgrad_fused(gs, zy, 0.0, y1, None, None, None, True, 0.0, 0, want_bias=has_b2, mask=mask)
        g1, _, _ = fused_linear_bwd(gs, zy, 0.0, w2, y1, None, None, True, 0.0, 0, None, mask)
        gw1, gb1 = wgrad(g1, out, want_bias=has_b1)
        gout, _, _ = fused_linear_bwd(g1, None, 0.0, w1, out, None, None, False, 0.0, 0, acc_in=gs)   # gs + rFF branch
        return gout, gw1, gb1, gw2, gb2, dg, db, None, None, None


def pma_residual_ff(out: Tensor, w1, b1, w2, b2, gamma, beta, eps: float = 1e-5, relu_post: bool = False, p: float = 0.0) -> Tensor:
    return _PmaResidualFF.apply(out, w1, b1, w2, b2, gamma, beta, float(eps), bool(relu_post), float(p))


# ---- the PMA tail with ln0 / ln1 inside its two rFF Linears (round 5; csrc/fused_fwd2.hip modes 1 and 2) ---------------------------
def pma_tail_supported(out_like: Tensor, d: int, w1: Tensor, w2: Tensor) -> bool:
    """``ln1(out + relu(rFF(out)))`` with ``out = ln0(pooled + att_r)`` (reference layers.py:153-157) runs as TWO forward kernels:
    device fp32, every width 128, the fp16x3 arithmetic not switched off."""
    return (out_like.is_cuda and out_like.dtype == torch.float32 and d == 128 and tuple(w1.shape) == (128, 128) and tuple(w2.shape) == (128, 128)
            and _arith != _lib.ARITH_BF16X6 and bool(_lib.load().allset_fused_linear_tail_supported(128, 128)))


def fused_linear_fwd_ln_side(x: Tensor, colb: Optional[Tensor], gamma: Tensor, beta: Tensor, eps: float, weight: Tensor,
                             bias: Optional[Tensor], relu_out: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """``(y, u, stats)``: ``u = LayerNorm(x + colb)``, ``y = relu_out?(u W^T + b)`` (include/allset_hip_ext.h allset_fused_linear_fwd_ln_side)."""
    dev = require_device(x, colb, gamma, beta, weight, bias)
    _check_f32(x, colb, gamma, beta, weight, bias)
    x = _rowmajor(x)
    n, K = x.shape
    N = weight.shape[0]
    y = torch.empty((n, N), dtype=torch.float32, device=dev)
    u = torch.empty((n, K), dtype=torch.float32, device=dev)
    stats = torch.empty((n, 2), dtype=torch.float32, device=dev)
    with on_device(dev), _timed("fused_linear_fwd", dev, n * (K + N + K) * 4):
        check(_lib.load().allset_fused_linear_fwd_ln_side(
            ptr(x), _ld(x), ptr(colb.contiguous() if colb is not None else None), ptr(gamma.contiguous()), ptr(beta.contiguous()), eps,
            ptr(weight.contiguous()), ptr(bias.contiguous() if bias is not None else None), int(relu_out), ptr(y), max(N, 1), ptr(u),
            max(K, 1), ptr(stats), n, K, N, stream_of(dev)), "allset_fused_linear_fwd_ln_side")
    return y, u, stats


def fused_linear_fwd_res_ln(x: Tensor, relu_in: bool, weight: Tensor, bias: Optional[Tensor], relu_out: bool, res: Tensor, gamma: Tensor,
                            beta: Tensor, eps: float, relu_post: bool, p: float, seed: int, seed_base: Optional[Tensor],
                            mask_out: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """``(y, s, stats)``: ``z = relu_out?(relu_in?(x) W^T + b)``, ``s = res + z``, ``y = dropout_p(relu_post?(LayerNorm(s)))``
    (include/allset_hip_ext.h allset_fused_linear_fwd_res_ln)."""
    dev = require_device(x, weight, bias, res, gamma, beta)
    _check_f32(x, weight, bias, res, gamma, beta)
    x, res = _rowmajor(x), _rowmajor(res)
    n, K = x.shape
    N = weight.shape[0]
    y = torch.empty((n, N), dtype=torch.float32, device=dev)
    s = torch.empty((n, N), dtype=torch.float32, device=dev)
    stats = torch.empty((n, 2), dtype=torch.float32, device=dev)
    with on_device(dev), _timed("fused_linear_fwd", dev, n * (K + 3 * N) * 4):
        check(_lib.load().allset_fused_linear_fwd_res_ln(
            ptr(x), _ld(x), int(relu_in), ptr(weight.contiguous()), ptr(bias.contiguous() if bias is not None else None), int(relu_out),
            ptr(res), _ld(res), ptr(gamma.contiguous()), ptr(beta.contiguous()), eps, int(relu_post), p, seed, ptr(seed_base), ptr(y),
            max(N, 1), ptr(s), max(N, 1), ptr(stats), ptr(mask_out), n, K, N, stream_of(dev)), "allset_fused_linear_fwd_res_ln")
    return y, s, stats


def pma_tail_fwd(pooled: Tensor, cb: Tensor, g0, b0, eps0, w1, b1, w2, b2, g1, bt1, eps1, relu_post: bool, p: float):
    """Forward of the whole tail in two kernels.  Returns ``(y, saved)``; ``saved`` goes to :func:`pma_tail_bwd`."""
    y1, out, stats0 = fused_linear_fwd_ln_side(pooled, cb, g0, b0, eps0, w1, b1, False)
    words = activation_mask_words(pooled.shape[0], w2.shape[0])
    mask = torch.empty(words, dtype=torch.int32, device=pooled.device)
    seed = _draw_seed() if p > 0.0 else 0
    base = _seed_base() if p > 0.0 else None
    y, s, stats1 = fused_linear_fwd_res_ln(y1, True, w2, b2, True, out, g1, bt1, eps1, relu_post, p, seed, base, mask)
    return y, (pooled, cb, stats0, g0, b0, out, y1, mask, s, stats1, w1, w2, g1, bt1), (bool(relu_post), float(p), seed, base,
                                                                                       b1 is not None, b2 is not None)


def fused_linear_bwd_pma_tail_supported(heads: int) -> bool:
    return _arith != _lib.ARITH_BF16X6 and bool(_lib.load().allset_fused_linear_bwd_pma_tail_supported(128, 128, int(heads)))


def fused_linear_bwd_pma_tail(gy: Tensor, weight: Tensor, pooled: Tensor, colb: Optional[Tensor], stats: Tensor, gamma: Tensor, beta: Tensor,
                              gres: Tensor, m: Tensor, l: Tensor, want_bias: bool = True, defer_to=None):
    """``(g_pooled, dgamma0, dbeta0, dcolb, gW, gb, pma_stats)``: the backward of the PMA tail's first rFF Linear with the residual
    branch's gradient added in front of ln0's backward, ln0's backward, and the pooling's backward statistics -- ONE pass
    (include/allset_hip_ext.h allset_fused_linear_bwd_pma_tail; until round 6: fused_linear_bwd_all(acc_in) + ln_res_bwd_pma)."""
    dev = require_device(gy, weight, pooled, colb, stats, gamma, beta, gres, m, l)
    _check_f32(gy, weight, pooled, colb, stats, gamma, beta, gres, m, l)
    gy, pooled, gres = _rowmajor(gy), _rowmajor(pooled), _rowmajor(gres)
    n, O = gy.shape
    I = pooled.shape[1]
    H = m.shape[1]
    lib = _lib.load()
    ns = c_int64(0)
    check(lib.allset_fused_linear_bwd_all_slices_for(n, O, I, 0, byref(ns)), "allset_fused_linear_bwd_all_slices_for")
    P = ns.value
    M = (O * I + O + 3 * I + 3) // 4 * 4
    part = torch.empty((P, M), dtype=torch.float32, device=dev)
    gx = torch.empty((n, I), dtype=torch.float32, device=dev)
    pstats = torch.empty((n, H, 2), dtype=torch.float32, device=dev)
    with on_device(dev), _timed("fused_linear_bwd_all", dev, n * (O + 3 * I) * 4 + n * H * 16):
        check(lib.allset_fused_linear_bwd_pma_tail(
            ptr(gy), _ld(gy), ptr(weight.contiguous()), ptr(pooled), _ld(pooled), ptr(colb.contiguous() if colb is not None else None),
            ptr(stats), ptr(gamma.contiguous()), ptr(beta.contiguous()), ptr(gres), _ld(gres), ptr(gx), max(I, 1), ptr(part), M, P,
            ptr(m.contiguous()), ptr(l.contiguous()), ptr(pstats), H, n, O, I, stream_of(dev)), "allset_fused_linear_bwd_pma_tail")
    o = O * I
    if defer_to is not None and colb is not None and _deferrable(part, *defer_to):
        # (weight, bias, gamma0, beta0, att_r) PARAMETERS: queued for the batched reduction of deferred_param_grads()
        w_p, b_p, g_p, bt_p, c_p = defer_to
        _defer(part, [(w_p, 0, (O, I)), (b_p if want_bias else None, o, (O,)), (g_p, o + O, (I,)), (bt_p, o + O + I, (I,)),
                      (c_p, o + O + 2 * I, tuple(c_p.shape))])
        return gx, None, None, None, None, None, pstats
    red = reduce_partials(part)
    return (gx, red[o + O:o + O + I], red[o + O + I:o + O + 2 * I], red[o + O + 2 * I:o + O + 3 * I], red[:o].view(O, I),
            red[o:o + O] if want_bias else None, pstats)


def fused_linear_bwd_ln_pro_supported() -> bool:
    return _arith != _lib.ARITH_BF16X6 and bool(_lib.load().allset_fused_linear_bwd_ln_pro_supported(128, 128))


def fused_linear_bwd_ln_pro(gy: Tensor, s: Tensor, stats2: Tensor, gamma2: Tensor, beta2: Tensor, relu_post: bool, p: float, seed: int,
                            seed_base: Optional[Tensor], mask: Tensor, weight: Tensor, x: Tensor, relu_in: bool, want_bias: bool = True,
                            defer_to=None):
    """``(gs, dgamma2, dbeta2, gx, gW, gb)``: ln1's backward (on the saved sum ``s``) as the gy prologue of the second rFF Linear's
    one-pass backward (include/allset_hip_ext.h allset_fused_linear_bwd_ln_pro; until round 6: ln_res_bwd + fused_linear_bwd_all)."""
    dev = require_device(gy, s, stats2, gamma2, beta2, mask, weight, x)
    _check_f32(gy, s, stats2, gamma2, beta2, weight, x)
    gy, s, x = _rowmajor(gy), _rowmajor(s), _rowmajor(x)
    n, O = gy.shape
    I = x.shape[1]
    lib = _lib.load()
    ns = c_int64(0)
    check(lib.allset_fused_linear_bwd_all_slices_for(n, O, I, 0, byref(ns)), "allset_fused_linear_bwd_all_slices_for")
    P = ns.value
    M = (O * I + 3 * O + 3) // 4 * 4
    part = torch.empty((P, M), dtype=torch.float32, device=dev)
    gs = torch.empty((n, O), dtype=torch.float32, device=dev)
    gx = torch.empty((n, I), dtype=torch.float32, device=dev)
    with on_device(dev), _timed("fused_linear_bwd_all", dev, n * (3 * O + 2 * I) * 4):
        check(lib.allset_fused_linear_bwd_ln_pro(
            ptr(gy), _ld(gy), ptr(s), _ld(s), ptr(stats2), ptr(gamma2.contiguous()), ptr(beta2.contiguous()), int(relu_post), float(p), int(seed),
            ptr(seed_base), ptr(mask), ptr(weight.contiguous()), ptr(x), _ld(x), int(relu_in), ptr(gs), max(O, 1), ptr(gx), max(I, 1), ptr(part), M, P,
            n, O, I, stream_of(dev)), "allset_fused_linear_bwd_ln_pro")
    o = O * I
    if defer_to is not None and _deferrable(part, *defer_to):
        w_p, b_p, g_p, bt_p = defer_to         # (weight, bias, gamma1, beta1) PARAMETERS
        _defer(part, [(w_p, 0, (O, I)), (b_p if want_bias else None, o, (O,)), (g_p, o + O, (O,)), (bt_p, o + 2 * O, (O,))])
        return gs, None, None, gx, None, None
    red = reduce_partials(part)
    return gs, red[o + O:o + 2 * O], red[o + 2 * O:o + 3 * O], gx, red[:o].view(O, I), (red[o:o + O] if want_bias else None)


def pma_tail_bwd(saved, cfg, gy: Tensor, m: Optional[Tensor] = None, l: Optional[Tensor] = None, params=None):
    """``(g_pooled, dcolb, dg0, db0, gw1, gb1, gw2, gb2, dg1, db1, pma_stats or None)``: ln1's backward on the saved sum, the two
    Linears' one-pass backward (the residual branch summed through ``acc_in``), ln0's backward -- with the pooling's backward
    statistics written by the same pass when the softmax statistics ``(m, l)`` are given."""
    pooled, cb, stats0, g0, b0, out, y1, mask, s, stats1, w1, w2, g1, bt1 = saved
    relu_post, p, seed, base, has_b1, has_b2 = cfg
    # ``params`` = (att_r, g0, b0, w1, b1, w2, b2, g1, bt1) as the node received them: inside deferred_param_grads() the two one-pass
    # kernels' parameter gradients are queued for the batched reduction and come back as None
    d1 = (params[5], params[6], params[7], params[8]) if params is not None else None
    d0 = (params[3], params[4], params[1], params[2], params[0]) if params is not None else None
    if fused_linear_bwd_ln_pro_supported():
        # ln1's backward inside the second Linear's one-pass backward (csrc/fused_bwd6.hip PT2): one pass instead of two
        gs, dg1, db1, gh, gw2, gb2 = fused_linear_bwd_ln_pro(gy.contiguous(), s, stats1, g1, bt1, relu_post, p, seed, base, mask, w2, y1, True,
                                                             want_bias=has_b2, defer_to=d1)
    else:
        gs, dg1, db1, _ = ln_res_bwd(gy.contiguous(), s, None, None, stats1, g1, bt1, relu_post, p, seed, base)
        gh, _, _, gw2, gb2 = fused_linear_bwd_all(gs, mask, 0.0, w2, y1, None, None, None, True, 0.0, 0, want_bias=has_b2)
    if m is None and fused_linear_bwd_pma_tail_supported(1):
        # the tail on its own (no pooling behind it): the same pass with one dummy head, its statistics discarded
        m1 = torch.zeros((pooled.shape[0], 1), dtype=torch.float32, device=pooled.device)
        g_pooled, dg0, db0, dc, gw1, gb1, _ = fused_linear_bwd_pma_tail(gh, w1, pooled, cb, stats0, g0, b0, gs, m1, torch.ones_like(m1), want_bias=has_b1)
        return g_pooled, dc, dg0, db0, gw1, gb1, gw2, gb2, dg1, db1, None
    if m is not None and fused_linear_bwd_pma_tail_supported(m.shape[1]):
        # the first Linear's backward, ln0's backward and the pooling's statistics in ONE pass (csrc/fused_bwd6.hip PT)
        g_pooled, dg0, db0, dc, gw1, gb1, pstats = fused_linear_bwd_pma_tail(gh, w1, pooled, cb, stats0, g0, b0, gs, m, l, want_bias=has_b1,
                                                                             defer_to=d0)
        return g_pooled, dc, dg0, db0, gw1, gb1, gw2, gb2, dg1, db1, pstats
    gout, _, _, gw1, gb1 = fused_linear_bwd_all(gh, None, 0.0, w1, out, None, None, None, False, 0.0, 0, acc_in=gs, want_bias=has_b1)
    if m is not None:
        g_pooled, dg0, db0, dc, pstats = ln_res_bwd_pma(gout, pooled, cb, stats0, g0, b0, m, l)
    else:
        g_pooled, dg0, db0, dc = ln_res_bwd(gout, pooled, cb, None, stats0, g0, b0, False, 0.0, 0, None)
        pstats = None
    return g_pooled, dc, dg0, db0, gw1, gb1, gw2, gb2, dg1, db1, pstats


def tail_defer_params(ctx, idx):
    """``ctx.params`` (att_r, g0, b0, w1, b1, w2, b2, g1, bt1) when the node runs inside ``deferred_param_grads()`` and every one of
    them (inputs ``idx`` of the node) wants its gradient; None otherwise."""
    if not _Deferred.active or getattr(ctx, "params", None) is None:
        return None
    if not all(ctx.needs_input_grad[i] for i, q in zip(idx, ctx.params) if q is not None):
        return None
    return ctx.params


class _PmaTail(torch.autograd.Function):
    """``dropout_p(relu_post?(ln1(out + relu(rFF(out)))))`` with ``out = ln0(pooled + att_r)`` as one autograd node on two forward
    kernels (:func:`pma_tail_fwd`)."""

    @staticmethod
    def forward(ctx, pooled, att_r, g0, b0, eps0, w1, b1, w2, b2, g1, bt1, eps1, relu_post, p):
        y, saved, cfg = pma_tail_fwd(pooled, att_r.reshape(-1), g0, b0, eps0, w1, b1, w2, b2, g1, bt1, eps1, relu_post, p)
        ctx.save_for_backward(*saved)
        ctx.cfg, ctx.cshape = cfg, att_r.shape
        ctx.params = (att_r, g0, b0, w1, b1, w2, b2, g1, bt1)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        g_pooled, dc, dg0, db0, gw1, gb1, gw2, gb2, dg1, db1, _ = pma_tail_bwd(ctx.saved_tensors, ctx.cfg, gy,
                                                                               params=tail_defer_params(ctx, (1, 2, 3, 5, 6, 7, 8, 9, 10)))
        return g_pooled, (dc.reshape(ctx.cshape) if dc is not None else None), dg0, db0, None, gw1, gb1, gw2, gb2, dg1, db1, None, None, None


def pma_tail(pooled: Tensor, att_r: Tensor, g0, b0, eps0, w1, b1, w2, b2, g1, bt1, eps1, relu_post: bool = False, p: float = 0.0) -> Tensor:
    return _PmaTail.apply(pooled, att_r, g0, b0, float(eps0), w1, b1, w2, b2, g1, bt1, float(eps1), bool(relu_post), float(p))


class _PmaFold(torch.autograd.Function):
    """``(w [H, K], b [H]) = fold(W_K [H C, K], b_K [H C], att_r [.., H, C])``: the weight of PMA's folded logits as ONE kernel each
    way (as torch ops: mul, sum, mul, sum forward and six more backward -- ~80 us per replayed dataset-scale step).  fp32, or bf16
    parameters (fp32 arithmetic, each output rounded once)."""

    @staticmethod
    def forward(ctx, Wk, bk, att):
        dev = require_device(Wk, att)
        HC, K = Wk.shape
        H = att.shape[-2]
        C = HC // H
        dt = Wk.dtype
        _check_dtype(dt, att, bk)
        Wk_c, att_c = Wk.contiguous(), att.contiguous()
        w = torch.empty((H, K), dtype=dt, device=dev)
        b = torch.empty((H,), dtype=dt, device=dev)
        lib = _lib.load()
        fn, name = ((lib.allset_pma_fold_fwd_bf16, "allset_pma_fold_fwd_bf16") if dt == torch.bfloat16
                    else (lib.allset_pma_fold_fwd, "allset_pma_fold_fwd"))
        with on_device(dev):
            check(fn(ptr(Wk_c), ptr(bk.contiguous() if bk is not None else None), ptr(att_c), ptr(w), ptr(b), H, C, K, stream_of(dev)), name)
        ctx.save_for_backward(Wk_c, bk, att_c)
        ctx.att_shape = att.shape
        return w, b

    @staticmethod
    @once_differentiable
    def backward(ctx, gw, gb):
        Wk, bk, att = ctx.saved_tensors
        dev = Wk.device
        HC, K = Wk.shape
        H = att.shape[-2]
        C = HC // H
        dt = Wk.dtype
        gWk = torch.empty_like(Wk)
        gbk = torch.empty((HC,), dtype=dt, device=dev) if bk is not None else None
        gatt = torch.empty((HC,), dtype=dt, device=dev)
        gw = gw.to(dt).contiguous()
        gb = gb.to(dt).contiguous() if gb is not None else None
        lib = _lib.load()
        fn, name = ((lib.allset_pma_fold_bwd_bf16, "allset_pma_fold_bwd_bf16") if dt == torch.bfloat16
                    else (lib.allset_pma_fold_bwd, "allset_pma_fold_bwd"))
        with on_device(dev):
            check(fn(ptr(Wk), ptr(bk.contiguous() if bk is not None else None), ptr(att), ptr(gw), ptr(gb), ptr(gWk), ptr(gbk), ptr(gatt),
                     H, C, K, stream_of(dev)), name)
        return gWk, gbk, gatt.view(ctx.att_shape)


def pma_fold(Wk: Tensor, bk: Optional[Tensor], att: Tensor) -> Tuple[Tensor, Tensor]:
    """fp32 or bf16 device parameters, all of one dtype (callers keep the torch expression for anything else)."""
    return _PmaFold.apply(Wk, bk, att)


# ---- the bf16 regime (BASELINE configs[4]): csrc/fused_bf16.hip ---------------------------------------------------------
def linear_bf16_supported(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> bool:
    """bf16 rows on the device, bf16 parameters, in / out features in {128, 256}."""
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and (bias is None or bias.dtype == torch.bfloat16)
            and bool(_lib.load().allset_linear_bf16_supported(weight.shape[1], weight.shape[0])))


def linear_bf16_fwd(x: Tensor, weight: Tensor, bias: Optional[Tensor], relu_out: bool = False, aux_w: Optional[Tensor] = None,
                    aux_b: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """``y = act(x W^T + b)`` (bf16) and, with ``aux_w`` [4, K], the fp32 logits ``x aux_w^T + aux_b`` [n, 4]."""
    dev = require_device(x, weight)
    _check_dtype(torch.bfloat16, x, weight)
    x = _rowmajor(x)
    n, K = x.shape
    N = weight.shape[0]
    y = torch.empty((n, N), dtype=torch.bfloat16, device=dev)
    aux = torch.empty((n, 4), dtype=torch.float32, device=dev) if aux_w is not None else None
    with on_device(dev), _timed("linear_bf16_fwd", dev, n * (K + N) * 2 + (n * 16 if aux is not None else 0)):
        check(_lib.load().allset_linear_bf16_fwd(ptr(x), _ld(x), ptr(weight.contiguous()), ptr(bias.contiguous() if bias is not None else None),
                                                 int(relu_out), ptr(aux_w.contiguous() if aux_w is not None else None),
                                                 ptr(aux_b.contiguous() if aux_b is not None else None), ptr(aux), ptr(y), N, n, K, N,
                                                 stream_of(dev)), "allset_linear_bf16_fwd")
    return y, aux


def linear_bf16_fwd_mask(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """``y = relu(x W^T + b)`` (bf16) and its relu mask as one bit per element (uint8 [n, N / 8]; the private bit order of
    ``allset_linear_bf16_fwd_mask``, consumed by :func:`linear_bf16_bwd_bits` and ``wgrad(..., bits=)``)."""
    dev = require_device(x, weight)
    _check_dtype(torch.bfloat16, x, weight)
    x = _rowmajor(x)
    n, K = x.shape
    N = weight.shape[0]
    lib = _lib.load()
    y = torch.empty((n, N), dtype=torch.bfloat16, device=dev)
    bits = torch.empty((n, int(lib.allset_linear_bf16_mask_pitch(N))), dtype=torch.uint8, device=dev)
    with on_device(dev), _timed("linear_bf16_fwd", dev, n * (K + N) * 2 + n * N // 8):
        check(lib.allset_linear_bf16_fwd_mask(ptr(x), _ld(x), ptr(weight.contiguous()), ptr(bias.contiguous() if bias is not None else None),
                                              ptr(y), N, ptr(bits), n, K, N, stream_of(dev)), "allset_linear_bf16_fwd_mask")
    return y, bits


def linear_bf16_bwd(gy: Tensor, weight: Tensor, ymask: Optional[Tensor] = None, want_ga: bool = False,
                    acc_in: Optional[Tensor] = None, galpha: Optional[Tensor] = None, aux_w: Optional[Tensor] = None
                    ) -> Tuple[Tensor, Tensor]:
    """``gx = (gy where ymask > 0) W [+ acc_in] [+ galpha aux_w]`` (one rounding), and the masked gradient ``ga`` (``gy``
    itself without a mask) for the weight-gradient kernel."""
    dev = require_device(gy, weight)
    _check_dtype(torch.bfloat16, gy, weight)
    gy = _rowmajor(gy)
    n, O = gy.shape
    I = weight.shape[1]
    gx = torch.empty((n, I), dtype=torch.bfloat16, device=dev)
    ga = torch.empty((n, O), dtype=torch.bfloat16, device=dev) if (ymask is not None and want_ga) else None
    if ymask is not None:
        ymask = _rowmajor(ymask)
    if acc_in is not None:
        acc_in = _rowmajor(acc_in)
    nbytes = n * (O + I) * 2 + (n * O * 2 if ymask is not None else 0) + (n * O * 2 if ga is not None else 0) \
        + (n * I * 2 if acc_in is not None else 0)
    with on_device(dev), _timed("linear_bf16_bwd", dev, nbytes):
        check(_lib.load().allset_linear_bf16_bwd(ptr(gy), _ld(gy), ptr(ymask), _ld(ymask) if ymask is not None else 0, ptr(ga), O,
                                                 ptr(weight.contiguous()), ptr(galpha.contiguous() if galpha is not None else None),
                                                 ptr(aux_w.contiguous() if aux_w is not None else None), ptr(acc_in),
                                                 _ld(acc_in) if acc_in is not None else 0, ptr(gx), I, n, O, I, stream_of(dev)),
              "allset_linear_bf16_bwd")
    return gx, (ga if ga is not None else gy)


def linear_bf16_bwd_bits(gy: Tensor, weight: Tensor, bits: Tensor, acc_in: Optional[Tensor] = None) -> Tensor:
    """``gx = (gy where bit) W [+ acc_in]`` behind a relu whose mask is :func:`linear_bf16_fwd_mask`'s bit mask.  The masked
    gradient is not written: the weight-gradient kernel applies the same bits (``wgrad(gy, u, bits=bits)``)."""
    dev = require_device(gy, weight, bits)
    _check_dtype(torch.bfloat16, gy, weight)
    gy = _rowmajor(gy)
    n, O = gy.shape
    I = weight.shape[1]
    if bits.dtype != torch.uint8 or tuple(bits.shape) != (n, O // 8) or not bits.is_contiguous():
        raise _lib.AllSetHipError("linear_bf16_bwd_bits: bits must be the contiguous uint8 [n, O / 8] mask of linear_bf16_fwd_mask")
    gx = torch.empty((n, I), dtype=torch.bfloat16, device=dev)
    if acc_in is not None:
        acc_in = _rowmajor(acc_in)
    nbytes = n * (O + I) * 2 + n * O // 8 + (n * I * 2 if acc_in is not None else 0)
    with on_device(dev), _timed("linear_bf16_bwd", dev, nbytes):
        check(_lib.load().allset_linear_bf16_bwd_bits(ptr(gy), _ld(gy), ptr(bits), ptr(weight.contiguous()), ptr(acc_in),
                                                      _ld(acc_in) if acc_in is not None else 0, ptr(gx), I, n, O, I, stream_of(dev)),
              "allset_linear_bf16_bwd_bits")
    return gx


def wgrad_bf16_ex2_supported(O: int, I: int, bits: bool, aux: bool) -> bool:
    return bool(_lib.load().allset_wgrad_bf16_ex2_supported(O, I, int(bits), int(aux)))


def wgrad_bf16_ex2(ga: Tensor, u: Tensor, bits: Optional[Tensor] = None, g4: Optional[Tensor] = None, want_bias: bool = True,
                   defer_to=None):
    """``(gW, gb[, gWa [4, I], gba [4]])`` of a bf16 Linear in one pass over ``ga`` and ``u``: ``bits`` = the relu bit mask of the
    Linear's output (``ga`` is the gradient BEFORE the mask), ``g4`` = the fp32 [n, 4] gradient of four auxiliary output columns.
    Results bf16 (summed in fp32, rounded once by the reduction).  ``defer_to`` = the (weight, bias) PARAMETERS: inside
    ``deferred_param_grads()`` gW / gb are queued for the batched reduction and come back as None (the auxiliary rows, whose
    weights are not parameters, are reduced at once)."""
    dev = require_device(ga, u)
    _check_dtype(torch.bfloat16, ga, u)
    ga, u = _rowmajor(ga), _rowmajor(u)
    n, O = ga.shape
    I = u.shape[1]
    lib = _lib.load()
    ns = c_int64(0)
    check(lib.allset_wgrad_bf16_slices(n, O, I, byref(ns)), "allset_wgrad_bf16_slices")
    M = O * I + (O if want_bias else 0) + (4 * I + 4 if g4 is not None else 0)
    part = torch.empty((ns.value, M), dtype=torch.float32, device=dev)
    if g4 is not None:
        g4 = g4.float().contiguous()
    nbytes = n * (O + I) * 2 + (n * O // 8 if bits is not None else 0) + (n * 16 if g4 is not None else 0)
    with on_device(dev), _timed("wgrad", dev, nbytes):
        check(lib.allset_wgrad_bf16_ex2(ptr(ga), _ld(ga), ptr(bits), ptr(g4), ptr(u), _ld(u), ptr(part), M, int(want_bias), ns.value,
                                        n, O, I, stream_of(dev)), "allset_wgrad_bf16_ex2")
    M_main = O * I + (O if want_bias else 0)
    if defer_to is not None and M_main % 4 == 0 and _deferrable(part, defer_to[0], defer_to[1] if want_bias else None, M=M_main):
        _defer(part, [(defer_to[0], 0, (O, I)), (defer_to[1] if want_bias else None, O * I, (O,))], M=M_main)
        if g4 is None:
            return None, None
        redx = reduce_partials_slice(part, M_main, 4 * I + 4, torch.bfloat16)
        return None, None, redx[:4 * I].view(4, I), redx[4 * I:]
    red = reduce_partials_to(part, M, torch.bfloat16)
    gw = red[:O * I].view(O, I)
    off = O * I
    gb = None
    if want_bias:
        gb = red[off:off + O]
        off += O
    if g4 is None:
        return gw, gb
    return gw, gb, red[off:off + 4 * I].view(4, I), red[off + 4 * I:off + 4 * I + 4]


class _LinearBf16(torch.autograd.Function):
    """``y = act(x W^T + b)`` in the bf16 regime: one kernel forward (relu in its epilogue; when a gradient will be asked for, its
    mask as one bit per element beside y), one backward-data kernel and the full-width bf16 weight-gradient kernel, both applying
    the bit mask to gy as they stage it."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu_out):
        O, I = weight.shape
        need_grad = any(ctx.needs_input_grad[:3])
        use_bits = bool(relu_out) and need_grad and wgrad_bf16_ex2_supported(O, I, True, False)
        if use_bits:
            y, bits = linear_bf16_fwd_mask(x, weight, bias)
            ctx.save_for_backward(x, weight, bits)
        else:
            y, _ = linear_bf16_fwd(x, weight, bias, relu_out)
            ctx.save_for_backward(x, weight, y if relu_out else None)
        ctx.has_bias = bias is not None
        ctx.use_bits = use_bits
        ctx.params = (weight, bias)               # (the objects themselves: deferred_param_grads assigns their .grad)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = gy.contiguous()
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        gx = gw = gb = None
        if ctx.use_bits:
            if ctx.needs_input_grad[0]:
                gx = linear_bf16_bwd_bits(gy, weight, y)
            if need_w:
                defer = ctx.params if (_Deferred.active and ctx.needs_input_grad[1] and (not ctx.has_bias or ctx.needs_input_grad[2])) else None
                gw, gb = wgrad_bf16_ex2(gy, x, bits=y, want_bias=ctx.has_bias, defer_to=defer)
            return gx, gw, gb, None
        ga = gy
        if ctx.needs_input_grad[0]:
            gx, ga = linear_bf16_bwd(gy, weight, y, want_ga=need_w)
        elif y is not None and need_w:
            ga = torch.where(y > 0, gy, torch.zeros_like(gy))
        if need_w:
            gw, gb = wgrad(ga, x, want_bias=ctx.has_bias)
        return gx, gw, gb, None


def linear_bf16(x: Tensor, weight: Tensor, bias: Optional[Tensor], relu_out: bool = False) -> Tensor:
    return _LinearBf16.apply(x, weight, bias, bool(relu_out))


class _PmaProjectBf16(torch.autograd.Function):
    """:class:`_PmaProject` in the bf16 regime (H <= 4): the folded logits are four fp32 auxiliary output columns of the value
    projection's kernel, and their input gradient a rank-4 term of its backward-data kernel."""

    @staticmethod
    def forward(ctx, x, w_v, b_v, w_a, b_a):
        H = w_a.shape[0]
        w4 = w_a if H == 4 else torch.cat([w_a, w_a.new_zeros(4 - H, w_a.shape[1])])
        b4 = None if b_a is None else (b_a if H == 4 else torch.cat([b_a, b_a.new_zeros(4 - H)]))
        x_v, a4 = linear_bf16_fwd(x, w_v, b_v, False, w4, b4)
        ctx.save_for_backward(x, w_v, w4)
        ctx.cfg = (b_v is not None, b_a is not None, H)
        ctx.params = (w_v, b_v)
        return x_v, (a4 if H == 4 else a4[:, :H].contiguous())

    @staticmethod
    @once_differentiable
    def backward(ctx, g_v, g_alpha):
        x, w_v, w4 = ctx.saved_tensors
        has_bv, has_ba, H = ctx.cfg
        g_v, g_alpha = g_v.contiguous(), g_alpha.float().contiguous()
        gx = gwv = gbv = gwa = gba = None
        if ctx.needs_input_grad[0]:
            g4 = g_alpha if H == 4 else torch.cat([g_alpha, g_alpha.new_zeros(g_alpha.shape[0], 4 - H)], dim=1)
            gx, _ = linear_bf16_bwd(g_v, w_v, galpha=g4, aux_w=w4)
        need_v = ctx.needs_input_grad[1] or (has_bv and ctx.needs_input_grad[2])
        need_a = ctx.needs_input_grad[3] or (has_ba and ctx.needs_input_grad[4])
        if need_v and need_a and wgrad_bf16_ex2_supported(w_v.shape[0], w_v.shape[1], False, True) and g_v.stride(1) == 1 \
                and g_v.stride(0) % 8 == 0 and x.stride(1) == 1 and x.stride(0) % 8 == 0:
            # the four logit rows ride in the value projection's weight-gradient pass (one read of x instead of two)
            g4 = g_alpha if H == 4 else torch.cat([g_alpha, g_alpha.new_zeros(g_alpha.shape[0], 4 - H)], dim=1)
            defer = ctx.params if (_Deferred.active and ctx.needs_input_grad[1] and (not has_bv or ctx.needs_input_grad[2])) else None
            gwv, gbv, gwa4, gba4 = wgrad_bf16_ex2(g_v, x, g4=g4, want_bias=has_bv, defer_to=defer)
            gwa, gba = gwa4[:H], (gba4[:H] if has_ba else None)
            return gx, gwv, gbv, gwa, gba
        if need_v:
            gwv, gbv = wgrad(g_v, x, want_bias=has_bv)
        if need_a:
            ga16 = g_alpha.to(torch.bfloat16)
            if wgrad_supported(ga16, x):
                gwa, gba = wgrad(ga16, x, want_bias=has_ba)
            else:
                gwa, gba = ga16.t() @ x, (ga16.sum(0) if has_ba else None)
        return gx, gwv, gbv, gwa, gba


def pma_project_bf16(x: Tensor, w_v: Tensor, b_v: Optional[Tensor], w_a: Tensor, b_a: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    return _PmaProjectBf16.apply(x, w_v, b_v, w_a, b_a)


class _PmaResidualFFBf16(torch.autograd.Function):
    """:class:`_PmaResidualFF` in the bf16 regime: ``y = dropout_p(relu_post(LN(out + relu(W2 relu(W1 out + b1) + b2))))``;
    both relus are Linear epilogues that also write their masks as one bit per element; the backward-data and weight-gradient
    kernels apply those bits to the incoming gradient as they stage it, and the two gradient branches of ``out`` are summed in
    the last backward-data kernel."""

    @staticmethod
    def forward(ctx, out, w1, b1, w2, b2, gamma, beta, eps, relu_post, p):
        use_bits = (any(ctx.needs_input_grad[:5]) and wgrad_bf16_ex2_supported(w1.shape[0], w1.shape[1], True, False)
                    and wgrad_bf16_ex2_supported(w2.shape[0], w2.shape[1], True, False))
        if use_bits:
            h, mh = linear_bf16_fwd_mask(out, w1, b1)
            z, mz = linear_bf16_fwd_mask(h, w2, b2)
        else:
            h, _ = linear_bf16_fwd(out, w1, b1, True)
            z, _ = linear_bf16_fwd(h, w2, b2, True)
            mh = mz = None
        seed = _draw_seed() if p > 0.0 else 0
        base = _seed_base() if p > 0.0 else None
        y, stats = ln_res_fwd(out, None, z, gamma, beta, eps, relu_post, p, seed, base)
        ctx.save_for_backward(out, h, z, stats, w1, w2, gamma, beta, mh, mz)
        ctx.cfg = (bool(relu_post), float(p), seed, base, b1 is not None, b2 is not None)
        ctx.params = (w1, b1, w2, b2, gamma, beta)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        out, h, z, stats, w1, w2, gamma, beta, mh, mz = ctx.saved_tensors
        relu_post, p, seed, base, has_b1, has_b2 = ctx.cfg
        need = ctx.needs_input_grad
        dfr = _Deferred.active and all(need[1:7][i] for i, q in enumerate(ctx.params) if q is not None)
        p_w1, p_b1, p_w2, p_b2, p_g, p_b = ctx.params
        gs, dg, db, _ = ln_res_bwd(gy.contiguous(), out, None, z, stats, gamma, beta, relu_post, p, seed, base)     # (2048 partial rows: its own reduction)
        if mh is not None:
            gh = linear_bf16_bwd_bits(gs, w2, mz)
            gw2, gb2 = wgrad_bf16_ex2(gs, h, bits=mz, want_bias=has_b2, defer_to=(p_w2, p_b2) if dfr else None)
            gout = linear_bf16_bwd_bits(gh, w1, mh, acc_in=gs)                  # gs + the rFF branch
            gw1, gb1 = wgrad_bf16_ex2(gh, out, bits=mh, want_bias=has_b1, defer_to=(p_w1, p_b1) if dfr else None)
            return gout, gw1, gb1, gw2, gb2, dg, db, None, None, None
        gh, ga2 = linear_bf16_bwd(gs, w2, z, want_ga=True)
        gw2, gb2 = wgrad(ga2, h, want_bias=has_b2)
        gout, ga1 = linear_bf16_bwd(gh, w1, h, want_ga=True, acc_in=gs)         # gs + the rFF branch
        gw1, gb1 = wgrad(ga1, out, want_bias=has_b1)
        return gout, gw1, gb1, gw2, gb2, dg, db, None, None, None


def pma_residual_ff_bf16(out: Tensor, w1, b1, w2, b2, gamma, beta, eps: float = 1e-5, relu_post: bool = False, p: float = 0.0) -> Tensor:
    return _PmaResidualFFBf16.apply(out, w1, b1, w2, b2, gamma, beta, float(eps), bool(relu_post), float(p))


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5, relu_in: bool = False, p: float = 0.0) -> Tensor:
    """``dropout_p(LayerNorm(relu(x) if relu_in else x))`` in one pass."""
    return _LayerNormFused.apply(x, gamma, beta, float(eps), bool(relu_in), float(p))


def relu_dropout(x: Tensor, p: float = 0.0) -> Tensor:
    """``dropout_p(relu(x))`` in one pass."""
    return _ReluDropout.apply(x, float(p))


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    return _Linear.apply(x, weight, bias)


def linear_narrow_supported(gy: Tensor, x: Tensor, weight: Tensor) -> bool:
    return (gy.is_cuda and gy.dtype == x.dtype == weight.dtype == torch.float32 and gy.dim() == 2 and x.dim() == 2
            and bool(_lib.load().allset_linear_narrow_supported(weight.shape[0], weight.shape[1]))
            and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)


def linear_narrow_bwd(gy: Tensor, x: Tensor, weight: Tensor, want_gx: bool = True, defer_to=None):
    """``(gx, gW, gb)`` of ``y = x W^T + b`` with at most 16 outputs (a classifier head) from one kernel (csrc/narrow_linear.hip).
    ``defer_to`` = the (weight, bias) PARAMETERS: inside ``deferred_param_grads()`` gW / gb are queued and come back as None."""
    dev = require_device(gy, x, weight)
    _check_f32(gy, x, weight)
    gy, x, weight = _rowmajor(gy), _rowmajor(x), weight.contiguous()
    n, N = gy.shape
    K = x.shape[1]
    lib = _lib.load()
    ns = c_int64(0)
    check(lib.allset_linear_narrow_slices(n, byref(ns)), "allset_linear_narrow_slices")
    M = (N * K + N + 3) // 4 * 4
    part = torch.empty((ns.value, M), dtype=torch.float32, device=dev)
    gx = torch.empty((n, K), dtype=torch.float32, device=dev) if want_gx else None
    with on_device(dev):
        check(lib.allset_linear_narrow_bwd(ptr(gy), _ld(gy), ptr(x), _ld(x), ptr(weight), n, N, K, ptr(gx), K, ptr(part), M, ns.value,
                                           stream_of(dev)), "allset_linear_narrow_bwd")
    if defer_to is not None and _deferrable(part, *defer_to):
        _defer(part, [(defer_to[0], 0, (N, K)), (defer_to[1], N * K, (N,))])
        return gx, None, None
    red = reduce_partials(part)
    return gx, red[:N * K].view(N, K), red[N * K:N * K + N]


# ---- the FIRST Linear of a model: [dropout ->] LayerNorm(raw features) -> Linear on an input without gradient (csrc/input_linear.hip) ----
def input_norm_linear_supported(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], weight: Tensor, bias: Optional[Tensor]) -> bool:
    """``x`` is a device fp32 matrix that needs no gradient (raw features), the LayerNorm has both affine parameters, width
    <= 4096.  Widths the fused kernels take (64 / 128) keep those."""
    if gamma is None or beta is None or x.dim() != 2 or not x.is_cuda or x.dtype != torch.float32:
        return False
    if x.requires_grad and torch.is_grad_enabled():
        return False
    if any(t is not None and (t.dtype != torch.float32 or not t.is_cuda) for t in (gamma, beta, weight, bias)):
        return False
    return bool(_lib.load().allset_input_linear_supported(x.shape[1])) and weight.shape[1] == x.shape[1]


def xhat_rows(x: Tensor, eps: float, p_pre: float, seed: int) -> Tensor:
    """``[LayerNorm_noaffine(dropout_p(x)) | 1 | 0...]``  [n, K], K = d + 1 rounded up to 16."""
    require_device(x); _check_f32(x)
    x = _rowmajor(x)
    n, d = x.shape
    K = int(_lib.load().allset_input_linear_k(d))
    xh = torch.empty((n, K), dtype=torch.float32, device=x.device)
    base = _seed_base()
    with on_device(x.device):
        check(_lib.load().allset_xhat_rows(ptr(x), _ld(x), n, d, float(eps), float(p_pre), seed, ptr(base) if base is not None else None,
                                           ptr(xh), K, stream_of(x.device)), "allset_xhat_rows")
    return xh


def fold_ln_linear(weight: Tensor, gamma: Tensor, beta: Tensor, bias: Optional[Tensor]) -> Tensor:
    """``[W * gamma | b + W beta | 0...]``  [O, K]."""
    require_device(weight); _check_f32(weight, gamma, beta, bias)
    weight, gamma, beta = _rowmajor(weight), gamma.contiguous(), beta.contiguous()
    O, d = weight.shape
    K = int(_lib.load().allset_input_linear_k(d))
    wp = torch.empty((O, K), dtype=torch.float32, device=weight.device)
    with on_device(weight.device):
        check(_lib.load().allset_fold_ln_linear(ptr(weight), _ld(weight), ptr(gamma), ptr(beta), ptr(bias.contiguous()) if bias is not None else None,
                                                O, d, ptr(wp), K, stream_of(weight.device)), "allset_fold_ln_linear")
    return wp


def unfold_ln_linear(M: Tensor, weight: Tensor, gamma: Tensor, beta: Tensor, want_bias: bool) -> Tuple[Tensor, Optional[Tensor], Tensor, Tensor]:
    """``M = gy^T [x_hat | 1]`` -> ``(gW, gb, ggamma, gbeta)``."""
    require_device(M); _check_f32(M, weight, gamma, beta)
    M, weight = _rowmajor(M), _rowmajor(weight)
    O, d = weight.shape
    out = torch.empty(O * d + O + 2 * d, dtype=torch.float32, device=M.device)       # one allocation: gW | gb | ggamma | gbeta
    gw, gb = out[:O * d].view(O, d), out[O * d:O * d + O]
    gg, gbt = out[O * d + O:O * d + O + d], out[O * d + O + d:]
    with on_device(M.device):
        check(_lib.load().allset_unfold_ln_linear(ptr(M), _ld(M), ptr(weight), _ld(weight), ptr(gamma.contiguous()), ptr(beta.contiguous()),
                                                  O, d, ptr(gw), d, ptr(gb) if want_bias else None, ptr(gg), ptr(gbt),
                                                  stream_of(M.device)), "allset_unfold_ln_linear")
    return gw, (gb if want_bias else None), gg, gbt


class _InputNormLinear(torch.autograd.Function):
    """``Linear(LayerNorm(dropout_p(x)))`` for an ``x`` that needs no gradient: x_hat (LayerNorm without its affine part, a ones
    column behind it) is written once and kept; forward = one GEMM against the folded weight ``[W * gamma | b + W beta]``; the
    backward is ONE GEMM ``M = gy^T [x_hat | 1]`` and a [O, d]-sized kernel that unfolds it into all four parameter gradients
    (csrc/input_linear.hip has the algebra).  Reference: models.py:473-476 + layers.py:571-573 (``x = self.normalizations[0](x)``
    ... ``self.lins[0](x)``) -- there a dropout, a LayerNorm, a Linear and, in the backward, the [n, d] input gradient of the
    Linear + the LayerNorm backward over it."""

    @staticmethod
    def forward(ctx, x, gamma, beta, weight, bias, eps, p_pre):
        seed = _draw_seed() if p_pre > 0.0 else 0
        xh = xhat_rows(x, eps, p_pre, seed)
        wp = fold_ln_linear(weight, gamma, beta, bias)
        ctx.save_for_backward(xh, gamma, beta, weight)
        ctx.has_bias = bias is not None
        return xh @ wp.t()

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xh, gamma, beta, weight = ctx.saved_tensors
        M = gy.contiguous().t() @ xh
        gw, gb, gg, gbt = unfold_ln_linear(M, weight, gamma, beta, ctx.has_bias)
        need = ctx.needs_input_grad
        return None, gg if need[1] else None, gbt if need[2] else None, gw if need[3] else None, gb if (ctx.has_bias and need[4]) else None, None, None


# ---- ... and on SPARSE raw features (csrc/sparse_input.hip) -----------------------------------------------------------------------------
class SparseRows:
    """CSR + CSC of a constant feature matrix (bag-of-words rows), built once per tensor: ``rowptr / col / val`` in row-major order,
    ``colptr / rowT / posT`` (row id and CSR position of each entry) in column-major order; int32 indices."""

    def __init__(self, x: Tensor):
        n, d = x.shape
        idx = (x != 0).nonzero()                                   # row-major order; one host sync (the count)
        rows, cols = idx[:, 0], idx[:, 1]
        self.n, self.d, self.nnz = n, d, int(idx.shape[0])
        self.val = x[rows, cols].contiguous()
        self.col = cols.to(torch.int32)
        self.rowptr = torch.zeros(n + 1, dtype=torch.int32, device=x.device)
        self.rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0).to(torch.int32)
        order = torch.argsort(cols, stable=True)
        self.rowT = rows[order].to(torch.int32)
        self.posT = order.to(torch.int32)
        self.colptr = torch.zeros(d + 1, dtype=torch.int32, device=x.device)
        self.colptr[1:] = torch.cumsum(torch.bincount(cols, minlength=d), 0).to(torch.int32)


_SPARSE_ROWS = {}           # data_ptr -> (weakref to the tensor, version, shape, SparseRows or None)
SPARSE_MAX_DENSITY = 0.10   # above this fraction of non-zeros the GEMM path wins


def sparse_rows(x: Tensor) -> Optional[SparseRows]:
    """The cached :class:`SparseRows` of ``x`` when at most 10 % of it is non-zero, else None.  Built on first sight of the tensor
    (one host read-back; keyed on identity and version, so an in-place update rebuilds it) -- never during graph capture: a tensor
    first seen there takes the dense path."""
    import weakref
    key = x.data_ptr()
    hit = _SPARSE_ROWS.get(key)
    if hit is not None:
        ref, ver, shape, sp = hit
        if ref() is x and ver == x._version and shape == tuple(x.shape):
            return sp
    if torch.cuda.is_current_stream_capturing():
        return None
    if len(_SPARSE_ROWS) > 64:
        for k in [k for k, v in _SPARSE_ROWS.items() if v[0]() is None]:
            del _SPARSE_ROWS[k]
    nnz = int(torch.count_nonzero(x))
    sp = SparseRows(x) if (x.numel() > 0 and nnz <= SPARSE_MAX_DENSITY * x.numel()) else None
    _SPARSE_ROWS[key] = (weakref.ref(x), x._version, tuple(x.shape), sp)
    return sp


class _SparseInputNormLinear(torch.autograd.Function):
    """:class:`_InputNormLinear` from the non-zeros of ``x`` (csrc/sparse_input.hip): forward = the transposed folded weight (one
    kernel) + one gather-sum kernel; backward = one gather-sum kernel over the CSC + the unfold kernel.  No [n, d] tensor at all."""

    @staticmethod
    def forward(ctx, x, sp, gamma, beta, weight, bias, eps, p_pre):
        lib = _lib.load()
        dev = x.device
        O, d = weight.shape
        n = x.shape[0]
        seed = _draw_seed() if p_pre > 0.0 else 0
        base = _seed_base() if p_pre > 0.0 else None
        weight_c, gamma_c, beta_c = _rowmajor(weight), gamma.contiguous(), beta.contiguous()
        wt = torch.empty((d + 2, O), dtype=torch.float32, device=dev)
        y = torch.empty((n, O), dtype=torch.float32, device=dev)
        w = torch.empty(max(sp.nnz, 1), dtype=torch.float32, device=dev)
        rm = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
        with on_device(dev):
            check(lib.allset_fold_ln_linear_t(ptr(weight_c), _ld(weight_c), ptr(gamma_c), ptr(beta_c), ptr(bias.contiguous()) if bias is not None else None,
                                              O, d, ptr(wt), stream_of(dev)), "allset_fold_ln_linear_t")
            check(lib.allset_sparse_ln_linear_fwd(ptr(sp.rowptr), ptr(sp.col), ptr(sp.val), n, d, ptr(wt), O, float(eps), float(p_pre), seed,
                                                  ptr(base), ptr(y), O, ptr(w), ptr(rm), stream_of(dev)), "allset_sparse_ln_linear_fwd")
        ctx.save_for_backward(w, rm, gamma, beta, weight)
        ctx.sp = sp
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        w, rm, gamma, beta, weight = ctx.saved_tensors
        sp = ctx.sp
        lib = _lib.load()
        dev = gy.device
        gy = _rowmajor(gy.contiguous())
        O, d = weight.shape
        n = gy.shape[0]
        slices = int(lib.allset_sparse_ln_linear_slices())
        M = torch.empty((O, d), dtype=torch.float32, device=dev)
        su = torch.empty((slices, 2, O), dtype=torch.float32, device=dev)
        out = torch.empty(O * d + O + 2 * d, dtype=torch.float32, device=dev)       # gW | gb | ggamma | gbeta
        gw, gb = out[:O * d].view(O, d), out[O * d:O * d + O]
        gg, gbt = out[O * d + O:O * d + O + d], out[O * d + O + d:]
        weight_c = _rowmajor(weight)
        with on_device(dev):
            check(lib.allset_sparse_ln_linear_bwd(ptr(sp.colptr), ptr(sp.rowT), ptr(sp.posT), ptr(w), ptr(rm), ptr(gy), _ld(gy), n, d, O,
                                                  ptr(M), d, ptr(su), stream_of(dev)), "allset_sparse_ln_linear_bwd")
            check(lib.allset_unfold_ln_linear_ex(ptr(M), d, ptr(weight_c), _ld(weight_c), ptr(gamma.contiguous()), ptr(beta.contiguous()), O, d,
                                                 ptr(gw), d, ptr(gb) if ctx.has_bias else None, ptr(gg), ptr(gbt), ptr(su), slices,
                                                 stream_of(dev)), "allset_unfold_ln_linear_ex")
        need = ctx.needs_input_grad
        return (None, None, gg if need[2] else None, gbt if need[3] else None, gw if need[4] else None,
                gb if (ctx.has_bias and need[5]) else None, None, None)


_STREAM_TICKETS = {}      # (device index, stream handle) -> zeroed uint32 a kernel counts its workgroups on and re-arms (losses.py has the twin)


def _stream_ticket(dev: torch.device) -> Tensor:
    key = (dev.index, int(torch.cuda.current_stream(dev).cuda_stream))
    t = _STREAM_TICKETS.get(key)
    if t is None:
        t = _STREAM_TICKETS[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t


class _SparsePmaProject(torch.autograd.Function):
    """``(x_V, alpha) = (dropout_p(x) W_V^T + b_V,  dropout_p(x) w_a^T + b_a)`` from the non-zeros of raw features ``x`` (no input
    gradient): PMA's value projection and its folded logits on the first conv of an AllSetTransformer (reference models.py:473,
    layers.py:126-131) -- three kernels forward (transposed stacked weight, gather-sum; the dropout is a hash site of the latter), one
    gather-sum over the CSC backward + the bias sums.  csrc/sparse_input.hip ``sparse_lin_*``."""

    @staticmethod
    def forward(ctx, x, sp, w_v, b_v, w_a, b_a, p_pre):
        lib = _lib.load()
        dev = x.device
        O1, d = w_v.shape
        H = w_a.shape[0]
        n = x.shape[0]
        seed = _draw_seed() if p_pre > 0.0 else 0
        base = _seed_base() if p_pre > 0.0 else None
        w_v_c, w_a_c = _rowmajor(w_v), _rowmajor(w_a)
        pitch = int(lib.allset_sparse_linear_pitch(O1, H))
        wt = torch.empty((d + 1, pitch), dtype=torch.float32, device=dev)
        y = torch.empty((n, O1), dtype=torch.float32, device=dev)
        y2 = torch.empty((n, pitch - O1), dtype=torch.float32, device=dev)            # (the auxiliary columns in whole 16-byte chunks)
        w = torch.empty(max(sp.nnz, 1), dtype=torch.float32, device=dev)
        with on_device(dev):
            check(lib.allset_sparse_linear_wt(ptr(w_v_c), _ld(w_v_c), O1, ptr(w_a_c), _ld(w_a_c), H, ptr(b_v.contiguous()) if b_v is not None else None,
                                              ptr(b_a.contiguous()) if b_a is not None else None, d, ptr(wt), stream_of(dev)), "allset_sparse_linear_wt")
            check(lib.allset_sparse_linear_fwd(ptr(sp.rowptr), ptr(sp.col), ptr(sp.val), n, d, ptr(wt), O1, H, float(p_pre), seed, ptr(base),
                                               ptr(y), O1, ptr(y2), ptr(w), stream_of(dev)), "allset_sparse_linear_fwd")
        ctx.save_for_backward(w)
        ctx.sp = sp
        ctx.cfg = (O1, H, d, b_v is not None, b_a is not None)
        return y, (y2 if H == y2.shape[1] else y2[:, :H].contiguous())

    @staticmethod
    @once_differentiable
    def backward(ctx, g_v, g_alpha):
        (w,) = ctx.saved_tensors
        sp = ctx.sp
        O1, H, d, has_bv, has_ba = ctx.cfg
        lib = _lib.load()
        dev = g_v.device
        g_v = _rowmajor(g_v.contiguous())
        n = g_v.shape[0]
        pitch = int(lib.allset_sparse_linear_pitch(O1, H))
        g4 = g_alpha.contiguous() if H == pitch - O1 else torch.cat([g_alpha, g_alpha.new_zeros(n, pitch - O1 - H)], dim=1)
        slices = int(lib.allset_sparse_ln_linear_slices())
        gw = torch.empty((O1 + H, d), dtype=torch.float32, device=dev)               # gW_V | gw_a: one allocation
        sb = torch.empty((slices + 1, pitch), dtype=torch.float32, device=dev)         # the slices | their total
        with on_device(dev):
            check(lib.allset_sparse_linear_bwd(ptr(sp.colptr), ptr(sp.rowT), ptr(sp.posT), ptr(w), ptr(g_v), _ld(g_v), ptr(g4), n, d, O1, H,
                                               ptr(gw), d, ptr(gw[O1:]), d, ptr(sb), ptr(_stream_ticket(dev)), ptr(sb[slices]), stream_of(dev)),
                  "allset_sparse_linear_bwd")
        need = ctx.needs_input_grad
        gb = sb[slices]                                # (summed by the last slice workgroup of the launch itself)
        return (None, None, gw[:O1] if need[2] else None, gb[:O1] if (has_bv and need[3]) else None, gw[O1:] if need[4] else None,
                gb[O1:O1 + H] if (has_ba and need[5]) else None, None)


def sparse_linear_supported(O1: int, H: int) -> bool:
    return bool(_lib.load().allset_sparse_linear_supported(int(O1), int(H)))


def sparse_pma_project(x: Tensor, sp: "SparseRows", w_v: Tensor, b_v: Optional[Tensor], w_a: Tensor, b_a: Optional[Tensor],
                       p_pre: float = 0.0) -> Tuple[Tensor, Tensor]:
    return _SparsePmaProject.apply(x, sp, w_v, b_v, w_a, b_a, float(p_pre))


class _ConstFeatures:
    depth = 0


@contextlib.contextmanager
def constant_features():
    """Inside, NO-GRAD forwards may also read raw features through their cached non-zero structure (:func:`sparse_rows`): the caller
    promises that ``data.x`` is not overwritten in place behind a captured graph (a training loop's evaluation of the same features:
    ``allset_amd/train.py``; ``graphs.GraphedForward(..., constant_features=True)``)."""
    _ConstFeatures.depth += 1
    try:
        yield
    finally:
        _ConstFeatures.depth -= 1


def constant_features_active() -> bool:
    return _ConstFeatures.depth > 0


def input_norm_linear(x: Tensor, gamma: Tensor, beta: Tensor, weight: Tensor, bias: Optional[Tensor], eps: float = 1e-5,
                      p_pre: float = 0.0) -> Tensor:
    # (NO-GRAD forwards keep the dense kernels: graphs.GraphedForward lets its caller overwrite the features in place between
    #  replays, a replay cannot rebuild the non-zero structure, and eager and replayed inference stay bit-identical this way; a
    #  training step's features are constants -- and so are an evaluation's inside ``constant_features()``)
    replaceable = not torch.is_grad_enabled() and not constant_features_active()
    if x.shape[1] >= 256 and not replaceable and _lib.load().allset_sparse_ln_linear_supported(weight.shape[0]):
        sp = sparse_rows(x)
        if sp is not None:
            return _SparseInputNormLinear.apply(x, sp, gamma, beta, weight, bias, float(eps), float(p_pre))
    return _InputNormLinear.apply(x, gamma, beta, weight, bias, float(eps), float(p_pre))


# ---- BatchNorm over a row-sharded batch (reference layers.py:499-562: MLP's default Normalization='bn') ----------------------------
# A sharded layer holds a block of the batch's rows per rank (plus zero pad rows).  torch's BatchNorm1d would take its statistics
# over that block alone -- a different model than the single-GPU one, and replicated running statistics that drift apart.  Inside
# `sync_bn_rows(valid_rows, group)` the BatchNorm1d modules of an MLP compute (count, sum) and then the centred sum of squares over
# the VALID rows of ALL ranks (two [d]-sized all-reduces: the two-pass form, no cancellation), normalise every local row with them
# and update the running statistics exactly as torch does (biased variance to normalise, unbiased into running_var).
import contextlib
import threading

_sync_bn_state = threading.local()


@contextlib.contextmanager
def sync_bn_rows(valid_rows: int, group=None):
    """BatchNorm1d modules called inside see a batch = the first ``valid_rows`` local rows of every rank of ``group``."""
    prev = getattr(_sync_bn_state, "scope", None)
    _sync_bn_state.scope = (int(valid_rows), group)
    try:
        yield
    finally:
        _sync_bn_state.scope = prev


def _host_all_reduce_sum(t: Tensor, group) -> Tensor:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":          # (tests: two ranks on one GPU; a [d]-sized message)
        h = t.cpu()
        dist.all_reduce(h, group=group)
        return h.to(t.device)
    dist.all_reduce(t, group=group)
    return t


class _SyncBatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, valid, group, eps):
        xs = x[:valid].float()
        d = x.shape[1]
        head = torch.cat([xs.sum(0), xs.new_full((1,), float(valid))])
        head = _host_all_reduce_sum(head, group)
        count = head[d].clamp(min=1.0)
        mean = head[:d] / count
        m2 = _host_all_reduce_sum(((xs - mean) ** 2).sum(0), group)
        var = m2 / count                                             # biased: what normalises (torch.nn.BatchNorm1d)
        rstd = torch.rsqrt(var + eps)
        xhat = (x.float() - mean) * rstd
        y = xhat * weight.float() + bias.float() if weight is not None else xhat
        ctx.save_for_backward(xhat, weight, rstd, count)
        ctx.cfg = (int(valid), group, bias is not None)
        ctx.mark_non_differentiable(mean, var, count)
        return y.to(x.dtype), mean, var, count

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gm, _gv, _gc):
        xhat, weight, rstd, count = ctx.saved_tensors
        valid, group, has_bias = ctx.cfg
        g = gy.float()
        gs, xs = g[:valid], xhat[:valid]
        loc = torch.cat([gs.sum(0), (gs * xs).sum(0)])               # this rank's share of (dbeta, dgamma)
        d = g.shape[1]
        glob = _host_all_reduce_sum(loc.clone(), group)
        w = weight.float() if weight is not None else torch.ones_like(rstd)
        gx = torch.zeros_like(g)
        gx[:valid] = (gs - glob[:d] / count - xs * (glob[d:] / count)) * (w * rstd)
        if valid < g.shape[0]:                                       # pad rows: constants of the batch as far as they are concerned
            gx[valid:] = g[valid:] * (w * rstd)
        # parameter gradients stay LOCAL partial sums: the caller's gradient all-reduce (dist.allreduce_grads) completes them
        gw = loc[d:].to(weight.dtype) if weight is not None else None
        gb = loc[:d].to(weight.dtype) if (weight is not None and has_bias) else None
        return gx.to(gy.dtype), gw, gb, None, None, None


# ---- training-mode BatchNorm1d in front of a fused Linear (csrc/batchnorm.hip) ----------------------------------------------------

def col_moments_supported(d: int) -> bool:
    return bool(_lib.load().allset_col_moments_supported(int(d)))


def col_moments(x: Tensor, relu_in: bool = False, center: Optional[Tensor] = None) -> Tensor:
    """[d] column sums of ``f(x)`` (``center`` None) or of ``(f(x) - center)^2``; f = relu if ``relu_in``."""
    dev = require_device(x, center)
    _check_f32(x, center)
    x = _rowmajor(x)
    n, d = x.shape
    lib = _lib.load()
    ns = c_int64(0)
    check(lib.allset_col_moments_slices(n, byref(ns)), "allset_col_moments_slices")
    part = torch.empty((ns.value, d), dtype=torch.float32, device=dev)
    with on_device(dev), _timed("col_moments", dev, n * d * 4):
        check(lib.allset_col_moments(ptr(x), _ld(x), n, d, int(relu_in), ptr(center.contiguous() if center is not None else None),
                                     ptr(part), ns.value, stream_of(dev)), "allset_col_moments")
    return reduce_partials(part) if ns.value > 1 else part[0]


def col_mean_var(x: Tensor, relu_in: bool = False) -> Tuple[Tensor, Tensor]:
    """(mean, biased variance) per column of ``f(x)`` from ONE read: fp64 sums of f and f^2 (``allset_col_moments2``), combined in
    fp64, returned as fp32."""
    dev = require_device(x)
    _check_f32(x)
    x = _rowmajor(x)
    n, d = x.shape
    lib = _lib.load()
    ns = c_int64(0)
    check(lib.allset_col_moments_slices(n, byref(ns)), "allset_col_moments_slices")
    part = torch.empty((ns.value, 2, d), dtype=torch.float64, device=dev)
    with on_device(dev), _timed("col_moments", dev, n * d * 4):
        check(lib.allset_col_moments2(ptr(x), _ld(x), n, d, int(relu_in), ptr(part), ns.value, stream_of(dev)), "allset_col_moments2")
    tot = part.sum(0) if ns.value > 1 else part[0]
    mean = tot[0] / n
    var = (tot[1] / n - mean * mean).clamp_(min=0.0)
    return mean.float(), var.float()


def col_affine_add_(gx: Tensor, x: Tensor, s: Tensor, t: Tensor, relu_mask: bool) -> Tensor:
    """``gx += [x > 0 if relu_mask] * (f(x) * s + t)`` in place (the gradient of the batch statistics)."""
    dev = require_device(gx, x, s, t)
    _check_f32(gx, x, s, t)
    n, d = x.shape
    with on_device(dev), _timed("col_affine_add", dev, 3 * n * d * 4):
        check(_lib.load().allset_col_affine_add(ptr(gx), _ld(gx), ptr(x), _ld(x), ptr(s.contiguous()), ptr(t.contiguous()),
                                                int(relu_mask), n, d, stream_of(dev)), "allset_col_affine_add")
    return gx


def bn_linear_supported(bn, lin, x: Optional[Tensor] = None) -> bool:
    """The HIP BatchNorm -> [dropout] -> Linear path takes this pair: affine fp32 BatchNorm1d of the Linear's input width, a width
    the fused Linear kernels and the one-pass backward are built for; ``x`` (optional): also the operand itself (device fp32
    16-byte aligned rows, at least two of them)."""
    ok = (bn.affine and bn.weight is not None and bn.weight.dtype == torch.float32 and lin.weight.dtype == torch.float32 and
          lin.bias is not None and lin.in_features == bn.num_features and
          fused_linear_supported(lin.in_features, lin.out_features) and
          bool(_lib.load().allset_fused_linear_bwd_all_supported(lin.out_features, lin.in_features, 1, 1, 1, 1, 0)) and
          col_moments_supported(lin.in_features))
    if ok and x is not None:
        ok = (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 2 and x.shape[1] == lin.in_features and
              x.data_ptr() % 16 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0)
    return ok


class _BatchNormLinear(torch.autograd.Function):
    """``y = epi( dropout_p_in( BatchNorm_train( f(x) ) ) @ W^T + b )``, f = relu if ``relu_in`` -- reference MLP.forward with
    ``Normalization='bn'`` (layers.py:571-579: ``x = lin(dropout(bn(relu(x))))``), batch statistics.

    Forward: two column reductions (mean, centred biased variance), then the fused Linear with the per-column affine prologue
    ``f(x) * a + b`` (a = gamma * rstd, b = beta - mean * a): no normalised tensor is written.  Backward: the one-pass kernel gives
    the direct input gradient, the weight / bias gradient and the column sums G_a = sum gu * f(x), G_b = sum gu; then
        dgamma = rstd * (G_a - mean * G_b),  dbeta = G_b,
        dmean = -a * G_b,  dvar = -1/2 * rstd^3 * gamma * (G_a - mean * G_b),
        gx += mask * (f(x) * s + t),  s = 2 dvar / n,  t = dmean / n - s * mean          (one in-place pass)
    Returns (y, mean, biased var) -- the caller updates the running statistics."""

    @staticmethod
    def forward(ctx, x, bn_weight, bn_bias, weight, bias, eps, relu_in, p_in, relu_out, p_out):
        n, d = x.shape
        mean, var = col_mean_var(x, relu_in)
        rstd = torch.rsqrt(var + eps)
        a = bn_weight.float() * rstd
        b = bn_bias.float() - mean * a
        seed_in = _draw_seed() if p_in > 0.0 else 0
        seed_out = _draw_seed() if p_out > 0.0 else 0
        base = _seed_base() if (p_in > 0.0 or p_out > 0.0) else None
        keep_y = relu_out or p_out > 0.0
        words = activation_mask_words(n, weight.shape[0]) if (keep_y and any(ctx.needs_input_grad)) else 0
        mask = torch.empty(words, dtype=torch.int32, device=x.device) if words > 0 else None
        y, stats = fused_linear_fwd(x, weight, bias, a, b, 1.0, relu_in, p_in, seed_in, relu_out, p_out, seed_out, base, mask,
                                    norm_mode=1)
        ctx.save_for_backward(x, stats, a, b, weight, mask, mean, rstd, bn_weight)
        ctx.cfg = (bool(relu_in), float(p_in), seed_in, float(p_out), base, keep_y)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gm, _gv):
        x, stats, a, b, weight, mask, mean, rstd, bn_weight = ctx.saved_tensors
        relu_in, p_in, seed_in, p_out, base, keep_y = ctx.cfg
        if keep_y and mask is None:
            raise _lib.AllSetHipError("_BatchNormLinear: the forward ran without gradients enabled (no activation mask was kept)")
        n = x.shape[0]
        gx, g_a, g_b, gw, gb = fused_linear_bwd_all(gy.contiguous(), mask, p_out, weight, x, stats, a, b, relu_in, p_in, seed_in, base,
                                                    want_bias=True, norm_mode=1)
        core = g_a - mean * g_b                                    # sum_r gu * (f(x) - mean)
        dgamma = rstd * core
        dvar = -0.5 * rstd * rstd * rstd * bn_weight.float() * core
        dmean = -a * g_b
        s = dvar * (2.0 / n)
        t = dmean / n - s * mean
        if ctx.needs_input_grad[0]:
            col_affine_add_(gx, x, s, t, relu_in)
        else:
            gx = None
        return gx, dgamma.to(bn_weight.dtype), g_b.to(bn_weight.dtype), gw, gb, None, None, None, None, None


def bn_linear(bn, lin, x: Tensor, relu_in: bool, p_in: float, relu_out: bool = False, p_out: float = 0.0) -> Tensor:
    """``lin(dropout(bn(relu?(x))))`` (+ relu / dropout epilogue) in training mode on the HIP kernels, with torch's bookkeeping of
    the running statistics (biased variance normalises, unbiased goes into ``running_var``)."""
    y, mean, var = _BatchNormLinear.apply(x, bn.weight, bn.bias, lin.weight, lin.bias, float(bn.eps), bool(relu_in), float(p_in),
                                          bool(relu_out), float(p_out))
    if bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            n = x.shape[0]
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1.0 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
            bn.running_var.mul_(1.0 - mom).add_((var * (n / max(n - 1.0, 1.0))).to(bn.running_var.dtype), alpha=mom)
    return y


def batch_norm(bn: torch.nn.modules.batchnorm._BatchNorm, x: Tensor) -> Tensor:
    """``bn(x)``; inside :func:`sync_bn_rows` with batch statistics in use, the cross-rank form above."""
    scope = getattr(_sync_bn_state, "scope", None)
    use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
    if scope is None or not use_batch:
        return bn(x)                      # eval with running statistics is row-wise: nothing to synchronise
    valid, group = scope
    valid = max(0, min(valid, x.shape[0]))
    y, mean, var, count = _SyncBatchNorm.apply(x, bn.weight, bn.bias, valid, group, float(bn.eps))
    if bn.training and bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            unbiased = var * (count / (count - 1.0).clamp(min=1.0))
            bn.running_mean.mul_(1.0 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
            bn.running_var.mul_(1.0 - mom).add_(unbiased.to(bn.running_var.dtype), alpha=mom)
    return y
