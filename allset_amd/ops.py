"""Thin tensor-level wrappers over the C ABI (one python function per entry point of
``include/allset_hip.h``).  No autograd here -- see ``functional.py``.  All tensors are ROCm device
tensors; outputs are allocated with torch (plumbing) and filled by the HIP kernels.
"""
from __future__ import annotations

from collections import defaultdict
from ctypes import byref, c_size_t
from typing import Dict, List, NamedTuple, Optional, Tuple

import torch

from . import _lib
from ._lib import check, ptr, require_device, stream_of, on_device

Tensor = torch.Tensor


class KernelTimer:
    """Opt-in per-entry-point timing with HIP events recorded on the stream the kernels are launched on
    (torch's current stream).  ``bench.py`` installs one over its timed region; when none is installed
    the wrappers below add no events.  ``algo_bytes`` is the ALGORITHMIC traffic of the call (SURVEY.md
    section 8(d3) gather model: every incidence reads its d-vector once; int32 CSR)."""

    def __init__(self):
        self.events: Dict[str, List[Tuple[torch.cuda.Event, torch.cuda.Event, int]]] = defaultdict(list)

    def summary(self) -> Dict[str, Dict[str, float]]:
        """Call after a device synchronise.  name -> {calls, avg_ms, total_ms, algo_bytes (per call)}."""
        out = {}
        for name, evs in self.events.items():
            ms = [s.elapsed_time(e) for s, e, _ in evs]
            out[name] = dict(calls=len(ms), total_ms=sum(ms), avg_ms=sum(ms) / len(ms),
                             algo_bytes=sum(b for _, _, b in evs) / len(evs))
        return out


_timer: Optional[KernelTimer] = None


def set_kernel_timer(timer: Optional[KernelTimer]) -> None:
    global _timer
    _timer = timer


class _NoTimer:
    """Shared do-nothing context: the common case (no KernelTimer installed) must cost nothing per launch."""
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_TIMER = _NoTimer()


class _Timed:
    __slots__ = ("t", "name", "st", "s", "bytes")

    def __init__(self, t, name, dev, algo_bytes):
        self.t, self.name, self.bytes = t, name, int(algo_bytes)
        self.st = torch.cuda.current_stream(dev)

    def __enter__(self):
        self.s = torch.cuda.Event(enable_timing=True)
        self.s.record(self.st)
        return None

    def __exit__(self, *exc):
        e = torch.cuda.Event(enable_timing=True)
        e.record(self.st)
        self.t.events[self.name].append((self.s, e, self.bytes))
        return False


def _timed(name: str, dev, algo_bytes: int):
    t = _timer
    return _NO_TIMER if t is None else _Timed(t, name, dev, algo_bytes)


class SizeSplit(NamedTuple):
    """A skewed CSR cut by row length (round 3): the few LONG rows (> ``threshold`` incidences) as a list for
    the one-wave-per-row kernels, the many SHORT rows as a compacted CSR (+ their ids) for the short-row kernels, which pack
    64 / LPR rows into a wave.  Pays where a row is at most one cache line (the column-sharded PMA layer: d / P columns): one wave
    per row leaves most lanes idle there, and the short-row kernels alone would serialise a 4096-member row in one lane group."""
    long_ids: Tensor          # int32[n_long]
    short_ids: Tensor         # int32[n_short]
    rowptr_short: Tensor      # int32[n_short + 1] into col_short
    col_short: Tensor         # int32[nnz_short]
    threshold: int


SIZE_SPLIT_THRESHOLD = int(__import__('os').environ.get('ALLSET_SIZE_SPLIT_T', '32'))      # (env: tuning sweeps only)


def size_split(rowptr: Tensor, col: Tensor, n_rows: int, max_deg: int, threshold: int = SIZE_SPLIT_THRESHOLD) -> Optional[SizeSplit]:
    """Built once per CSR (one host sync), only for skewed row lengths: some row longer than ``threshold`` and at most 1/8 of the
    rows long."""
    if n_rows <= 0 or max_deg <= threshold or col.numel() == 0:
        return None
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    is_long = deg > threshold
    n_long = int(is_long.sum())
    if n_long == 0 or n_long * 8 > n_rows:
        return None
    long_ids = is_long.nonzero().reshape(-1).to(torch.int32)
    short_ids64 = (~is_long).nonzero().reshape(-1)
    deg_s = deg[short_ids64]
    rowptr_short = torch.zeros(short_ids64.numel() + 1, dtype=torch.int32, device=rowptr.device)
    rowptr_short[1:] = torch.cumsum(deg_s, 0).to(torch.int32)
    keep = ~torch.repeat_interleave(is_long, deg)                  # per incidence: its row is short
    return SizeSplit(long_ids, short_ids64.to(torch.int32), rowptr_short, col[keep].contiguous(), threshold)


class CSR(NamedTuple):
    """rowptr int32[n_rows+1], col int32[nnz], perm int32[nnz] (CSR position -> edge-list position);
    ``max_deg`` = longest row (read back once when the CSR is built; used to pick kernel variants)."""
    rowptr: Tensor
    col: Tensor
    perm: Tensor
    n_rows: int
    n_cols: int
    max_deg: int = 0
    row_order: Optional[Tensor] = None      # int32[n_rows] processing order (long rows first per XCD range) or None
    short_tail: int = -1                    # rows >= short_tail all have <= 2 incidences (Add_Self_Loops' singleton
                                            # hyperedges sit at the end of the id range) and are worth their own launch
    sizes: Optional[SizeSplit] = None       # skewed row lengths: long-row list + compacted short rows (narrow-row PMA kernels)

    def variant(self, kind: str, n_rows: Optional[int] = None) -> int:
        """Kernel variant for this orientation: 2 = short-row kernel, 1 = one wavefront per row.  Thresholds from
        profiles/r01_kernel_bench*.txt: the short-row kernels win below a mean degree of ~6 (segreduce, pma_fwd) and
        up to ~24 for pma_bwd_src, as long as no row is long enough to serialise a half-wave."""
        rows = max(int(self.n_rows if n_rows is None else n_rows), 1)
        if kind == "segreduce" and rows <= 16384:
            # dataset scale: every row gets its own lane group on a machine this size; the one-group-per-row kernel is one dependent
            # round trip shorter than a slot walking seven rows as a stream (the same rule as the library's AUTO, kFlatMinRows)
            return 1
        mean = self.col.numel() / rows
        limit = 24.0 if kind == "pma_bwd_src" else 6.0
        return 2 if (mean < limit and self.max_deg <= 512) else 1

    @property
    def nnz(self) -> int:
        return int(self.col.numel())


def _rowmajor(t: Tensor) -> Tensor:
    """Kernels take row-major matrices with an explicit leading dimension (stride(1) == 1)."""
    if t.dim() != 2:
        raise _lib.AllSetHipError(f"expected a 2-D matrix, got shape {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        return t.contiguous()
    return t


def _ld(t: Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def _f32(t: Tensor, what: str) -> None:
    if t.dtype != torch.float32:
        raise _lib.AllSetHipError(f"{what}: float32 required (got {t.dtype})")


def _dtype_code(t: Tensor, what: str) -> int:
    """Storage dtype of a feature matrix: fp32, or bf16 (accumulation is fp32 in both cases)."""
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.bfloat16:
        return _lib.BF16
    raise _lib.AllSetHipError(f"{what}: float32 or bfloat16 storage required (got {t.dtype})")


def csr_build(row_ids: Tensor, col_ids: Tensor, row_base: int, col_base: int, n_rows: int, n_cols: int) -> CSR:
    dev = require_device(row_ids, col_ids)
    if row_ids.dtype != torch.int64 or col_ids.dtype != torch.int64:
        raise _lib.AllSetHipError("csr_build: int64 ids required (the reference's edge_index dtype)")
    row_ids, col_ids = row_ids.contiguous(), col_ids.contiguous()
    nnz = row_ids.numel()
    lib = _lib.load()
    need = c_size_t(0)
    with on_device(dev):
        check(lib.allset_csr_build_workspace_bytes(nnz, n_rows, byref(need)), "allset_csr_build_workspace_bytes")
        rowptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
        col = torch.empty(nnz, dtype=torch.int32, device=dev)
        perm = torch.empty(nnz, dtype=torch.int32, device=dev)
        ws = torch.empty(max(need.value, 1), dtype=torch.uint8, device=dev)   # caching allocator: 512-B aligned
        check(lib.allset_csr_build(ptr(row_ids), ptr(col_ids), nnz, row_base, col_base, n_rows,
                                   ptr(rowptr), ptr(col), ptr(perm), ptr(ws), need.value, stream_of(dev)),
              "allset_csr_build")
    max_deg, short_tail = 0, -1
    if n_rows > 0 and nnz > 0:                                                            # one-time sync at build
        deg = rowptr[1:] - rowptr[:-1]
        big = (deg > 2).nonzero()
        stats = torch.stack([deg.max().to(torch.int64), (big[-1, 0] + 1) if big.numel() else torch.zeros((), dtype=torch.int64, device=dev)])
        max_deg, first_short = (int(v) for v in stats.tolist())
        head_nnz = int(rowptr[first_short]) if first_short > 0 else 0
        # a block of singleton rows behind regular rows (the reference's default Add_Self_Loops layout): one wave per
        # row wastes the launch on them, the short-row kernel wastes the long rows -- give each block its kernel
        if 0 < first_short < n_rows and (n_rows - first_short) * 16 >= n_rows and head_nnz >= 6 * first_short and max_deg <= 100000:
            short_tail = first_short
    return CSR(rowptr, col, perm, n_rows, n_cols, max_deg, long_rows_first_order(rowptr, n_rows, nnz, max_deg), short_tail,
               size_split(rowptr, col, n_rows, max_deg) if short_tail < 0 else None)


def long_rows_first_order(rowptr: Tensor, n_rows: int, nnz: int, max_deg: int) -> Optional[Tensor]:
    """Processing order for skewed degree distributions, or None when the distribution is not skewed.

    The one-wave-per-row kernels map workgroup b to XCD b % 8 and give every XCD a contiguous range of row slots.
    A row with thousands of incidences keeps one wave busy for ~0.3 ms; if it is dispatched late it sets the end
    of the launch (profiles: Zipf sizes up to 4096 run at 80 % of the uniform-size rate).  This permutation lists,
    for each XCD's slot range, that XCD's share of the long rows first (round-robin, longest first) and then the
    remaining rows in natural order -- the results are unchanged, only the dispatch order moves."""
    if n_rows < 64 or nnz == 0:
        return None
    mean = nnz / n_rows
    if not (max_deg > 256 and max_deg > 8.0 * mean):
        return None
    dev = rowptr.device
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    long_mask = deg > max(64, int(4 * mean))
    long_rows = long_mask.nonzero().reshape(-1)
    if long_rows.numel() == 0 or long_rows.numel() > n_rows // 4:
        return None
    long_rows = long_rows[torch.argsort(deg[long_rows], descending=True, stable=True)]
    rest = (~long_mask).nonzero().reshape(-1)
    nb = (n_rows + 3) // 4                                   # workgroups (4 rows each)
    q, r = nb // 8, nb % 8
    order = torch.empty(n_rows, dtype=torch.int64, device=dev)
    slot, taken = 0, 0
    for xcd in range(8):
        cnt = min(4 * (q + 1 if xcd < r else q), n_rows - slot)      # row slots of this XCD
        mine = long_rows[xcd::8]
        k = int(mine.numel())
        if k > cnt:
            return None                                        # more long rows than slots: not a skew problem
        order[slot:slot + k] = mine
        order[slot + k:slot + cnt] = rest[taken:taken + cnt - k]
        taken += cnt - k
        slot += cnt
    if slot != n_rows or taken != rest.numel():
        return None
    return order.to(torch.int32).contiguous()


def segreduce(reduce: int, rowptr: Tensor, col: Tensor, w: Optional[Tensor], x: Tensor, n_t: int,
              want_arg: bool = False, variant: int = 0, row_order: Optional[Tensor] = None, split: int = -1
              ) -> Tuple[Tensor, Optional[Tensor]]:
    """``variant``: 0 auto (short-row kernel when nnz / n_t < 6), 1 one wave per row, 2 short-row kernel.
    ``split`` (``CSR.short_tail``): rows >= split go to the short-row kernel in a second launch."""
    dev = require_device(rowptr, col, w, x)
    code = _dtype_code(x, "segreduce")
    es = x.element_size()
    x = _rowmajor(x)
    n_s, d = x.shape
    out = torch.empty((n_t, d), dtype=x.dtype, device=dev)
    arg = torch.empty((n_t, d), dtype=torch.int32, device=dev) if want_arg else None
    if w is not None:
        _f32(w, "segreduce weights")
        w = w.contiguous()
    nnz = col.numel()
    algo = nnz * (es * d + 4 + (4 if w is not None else 0)) + (n_t + 1) * 4 + n_t * d * es
    with on_device(dev), _timed("segreduce_fwd", dev, algo):
        if row_order is not None and row_order.numel() != n_t:
            row_order = None                       # order was built for a different row count (prefix views)
        lib = _lib.load()
        if 0 < split < n_t and not want_arg and rowptr.numel() == n_t + 1:
            # regular rows: one wave per row; the singleton tail: short-row kernel (same rowptr / col / out buffers, the
            # row pointers are absolute positions so an offset view is all the second launch needs)
            check(lib.allset_segreduce_fwd_ex(reduce, code, 1, nnz, None, ptr(rowptr), ptr(col), ptr(w), ptr(x), _ld(x),
                                              ptr(out), max(d, 1), None, split, n_s, d, stream_of(dev)), "allset_segreduce_fwd_ex")
            check(lib.allset_segreduce_fwd_ex(reduce, code, 2, nnz, None, ptr(rowptr[split:]), ptr(col), ptr(w), ptr(x), _ld(x),
                                              ptr(out[split:]), max(d, 1), None, n_t - split, n_s, d, stream_of(dev)),
                  "allset_segreduce_fwd_ex")
        else:
            check(lib.allset_segreduce_fwd_ex(reduce, code, variant, nnz, ptr(row_order), ptr(rowptr), ptr(col), ptr(w), ptr(x), _ld(x),
                                              ptr(out), max(d, 1), ptr(arg), n_t, n_s, d, stream_of(dev)),
                  "allset_segreduce_fwd_ex")
    return out, arg


def segmax_bwd(rowptrT: Tensor, colT: Tensor, posT: Tensor, wT: Optional[Tensor], argext: Tensor, gout: Tensor,
               n_s: int) -> Tensor:
    dev = require_device(rowptrT, colT, posT, wT, argext, gout)
    _f32(gout, "segmax_bwd")
    gout = _rowmajor(gout)
    n_t, d = gout.shape
    if colT.numel() == 0:                       # no incidences: no row received anything (the ABI takes no null index arrays)
        return torch.zeros((n_s, d), dtype=gout.dtype, device=dev)
    gx = torch.empty((n_s, d), dtype=gout.dtype, device=dev)
    algo = colT.numel() * (4 * d + 4 * d + 8) + (n_s + 1) * 4 + n_s * d * 4
    with on_device(dev), _timed("segmax_bwd", dev, algo):
        check(_lib.load().allset_segmax_bwd(ptr(rowptrT), ptr(colT), ptr(posT), ptr(wT), ptr(argext), ptr(gout),
                                            _ld(gout), ptr(gx), max(d, 1), n_s, n_t, d, stream_of(dev)),
              "allset_segmax_bwd")
    return gx


def sddmm_rowdot(reduce: int, rowptr: Tensor, col: Tensor, x: Tensor, gout: Tensor, argext: Optional[Tensor]) -> Tensor:
    dev = require_device(rowptr, col, x, gout, argext)
    _f32(x, "sddmm_rowdot")
    x, gout = _rowmajor(x), _rowmajor(gout)
    n_s, d = x.shape
    n_t = gout.shape[0]
    gw = torch.empty(col.numel(), dtype=torch.float32, device=dev)
    if col.numel() == 0:
        return gw
    algo = col.numel() * (4 * d + 8) + (n_t + 1) * 4 + n_t * d * 4
    with on_device(dev), _timed("sddmm_rowdot", dev, algo):
        check(_lib.load().allset_sddmm_rowdot(reduce, ptr(rowptr), ptr(col), ptr(x), _ld(x), ptr(gout), _ld(gout),
                                              ptr(argext), ptr(gw), n_t, n_s, d, stream_of(dev)),
              "allset_sddmm_rowdot")
    return gw


def pma_fwd(rowptr: Tensor, col: Tensor, alpha: Tensor, V: Tensor, heads: int, slope: float, n_t: int,
            variant: int = 0, row_order: Optional[Tensor] = None, split: int = -1, sizes: Optional[SizeSplit] = None
            ) -> Tuple[Tensor, Tensor, Tensor]:
    """``sizes`` (``CSR.sizes``): two launches -- the long rows one wave each, the short rows through the short-row kernel on their
    compacted CSR -- instead of ``variant``."""
    dev = require_device(rowptr, col, alpha, V)
    code = _dtype_code(V, "pma_fwd")
    es = V.element_size()
    _f32(alpha, "pma_fwd logits")
    V = _rowmajor(V)
    if alpha.dim() != 2 or (alpha.shape[1] > 1 and alpha.stride(1) != 1) or 0 < split:
        alpha = alpha.contiguous()           # (a row-strided view is fine: the logits may sit beside the V rows)
    n_s, d = V.shape
    if d % heads != 0 or alpha.shape != (n_s, heads):
        raise _lib.AllSetHipError(f"pma_fwd: V {tuple(V.shape)} / alpha {tuple(alpha.shape)} inconsistent with heads={heads}")
    out = torch.empty((n_t, d), dtype=V.dtype, device=dev)
    m = torch.empty((n_t, heads), dtype=torch.float32, device=dev)
    l = torch.empty((n_t, heads), dtype=torch.float32, device=dev)
    algo = col.numel() * (es * d + 4 + 4 * heads) + (n_t + 1) * 4 + n_t * (d * es + 8 * heads)
    with on_device(dev), _timed("pma_fwd", dev, algo):
        if row_order is not None and row_order.numel() != n_t:
            row_order = None
        lib = _lib.load()
        if 0 < split < n_t and rowptr.numel() == n_t + 1:      # see segreduce: regular rows / singleton tail
            check(lib.allset_pma_fwd_ex(code, 1, col.numel(), None, ptr(rowptr), ptr(col), ptr(alpha), ptr(V), _ld(V), slope,
                                        ptr(out), max(d, 1), ptr(m), ptr(l), split, n_s, heads, d // heads, stream_of(dev)),
                  "allset_pma_fwd_ex")
            check(lib.allset_pma_fwd_ex(code, 2, col.numel(), None, ptr(rowptr[split:]), ptr(col), ptr(alpha), ptr(V), _ld(V), slope,
                                        ptr(out[split:]), max(d, 1), ptr(m[split:]), ptr(l[split:]), n_t - split, n_s, heads,
                                        d // heads, stream_of(dev)), "allset_pma_fwd_ex")
        elif sizes is not None and rowptr.numel() == n_t + 1:
            lda = alpha.stride(0) if n_s > 1 else heads
            check(lib.allset_pma_fwd_ld(code, 1, col.numel(), ptr(sizes.long_ids), ptr(rowptr), ptr(col), ptr(alpha), lda, ptr(V),
                                        _ld(V), slope, ptr(out), max(d, 1), ptr(m), ptr(l), sizes.long_ids.numel(), n_s, heads,
                                        d // heads, stream_of(dev)), "allset_pma_fwd_ld")
            check(lib.allset_pma_fwd_ld(code, 2, sizes.col_short.numel(), ptr(sizes.short_ids), ptr(sizes.rowptr_short),
                                        ptr(sizes.col_short), ptr(alpha), lda, ptr(V), _ld(V), slope, ptr(out), max(d, 1), ptr(m),
                                        ptr(l), sizes.short_ids.numel(), n_s, heads, d // heads, stream_of(dev)), "allset_pma_fwd_ld")
        else:
            lda = alpha.stride(0) if n_s > 1 else heads
            if variant == 2:
                row_order = None             # (the short-row kernel reads a row list as the ids of a COMPACTED CSR: see ``sizes``)
            check(lib.allset_pma_fwd_ld(code, variant, col.numel(), ptr(row_order), ptr(rowptr), ptr(col), ptr(alpha), lda, ptr(V),
                                        _ld(V), slope, ptr(out), max(d, 1), ptr(m), ptr(l), n_t, n_s, heads, d // heads,
                                        stream_of(dev)), "allset_pma_fwd_ld")
    return out, m, l


def pma_attention(rowptr: Tensor, col: Tensor, alpha: Tensor, m: Tensor, l: Tensor, slope: float) -> Tensor:
    dev = require_device(rowptr, col, alpha, m, l)
    n_t, heads = m.shape
    p = torch.empty((col.numel(), heads), dtype=torch.float32, device=dev)
    with on_device(dev):
        check(_lib.load().allset_pma_attention(ptr(rowptr), ptr(col), ptr(alpha.contiguous()), ptr(m), ptr(l), slope,
                                               ptr(p), n_t, heads, stream_of(dev)), "allset_pma_attention")
    return p


def pma_bwd_stats(out: Tensor, gout: Tensor, m: Tensor, l: Tensor, stats: Optional[Tensor] = None) -> Tensor:
    """``stats`` (optional): a float32 [n_t, H, 2] destination, possibly a row-strided view (e.g. beside a copy of the
    ``gout`` rows, see ``pma_bwd_src``)."""
    dev = require_device(out, gout, m, l)
    code = _dtype_code(out, "pma_bwd_stats")
    if gout.dtype != out.dtype:
        gout = gout.to(out.dtype)
    es = out.element_size()
    out, gout = _rowmajor(out), _rowmajor(gout)
    n_t, d = out.shape
    heads = m.shape[1]
    if stats is None:
        stats = torch.empty((n_t, heads, 2), dtype=torch.float32, device=dev)
    elif stats.shape != (n_t, heads, 2) or stats.dtype != torch.float32 or stats.stride(2) != 1 or stats.stride(1) != 2:
        raise _lib.AllSetHipError("pma_bwd_stats: stats must be float32 [n_t, H, 2] with contiguous rows")
    lds = stats.stride(0) if n_t > 1 else 2 * heads
    algo = n_t * (2 * d * es + 8 * heads + 8 * heads)
    with on_device(dev), _timed("pma_bwd_stats", dev, algo):
        check(_lib.load().allset_pma_bwd_stats_ld(code, ptr(out), _ld(out), ptr(gout), _ld(gout), ptr(m.contiguous()),
                                                  ptr(l.contiguous()), ptr(stats), lds, n_t, heads, d // heads, stream_of(dev)),
              "allset_pma_bwd_stats_ld")
    return stats


def pma_bwd_src(rowptrT: Tensor, colT: Tensor, alpha: Tensor, V: Tensor, gout: Tensor, stats: Tensor, slope: float,
                variant: int = 0, row_order: Optional[Tensor] = None, split: int = -1, sizes: Optional[SizeSplit] = None
                ) -> Tuple[Tensor, Tensor]:
    dev = require_device(rowptrT, colT, alpha, V, gout, stats)
    code = _dtype_code(V, "pma_bwd_src")
    if gout.dtype != V.dtype:
        gout = gout.to(V.dtype)
    es = V.element_size()
    _f32(alpha, "pma_bwd_src logits")
    V, gout = _rowmajor(V), _rowmajor(gout)
    alpha = alpha.contiguous()
    n_s, d = V.shape
    n_t = gout.shape[0]
    heads = alpha.shape[1]
    gV = torch.empty((n_s, d), dtype=V.dtype, device=dev)
    galpha = torch.empty((n_s, heads), dtype=torch.float32, device=dev)
    stats = stats.view(n_t, heads, 2) if stats.is_contiguous() else stats
    if (stats.dim() != 3 or stats.stride(2) != 1 or stats.stride(1) != 2 or 0 < split < n_s):
        stats = stats.contiguous().view(n_t, heads, 2)           # (a row-strided view is fine: stats may sit beside gout rows)
    lds = stats.stride(0) if n_t > 1 else 2 * heads
    algo = colT.numel() * (es * d + 4 + 8 * heads) + (n_s + 1) * 4 + n_s * (2 * d * es + 8 * heads)
    with on_device(dev), _timed("pma_bwd_src", dev, algo):
        if row_order is not None and row_order.numel() != n_s:
            row_order = None
        lib = _lib.load()
        if 0 < split < n_s and rowptrT.numel() == n_s + 1:     # rows of this CSR are the SOURCES (alpha / V / gV rows)
            check(lib.allset_pma_bwd_src_ex(code, 1, colT.numel(), None, ptr(rowptrT), ptr(colT), ptr(alpha), ptr(V), _ld(V),
                                            ptr(gout), _ld(gout), ptr(stats), slope, ptr(gV), max(d, 1), ptr(galpha), split, n_t,
                                            heads, d // heads, stream_of(dev)), "allset_pma_bwd_src_ex")
            check(lib.allset_pma_bwd_src_ex(code, 2, colT.numel(), None, ptr(rowptrT[split:]), ptr(colT), ptr(alpha[split:]),
                                            ptr(V[split:]), _ld(V), ptr(gout), _ld(gout), ptr(stats), slope, ptr(gV[split:]),
                                            max(d, 1), ptr(galpha[split:]), n_s - split, n_t, heads, d // heads, stream_of(dev)),
                  "allset_pma_bwd_src_ex")
        elif sizes is not None and rowptrT.numel() == n_s + 1:
            check(lib.allset_pma_bwd_src_ld(code, 1, colT.numel(), ptr(sizes.long_ids), ptr(rowptrT), ptr(colT), ptr(alpha), ptr(V),
                                            _ld(V), ptr(gout), _ld(gout), ptr(stats), lds, slope, ptr(gV), max(d, 1),
                                            ptr(galpha), sizes.long_ids.numel(), n_t, heads, d // heads, stream_of(dev)),
                  "allset_pma_bwd_src_ld")
            check(lib.allset_pma_bwd_src_ld(code, 2, sizes.col_short.numel(), ptr(sizes.short_ids), ptr(sizes.rowptr_short),
                                            ptr(sizes.col_short), ptr(alpha), ptr(V), _ld(V), ptr(gout), _ld(gout), ptr(stats), lds,
                                            slope, ptr(gV), max(d, 1), ptr(galpha), sizes.short_ids.numel(), n_t, heads, d // heads,
                                            stream_of(dev)), "allset_pma_bwd_src_ld")
        else:
            if variant == 2:
                row_order = None
            check(lib.allset_pma_bwd_src_ld(code, variant, colT.numel(), ptr(row_order), ptr(rowptrT), ptr(colT), ptr(alpha), ptr(V),
                                            _ld(V), ptr(gout), _ld(gout), ptr(stats), lds, slope, ptr(gV), max(d, 1),
                                            ptr(galpha), n_s, n_t, heads, d // heads, stream_of(dev)),
                  "allset_pma_bwd_src_ld")
    return gV, galpha


def block_transpose(x: Tensor, world: int, to_blocks: bool) -> Tensor:
    """The pack / unpack copy of the column-sharded layer's all-to-all.  ``to_blocks``: [rows, P*dc] row-major ->
    [P, rows, dc]; else [P, rows, dc] -> [rows, P*dc].  Needs dc * element_size to be a multiple of 16."""
    dev = require_device(x)
    es = x.element_size()
    if to_blocks:
        x = _rowmajor(x)
        rows, d = x.shape
        dc = d // world
        out = torch.empty((world, rows, dc), dtype=x.dtype, device=dev)
        ld = _ld(x) * es
    else:
        x = x.contiguous()
        _, rows, dc = x.shape
        out = torch.empty((rows, world * dc), dtype=x.dtype, device=dev)
        ld = world * dc * es
    with on_device(dev), _timed("block_transpose", dev, 2 * rows * world * dc * es):
        check(_lib.load().allset_block_transpose(ptr(x), ptr(out), rows, world, dc * es, ld, int(to_blocks), stream_of(dev)),
              "allset_block_transpose")
    return out


def block_transpose_supported(x: Tensor, world: int, to_blocks: bool) -> bool:
    if not x.is_cuda:
        return False
    dc = (x.shape[1] // world) if to_blocks else x.shape[2]
    return (dc * x.element_size()) % 16 == 0 and dc > 0 and (not to_blocks or x.shape[1] % world == 0)


def pma_merge_pack(out_loc: Tensor, m_loc: Tensor, l_loc: Tensor, m_glob: Tensor, heads: int) -> Tensor:
    """[n, d + H] rows ``[out_loc * w | w]`` with ``w = l_loc * exp(m_loc - m_glob)`` (0 where ``l_loc == 0``): this rank's
    numerators / denominators relative to the global row maximum, ready for a sum-reduce-scatter."""
    dev = require_device(out_loc, m_loc, l_loc, m_glob)
    _f32(out_loc, "pma_merge_pack")
    out_loc = _rowmajor(out_loc)
    n, d = out_loc.shape
    width = d + heads
    ldp = (width + 3) // 4 * 4
    buf = torch.empty((n, ldp), dtype=torch.float32, device=dev)
    with on_device(dev), _timed("pma_merge_pack", dev, n * (2 * d + 4 * heads) * 4):
        check(_lib.load().allset_pma_merge_pack(ptr(out_loc), _ld(out_loc), ptr(m_loc.contiguous()), ptr(l_loc.contiguous()),
                                                ptr(m_glob.contiguous()), ptr(buf), ldp, n, heads, d // heads, stream_of(dev)),
              "allset_pma_merge_pack")
    return buf if ldp == width else buf[:, :width]
