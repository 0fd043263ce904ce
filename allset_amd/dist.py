"""Hyperedge-sharded execution of the AllSet layer across the GPUs of one node (one process per GPU,
``torch.distributed``; backend ``nccl`` == RCCL over xGMI on ROCm).  The reference has no distributed
code at all; this is new, and it is the one real exchange step of the path:

* hyperedges are partitioned into P nnz-balanced shards; rank r owns its hyperedges' rows end to end;
* vertices are partitioned into P equal blocks for the DENSE per-vertex work (owner computes);
* V->E : owned vertex rows --all-gather--> full [n_V, d] table --local gather/reduce--> owned hyperedges;
* E->V : owned hyperedges --local gather/reduce--> PARTIAL sums for all n_V vertices --reduce-scatter-->
         owned vertex rows.   Backward mirrors: all-gather <-> reduce-scatter swap roles.

So a layer forward is one all-gather + one reduce-scatter of [n_V, d] (and the same pair in backward),
i.e. the "all-reduce of boundary-vertex embeddings" of the north star split into its two halves so the
dense tail runs on n_V/P rows instead of n_V.  On a uniformly random hypergraph practically every vertex is
a boundary vertex, so the exchange is dense.

The local aggregation is pluggable (``aggregate=``): the product passes the HIP-backed functions of
``functional.py``; the CPU (gloo) tests pass the oracle so the partition + exchange logic is testable
without a GPU.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F

Tensor = torch.Tensor


# ---------------------------------------------------------------------------------------------
# partitioning (pure index arithmetic; runs on whatever device the ids live on)
# ---------------------------------------------------------------------------------------------

def vertex_block(n_v: int, world: int, rank: int) -> Tuple[int, int, int]:
    """Equal vertex blocks of the padded vertex range.  Returns (lo, hi, n_v_padded)."""
    per = (n_v + world - 1) // world
    return rank * per, (rank + 1) * per, per * world


def partition_hyperedges(sizes: Tensor, world: int, method: str = "contiguous") -> Tensor:
    """Assign each hyperedge to a rank so that incidences (nnz), not hyperedge counts, are balanced.

    ``contiguous``: split the id range where the running nnz crosses k/P of the total (keeps locality of
    neighbouring ids; right for near-uniform sizes).  ``lpt``: longest-processing-time greedy over sizes
    (right for power-law sizes, BASELINE.json configs[4]).  Returns int64[n_e] owner ranks."""
    sizes = sizes.to(torch.int64)
    n_e = sizes.numel()
    if method == "contiguous":
        csum = torch.cumsum(sizes, 0)
        total = int(csum[-1]) if n_e else 0
        # owner = number of split points strictly below the running midpoint of each hyperedge
        mid = csum - sizes // 2
        bounds = torch.tensor([total * (k + 1) // world for k in range(world - 1)], dtype=torch.int64, device=sizes.device)
        return torch.bucketize(mid, bounds, right=False).clamp_(max=world - 1)
    if method == "lpt":
        order = torch.argsort(sizes, descending=True).tolist()
        load = [0] * world
        owner = torch.empty(n_e, dtype=torch.int64)
        sz = sizes.tolist()
        for e in order:
            r = min(range(world), key=load.__getitem__)
            owner[e] = r
            load[r] += sz[e]
        return owner.to(sizes.device)
    raise ValueError(method)


def local_shard(edge_index: Tensor, owner: Tensor, rank: int) -> Tuple[Tensor, Tensor]:
    """Incidences of the hyperedges owned by ``rank`` with hyperedge ids renumbered 0..n_e_local-1 in
    increasing global-id order.  ``edge_index``: [2, nnz] (row 0 global vertex ids, row 1 global 0-based
    hyperedge ids).  Returns (local edge_index, int64[n_e_local] global ids of the local hyperedges)."""
    mine = owner == rank
    gids = mine.nonzero().reshape(-1)
    new_id = torch.full_like(owner, -1)
    new_id[gids] = torch.arange(gids.numel(), device=owner.device)
    keep = mine[edge_index[1]]
    loc = torch.stack([edge_index[0][keep], new_id[edge_index[1][keep]]], dim=0)
    return loc.contiguous(), gids


# ---------------------------------------------------------------------------------------------
# collectives with autograd mirrors
# ---------------------------------------------------------------------------------------------

def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _skip_collective(group=None) -> bool:
    """World size 1 needs no exchange -- unless ALLSET_FORCE_COLLECTIVES=1, which runs the very same RCCL
    calls on a 1-rank group (the only way to exercise them on a 1-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and os.environ.get("ALLSET_FORCE_COLLECTIVES", "0") != "1"


# ---- opt-in reduced-precision WIRE format (SURVEY section 7 mitigation (b)) -----------------------------------------------
# fp32 activations are rounded to this dtype for the exchange only and widened again on arrival; every sum -- including the
# cross-rank sum of the row scheme's reduce-scatter, which becomes an all-to-all of pieces + a local fp32 sum -- stays fp32.
# Halves the bytes per link.  NOT the default and never the headline: it changes results (one bf16 rounding, 2^-9 relative, per
# exchanged activation; tests/test_dist_cpu.py states the tolerance) -- bench.py reports it as its own `partitions` entry.
# Covers the blocking exchanges; the chunked asynchronous exchange keeps fp32.
_WIRE_DTYPE: Optional[torch.dtype] = torch.bfloat16 if os.environ.get("ALLSET_WIRE_DTYPE", "") == "bf16" else None


def set_wire_dtype(dtype: Optional[torch.dtype]) -> Optional[torch.dtype]:
    """``torch.bfloat16`` (or ``None`` = exact fp32 wire, the default).  Returns the previous setting."""
    global _WIRE_DTYPE
    if dtype not in (None, torch.bfloat16, torch.float16):
        raise ValueError(f"wire dtype {dtype}")
    prev, _WIRE_DTYPE = _WIRE_DTYPE, dtype
    return prev


def _narrow(t: Tensor) -> Tensor:
    return t.to(_WIRE_DTYPE) if (_WIRE_DTYPE is not None and t.dtype == torch.float32) else t


def _host_staged(group, *tensors) -> bool:
    """gloo moves host memory only: device tensors are staged through the host around the call.  That is how two real
    ranks run the HIP path on ONE GPU (tests/test_gpu_two_ranks.py: RCCL refuses two ranks on the same device) -- the
    autograd Functions, the kernels and the exchange wiring are the product's, only the wire is the host's."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_backend(group) == "gloo" and any(t.is_cuda for t in tensors)


def _all_reduce_(t: Tensor, op=None, group=None) -> Tensor:
    """In-place all-reduce of ``t`` (sum by default) on whatever backend the group has."""
    op = dist.ReduceOp.SUM if op is None else op
    if _host_staged(group, t):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def _all_to_all_single(recv: Tensor, send: Tensor, group=None) -> None:
    if _host_staged(group, recv, send):
        hs = send.cpu()
        hr = torch.empty_like(hs)
        dist.all_to_all_single(hr, hs, group=group)
        recv.copy_(hr)
    else:
        dist.all_to_all_single(recv, send, group=group)


def _all_to_all_rows(send: Tensor, in_splits, out_splits, group=None) -> Tensor:
    """All-to-all of ROW RANGES with per-peer counts: ``send`` [sum(in_splits), ...] holds the rows for peer 0, 1, ... back to
    back; returns [sum(out_splits), ...] in peer order.  (The halo exchange of the row partition; gloo: staged through the host.)"""
    send = send.contiguous()
    out_shape = (int(sum(out_splits)),) + tuple(send.shape[1:])
    if _host_staged(group, send):
        hs = send.cpu()
        hr = torch.empty(out_shape, dtype=hs.dtype)
        dist.all_to_all_single(hr, hs, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits), group=group)
        return hr.to(send.device)
    recv = send.new_empty(out_shape)
    dist.all_to_all_single(recv, send, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits), group=group)
    return recv


def _all_gather_rows(x: Tensor, group=None) -> Tensor:
    if _skip_collective(group):
        return x
    w = _world(group)
    x = x.contiguous()
    if _narrow(x) is not x:                       # reduced-precision wire: gather the narrow rows, widen on arrival
        xn = _narrow(x)
        outn = xn.new_empty((w * x.shape[0],) + tuple(x.shape[1:]))
        if _host_staged(group, xn):
            ho = torch.empty(outn.shape, dtype=outn.dtype)
            dist.all_gather_into_tensor(ho, xn.cpu(), group=group)
            outn.copy_(ho)
        else:
            dist.all_gather_into_tensor(outn, xn, group=group)
        return outn.to(x.dtype)
    out = x.new_empty((w * x.shape[0],) + tuple(x.shape[1:]))
    if _host_staged(group, x):
        ho = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(ho, x.cpu(), group=group)
        out.copy_(ho)
    else:
        dist.all_gather_into_tensor(out, x, group=group)
    return out


def _reduce_scatter_rows(x: Tensor, group=None) -> Tensor:
    if _skip_collective(group):
        return x
    w = _world(group)
    x = x.contiguous()
    per = x.shape[0] // w
    if _narrow(x) is not x:                       # reduced-precision wire: the P pieces travel narrow, the sum over ranks is fp32 here
        send = _narrow(x).view((w, per) + tuple(x.shape[1:]))
        recv = torch.empty_like(send)
        _all_to_all_single(recv, send, group)
        return recv.to(x.dtype).sum(dim=0)
    if dist.get_backend(group) == "gloo":            # gloo has no reduce_scatter: all-reduce + slice (tests only)
        buf = _all_reduce_(x.clone(), group=group)
        r = dist.get_rank(group)
        return buf[r * per:(r + 1) * per].contiguous()
    out = x.new_empty((per,) + tuple(x.shape[1:]))
    dist.reduce_scatter_tensor(out, x, op=dist.ReduceOp.SUM, group=group)
    return out


class _AllGatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return _all_gather_rows(x, group)

    @staticmethod
    def backward(ctx, g):
        return _reduce_scatter_rows(g, ctx.group), None


class _ReduceScatterRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return _reduce_scatter_rows(x, group)

    @staticmethod
    def backward(ctx, g):
        return _all_gather_rows(g, ctx.group), None


def all_gather_rows(x: Tensor, group=None) -> Tensor:
    """[n/P, ...] owned block -> [n, ...] full table; backward = sum-reduce-scatter of the gradient."""
    return _AllGatherRows.apply(x, group)


def reduce_scatter_rows(x: Tensor, group=None) -> Tensor:
    """[n, ...] per-rank partial sums -> [n/P, ...] owned block of the total; backward = all-gather."""
    return _ReduceScatterRows.apply(x, group)


# ---------------------------------------------------------------------------------------------
# boundary-vertex ("halo") exchange of the row partition
# ---------------------------------------------------------------------------------------------
# The all-gather / reduce-scatter pair above moves the WHOLE vertex table, which is what a hypergraph without locality needs
# (the synthetic benchmark: a rank's hyperedges touch 86 % of all vertices at P = 8).  A hypergraph WITH locality -- most members of
# a hyperedge live in the block of vertices its rank owns -- only needs the rows of the vertices its local hyperedges actually touch:
# the north star's "all-reduce of boundary-vertex embeddings", split into its two halves.  ``Halo`` is that exchange, built once
# per sharded hypergraph:
#   needed      the sorted global ids of the vertices this rank's hyperedges touch (its own block's included); the local incidence is
#               renumbered onto 0 .. len(needed)-1, so the V->E gather reads a COMPACT table;
#   gather      owned rows --index_select by what each peer asked for--> all-to-all (per-peer row counts) --> the compact table;
#   scatter_add the transpose: compact partial sums --all-to-all--> rows addressed to my owned vertices --> summed per owned row
#               in a FIXED order (a CSR over the received rows, the library's own segment-sum kernel on the device; no atomics).
# Forward and backward of the two autograd Functions are each other.  Bytes per exchange: sum over peers of the rows asked for,
# instead of (P-1)/P of the table.

class Halo:
    def __init__(self, local_vertex_ids: Tensor, n_v_pad: int, world: int, rank: int, group=None):
        dev = local_vertex_ids.device
        self.world, self.rank, self.group = int(world), int(rank), group
        self.block = n_v_pad // world
        self.needed = torch.unique(local_vertex_ids)                              # sorted global ids
        self.compact_ids = torch.searchsorted(self.needed, local_vertex_ids)      # local incidence -> rows of the compact table
        owner = torch.div(self.needed, self.block, rounding_mode="floor")
        need_counts = torch.bincount(owner, minlength=world)[:world]
        ask = self.needed - owner * self.block                                     # row inside its owner's block, peer-major already
        if world == 1 or _skip_collective(group):
            send_counts, send_idx = need_counts.clone(), ask
        else:
            send_counts = _all_to_all_rows(need_counts.view(world, 1), [1] * world, [1] * world, group).view(-1)
            send_idx = _all_to_all_rows(ask, [int(c) for c in need_counts.tolist()], [int(c) for c in send_counts.tolist()], group)
        self.need_counts = [int(c) for c in need_counts.tolist()]                 # rows I receive from each peer (gather)
        self.send_counts = [int(c) for c in send_counts.tolist()]                 # rows I send to each peer (gather)
        self.send_idx = send_idx.to(dev)                                          # int64 [sum(send_counts)]: my owned rows, peer-major
        if self.send_idx.numel() and (int(self.send_idx.min()) < 0 or int(self.send_idx.max()) >= self.block):
            raise ValueError("halo: a peer asked for a row outside this rank's vertex block")
        # fixed-order sum of the rows that come back in scatter_add: CSR over my owned rows, entries = positions in the received buffer
        order = torch.argsort(self.send_idx, stable=True)
        counts = torch.bincount(self.send_idx, minlength=self.block)
        self.sa_rowptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32)
        self.sa_col = order.to(torch.int32)
        self.n_needed = int(self.needed.numel())

    # bytes one rank receives per gather (or sends per scatter_add) for rows of ``row_bytes`` bytes -- own block excluded
    def exchange_rows(self) -> int:
        return sum(c for q, c in enumerate(self.need_counts) if q != self.rank)

    def _gather(self, x_owned: Tensor, narrow: bool = True) -> Tensor:
        """``narrow=False``: the rows travel in their own dtype whatever the wire format (softmax statistics: a few fp32 columns whose
        rounding would not cancel between the owner's copy and the travelled one)."""
        send = x_owned.index_select(0, self.send_idx)
        if self.world == 1 or _skip_collective(self.group):
            return send
        wire = _narrow(send) if narrow else send
        return _all_to_all_rows(wire, self.send_counts, self.need_counts, self.group).to(x_owned.dtype)

    def _scatter_add(self, part: Tensor) -> Tensor:
        """Fixed-order sum (no atomics) of what the ranks sent, per owned row.  On a device the sum ALWAYS runs through the library's
        segment-sum kernel: any width (padded to a multiple of 4 here) and any floating dtype (summed in fp32, returned in the
        input's dtype); torch's ``index_add_`` (atomics on a GPU: run-to-run nondeterministic) is the CPU path only."""
        if self.world == 1 or _skip_collective(self.group):
            recv = part
        else:
            recv = _all_to_all_rows(_narrow(part.contiguous()), self.need_counts, self.send_counts, self.group).to(part.dtype)
        if recv.is_cuda:
            if recv.dim() != 2 or not recv.is_floating_point():
                raise ValueError("halo scatter_add on a device takes a 2-D floating tensor (fixed-order segment sum)")
            from . import ops
            c = recv.shape[1]
            r32 = recv.float()
            pad = (-c) % 4
            if pad:
                r32 = torch.cat([r32, r32.new_zeros(r32.shape[0], pad)], dim=1)
            out = ops.segreduce(0, self.sa_rowptr, self.sa_col, None, r32.contiguous(), self.block)[0]
            if pad:
                out = out[:, :c].contiguous()
            return out.to(recv.dtype)
        out = recv.new_zeros((self.block,) + tuple(recv.shape[1:]))
        return out.index_add_(0, self.send_idx, recv)                             # CPU (gloo tests): sequential, deterministic


    def _scatter_max(self, part: Tensor) -> Tensor:
        """[len(needed), c] -> [block, c]: per owned row the maximum over what the ranks sent (rows nobody sent: 0)."""
        if self.world == 1 or _skip_collective(self.group):
            recv = part
        else:
            recv = _all_to_all_rows(part.contiguous(), self.need_counts, self.send_counts, self.group)
        c = recv.shape[1]
        if recv.is_cuda and recv.dtype == torch.float32:
            from . import ops
            pad = (-c) % 4
            if pad:
                recv = torch.cat([recv, recv.new_zeros(recv.shape[0], pad)], dim=1)
            return ops.segreduce(2, self.sa_rowptr, self.sa_col, None, recv.contiguous(), self.block)[0][:, :c].contiguous()
        out = recv.new_zeros((self.block, c))
        idx = self.send_idx.view(-1, 1).expand_as(recv)
        return out.scatter_reduce(0, idx, recv, "amax", include_self=False)

    def gather_narrow(self, x_owned: Tensor) -> Tensor:
        """``_gather`` for a few fp32 columns (softmax maxima / statistics): its own all-to-all, no autograd, and NEVER the narrow
        wire format -- the owner keeps the unrounded maximum for its backward (M = m + log l), so the copy that travels must be the
        same number (the all-to-all is only H columns wide)."""
        return self._gather(x_owned.contiguous(), narrow=False)

    def scatter_add_narrow(self, part: Tensor) -> Tensor:
        """``_scatter_add`` for a tensor whose width is not a multiple of 4 (kept for callers; ``_scatter_add`` pads itself now)."""
        return self._scatter_add(part)


class _HaloGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_owned, halo):
        ctx.halo = halo
        return halo._gather(x_owned)

    @staticmethod
    def backward(ctx, g):
        return ctx.halo._scatter_add(g.contiguous()), None


class _HaloScatterAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, part, halo):
        ctx.halo = halo
        return halo._scatter_add(part)

    @staticmethod
    def backward(ctx, g):
        return ctx.halo._gather(g.contiguous()), None


def halo_gather(x_owned: Tensor, halo: Halo) -> Tensor:
    """[n_v/P, d] owned rows -> [len(halo.needed), d] rows of the vertices this rank's hyperedges touch; backward = scatter_add."""
    return _HaloGather.apply(x_owned, halo)


def halo_scatter_add(part: Tensor, halo: Halo) -> Tensor:
    """[len(halo.needed), d] partial sums -> [n_v/P, d] totals of the owned rows over all ranks; backward = gather."""
    return _HaloScatterAdd.apply(part, halo)


# ---------------------------------------------------------------------------------------------
# the sharded layer
# ---------------------------------------------------------------------------------------------

class ShardedHypergraph:
    """One rank's view: its hyperedges' incidences over the GLOBAL (padded) vertex range.

    ``local_edge_index``: int64 [2, nnz_local], row 0 global vertex ids, row 1 local hyperedge ids.
    ``v2e`` / ``e2v`` are whatever the ``aggregate`` callables accept as their incidence argument -- the
    product builds :class:`allset_amd.incidence.Incidence` objects (``build_incidences``)."""

    def __init__(self, local_edge_index: Tensor, n_v: int, n_e_local: int, world: int, rank: int,
                 norm: Optional[Tensor] = None, inc_ids: Optional[Tensor] = None, halo: bool = False, group=None):
        """``halo=True``: exchange only the rows of the vertices the local hyperedges touch (:class:`Halo`; every Deep Sets
        ``aggr`` and the PMA layer) instead of all-gather / reduce-scatter of the whole vertex table.  Construction then runs two small all-to-alls on ``group`` (every rank must construct at the same time)."""
        self.local_edge_index = local_edge_index
        # positions of the local incidences in the GLOBAL edge list (what a replicated per-incidence parameter such as
        # SetGNN.Importance, reference models.py:336-337, is indexed by); only LearnMask needs them
        self.inc_ids = inc_ids
        self.n_v, self.n_e_local, self.world, self.rank = int(n_v), int(n_e_local), int(world), int(rank)
        self.v_lo, self.v_hi, self.n_v_pad = vertex_block(n_v, world, rank)
        self.norm = norm
        self.v2e = None
        self.e2v = None
        self._vdeg_owned: Optional[Tensor] = None
        self.halo: Optional[Halo] = Halo(local_edge_index[0], self.n_v_pad, self.world, self.rank, group) if halo else None
        self.halo_v2e = None
        self.halo_e2v = None

    def halo_edge_index(self) -> Tensor:
        """The local incidence with vertex ids renumbered onto the compact table of ``halo.needed``."""
        return torch.stack([self.halo.compact_ids, self.local_edge_index[1]])

    def build_incidences(self) -> "ShardedHypergraph":
        from .incidence import Incidence
        self.v2e = Incidence.from_edge_index(self.local_edge_index, n_src=self.n_v_pad, n_dst=self.n_e_local)
        self.e2v = self.v2e.reversed(n_dst=self.n_v_pad)          # partial rows for EVERY vertex
        if self.halo is not None:
            self.halo_v2e = Incidence.from_edge_index(self.halo_edge_index(), n_src=self.halo.n_needed, n_dst=self.n_e_local)
            self.halo_e2v = self.halo_v2e.reversed(n_dst=self.halo.n_needed)      # partial rows for the touched vertices only
        return self

    def local_vertex_has_incidence(self) -> Tensor:
        """[n_v_pad] bool: the vertex is a member of at least one of THIS rank's hyperedges (cached)."""
        if getattr(self, "_vhas", None) is None:
            self._vhas = torch.bincount(self.local_edge_index[0], minlength=self.n_v_pad) > 0
        return self._vhas

    def owned_vertex_degree(self, group=None) -> Tensor:
        """Global degree of the owned vertices (for E->V 'mean'); one reduce-scatter, cached."""
        if self._vdeg_owned is None:
            if self.halo is not None:
                deg = torch.bincount(self.halo.compact_ids, minlength=self.halo.n_needed).to(torch.float32)
                self._vdeg_owned = self.halo._scatter_add(deg.view(-1, 1).repeat(1, 4))[:, 0].contiguous()
            else:
                deg = torch.bincount(self.local_edge_index[0], minlength=self.n_v_pad).to(torch.float32)
                self._vdeg_owned = _reduce_scatter_rows(deg.view(-1, 1), group).view(-1)
        return self._vdeg_owned


def _ordered_key(v: Tensor) -> Tensor:
    """fp32 -> int64 whose integer order is the float order (NaN excluded): flip the sign bit of non-negatives, all
    bits of negatives."""
    b = v.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    return torch.where(b >= 0x80000000, 0xFFFFFFFF - b, b + 0x80000000)


class _ShardedExtremeMerge(torch.autograd.Function):
    """Cross-shard ``max`` / ``min`` of per-rank partial extremes ("max needs an all-reduce(max)",
    bit-exact).  ``part`` [n_v_pad, d]: this rank's extreme over its local incidences (0 where a vertex has none),
    ``has`` [n_v_pad] bool: the vertex has a local incidence.

    forward : key = (order-preserving 32-bit image of the value) << 8 | (255 - rank) for rows with local incidences,
              -1 otherwise; ONE max-collective over the int64 keys gives the global extreme AND a unique winner (lowest
              rank among exact ties) for every (vertex, feature); owned rows are decoded, vertices without any
              incidence give 0 as in torch_scatter.
    backward: the all-gathered gradient, masked to the elements this rank won, continues into the local aggregation.
    """

    @staticmethod
    def forward(ctx, part, has, hg, group, is_min):
        w, r = (1, 0) if _skip_collective(group) else (_world(group), dist.get_rank(group))
        v = -part if is_min else part
        key = (_ordered_key(v) << 8) | (255 - r)
        key = torch.where(has.view(-1, 1), key, torch.full_like(key, -1))
        if _skip_collective(group):
            best = key
        else:
            best = _all_reduce_(key.clone(), dist.ReduceOp.MAX, group)   # [n_v_pad, d] int64: value and winner in one pass
        won = (best == key) & has.view(-1, 1)
        lo, hi = hg.v_lo, hg.v_hi
        kb = best[lo:hi]
        any_inc = kb >= 0
        bits = (kb >> 8) & 0xFFFFFFFF
        bits = torch.where(bits >= 0x80000000, bits - 0x80000000, 0xFFFFFFFF - bits)
        val = torch.where(bits >= 0x80000000, bits - 0x100000000, bits).to(torch.int32).view(torch.float32)
        out = torch.where(any_inc, -val if is_min else val, torch.zeros_like(val))
        ctx.save_for_backward(won)
        ctx.group = group
        return out

    @staticmethod
    def backward(ctx, gout):
        (won,) = ctx.saved_tensors
        g_full = _all_gather_rows(gout.contiguous(), ctx.group)
        return torch.where(won, g_full, torch.zeros_like(g_full)), None, None, None, None


class _HaloExtremeMerge(torch.autograd.Function):
    """:class:`_ShardedExtremeMerge` through the boundary-vertex exchange: ``part`` [len(halo.needed), d] is this rank's extreme
    for the vertices its hyperedges touch (each has a local incidence by construction).  The same keys -- (order-preserving
    image of the value) << 8 | (255 - rank) -- travel to the vertices' owners, who keep the maximum per (vertex, feature): the
    global extreme and a unique winner; the winning keys travel back so that every rank knows what it won; the backward gathers the
    owners' gradient rows for the touched vertices and masks them to the elements won."""

    @staticmethod
    def forward(ctx, part, hg, is_min):
        halo = hg.halo
        r = halo.rank
        v = -part if is_min else part
        key = (_ordered_key(v) << 8) | (255 - r)
        if halo.world == 1 or _skip_collective(halo.group):
            recv = key
        else:
            recv = _all_to_all_rows(key.contiguous(), halo.need_counts, halo.send_counts, halo.group)
        best = torch.full((halo.block, key.shape[1]), -1, dtype=torch.int64, device=key.device)
        if recv.shape[0]:
            best = best.scatter_reduce(0, halo.send_idx.view(-1, 1).expand_as(recv), recv, "amax", include_self=True)
        best_c = best.index_select(0, halo.send_idx)
        if not (halo.world == 1 or _skip_collective(halo.group)):
            best_c = _all_to_all_rows(best_c, halo.send_counts, halo.need_counts, halo.group)
        won = best_c == key
        any_inc = best >= 0
        bits = (best >> 8) & 0xFFFFFFFF
        bits = torch.where(bits >= 0x80000000, bits - 0x80000000, 0xFFFFFFFF - bits)
        val = torch.where(bits >= 0x80000000, bits - 0x100000000, bits).to(torch.int32).view(torch.float32)
        out = torch.where(any_inc, -val if is_min else val, torch.zeros_like(val))
        ctx.save_for_backward(won)
        ctx.halo = halo
        return out

    @staticmethod
    def backward(ctx, gout):
        (won,) = ctx.saved_tensors
        g_c = ctx.halo._gather(gout.contiguous())
        return torch.where(won, g_c, torch.zeros_like(g_c)), None, None


def _hip_deepsets(x, inc, norm, aggr):
    from . import functional as AF
    return AF.deepsets_aggregate(x, inc, norm, aggr)


class HipPmaKernels:
    """The four local PMA primitives the sharded layer needs, on the HIP kernels.  The gloo tests swap in a
    torch-CPU object with the same four methods."""

    @staticmethod
    def aggregate(V, alpha, inc, heads, slope):            # differentiable, complete targets (V->E)
        from . import functional as AF
        return AF.pma_aggregate(V, alpha, inc, heads, slope)[0]

    @staticmethod
    def fwd(V, alpha, inc, heads, slope):                   # raw: (out, m, l), local softmax statistics
        from . import ops
        from .functional import _variant, _sizes
        csr = inc.by_dst
        return ops.pma_fwd(csr.rowptr, csr.col, alpha, V, heads, slope, inc.n_dst,
                           variant=_variant(csr, "pma_fwd", inc.n_dst, V, heads), row_order=csr.row_order, sizes=_sizes(csr, V, heads))

    @staticmethod
    def bwd_stats(out, gout, m, l):
        from . import ops
        return ops.pma_bwd_stats(out, gout, m, l)

    @staticmethod
    def merge_pack(o_loc, m_loc, l_loc, m_glob, heads):     # [o_loc * w | w], w = l_loc * exp(m_loc - m_glob)
        from . import ops
        return ops.pma_merge_pack(o_loc, m_loc, l_loc, m_glob, heads)

    @staticmethod
    def bwd_src(inc, alpha, V, gout, stats, slope):
        from . import ops
        from .functional import _variant, _sizes
        T = inc.by_src
        return ops.pma_bwd_src(T.rowptr, T.col, alpha, V, gout, stats, slope,
                               variant=_variant(T, "pma_bwd_src", V.shape[0], V, alpha.shape[1]), row_order=T.row_order,
                               sizes=_sizes(T, V, alpha.shape[1]))


def _bn_scope(valid_rows: int, group):
    """Row-sharded BatchNorm (the reference MLP's default ``Normalization='bn'``, layers.py:499-562): per-rank batch statistics
    (zero pad rows included) would be a different model than the single-GPU one and let the replicated running statistics drift
    apart.  Inside this scope the MLP's BatchNorm1d modules reduce (count, sum, centred sum of squares) over the VALID rows of all
    ranks (``dense.sync_bn_rows``); round 3 -- rounds 1-2 refused 'bn' in sharded mode."""
    from . import dense
    # Entered at world 1 too: the owned blocks may carry zero PAD rows there as well (ColumnShardedHypergraph pads them to a
    # multiple of `chunks`), and only the valid-row mask keeps those out of the batch and running statistics; the all-reduces
    # inside are no-ops on one rank.
    return dense.sync_bn_rows(valid_rows, group)


def _valid_rows(lo: int, hi: int, n: int) -> int:
    return max(0, min(int(hi), int(n)) - int(lo))


def sharded_deepsets_layer(v2e_conv, e2v_conv, x_owned: Tensor, hg: ShardedHypergraph, aggr: str = "add",
                           dropout: float = 0.0, training: bool = False, group=None,
                           aggregate: Callable = _hip_deepsets, norm: Optional[Tensor] = None,
                           dropout_out: Optional[float] = None) -> Tensor:
    """One V->E->V AllDeepSets layer (reference models.py:475-481 with layers.py:630-634 inlined) on a
    hyperedge shard.  ``x_owned``: this rank's block of vertex rows [n_v_pad/P, F].  Returns the owned block
    of the layer output.  ``v2e_conv`` / ``e2v_conv`` are :class:`allset_amd.layers.HalfNLHconv` (Deep Sets
    variant); their parameters are replicated, so parameter gradients must be all-reduced by the caller
    (``allreduce_grads``) -- each rank sees only its rows."""
    if aggr not in ("add", "sum", "mean", "max", "min"):
        raise ValueError(f"aggr {aggr!r}")
    vv = _valid_rows(hg.v_lo, hg.v_hi, hg.n_v)                 # real (not pad) rows of this rank's vertex block; every local hyperedge is real
    norm = hg.norm if norm is None else norm                   # ``norm``: per-incidence weights of the LOCAL incidences
    p_out = dropout if dropout_out is None else dropout_out    # GPR applies the last dropout itself (models.py:466-469)
    # ---- V -> E: dense on owned vertices, all-gather, local reduce over owned hyperedges
    # ``training`` must agree with the convs' own mode (the fused MLP kernels read conv.training)
    use_halo = hg.halo is not None and hg.halo_v2e is not None
    with _bn_scope(vv, group):
        h = v2e_conv._mlp_act(v2e_conv.f_enc, x_owned, v2e_conv.dropout)
    if use_halo:                                               # only the rows the local hyperedges touch travel
        e = aggregate(halo_gather(h, hg.halo), hg.halo_v2e, norm, aggr)
    else:
        h_full = all_gather_rows(h, group)
        e = aggregate(h_full, hg.v2e, norm, aggr)
    with _bn_scope(e.shape[0], group):
        e = v2e_conv._mlp_act(v2e_conv.f_dec, e, dropout)       # conv's relu (SetGNN's outer relu is idempotent) + dropout
        # ---- E -> V: dense on owned hyperedges, local partial sums for all vertices, reduce-scatter
        g = e2v_conv._mlp_act(e2v_conv.f_enc, e, e2v_conv.dropout)
    if aggr in ("max", "min") and use_halo:
        v = _HaloExtremeMerge.apply(aggregate(g, hg.halo_e2v, norm, aggr), hg, aggr == "min")
    elif aggr in ("max", "min"):
        # local extreme over this rank's hyperedges (autograd routes to the local arg-extreme), then the key merge
        partial = aggregate(g, hg.e2v, norm, aggr)
        v = _ShardedExtremeMerge.apply(partial, hg.local_vertex_has_incidence(), hg, group, aggr == "min")
    elif use_halo:
        v = halo_scatter_add(aggregate(g, hg.halo_e2v, norm, "add"), hg.halo)
    else:
        partial = aggregate(g, hg.e2v, norm, "add")
        v = reduce_scatter_rows(partial, group)
    if aggr == "mean":
        v = v / hg.owned_vertex_degree(group).clamp(min=1).view(-1, 1)
    with _bn_scope(vv, group):
        return e2v_conv._mlp_act(e2v_conv.f_dec, v, p_out)


class _ShardedPmaE2V(torch.autograd.Function):
    """E->V attention pooling when a vertex's hyperedges live on several ranks.

    forward : local fused pass -> (o_loc, m_loc, l_loc) for every vertex; all-reduce(max) of m;
              rescale w = l_loc * exp(m_loc - m_glob); reduce-scatter(sum) of [o_loc * w | w] -> owned rows;
              out = numer / (l_glob + 1e-16).
    backward: stats {M = m_glob + log l_glob, delta = <out, gO>} on owned rows; all-gather [gO | stats];
              the ordinary one-pass source-major kernel with the GLOBAL stats gives exact gV / galpha for
              the local hyperedges (p_j = exp(a_j - M_t) is the global softmax weight).
    """

    @staticmethod
    def forward(ctx, V, alpha, hg, heads, slope, group, K, use_halo=False):
        halo = hg.halo if use_halo else None
        inc = hg.halo_e2v if halo is not None else hg.e2v
        in_dtype = V.dtype
        alpha = alpha.float()                                             # logits / softmax statistics are fp32 throughout
        o_loc, m_loc, l_loc = K.fwd(V, alpha, inc, heads, slope)          # [n_v_pad, d], [n_v_pad, H] x2 (halo: rows of the touched vertices)
        o_loc = o_loc.float()                                             # (bf16 storage: the cross-rank merge runs in fp32)
        n, d = o_loc.shape
        C = d // heads
        has = l_loc > 0
        neg_inf = torch.full_like(m_loc, float("-inf"))
        m_eff = torch.where(has, m_loc, neg_inf)
        if halo is not None:
            # boundary-vertex form of the same merge: every touched vertex has a local incidence (has is all true), the owners
            # take the maximum of what their vertices' ranks report and hand it back; then the weighted sums travel to the owners
            m_owned = halo._scatter_max(m_loc.contiguous())                            # [block, H]
            m_gc = halo.gather_narrow(m_owned)                                         # [needed, H]
            if hasattr(K, "merge_pack"):
                packed = K.merge_pack(o_loc, m_loc, l_loc, m_gc, heads)
            else:
                w = l_loc * torch.exp(m_loc - m_gc)
                packed = torch.cat([(o_loc.view(n, heads, C) * w.unsqueeze(-1)).view(n, d), w], dim=1)
            red = halo.scatter_add_narrow(packed)                                      # owned rows
            numer, l_g = red[:, :d], red[:, d:].contiguous()
            inv = torch.where(l_g > 0, 1.0 / (l_g + 1e-16), torch.zeros_like(l_g))
            out = (numer.reshape(-1, heads, C) * inv.unsqueeze(-1)).reshape(-1, d).contiguous()
            m_g_owned = torch.where(l_g > 0, m_owned, torch.zeros_like(l_g)).contiguous()
            ctx.save_for_backward(V, alpha, out, m_g_owned, l_g)
            ctx.hg, ctx.heads, ctx.slope, ctx.group, ctx.K = hg, heads, slope, group, K
            ctx.in_dtype, ctx.use_halo = in_dtype, True
            return out.to(in_dtype)
        ctx.use_halo = False
        m_g = m_eff.clone()
        if not _skip_collective(group):
            _all_reduce_(m_g, dist.ReduceOp.MAX, group)
        if hasattr(K, "merge_pack"):                 # one kernel; rows without local incidences have l_loc == 0 -> w = 0
            packed = K.merge_pack(o_loc, m_loc, l_loc, torch.where(has, m_g, m_loc), heads)
        else:
            w = torch.where(has, l_loc * torch.exp(m_eff - torch.where(has, m_g, m_eff)), torch.zeros_like(l_loc))
            packed = torch.cat([(o_loc.view(n, heads, C) * w.unsqueeze(-1)).view(n, d), w], dim=1)
        red = _reduce_scatter_rows(packed, group)                           # owned rows
        numer, l_g = red[:, :d], red[:, d:].contiguous()
        inv = torch.where(l_g > 0, 1.0 / (l_g + 1e-16), torch.zeros_like(l_g))
        out = (numer.reshape(-1, heads, C) * inv.unsqueeze(-1)).reshape(-1, d).contiguous()
        lo, hi = hg.v_lo, hg.v_hi
        m_g_owned = torch.where(l_g > 0, m_g[lo:hi], torch.zeros_like(l_g)).contiguous()
        ctx.save_for_backward(V, alpha, out, m_g_owned, l_g)
        ctx.hg, ctx.heads, ctx.slope, ctx.group, ctx.K = hg, heads, slope, group, K
        ctx.in_dtype = in_dtype
        return out.to(in_dtype)

    @staticmethod
    def backward(ctx, gout):
        V, alpha, out, m_g, l_g = ctx.saved_tensors
        hg, H, K = ctx.hg, ctx.heads, ctx.K
        gout = gout.contiguous().float()
        stats = K.bwd_stats(out, gout, m_g, l_g)                            # [n_own, H, 2]
        d = gout.shape[1]
        # [gout | stats | pad]: the gradient rows are gathered through a strided view of the packed table, so its row pitch
        # must keep them 16-byte aligned (d + 2H floats is not a multiple of 4 for odd head counts: H = 1 gives 66 -- found by
        # the two-rank GPU test, a 1-rank group never packs)
        pad = (-(d + 2 * H)) % 4
        parts = [gout, stats.reshape(gout.shape[0], 2 * H)] + ([gout.new_zeros(gout.shape[0], pad)] if pad else [])
        if ctx.use_halo:                              # the rows of the touched vertices only
            if _WIRE_DTYPE is None:
                full = hg.halo._gather(torch.cat(parts, dim=1))
            else:
                # narrow wire (opt-in): only the gradient rows are rounded; the softmax statistics {m + log l, delta} travel in fp32 in
                # their own all-to-all, as the forward's maxima do (Halo.gather_narrow) -- rounded to 2^-9, exp(a - M) is off by ~1 %
                # for |M| ~ 5 and forward and backward would use different maxima
                full = torch.cat([hg.halo._gather(gout), hg.halo.gather_narrow(torch.cat(parts[1:], dim=1))], dim=1)
            inc = hg.halo_e2v
        else:
            full = _all_gather_rows(torch.cat(parts, dim=1), ctx.group)
            inc = hg.e2v
        g_full = full[:, :d]                          # strided view: the kernels take a leading dimension, no copy
        stats_full = full[:, d:d + 2 * H].contiguous().view(-1, H, 2)
        gV, galpha = K.bwd_src(inc, alpha, V, g_full.to(V.dtype) if g_full.dtype != V.dtype else g_full, stats_full, ctx.slope)
        return gV, galpha.to(ctx.in_dtype) if galpha.dtype != ctx.in_dtype else galpha, None, None, None, None, None, None


def sharded_pma_layer(v2e_conv, e2v_conv, x_owned: Tensor, hg: ShardedHypergraph, dropout: float = 0.0,
                      training: bool = False, group=None, kernels=HipPmaKernels,
                      dropout_out: Optional[float] = None) -> Tensor:
    """One V->E->V AllSetTransformer layer (reference models.py:475-481 with PMA.forward, layers.py:120-157,
    inlined) on a hyperedge shard.  ``v2e_conv`` / ``e2v_conv``: :class:`allset_amd.layers.HalfNLHconv` with
    ``attention=True``.  V->E targets (hyperedges) are complete on their owner, so that direction is the local
    kernel behind an all-gather of [V | alpha]; E->V needs the cross-shard softmax merge above."""
    from .layers import _linear, _on_hip
    from . import dense as _dense

    def _lin_v(pma, t):
        if _on_hip(t) and _dense.fused_linear_supported(pma.lin_V.in_features, pma.lin_V.out_features):
            return _dense.fused_norm_linear(t, None, None, pma.lin_V.weight, pma.lin_V.bias)
        return _linear(pma.lin_V, t)
    K = kernels
    # ---- V -> E
    p = v2e_conv.prop
    H, C = p.heads, p.hidden
    # dense on owned vertices, then two all-gathers (no concatenate / split copies of the [n_V, d] table)
    V, alpha = p.project(x_owned)
    use_halo = hg.halo is not None and hg.halo_v2e is not None and not _skip_collective(group)
    if use_halo:                # only the rows of the vertices the local hyperedges touch travel (logits padded to 16-byte rows)
        Hp = (-alpha.shape[1]) % 4
        a4 = torch.cat([alpha, alpha.new_zeros(alpha.shape[0], Hp)], dim=1) if Hp else alpha
        V, alpha = halo_gather(V, hg.halo), halo_gather(a4, hg.halo)[:, :alpha.shape[1]]
        inc_v2e = hg.halo_v2e
    else:
        V, alpha = all_gather_rows(V, group), all_gather_rows(alpha, group)
        inc_v2e = hg.v2e
    if K is HipPmaKernels:      # targets complete on their owner: the module's own joint pooling + ln0 node applies
        e = p.pool_tail(V.contiguous(), alpha.contiguous(), inc_v2e, dropout if training else 0.0)[0]
    else:
        o = K.aggregate(V.contiguous(), alpha.contiguous(), inc_v2e, H, p.negative_slope)
        e = p.tail(o, _post=dropout if training else 0.0)            # relu -> dropout inside ln1's pass
    # ---- E -> V
    p = e2v_conv.prop
    H, C = p.heads, p.hidden
    V, alpha = p.project(e)                                                         # dense on owned hyperedges
    p_out = dropout if dropout_out is None else dropout_out
    if _skip_collective(group) and K is HipPmaKernels and hg.v_lo == 0 and hg.v_hi == hg.e2v.n_dst:
        return p.pool_tail(V.contiguous(), alpha.contiguous(), hg.e2v, p_out if training else 0.0)[0]
    if _skip_collective(group):      # one rank owns every vertex: the local fused pooling IS the answer, no (m,l,o) merge
        o = K.aggregate(V.contiguous(), alpha.contiguous(), hg.e2v, H, p.negative_slope)
        if hg.v_lo != 0 or hg.v_hi != o.shape[0]:          # (a no-op slice would still cost a zero-fill + copy backward)
            o = o[hg.v_lo:hg.v_hi]
    else:
        o = _ShardedPmaE2V.apply(V.contiguous(), alpha.contiguous(), hg, H, p.negative_slope, group, K, use_halo)
    return p.tail(o, _post=p_out if training else 0.0)


# ---------------------------------------------------------------------------------------------
# column-sharded aggregation: the exchange volume that does not grow with the number of ranks
# ---------------------------------------------------------------------------------------------
#
# The hyperedge-sharded layer above moves the whole [n_V, d] table through every rank on every exchange
# ((P-1)/P * n_V * d elements in, per rank): on a hypergraph without locality -- every vertex a boundary vertex,
# SURVEY section 7 -- that volume grows linearly with P under weak scaling and dominates the step beyond 2 ranks
# (DESIGN.md section 7).  The aggregation is independent per feature column, so the same layer can be cut the other
# way: every rank holds the FULL incidence (a few bytes per incidence; trivial next to 288 GB) and d/P of the
# columns of every row.  Dense work stays row-sharded (owner computes n/P rows, full d); around each aggregation the
# activations change layout with one all-to-all,
#       [n/P, d] (my rows, all columns)  <->  [n, d/P] (all rows, my columns),
# which sends n/P * d/P elements to each peer: (P-1)/P * n/P * d per rank, 1/P of the all-gather's volume, and
# constant per rank under weak scaling.  Every target row is complete on every rank, so max / min / mean and PMA's
# segment softmax need no cross-rank merge: the ordinary differentiable local aggregates run on the column slice
# (PMA: whole heads per rank while P <= H, a fraction of one head beyond -- the logits of that head are sent along).
# Work per rank is identical by construction, so power-law hyperedge sizes (configs[4]) need no bin packing.
# The price: gathers of d/P-wide rows; below 32 fp32 columns (128 B) a gathered row is less than a cache line and the
# gather kernels lose efficiency (measured: profiles/r01_colshard_kernels.txt).

def _pack(x: Tensor, w: int) -> Tensor:
    """[rows, P*dc] -> [P, rows, dc] contiguous (what an all-to-all sends); one HIP copy kernel on the device."""
    from . import ops
    if ops.block_transpose_supported(x, w, True):
        return ops.block_transpose(x, w, True)
    return x.view(x.shape[0], w, x.shape[1] // w).permute(1, 0, 2).contiguous()


def _unpack(x: Tensor) -> Tensor:
    """[P, rows, dc] (what an all-to-all received) -> [rows, P*dc]."""
    from . import ops
    if ops.block_transpose_supported(x, x.shape[0], False):
        return ops.block_transpose(x, x.shape[0], False)
    return x.permute(1, 0, 2).reshape(x.shape[1], x.shape[0] * x.shape[2])


def _rows_to_cols(x: Tensor, group=None) -> Tensor:
    """[n/P, d] -> [n, d/P]: rank r ends with columns [r*d/P, (r+1)*d/P) of every rank's rows, in rank order."""
    if _skip_collective(group):
        return x
    w = _world(group)
    r, d = x.shape
    if d % w:
        raise ValueError(f"column sharding needs the width ({d}) to be a multiple of the world size ({w})")
    send = _narrow(_pack(x, w))                                            # [P, n/P, d/P]: chunk j goes to rank j
    recv = torch.empty_like(send)
    _all_to_all_single(recv, send, group)
    return recv.view(w * r, d // w).to(x.dtype)                            # chunk i = rank i's rows: already row-major


def _cols_to_rows(x: Tensor, group=None) -> Tensor:
    """[n, d/P] -> [n/P, d]: the inverse layout change."""
    if _skip_collective(group):
        return x
    w = _world(group)
    n, dc = x.shape
    if n % w:
        raise ValueError(f"column sharding needs the (padded) row count ({n}) to be a multiple of the world size ({w})")
    send = _narrow(x.contiguous())                                         # rows of block j (my columns) go to rank j
    recv = torch.empty_like(send)
    _all_to_all_single(recv, send, group)
    return _unpack(recv.view(w, n // w, dc).to(x.dtype))


def _exchange_blocks(x: Tensor, group=None) -> Tensor:
    """The all-to-all of :func:`_rows_to_cols` / :func:`_cols_to_rows` on a tensor that is ALREADY in the exchange layout: ``x``
    [P * r, c] is P blocks of r rows, block j goes to rank j, block i of the result came from rank i.  Row block -> column slice:
    ``x`` is column-blocked (block j = columns [j c, (j+1) c) of my r rows, what a fused Linear writes with ``out_cb = c``) and
    the result is the row-major [n, c] table of my column slice.  Column slice -> row block: ``x`` is that table and the result is
    my rows column-blocked (what a fused Linear reads with ``in_cb = c``).  No pack / unpack pass on either side."""
    if _skip_collective(group):
        return x
    send = _narrow(x.contiguous())
    recv = torch.empty_like(send)
    _all_to_all_single(recv, send, group)
    return recv.to(x.dtype)


class _ExchangeBlocks(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return _exchange_blocks(x, group)

    @staticmethod
    def backward(ctx, g):                          # the exchange is its own transpose: block (i <- j) back along (j <- i)
        return _exchange_blocks(g, ctx.group), None


def exchange_blocks(x: Tensor, group=None) -> Tensor:
    return _ExchangeBlocks.apply(x, group)


def _blocked_exchange_width(x_owned: Tensor, w: int, *mlps) -> int:
    """Column-block width d / P if every MLP around the two exchanges of the column-sharded layer can read / write the exchange
    layout directly (fused Linear kernels of csrc/fused_fwd2.hip / fused_bwd4.hip: fp32, width 128); 0 = pack / unpack path.
    ``ALLSET_BLOCKED_EXCHANGE=0`` forces the pack / unpack path."""
    if os.environ.get("ALLSET_BLOCKED_EXCHANGE", "1") == "0" or w <= 1:
        return 0
    from .layers import MLP
    if not all(isinstance(m, MLP) for m in mlps):
        return 0
    d = mlps[0].lins[-1].out_features
    if d % w or any(m.lins[0].in_features != d or m.lins[-1].out_features != d for m in mlps[1:]):
        return 0
    cb = d // w
    return cb if all(m.blockable(x_owned, cb) for m in mlps) else 0


class _RowsToCols(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return _rows_to_cols(x, group)

    @staticmethod
    def backward(ctx, g):
        return _cols_to_rows(g, ctx.group), None


class _ColsToRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return _cols_to_rows(x, group)

    @staticmethod
    def backward(ctx, g):
        return _rows_to_cols(g, ctx.group), None


def rows_to_cols(x: Tensor, group=None) -> Tensor:
    """Owned rows, all columns -> all rows, owned columns (one all-to-all); backward = :func:`cols_to_rows`."""
    return _RowsToCols.apply(x, group)


def cols_to_rows(x: Tensor, group=None) -> Tensor:
    """All rows, owned columns -> owned rows, all columns (one all-to-all); backward = :func:`rows_to_cols`."""
    return _ColsToRows.apply(x, group)


class ColumnShardedHypergraph:
    """One rank's view for column-sharded aggregation: the FULL incidence (``edge_index`` int64 [2, nnz], row 0 global
    vertex ids, row 1 global 0-based hyperedge ids; identical on every rank) plus the row blocks this rank owns for the
    dense work -- vertices ``v_lo:v_hi`` of the padded vertex range, hyperedges ``e_lo:e_hi`` of the padded hyperedge
    range (blocks padded to a multiple of ``chunks``, the chunk count of the overlapped exchange).  ``norm``:
    per-incidence weights in ``edge_index`` order, or None.

    ``row_groups = R > 1`` (round 5): the HYBRID partition -- the world is R target groups x C = world / R column groups, rank
    r = a C + b.  Rank (a, b) aggregates, for the TARGETS of group a (hyperedges ``a n_E/R .. (a+1) n_E/R`` in V->E, vertices
    likewise in E->V), the d / C columns of slice b: it holds the incidences of those targets only (half of them at R = 2) and gathers
    rows of d / C columns -- 128 bytes at d = 128, C = 4, where the pure column partition at 8 ranks gathers 64-byte rows at half
    the fabric's efficiency.  Exchanges per aggregation: rows -> columns as ONE all-to-all over the whole world in
    which every rank sends column slice b' of its rows to BOTH (all R) ranks (., b') -- every target group needs every source row,
    and a direct send loads all world - 1 point-to-point links evenly, where an all-gather across a 2-rank group would put the
    whole table on one xGMI link --, the local aggregation, columns -> rows inside the column group (an all-to-all over the C ranks
    of group a: ``col_group``).  Backward: the transposes; the R copies' gradients are summed on arrival (fixed order).  Groups from
    :func:`hybrid_groups` (every rank must create all of them, in the same order)."""

    def __init__(self, edge_index: Tensor, n_v: int, n_e: int, world: int, rank: int, norm: Optional[Tensor] = None,
                 chunks: int = 1, row_groups: int = 1, col_group=None, gather_group=None):
        self.edge_index = edge_index
        self.n_v, self.n_e, self.world, self.rank = int(n_v), int(n_e), int(world), int(rank)
        self.row_groups = max(int(row_groups), 1)
        if self.world % self.row_groups:
            raise ValueError(f"hybrid partition: the world size ({world}) must be a multiple of row_groups ({row_groups})")
        self.col_world = self.world // self.row_groups
        self.group_a, self.slice_b = self.rank // self.col_world, self.rank % self.col_world
        self.col_group, self.gather_group = col_group, gather_group
        self.chunks = max(int(chunks), 1) if self.row_groups == 1 else 1     # owned blocks are padded to a multiple of this (overlapped exchange)

        def block(n):
            per = (n + world - 1) // world
            per = (per + self.chunks - 1) // self.chunks * self.chunks
            return rank * per, (rank + 1) * per, per * world
        self.v_lo, self.v_hi, self.n_v_pad = block(self.n_v)
        self.e_lo, self.e_hi, self.n_e_pad = block(self.n_e)
        self.norm = norm
        self.v2e = None
        self.e2v = None
        self.ids_v2e = self.ids_e2v = None        # hybrid: positions (in edge_index) of the incidences each direction keeps

    @property
    def hybrid(self) -> bool:
        return self.row_groups > 1

    def target_slices(self):
        """Hybrid: ``((ei_v2e, n_dst, ids), (ei_e2v, n_dst, ids))`` -- per direction the [2, k] (source id, LOCAL target id) list of the
        incidences whose target lies in this rank's target group, the group's target count and the positions of those incidences in
        ``edge_index`` (what routes ``norm`` / ``Importance``)."""
        R, a = self.row_groups, self.group_a
        ne, nv = self.n_e_pad // R, self.n_v_pad // R
        v, e = self.edge_index[0], self.edge_index[1]
        ke = torch.nonzero((e >= a * ne) & (e < (a + 1) * ne)).reshape(-1)
        kv = torch.nonzero((v >= a * nv) & (v < (a + 1) * nv)).reshape(-1)
        return ((torch.stack([v[ke], e[ke] - a * ne]), ne, ke), (torch.stack([e[kv], v[kv] - a * nv]), nv, kv))

    def norm_of(self, norm: Optional[Tensor], direction: str) -> Optional[Tensor]:
        """``norm`` (edge_index order) restricted to the incidences direction 'v2e' / 'e2v' keeps on this rank (hybrid); else as is."""
        if norm is None or not self.hybrid:
            return norm
        ids = self.ids_v2e if direction == "v2e" else self.ids_e2v
        return norm[ids]

    def build_incidences(self) -> "ColumnShardedHypergraph":
        from .incidence import Incidence
        if self.hybrid:
            (ei1, n1, k1), (ei2, n2, k2) = self.target_slices()
            self.v2e = Incidence.from_edge_index(ei1.contiguous(), n_src=self.n_v_pad, n_dst=n1)
            self.e2v = Incidence.from_edge_index(ei2.contiguous(), n_src=self.n_e_pad, n_dst=n2)
            self.ids_v2e, self.ids_e2v = k1, k2
            return self
        self.v2e = Incidence.from_edge_index(self.edge_index, n_src=self.n_v_pad, n_dst=self.n_e_pad)
        self.e2v = self.v2e.reversed(n_dst=self.n_v_pad)
        return self


def hybrid_groups(world: int, row_groups: int, rank: int):
    """``(col_group, None)`` of ``rank`` for the hybrid partition (rank = a C + b): the C ranks of target group a (the second
    slot was the cross-group gather group of the first design; the spread is a world all-to-all now).  Collective: EVERY rank
    creates every group, in the same order (``dist.new_group``)."""
    R, C = int(row_groups), world // int(row_groups)
    col = None
    for a in range(R):
        g = dist.new_group(ranks=[a * C + b for b in range(C)])
        if rank // C == a:
            col = g
    return col, None


class _SpreadBlocks(torch.autograd.Function):
    """Hybrid partition, rows -> columns: ``x`` [C * r, c] = C blocks of this rank's r rows (block b' = column slice b' of them:
    what ``_pack`` or a fused Linear with ``out_cb = c`` produces); every block goes to all R ranks (a', b') in ONE all-to-all over
    ``group`` (the world); result [world * r, c] = column slice ``b`` of EVERY rank's rows, in rank (= global row block) order.
    Backward: the transpose all-to-all, the R gradient copies of a block summed in a fixed order."""

    @staticmethod
    def forward(ctx, x, R, group):
        ctx.R, ctx.group, ctx.shape = R, group, x.shape
        send = _narrow(x.contiguous()).repeat(R, 1)                  # [R][C * r, c]: destination rank a' C + b' gets block b'
        recv = torch.empty_like(send)
        _all_to_all_single(recv, send, group)
        return recv.to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        send = _narrow(g.contiguous())
        recv = torch.empty_like(send)
        _all_to_all_single(recv, send, ctx.group)                      # piece from rank (a', b') = its gradient of my rows' slice b'
        return recv.to(g.dtype).view((ctx.R,) + tuple(ctx.shape)).sum(dim=0), None, None


def spread_blocks(x: Tensor, row_groups: int, group=None) -> Tensor:
    return _SpreadBlocks.apply(x, int(row_groups), group)


class _PackCols(torch.autograd.Function):
    """[r, C * c] -> [C * r, c] (block b' = column slice b'); backward = the inverse copy."""

    @staticmethod
    def forward(ctx, x, C):
        ctx.C = C
        return _pack(x, C).reshape(C * x.shape[0], x.shape[1] // C)

    @staticmethod
    def backward(ctx, g):
        C = ctx.C
        return _unpack(g.contiguous().view(C, g.shape[0] // C, g.shape[1])), None


def head_slots(heads: int, world: int) -> Tuple[int, list]:
    """How PMA's heads fall on the column slices.  Returns (heads per rank, for every rank in order the global head
    ids of its slice).  P <= H: H/P whole heads per rank; P > H: P/H ranks share one head, each holding C*H/P of its
    columns (the softmax statistics of that head are then computed on each of them -- they only need the logits)."""
    if heads % world == 0:
        hl = heads // world
        return hl, [r * hl + j for r in range(world) for j in range(hl)]
    if world % heads == 0:
        per = world // heads
        return 1, [r // per for r in range(world)]
    raise ValueError(f"column sharding of PMA needs heads ({heads}) and world size ({world}) to divide one another")


def _chunking(hg, x_owned: Tensor, chunks: int, world: int, group, *convs) -> int:
    """Chunk count for the overlapped exchange: divides both owned row counts; 1 (plain path) when there is no exchange
    or with BatchNorm in a conv (its batch statistics are not row-wise)."""
    if _skip_collective(group) or chunks <= 1 or _has_batchnorm(*convs):
        return 1
    import math
    return pipeline_chunks(math.gcd(x_owned.shape[0], hg.n_e_pad // world), chunks)


def colsharded_deepsets_layer(v2e_conv, e2v_conv, x_owned: Tensor, hg: ColumnShardedHypergraph, aggr: str = "add",
                              dropout: float = 0.0, training: bool = False, group=None,
                              aggregate: Callable = _hip_deepsets, chunks: int = 1, norm: Optional[Tensor] = None,
                              dropout_out: Optional[float] = None) -> Tensor:
    """The layer of :func:`sharded_deepsets_layer` with column-sharded aggregation: four all-to-alls, no reduction
    across ranks; every ``aggr`` of the reference (layers.py:641-656) is the plain local one on the column slice.
    ``chunks`` > 1: the owned rows go through the MLPs in that many chunks, each chunk's exchange overlapping the
    others' dense work (chunk k of rank i lands at rows i*n/P + k*rc of the column table: the natural row order)."""
    if aggr not in ("add", "sum", "mean", "max", "min"):
        raise ValueError(f"aggr {aggr!r}")
    norm = hg.norm if norm is None else norm
    p_out = dropout if dropout_out is None else dropout_out
    hyb = bool(getattr(hg, "hybrid", False))
    cg = hg.col_group if hyb else group             # the ranks the columns -> rows all-to-all spans: the world, or one target group
    norm1, norm2 = (hg.norm_of(norm, "v2e"), hg.norm_of(norm, "e2v")) if hyb else (norm, norm)
    w = 1 if _skip_collective(cg) else _world(cg)
    K = 1 if hyb else _chunking(hg, x_owned, chunks, w, group, v2e_conv, e2v_conv)
    if hyb:        # rows -> columns: one world all-to-all, every block to all R target groups (every group needs every source row)
        to_cols = lambda t: spread_blocks(_PackCols.apply(t, w), hg.row_groups, group)
        to_cols_blocked = lambda t: spread_blocks(t, hg.row_groups, group)
    else:
        to_cols = lambda t: rows_to_cols(t, cg)
        to_cols_blocked = lambda t: exchange_blocks(t, cg)
    enc1 = lambda t: v2e_conv._mlp_act(v2e_conv.f_enc, t, v2e_conv.dropout)
    mid = lambda t: e2v_conv._mlp_act(e2v_conv.f_enc, v2e_conv._mlp_act(v2e_conv.f_dec, t, dropout), e2v_conv.dropout)
    dec2 = lambda t: e2v_conv._mlp_act(e2v_conv.f_dec, t, p_out)
    cb = _blocked_exchange_width(x_owned, w, v2e_conv.f_enc, v2e_conv.f_dec, e2v_conv.f_enc, e2v_conv.f_dec) if K == 1 else 0
    if cb:
        # repack-free: the MLPs on either side of each all-to-all write / read its buffer layout themselves
        h = v2e_conv._mlp_act(v2e_conv.f_enc, x_owned, v2e_conv.dropout, out_cb=cb)                   # [P * n_V/P, d/P] blocked
        e = exchange_blocks(aggregate(to_cols_blocked(h), hg.v2e, norm1, aggr), cg)                   # [P * n_E/P, d/P] blocked
        g = e2v_conv._mlp_act(e2v_conv.f_enc, v2e_conv._mlp_act(v2e_conv.f_dec, e, dropout, in_cb=cb), e2v_conv.dropout, out_cb=cb)
        v = exchange_blocks(aggregate(to_cols_blocked(g), hg.e2v, norm2, aggr), cg)
        return e2v_conv._mlp_act(e2v_conv.f_dec, v, p_out, in_cb=cb)
    if K == 1:
        vv, ve = _valid_rows(hg.v_lo, hg.v_hi, hg.n_v), _valid_rows(hg.e_lo, hg.e_hi, hg.n_e)     # real rows of the two owned blocks
        with _bn_scope(vv, group):
            h = enc1(x_owned)
        e = cols_to_rows(aggregate(to_cols(h), hg.v2e, norm1, aggr), cg)      # [n_E/P, d]
        with _bn_scope(ve, group):
            g = mid(e)
        v = cols_to_rows(aggregate(to_cols(g), hg.e2v, norm2, aggr), cg)
        with _bn_scope(vv, group):
            return dec2(v)
    rc_v, rc_e = x_owned.shape[0] // K, hg.n_e_pad // w // K
    cbk = _blocked_exchange_width(x_owned[:rc_v], w, v2e_conv.f_enc, v2e_conv.f_dec, e2v_conv.f_enc, e2v_conv.f_dec)
    if cbk:
        # chunked AND repack-free: every chunk's MLPs write / read the [P][rc][d/P] buffers of its own all-to-all
        enc1 = lambda t: v2e_conv._mlp_act(v2e_conv.f_enc, t, v2e_conv.dropout, out_cb=cbk)
        mid = lambda t: e2v_conv._mlp_act(e2v_conv.f_enc, v2e_conv._mlp_act(v2e_conv.f_dec, t, dropout, in_cb=cbk), e2v_conv.dropout,
                                          out_cb=cbk)
        dec2 = lambda t: e2v_conv._mlp_act(e2v_conv.f_dec, t, p_out, in_cb=cbk)
    bl = bool(cbk)
    (hc,) = _stage(torch.split(x_owned, rc_v), enc1, rc_v, K, w, group, bl)
    ec = aggregate(hc, hg.v2e, norm, aggr)
    (gc,) = _stage(_unstage(ec, K, w, group, bl), lambda get: mid(get()), rc_e, K, w, group, bl)
    vc = aggregate(gc, hg.e2v, norm, aggr)
    return torch.cat([dec2(get()) for get in _unstage(vc, K, w, group, bl)])


def colsharded_pma_layer(v2e_conv, e2v_conv, x_owned: Tensor, hg: ColumnShardedHypergraph, dropout: float = 0.0,
                         training: bool = False, group=None, kernels=HipPmaKernels, chunks: int = 1,
                         dropout_out: Optional[float] = None) -> Tensor:
    """The layer of :func:`sharded_pma_layer` with column-sharded pooling: per direction one all-to-all of the values,
    one of the (few) logit columns each slice needs, the ordinary local fused pooling, one all-to-all back.
    ``chunks``: as in :func:`colsharded_deepsets_layer`."""
    hyb = bool(getattr(hg, "hybrid", False))
    cg = hg.col_group if hyb else group
    w = 1 if _skip_collective(cg) else _world(cg)
    to_cols = (lambda t: spread_blocks(_PackCols.apply(t, w), hg.row_groups, group)) if hyb else (lambda t: rows_to_cols(t, cg))
    K = 1 if hyb else _chunking(hg, x_owned, chunks, w, group, v2e_conv, e2v_conv)
    post = dropout if training else 0.0
    post_out = (dropout if dropout_out is None else dropout_out) if training else 0.0

    def project(p):
        hl, slots = head_slots(p.heads, w)
        per = 1 if slots == list(range(p.heads)) else w // p.heads          # ranks sharing one head (P > H)

        def f(t):
            V, alpha = p.project(t)                                  # [rows, H*C], [rows, H]: dense, owned rows
            if per > 1:
                # a shared head's logits go to each sharer: every column `per` times in a row.  As expand + reshape, whose
                # backward is a sum over the copies -- `alpha[:, idx]` with an index tensor costs a sort-based index_put in
                # backward (2.8 ms per direction at 1M rows, the third-largest kernel of the N = 8 step)
                alpha = alpha.unsqueeze(2).expand(-1, p.heads, per).reshape(alpha.shape[0], p.heads * per)
            return V, alpha.contiguous()
        return f, hl

    p1, p2 = v2e_conv.prop, e2v_conv.prop
    f1, hl1 = project(p1)
    f2, hl2 = project(p2)
    if K == 1:
        def pool(p, f, hl, t, inc, pp):
            V, alpha = f(t)
            o = kernels.aggregate(to_cols(V).contiguous(), to_cols(alpha).contiguous(), inc, hl, p.negative_slope)
            return p.tail(cols_to_rows(o, cg), _post=pp)
        return pool(p2, f2, hl2, pool(p1, f1, hl1, x_owned, hg.v2e, post), hg.e2v, post_out)
    rv, re = x_owned.shape[0] // K, hg.n_e_pad // w // K
    Vc, ac = _stage(torch.split(x_owned, rv), f1, rv, K, w, group)
    oc = kernels.aggregate(Vc, ac, hg.v2e, hl1, p1.negative_slope)
    Vc, ac = _stage(_unstage(oc, K, w, group), lambda get: f2(p1.tail(get(), _post=post)), re, K, w, group)
    oc = kernels.aggregate(Vc, ac, hg.e2v, hl2, p2.negative_slope)
    return torch.cat([p2.tail(get(), _post=post_out) for get in _unstage(oc, K, w, group)])


# ---- the same exchanges, chunked and overlapped with the row-sharded dense work ---------------------------------
#
# Both neighbours of an all-to-all are row-wise MLPs, so the owned rows are cut into K chunks and chunk k's exchange
# runs (on RCCL's stream) while chunk k+1 is still in -- or chunk k-1 already past -- the MLP.  The autograd wiring
# keeps that true in backward, where the engine would otherwise serialise "exchange k, MLP k, exchange k-1, ...":
#   producer side   MLP chunk --_SendRowsChunk--> token  ... all tokens --_AssembleCols--> [n, d/P]
#       forward : every chunk's all-to-all is issued as soon as the chunk exists; _AssembleCols waits for all of them.
#       backward: _AssembleCols issues ALL K reverse all-to-alls at once; each _SendRowsChunk.backward only waits for its own.
#   consumer side   [n, d/P] --_ScatterCols--> K tokens ... token --_RecvRowsChunk--> chunk for the MLP
#       forward : all K all-to-alls issued at once, each chunk waits for its own; backward: each chunk issues its reverse
#       exchange as soon as its MLP backward is done, _ScatterCols.backward waits for all.
# Tokens are zero-size tensors that only carry the graph edges; the data moves through the shared ``_Pipe``.

class _PendingCopy:
    """gloo (tests) has no list all-to-all: exchange through a contiguous buffer, copy out after the wait."""

    def __init__(self, work, tmp, views):
        self.work, self.tmp, self.views = work, tmp, views

    def wait(self):
        self.work.wait()
        for t, v in zip(self.tmp.unbind(0), self.views):
            v.copy_(t)


def _a2a_async(out_views, in_views, group):
    """All-to-all of P equal contiguous pieces, asynchronous; ``wait()`` orders the current stream behind it."""
    if dist.get_backend(group) == "nccl":
        return dist.all_to_all(list(out_views), list(in_views), group=group, async_op=True)
    send = torch.stack(list(in_views))
    if send.is_cuda:                      # host-staged (two ranks on one GPU): the exchange itself runs on host memory
        send = send.cpu()
    tmp = torch.empty_like(send)
    return _PendingCopy(dist.all_to_all_single(tmp, send, group=group, async_op=True), tmp, list(out_views))


class _Pipe:
    def __init__(self, K: int, world: int, rc: int, group, blocked: bool = False):
        self.K, self.P, self.rc, self.group = K, world, rc, group
        # blocked: the row-side tensors of a chunk are ALREADY in the exchange layout [P][rc][dc] (column-blocked operands of
        # the fused Linear kernels, ``_blocked_exchange_width``): no pack before a send, no unpack after a receive
        self.blocked = blocked
        self.work = [None] * K
        self.bwork = [None] * K
        self.buf = [None] * K          # per-chunk [P, rc, dc] buffers (kept alive until their exchange was waited for)
        self.bbuf = [None] * K
        self.full = None               # [n, dc] assembled forward table / source of the scatter
        self.bfull = None              # its gradient

    def blocks(self, full: Tensor, k: int):
        """The P contiguous [rc, dc] pieces of chunk k inside a [P*K*rc, dc] table (rank-major, chunk, row)."""
        v = full.view(self.P, self.K, self.rc, full.shape[1])
        return [v[i, k] for i in range(self.P)]


class _SendRowsChunk(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, pipe, k):
        if pipe.blocked:                                   # [P * rc, dc]: block j = columns of slice j of my rc rows
            rc, dc = h.shape[0] // pipe.P, h.shape[1]
            send = h.view(pipe.P, rc, dc)
        else:
            rc, d = h.shape
            dc = d // pipe.P
            send = _pack(h, pipe.P)
        if pipe.full is None:
            pipe.full = h.new_empty((pipe.P * pipe.K * rc, dc))
        pipe.buf[k] = send
        pipe.work[k] = _a2a_async(pipe.blocks(pipe.full, k), send.unbind(0), pipe.group)
        ctx.pipe, ctx.k = pipe, k
        return h.new_empty(0)

    @staticmethod
    def backward(ctx, _g):
        pipe, k = ctx.pipe, ctx.k
        pipe.bwork[k].wait()
        recv, pipe.bbuf[k], pipe.bwork[k] = pipe.bbuf[k], None, None
        return (recv.view(pipe.P * pipe.rc, recv.shape[2]) if pipe.blocked else _unpack(recv)), None, None


class _AssembleCols(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pipe, *tokens):
        for k in range(pipe.K):
            pipe.work[k].wait()
            pipe.work[k] = pipe.buf[k] = None
        ctx.pipe = pipe
        ctx.token_like = tokens[0]
        full, pipe.full = pipe.full, None
        return full

    @staticmethod
    def backward(ctx, g):
        pipe = ctx.pipe
        g = g.contiguous()
        pipe.bfull = g                                               # alive until the last chunk was received
        for k in reversed(range(pipe.K)):                            # the engine runs the chunks last-created first
            recv = g.new_empty((pipe.P, pipe.rc, g.shape[1]))
            pipe.bbuf[k] = recv
            pipe.bwork[k] = _a2a_async(recv.unbind(0), pipe.blocks(g, k), pipe.group)
        return (None,) + tuple(ctx.token_like.new_zeros(0) for _ in range(pipe.K))


class _ScatterCols(torch.autograd.Function):
    @staticmethod
    def forward(ctx, full, pipe):
        full = full.contiguous()
        pipe.full = full
        for k in range(pipe.K):
            recv = full.new_empty((pipe.P, pipe.rc, full.shape[1]))
            pipe.buf[k] = recv
            pipe.work[k] = _a2a_async(recv.unbind(0), pipe.blocks(full, k), pipe.group)
        ctx.pipe = pipe
        return tuple(full.new_empty(0) for _ in range(pipe.K))

    @staticmethod
    def backward(ctx, *_g):
        pipe = ctx.pipe
        for k in range(pipe.K):
            pipe.bwork[k].wait()
            pipe.bwork[k] = pipe.bbuf[k] = None
        g, pipe.bfull, pipe.full = pipe.bfull, None, None
        return g, None


class _RecvRowsChunk(torch.autograd.Function):
    @staticmethod
    def forward(ctx, token, pipe, k):
        pipe.work[k].wait()
        recv, pipe.buf[k], pipe.work[k] = pipe.buf[k], None, None
        ctx.pipe, ctx.k = pipe, k
        ctx.token_like = token
        return recv.view(pipe.P * pipe.rc, recv.shape[2]) if pipe.blocked else _unpack(recv)

    @staticmethod
    def backward(ctx, g):
        pipe, k = ctx.pipe, ctx.k
        if pipe.blocked:
            g = g.contiguous()
            rc, dc = g.shape[0] // pipe.P, g.shape[1]
            send = g.view(pipe.P, rc, dc)
        else:
            rc, d = g.shape
            dc = d // pipe.P
            send = _pack(g, pipe.P)
        if pipe.bfull is None:
            pipe.bfull = g.new_empty((pipe.P * pipe.K * rc, dc))
        pipe.bbuf[k] = send
        pipe.bwork[k] = _a2a_async(pipe.blocks(pipe.bfull, k), send.unbind(0), pipe.group)
        return ctx.token_like.new_zeros(0), None, None


def auto_chunks(rows: int) -> int:
    """Chunk count of the overlapped exchange for ``rows`` owned rows.  Every chunk repeats the layer's ~25 dense launches
    and adds 2 collectives per exchange; measured on one GPU (1-rank RCCL group, profiles/r01_colshard_chunks.txt) that
    costs 3.3 ms per step at 4 x 250k fp32 rows and 7 ms at 4 x 62k bf16 rows (launch-bound), against the dense time it
    can hide an exchange behind (~7 ms per step at 1M x 128): worth it only while a chunk keeps >= 250k rows."""
    return max(1, min(4, int(rows) // 250_000))


def pipeline_chunks(rows: int, want: int) -> int:
    """Largest chunk count <= ``want`` that divides the owned row count (1 = no pipelining)."""
    for k in range(max(int(want), 1), 0, -1):
        if rows % k == 0:
            return k
    return 1


def _stage(items, fn, rc: int, K: int, world: int, group, blocked: bool = False):
    """``fn(item)`` (row-wise; one tensor or a tuple of tensors [rc, width] per chunk -- or, ``blocked``, [P*rc, width/P] in the
    exchange layout) for every chunk, each result handed to its all-to-all at once.  Returns the assembled [P*K*rc, width/P]
    table(s), rows in global order."""
    pipes, tokens = None, None
    for k, item in enumerate(items):
        ys = fn(item)
        ys = ys if isinstance(ys, tuple) else (ys,)
        if pipes is None:
            pipes = [_Pipe(K, world, rc, group, blocked) for _ in ys]
            tokens = [[] for _ in ys]
        for pipe, toks, y in zip(pipes, tokens, ys):
            toks.append(_SendRowsChunk.apply(y.contiguous(), pipe, k))
    return tuple(_AssembleCols.apply(pipe, *toks) for pipe, toks in zip(pipes, tokens))


def _unstage(table: Tensor, K: int, world: int, group, blocked: bool = False):
    """[P*K*rc, dc] -> K thunks; thunk k returns the owned row chunk [rc, P*dc] (``blocked``: [P*rc, dc], the exchange layout
    as it arrived) once its all-to-all (all K are already in flight) has arrived."""
    rc = table.shape[0] // (world * K)
    pipe = _Pipe(K, world, rc, group, blocked)
    tokens = _ScatterCols.apply(table, pipe)
    return [(lambda tok=tok, k=k: _RecvRowsChunk.apply(tok, pipe, k)) for k, tok in enumerate(tokens)]


def _has_batchnorm(*mods) -> bool:
    return any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for mod in mods for m in mod.modules())


def exchange_bytes_per_rank(mode: str, world: int, n_v: int, n_e: int, d: int, elem: int = 4) -> int:
    """Bytes one rank RECEIVES per V->E->V layer, forward + backward (the dense case: every vertex a boundary vertex).
    ``rows`` (hyperedge shards): all-gather + reduce-scatter of [n_V, d] each way; ``columns``: four all-to-alls each
    way over [n_V, d] and [n_E, d] slices."""
    if world <= 1:
        return 0
    f = (world - 1) / world
    if mode == "rows":
        return int(4 * f * n_v * d * elem)
    if mode == "columns":
        return int(4 * f * (n_v + n_e) / world * d * elem)
    if mode.startswith("hybrid"):       # hybridRxC: per aggregation the spread (rows from all other ranks, d / C columns) + the return
        R = int(mode[len("hybrid"):].split("x")[0])          # inside the column group; forward + backward, both directions
        C = world // R
        spread = f * (n_v + n_e) * (d / C) * elem
        back = (C - 1) / C * (n_v + n_e) / world * d * elem
        return int(2 * (spread + back))
    raise ValueError(mode)


def exchange_bytes_per_link(mode: str, world: int, n_v: int, n_e: int, d: int, elem: int = 4) -> int:
    """Bytes on the BUSIEST point-to-point link of a rank per layer (xGMI: one link per pair of GPUs): rows / columns load all
    ``world - 1`` links evenly; the hybrid partition's spread does too, its return only the C - 1 links of the column group."""
    if world <= 1:
        return 0
    if mode.startswith("hybrid"):
        R = int(mode[len("hybrid"):].split("x")[0])
        C = world // R
        per_pair_spread = (n_v + n_e) / world * (d / C) * elem
        per_pair_back = (n_v + n_e) / world * (d / C) * elem
        return int(2 * (per_pair_spread + per_pair_back))
    return exchange_bytes_per_rank(mode, world, n_v, n_e, d, elem) // (world - 1)


def choose_sharding(world: int, d: int, heads: Optional[int] = None, elem: int = 4) -> str:
    """``"rows"`` or ``"columns"`` for a hypergraph WITHOUT locality (the synthetic benchmarks; every vertex a boundary
    vertex).  Per link and step the row scheme moves 4*n_loc*d*elem bytes whatever P is, the column scheme 8/P of
    that: equal at P = 2, half at P = 4, a quarter at P = 8; the column scheme pays for it with d/P-wide gathers
    (no loss down to 128-byte rows, ~2.3x slower aggregation at 64-byte rows: profiles/r01_colshard_kernels.txt) and
    two layout copies per exchange.  With the measured per-rank compute (profiles/r01_sim_rank.txt) the column scheme
    wins from P = 4 on; at P = 2 the two move the same bytes over the one link and the column scheme is taken for its
    overlapped exchange (the row scheme's collectives run back to back with the compute).  A hypergraph whose
    partitions have few boundary vertices wants the row scheme regardless -- pass the mode explicitly there."""
    if world < 2 or d % world:
        return "rows"
    dc = d // world
    if dc * elem < 64 or dc % (16 // elem):            # below one 64-byte sector / not 16-byte packets: gather kernels degrade
        return "rows"
    if heads is not None:
        try:
            hl, _ = head_slots(heads, world)
        except ValueError:
            return "rows"
        if dc % hl or (dc // hl) % (16 // elem):
            return "rows"
    return "columns"


def _rank_dropout(x: Tensor, p: float, training: bool) -> Tensor:
    """``F.dropout`` whose mask differs between ranks.  The ranks of a sharded job seed torch identically (replicated initial
    weights), so torch's device generator would drop the SAME positions of every rank's row block; device fp32 tensors go through
    the library's hash dropout, whose seed carries the rank (``dense._draw_seed``), anything else through a generator forked
    per rank."""
    if not training or p <= 0.0:
        return x
    if x.is_cuda and x.dtype == torch.float32:
        from . import dense
        return _Dropout.apply(x, float(p), dense._draw_seed())
    from . import dense
    gen = torch.Generator(device=x.device)
    gen.manual_seed(dense._draw_seed())                     # a draw from torch's CPU generator with the rank mixed in
    keep = (torch.rand(x.shape, device=x.device, generator=gen) >= p).to(x.dtype)
    return x * keep / (1.0 - p)


class _Dropout(torch.autograd.Function):
    """y = x * keep / (1 - p) with the library's hash mask (kept as the 0 / scale pattern of one extra tensor)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        from . import dense
        scale = dense.dropout_scale(x.shape, p, seed, x.device)          # 0 or 1 / (1 - p) per element
        ctx.save_for_backward(scale)
        return x * scale

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return g * scale, None, None


class ShardedSetGNN(torch.nn.Module):
    """A :class:`allset_amd.SetGNN` executed on a hyperedge shard or with column-sharded aggregation (reference
    models.py:450-484, both branches: stock and GPR; ``LearnMask`` included).

    Holds the SAME module (same parameters / ``state_dict``); ``forward(x_owned)`` takes this rank's block of vertex
    rows (``hg.v_lo:hg.v_hi`` of the padded vertex range) and returns the logits of those rows.  Every rank must call
    it with its own block; replicated-parameter gradients are summed with :func:`allreduce_grads` after backward.

    ``LearnMask`` (models.py:336-337,451-452): ``Importance`` is a replicated [nnz] parameter in the caller's edge-list
    order.  A hyperedge shard multiplies its own incidences' entries (``hg.inc_ids``) into its local ``norm``; every
    incidence lives on exactly one rank, so the gradient all-reduce adds zeros from the others.  Column shards hold every
    incidence and d/P of the columns: each rank's weight gradient is a partial sum over its columns and the same
    all-reduce completes it.  ``Normalization='bn'``: cross-rank batch statistics (``_bn_scope``)."""

    def __init__(self, model, hg, group=None, aggregate: Callable = _hip_deepsets, kernels=HipPmaKernels):
        super().__init__()
        if getattr(model, "LearnMask", False) and isinstance(hg, ShardedHypergraph) and hg.inc_ids is None \
                and model.Importance.numel() != hg.local_edge_index.shape[1]:
            raise ValueError("LearnMask on a hyperedge shard needs ShardedHypergraph(inc_ids=...): the positions of the "
                             "local incidences in the global edge list")
        self.model, self.hg, self.group = model, hg, group
        self._aggregate, self._kernels = aggregate, kernels

    def _layer(self, v2e, e2v, x, norm, dropout_out=None):
        m, hg = self.model, self.hg
        cols = isinstance(hg, ColumnShardedHypergraph)
        extra = {"chunks": hg.chunks} if cols else {}          # overlapped exchange: the chunking the blocks were padded for
        if v2e.attention:                                      # PMA ignores norm (reference layers.py:628-629)
            layer = colsharded_pma_layer if cols else sharded_pma_layer
            return layer(v2e, e2v, x, hg, dropout=m.dropout, training=m.training, group=self.group, kernels=self._kernels,
                         dropout_out=dropout_out, **extra)
        layer = colsharded_deepsets_layer if cols else sharded_deepsets_layer
        return layer(v2e, e2v, x, hg, aggr=m.aggr, dropout=m.dropout, training=m.training, group=self.group,
                     aggregate=self._aggregate, norm=norm, dropout_out=dropout_out, **extra)

    def forward(self, x_owned: Tensor) -> Tensor:
        # BatchNorm1d in the classifier / GPR MLP: statistics over the real vertex rows of all ranks (the convs' own MLPs set
        # their scopes inside the layer functions)
        with _bn_scope(_valid_rows(self.hg.v_lo, self.hg.v_hi, self.hg.n_v), self.group):
            return self._forward(x_owned)

    def _forward(self, x_owned: Tensor) -> Tensor:
        m, hg = self.model, self.hg
        norm = hg.norm
        if getattr(m, "LearnMask", False):                     # norm = Importance * norm (models.py:451-452), local slice
            ids = getattr(hg, "inc_ids", None)
            norm = (m.Importance if ids is None else m.Importance[ids]) * norm
        if getattr(m, "GPR", False):                           # models.py:457-471
            x = x_owned
            xs = [F.relu(m.MLP(x))]
            for v2e, e2v in zip(m.V2EConvs, m.E2VConvs):
                x = self._layer(v2e, e2v, x, norm, dropout_out=0.0)      # relu(E2V(.)) -- the dropout comes after the tap
                xs.append(x)
                x = _rank_dropout(x, m.dropout, m.training)
            from .models import _WeightedSum   # the weighted sum of the layer outputs (models.SetGNN.forward says why not a matmul)
            x = _WeightedSum.apply(m.GPRweights.weight, *xs)
            return m.classifier(x)
        x = _rank_dropout(x_owned, 0.2, m.training)                        # hard-coded input dropout (models.py:473)
        for v2e, e2v in zip(m.V2EConvs, m.E2VConvs):
            x = self._layer(v2e, e2v, x, norm)
        return m.classifier(x)

    def allreduce_grads(self) -> None:
        allreduce_grads(list(self.model.parameters()), self.group)


def allreduce_grads(params, group=None) -> None:
    """Sum replicated-parameter gradients over ranks with ONE flat all-reduce (the layer's parameters are a
    few hundred KB; bucketing them into a single message keeps this off the per-link latency floor).

    The flat buffer covers EVERY parameter that requires grad, in list order, with zeros where a rank has no gradient
    (a branch that saw no rows on this rank): the message size is then the same on every rank whatever each one's set
    of ``None`` grads is -- a per-rank selection would hang or mix parameters up.  A parameter whose gradient is
    ``None`` here but not on a peer receives the peers' sum."""
    if _skip_collective(group):
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    _all_reduce_(flat, group=group)
    off = 0
    for p in params:
        n = p.numel()
        piece = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = piece.clone()
        else:
            p.grad.copy_(piece)
        off += n
