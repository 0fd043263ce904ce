#!/usr/bin/env python
"""Per-rank COMPUTE time of an N-rank weak-scaling step, measured on one GPU: rank 0's shard of the N-rank job
(N*1M vertices, 1M local hyperedges) with the two collectives replaced by local stand-ins of identical shapes
(all-gather -> tile the owned block N times, reduce-scatter -> keep the owned slice).  What is left out is exactly the
xGMI time; the output bounds the scaling the driver can measure:  efficiency <= t(1) / (t_compute(N) + t_comm(N)).

usage: sim_rank.py [deepsets|pma] [rows|columns|hybrid]
``hybrid`` (round 5): the 2 x (N / 2) hybrid partition (dist.ColumnShardedHypergraph(row_groups=2)): rank 0 holds the incidences of
target group 0 and gathers rows of d / (N / 2) columns; its world all-to-all is replaced by a stand-in of the same shapes.
``columns``: the column-sharded layer (full incidence on the rank, d/N columns); the all-to-alls are replaced by their
own pack / unpack copies, which produce tensors of exactly the exchanged shapes.
The xGMI estimate printed beside it is per LINK: every pair of GPUs is joined by one link (~77 GB/s per direction peak,
60 GB/s assumed achievable), and both schemes load all N-1 links of a rank evenly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allset_amd import dist as adist
from allset_amd.layers import HalfNLHconv
from allset_amd.synthetic import random_hypergraph
dev = torch.device("cuda:0")
d, n_loc = int(os.environ.get("SIM_D", "128")), int(os.environ.get("SIM_N", "1000000"))
DT = torch.bfloat16 if os.environ.get("SIM_DTYPE") == "bf16" else torch.float32
DIST = os.environ.get("SIM_DIST", "fixed")          # fixed | zipf (BASELINE configs[4]: SIM_D=256 SIM_N=250000 SIM_DTYPE=bf16 SIM_DIST=zipf)
model = sys.argv[1] if len(sys.argv) > 1 else "deepsets"
mode = sys.argv[2] if len(sys.argv) > 2 else "rows"
LINK = 60e9
LOCALITY = float(os.environ.get("SIM_LOCALITY", "0"))   # rows: draw that share of every membership from the rank's own vertex block (dist.Halo exchange)
CHUNKS = int(os.environ.get("SIM_CHUNKS", "1"))     # columns: chunks of owned rows (the overlapped exchange's machinery; its all-to-alls become local copies)


class _Done:
    def wait(self):
        pass


def _a2a_local(out_views, in_views, group):
    # SIM_A2A_COPY=1: move the pieces with device copies (correct data, but 2 x 4 GB of copy traffic per step that a real
    # all-to-all does on the links, not in HBM time -- the K = 1 path's stand-in moves nothing either); default: move nothing,
    # the received buffers keep whatever they held (timing only: every kernel runs on the same shapes)
    if os.environ.get("SIM_A2A_COPY") == "1":
        for o, i in zip(out_views, in_views):
            o.copy_(i)
    else:
        for o in out_views[:1]:
            o.zero_()                       # (keeps NaN garbage of a fresh allocation out of the LayerNorms)
    return _Done()


adist._a2a_async = _a2a_local
for world in tuple(int(w) for w in os.environ.get("SIM_WORLDS", "1,2,4,8").split(",")):
    adist._rows_to_cols = (lambda x, group=None, w=world: x if w == 1 else adist._pack(x, w).view(w * x.shape[0], x.shape[1] // w))
    adist._cols_to_rows = (lambda x, group=None, w=world: x if w == 1 else adist._unpack(x.view(w, x.shape[0] // w, x.shape[1])))
    adist._exchange_blocks = (lambda x, group=None: x)                # repack-free exchange: the buffer IS the layout on both sides
    adist._all_gather_rows = (lambda x, group=None, w=world: x if w == 1 else x.repeat(w, *([1] * (x.dim() - 1))))
    adist._reduce_scatter_rows = (lambda x, group=None, w=world: x if w == 1 else x[:x.shape[0] // w].contiguous())
    adist._world = lambda group=None, w=world: w
    adist._skip_collective = lambda group=None, w=world: w == 1      # world > 1: take the real merge paths ...
    adist.dist.all_reduce = lambda *a, **k: None                      # ... with the small max-all-reduce stubbed out
    # halo exchange (rows, SIM_LOCALITY > 0): the per-peer all-to-all returns rows of the asked-for COUNT (by symmetry of the
    # generator every peer asks this rank for as many rows as it asks them)
    adist._all_to_all_rows = (lambda send, ins, outs, group=None: send[:sum(outs)] if send.shape[0] >= sum(outs) else
                              torch.cat([send, send.new_zeros((sum(outs) - send.shape[0],) + tuple(send.shape[1:]))]))
    n_v = n_loc * world
    if mode == "hybrid":
        if world < 4:
            continue
        R, C = 2, world // 2
        adist._world = lambda group=None, w=world, C=C: C if group == "COL" else w
        adist._skip_collective = lambda group=None: False

        def _a2a_single(recv, send, group=None):       # moves nothing (timing only); the head of the buffer zeroed against stale NaNs
            recv[:max(recv.shape[0] // 64, 1)].zero_()
        adist._all_to_all_single = _a2a_single
        adist._rows_to_cols = (lambda x, group=None, C=C: adist._pack(x, C).view(C * x.shape[0], x.shape[1] // C))
        adist._cols_to_rows = (lambda x, group=None, C=C: adist._unpack(x.view(C, x.shape[0] // C, x.shape[1])))
        blocks = [random_hypergraph(n_v, n_loc, 16, seed=5 + r, device=dev, e_offset=r * n_loc, dist=DIST) for r in range(world)]
        ei = torch.cat([b.edge_index for b in blocks], dim=1)
        shard = blocks[0]
        hg = adist.ColumnShardedHypergraph(ei, n_v, n_loc * world, world, 0, norm=torch.cat([b.norm for b in blocks]), row_groups=R,
                                           col_group="COL").build_incidences()
        del blocks, ei
    elif mode == "columns":
        blocks = [random_hypergraph(n_v, n_loc, 16, seed=5 + r, device=dev, e_offset=r * n_loc, dist=DIST) for r in range(world)]
        ei = torch.cat([b.edge_index for b in blocks], dim=1)
        shard = blocks[0]
        hg = adist.ColumnShardedHypergraph(ei, n_v, n_loc * world, world, 0,
                                           norm=torch.cat([b.norm for b in blocks])).build_incidences()
        del blocks, ei
    else:
        shard = random_hypergraph(n_v, n_loc, 16, seed=5, device=dev, dist=DIST, locality=LOCALITY, home=(0, world) if LOCALITY > 0 else None)
        hg = adist.ShardedHypergraph(shard.edge_index, n_v, n_loc, world, 0, norm=shard.norm, halo=LOCALITY > 0 and world > 1).build_incidences()
    attn = model == "pma"
    torch.manual_seed(0)
    a = HalfNLHconv(d, d, d, 2, 0.5, "ln", True, heads=4, attention=attn).to(dev).to(DT).train()
    b = HalfNLHconv(d, d, d, 2, 0.5, "ln", True, heads=4, attention=attn).to(dev).to(DT).train()
    params = list(a.parameters()) + list(b.parameters())
    opt = torch.optim.Adam(params, lr=1e-3, fused=True)
    x = torch.randn(n_loc, d, device=dev).to(DT).requires_grad_(True)
    G = torch.randn(n_loc, d, device=dev).to(DT)
    def step():
        opt.zero_grad(set_to_none=True); x.grad = None
        if mode in ("columns", "hybrid"):
            out = adist.colsharded_pma_layer(a, b, x, hg, dropout=0.5, training=True, chunks=CHUNKS) if attn else \
                adist.colsharded_deepsets_layer(a, b, x, hg, aggr="add", dropout=0.5, training=True, chunks=CHUNKS)
        else:
            out = adist.sharded_pma_layer(a, b, x, hg, dropout=0.5, training=True) if attn else \
                adist.sharded_deepsets_layer(a, b, x, hg, aggr="add", dropout=0.5, training=True)
        out.backward(G); opt.step()
    for _ in range(3): step()
    from allset_amd import ops as _ops
    timer = _ops.KernelTimer() if os.environ.get("SIM_KERNELS") else None     # per-kernel HIP-event times of the 10 steps
    import gc
    gc.collect(); gc.freeze()            # (as bench.py does: a full collection inside the timed steps costs milliseconds once the process holds several worlds' objects)
    torch.cuda.synchronize(); _ops.set_kernel_timer(timer); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    if os.environ.get("SIM_ALLOC_STATS"):
        st = torch.cuda.memory_stats()
        print(f"      allocator: device mallocs {st.get('num_device_alloc')}, frees {st.get('num_device_free')}, retries {st.get('num_alloc_retries')}, "
              f"reserved {st.get('reserved_bytes.all.current', 0) / 2**30:.1f} GiB, gc counts {__import__('gc').get_count()}")
    _ops.set_kernel_timer(None)
    gc.unfreeze()
    if timer is not None:
        for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
            print(f"      {k:22s} {v['calls'] / 10:5.1f} calls/step  {v['avg_ms']:7.3f} ms avg  {v['total_ms'] / 10:7.3f} ms/step")
    if world == 1 or "t1" not in globals():
        t1 = ms if world == 1 else float(os.environ.get("SIM_T1_MS", "nan"))
    mname = f"hybrid2x{world // 2}" if mode == "hybrid" else mode
    per_rank = adist.exchange_bytes_per_rank(mname, world, n_v, n_loc * world, d, 2 if DT == torch.bfloat16 else 4)     # received per step (fwd + bwd)
    if mode == "rows" and getattr(hg, "halo", None) is not None:
        per_rank = 4 * hg.halo.exchange_rows() * d * (2 if DT == torch.bfloat16 else 4)       # four exchanges of the asked-for rows per step
        print(f"      halo: {hg.halo.n_needed} of {n_v} vertices touched, {hg.halo.exchange_rows()} rows from peers per exchange (locality {LOCALITY:g})")
    per_link = adist.exchange_bytes_per_link(mname, world, n_v, n_loc * world, d, 2 if DT == torch.bfloat16 else 4) if world > 1 else 0
    comm = per_link / LINK * 1e3
    print(f"{model} {mode}{f' chunks={CHUNKS}' if CHUNKS > 1 else ''} world={world}: per-rank compute {ms:7.2f} ms   exchange {per_rank/1e9:5.2f} GB per rank and step, "
          f"{per_link/1e9:4.2f} GB per link -> {comm:5.1f} ms at 60 GB/s   speed-up if serial {world*t1/(ms+comm) if world > 1 else 1.0:4.2f}x, "
          f"if fully overlapped {world*t1/max(ms, comm) if world > 1 else 1.0:4.2f}x", flush=True)
    del hg, shard, a, b, x, G, opt
    torch.cuda.empty_cache()
