#!/usr/bin/env python
"""bench.py -- edges*d aggregated / sec for one V->E->V AllSet layer, forward + backward, on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N > 1: launched under
``python -m torch.distributed.run --nproc-per-node N``, one rank per GPU, RCCL).  Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
synthetic random hypergraph, |V| = |E| = 1M per GPU, every hyperedge has 16 distinct uniformly drawn members
(nnz = 16M per GPU), d = 128, fp32, AllDeepSets (HalfNLHconv x2 with 2-layer LayerNorm MLPs, aggr = add,
all-ones norm), weights from ``reset_parameters()`` under a fixed seed.  Weak scaling: rank r owns its own 1M
hyperedges over the global N*1M vertex range; the exchange is one all-gather + one reduce-scatter of
[n_V, d] per direction (allset_amd/dist.py).

One STEP = zero_grad + full layer forward (dense tail included, dropout active as in train.py) + backward
(gradients w.r.t. the input features and all parameters: four gather passes) + gradient all-reduce + Adam step.
``value`` = N * nnz_local * d / (seconds per step), inputs resident in HBM.  Nothing on the path is skipped.

Extra objects in the JSON line:
  roofline      the dominant kernel (segreduce_fwd): algorithmic bytes per launch (SURVEY section 8(d3):
                nnz*(4d+4) + (n_t+1)*4 + n_t*4d) / its mean launch time from HIP events recorded on the
                launch stream INSIDE the timed region; peak = 8 TB/s HBM3E.
  aggregation   the aggregation-only figure (all allset kernel time per step), which is what the
                north star's "% of HBM roofline on the V->E->V aggregation" refers to.
  cpu_baseline  the oracle (a restatement of the reference's CPU torch_scatter path) timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-per-gpu", type=int, default=1_000_000, help="|V| and |E| per GPU")
    ap.add_argument("--degree", type=int, default=16)
    ap.add_argument("--d", "--feature-dim", dest="d", type=int, default=128,
                    help="feature width (under torch.distributed.run use --feature-dim: the launcher's argparse takes a bare "
                         "'--d' for an abbreviation of its own --duplicate-* options)")
    ap.add_argument("--degree-dist", default="fixed", choices=["fixed", "poisson", "zipf"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: BASELINE configs[4] regime -- bf16 tensors end to end, bf16 instantiations of the gather kernels "
                         "(fp32 accumulation / softmax statistics), dense tail through torch's bf16 modules")
    ap.add_argument("--shard", default="auto", choices=["auto", "rows", "columns"],
                    help="N > 1: 'rows' = hyperedge shards with all-gather / reduce-scatter of the vertex table; 'columns' = "
                         "column-sharded aggregation with all-to-all layout changes (1/N of the exchange volume); 'auto' = "
                         "allset_amd.dist.choose_sharding (DESIGN.md section 7)")
    ap.add_argument("--pipeline-chunks", type=int, default=0,
                    help="--shard columns: chunks of owned rows whose all-to-alls overlap the other chunks' dense work "
                         "(1 = off, 0 = allset_amd.dist.auto_chunks: 4 at 1M rows per GPU, 1 below 500k)")
    ap.add_argument("--self-loops", action="store_true",
                    help="variant (SURVEY 8(d1)): add one singleton hyperedge per vertex as Add_Self_Loops does (single GPU only)")
    ap.add_argument("--model", default="deepsets", choices=["deepsets", "pma"],
                    help="deepsets = AllDeepSets (the headline, BASELINE configs[2]); pma = AllSetTransformer "
                         "(configs[3] per-GPU shape), not the driver's default")
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--seed", type=int, default=20260928)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-n", type=int, default=100_000, help="|V| = |E| of the CPU-baseline sample")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="host threads for the CPU baseline (32 is the fastest setting for torch's scatter_add_/"
                         "index_select on the 2x128-thread GPU box: profiles/r01_cpu_threads_probe.txt)")
    return ap.parse_args()


def cpu_baseline(args, d, degree):
    """Time the oracle (oracle/allset_oracle.py: index_select -> mul -> scatter_add_ + autograd, the ops
    torch_scatter 2.0.4 dispatches to for reference layers.py:633-656) on the host cores, on a bounded
    sample of the same workload.  Only this leg of bench.py touches oracle/."""
    from oracle import allset_oracle as oracle
    from allset_amd.synthetic import random_hypergraph
    from allset_amd.layers import HalfNLHconv
    n = args.cpu_sample_n
    cores = max(1, min(args.cpu_threads, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    hg = random_hypergraph(n, n, degree, seed=args.seed + 1, device="cpu", dist=args.degree_dist)
    ei, norm = hg.edge_index, hg.norm                       # int64 all-ones norm: the reference default (Q3)
    gen = torch.Generator().manual_seed(args.seed)
    x = torch.randn(n, d, generator=gen)
    # (a) aggregation only
    agg = []
    for it in range(args.cpu_iters + 1):
        t0 = time.perf_counter()
        oracle.v2e2v_aggregation_fwd_bwd(x, ei, norm, "add")
        if it:
            agg.append(time.perf_counter() - t0)
    # (b) the full layer, same state_dict layout as the GPU step (eval-mode: the oracle has no dropout)
    torch.manual_seed(args.seed)
    a = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, attention=False)
    b = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, attention=False)
    sd = {f"V2EConvs.0.{k}": v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in a.state_dict().items()}
    sd.update({f"E2VConvs.0.{k}": v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in b.state_dict().items()})
    rev = torch.stack([ei[1], ei[0]])
    full = []
    for it in range(args.cpu_iters + 1):
        t0 = time.perf_counter()
        xr = x.clone().requires_grad_(True)
        e = torch.relu(oracle.halfnlhconv_forward(sd, "V2EConvs.0.", xr, ei, norm, "add", False, 1, "ln"))
        v = torch.relu(oracle.halfnlhconv_forward(sd, "E2VConvs.0.", e, rev, norm, "add", False, 1, "ln"))
        v.backward(torch.ones_like(v))
        for t in sd.values():
            t.grad = None
        if it:
            full.append(time.perf_counter() - t0)
    unit = hg.nnz * d
    return {
        "value": unit / statistics.median(full), "unit": "edges*d/s", "cores": cores, "kind": "port",
        "sample": f"|V|=|E|={n}, deg {degree} ({args.degree_dist}), d={d}, nnz={hg.nnz}, int64 all-ones norm; full "
                  f"AllDeepSets layer fwd+bwd (eval-mode, no dropout), median of {args.cpu_iters} after 1 warm-up; "
                  f"torch {torch.__version__} CPU, {cores} threads of {os.cpu_count()} logical CPUs",
        "seconds_per_iter": statistics.median(full),
        "aggregation_only": {"value": unit / statistics.median(agg), "seconds_per_iter": statistics.median(agg)},
    }


def hbm_traffic_from_profile(kernel="segreduce_fwd"):
    """HBM bytes per launch of ``kernel`` from the committed rocprofv3 --pmc passes (profiles/), or None."""
    name = "hbm_traffic.json" if kernel == "segreduce_fwd" else "hbm_traffic_pma.json"
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            return json.load(open(path)).get(f"{kernel}_bytes_per_launch")
        except Exception:
            return None
    return None


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torchrun with {args.gpus} ranks (WORLD_SIZE is 1)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force = os.environ.get("ALLSET_FORCE_COLLECTIVES", "0") == "1"      # 1-rank RCCL group: API check on a 1-GPU box
    if world > 1 or (force and "RANK" in os.environ):
        dist.init_process_group("nccl", device_id=dev)

    from allset_amd import _lib, ops
    from allset_amd import dist as adist
    from allset_amd.layers import HalfNLHconv
    from allset_amd.synthetic import random_hypergraph
    _lib.load()                                                       # fail loudly if the HIP library is absent

    d, n_loc = args.d, args.n_per_gpu
    n_v = n_loc * world                                               # weak scaling: global vertex range grows with N
    if args.pipeline_chunks <= 0:
        args.pipeline_chunks = adist.auto_chunks(args.n_per_gpu)
    mode = args.shard if args.shard != "auto" else adist.choose_sharding(world, d, args.heads if args.model == "pma" else None)
    if mode == "columns" and world == 1 and os.environ.get("ALLSET_FORCE_COLLECTIVES", "0") != "1":
        mode = "rows"                                                  # one rank: the two layouts coincide
    shard = random_hypergraph(n_v, n_loc, args.degree, seed=args.seed + 1 + rank, device=dev, dist=args.degree_dist)
    n_e_loc = n_loc
    if args.self_loops:
        if world != 1:
            raise SystemExit("--self-loops is a single-GPU variant")
        vs = torch.arange(n_v, device=dev, dtype=torch.int64)
        ei = torch.cat([shard.edge_index, torch.stack([vs, n_loc + vs])], dim=1)
        ei = ei[:, torch.argsort(ei[0], stable=True)].contiguous()
        shard.edge_index, shard.nnz, n_e_loc = ei, int(ei.shape[1]), n_loc + n_v
        shard.norm = torch.ones(shard.nnz, dtype=torch.int64, device=dev)
    if mode == "columns":
        # every rank holds the whole incidence: the same P blocks of hyperedges the row mode deals out one per rank
        blocks = [shard if r == rank else random_hypergraph(n_v, n_loc, args.degree, seed=args.seed + 1 + r, device=dev,
                                                            dist=args.degree_dist) for r in range(world)]
        ei = torch.cat([torch.stack([b.edge_index[0], b.edge_index[1] + r * n_loc]) for r, b in enumerate(blocks)], dim=1)
        nnz_global = int(ei.shape[1])
        hg = adist.ColumnShardedHypergraph(ei, n_v, n_loc * world, world, rank, norm=torch.cat([b.norm for b in blocks]),
                                           chunks=args.pipeline_chunks).build_incidences()
        del blocks, ei
        nnz_local = nnz_global / world                   # each rank aggregates every incidence over d/N of the columns
    else:
        hg = adist.ShardedHypergraph(shard.edge_index, n_v, n_e_loc, world, rank, norm=shard.norm).build_incidences()
        nnz_local = shard.nnz

    torch.manual_seed(args.seed)                                       # identical replicated weights on every rank
    attn = args.model == "pma"
    v2e = HalfNLHconv(d, d, d, 2, args.dropout, "ln", True, heads=args.heads, attention=attn)
    e2v = HalfNLHconv(d, d, d, 2, args.dropout, "ln", True, heads=args.heads, attention=attn)
    v2e.reset_parameters(); e2v.reset_parameters()
    v2e.to(dev).train(); e2v.to(dev).train()
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if tdt != torch.float32:
        v2e.to(tdt); e2v.to(tdt)
    params = list(v2e.parameters()) + list(e2v.parameters())
    opt = torch.optim.Adam(params, lr=1e-3, fused=True)     # same Adam math, one multi-tensor kernel for the 24 small parameters

    gen = torch.Generator(device=dev).manual_seed(args.seed + 100 + rank)
    rows = hg.v_hi - hg.v_lo
    x = torch.randn(rows, d, device=dev, generator=gen).to(tdt).requires_grad_(True)       # owned vertex block
    G = torch.randn(rows, d, device=dev, generator=gen).to(tdt)

    def step():
        opt.zero_grad(set_to_none=True)
        x.grad = None
        if mode == "columns":
            out = (adist.colsharded_pma_layer(v2e, e2v, x, hg, dropout=args.dropout, training=True, chunks=args.pipeline_chunks)
                   if attn else
                   adist.colsharded_deepsets_layer(v2e, e2v, x, hg, aggr="add", dropout=args.dropout, training=True,
                                                   chunks=args.pipeline_chunks))
        elif attn:
            out = adist.sharded_pma_layer(v2e, e2v, x, hg, dropout=args.dropout, training=True)
        else:
            out = adist.sharded_deepsets_layer(v2e, e2v, x, hg, aggr="add", dropout=args.dropout, training=True)
        out.backward(G)
        adist.allreduce_grads(params)
        opt.step()

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    timer = ops.KernelTimer()
    fence()
    ops.set_kernel_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    ops.set_kernel_timer(None)

    stats = torch.tensor([elapsed, float(nnz_local)], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        tmax = stats.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stats.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, nnz_total = float(tmax[0]), float(tsum[1])
    else:
        nnz_total = float(nnz_local)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = nnz_total * d / (elapsed / args.steps)
        ks = timer.summary()
        AGG = ("segreduce_fwd", "segmax_bwd", "sddmm_rowdot", "pma_fwd", "pma_bwd_stats", "pma_bwd_src")
        agg_ks = {k: v for k, v in ks.items() if k in AGG}                       # HBM-bound gather kernels
        dense_ks = {k: v for k, v in ks.items() if k not in AGG}                 # dense tail (MFMA / streaming)
        dom = max(agg_ks, key=lambda k: agg_ks[k]["total_ms"]) if agg_ks else None   # dominant aggregation kernel
        seg = agg_ks.get(dom) if dom else None
        agg_ms = sum(v["total_ms"] for v in agg_ks.values()) / args.steps
        dense_ms = sum(v["total_ms"] for v in dense_ks.values()) / args.steps
        # the PMC passes were taken at exactly this shape (tools/pmc_probe.py); any other shape reports null
        traffic = (hbm_traffic_from_profile(dom) if (world == 1 and args.n_per_gpu == 1_000_000 and d == 128 and args.degree == 16
                                                     and args.degree_dist == "fixed" and args.dtype == "f32" and not args.self_loops
                                                     and (not attn or args.heads == 4)) else None)
        roofline = None
        if seg:
            achieved = seg["algo_bytes"] / (seg["avg_ms"] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": f"allset_{dom}",
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "algo_bytes_per_launch": seg["algo_bytes"], "avg_launch_ms": seg["avg_ms"],
                        "launches": seg["calls"]}
        line = {
            "metric": "edges*d aggregated / sec (V->E->V layer fwd+bwd)", "value": value, "unit": "edges*d/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "dtype_note": ("fp32 tensors end to end; aggregation kernels: plain fp32 adds; dense tail: every fp32 operand split exactly "
                           "into three bf16, six of the nine partial products formed on the bf16 matrix pipe and accumulated in "
                           "fp32 -- measured error vs float64 <= that of the native fp32 MFMA / hipBLASLt (DESIGN.md section 6)")
            if args.dtype == "f32" else
            ("bf16 tensors end to end (BASELINE configs[4] regime): bf16 instantiations of the gather kernels with fp32 "
             "accumulation and fp32 softmax statistics; dense tail: this library's bf16 LayerNorm / add+LayerNorm+relu+dropout / "
             "split-K weight-gradient kernels (fp32 arithmetic), Linear forward and backward-data through the library's bf16 "
             "GEMMs"),
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{3 if attn else 2}]{' per-GPU shape' if attn else ''}: synthetic random hypergraph |V|=|E|={n_loc} per GPU, "
                                   f"hyperedge size {args.degree} ({args.degree_dist}), nnz={int(nnz_total)}, d={d}, " +
                                   (f"AllSetTransformer layer (PMA x2, heads={args.heads}, dropout {args.dropout}), " if attn else
                                    f"AllDeepSets layer (HalfNLHconv x2, 2-layer LN MLPs, aggr=add, dropout {args.dropout}), ") +
                                   f"fwd+bwd+Adam" + (" + one singleton self-loop hyperedge per vertex" if args.self_loops else ""),
                       "n_v": n_v, "n_e": n_e_loc * world, "nnz": int(nnz_total), "d": d,
                       "parallelism": ("single GPU" if world == 1 else f"hyperedge-shard x{world}" if mode == "rows"
                                       else f"column-shard x{world} (rows for the dense tail, d/{world} columns for the aggregation; "
                                            f"exchange in {args.pipeline_chunks} overlapped chunks)"), "seed": args.seed},
            "roofline": roofline,
            "aggregation": {"ms_per_step": agg_ms, "value": nnz_total * d / (agg_ms * 1e-3) if world == 1 else None,
                            "unit": "edges*d/s", "note": "gather/segment-reduce kernel time per step (HIP events, rank 0): "
                            "the aggregation-only V->E->V fwd+bwd", "kernels": {k: {"calls_per_step": v["calls"] / args.steps,
                                                                          "avg_ms": v["avg_ms"]} for k, v in agg_ks.items()}},
            "dense_tail": {"ms_per_step": dense_ms, "note": "HIP dense-tail kernels per step (fused norm+Linear fwd, "
                           "backward-data with LayerNorm-backward epilogue, split-K weight gradient). fp32 in, fp32 out, "
                           "fp32-accurate arithmetic on the bf16 matrix pipe: operands split exactly into 3 bf16, 6 of 9 "
                           "products accumulated in fp32 (error <= native fp32 MFMA, tests/test_gpu_dense.py); these "
                           "kernels are HBM-bound: gbps = algorithmic activation bytes / time (peak 8000)",
                           "kernels": {k: {"calls_per_step": v["calls"] / args.steps, "avg_ms": v["avg_ms"],
                                           "gbps": (v["algo_bytes"] / (v["avg_ms"] * 1e-3) / 1e9) if v.get("algo_bytes") else None,
                                           "tflops": (2.0 * rows * d * d / (v["avg_ms"] * 1e-3) / 1e12)
                                           if k in ("fused_linear_fwd", "fused_linear_bwd", "wgrad_fused", "wgrad") else None}
                                       for k, v in dense_ks.items()}},
        }
        if world == 1 and not args.no_cpu_baseline and not attn:
            line["cpu_baseline"] = cpu_baseline(args, d, args.degree)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
