#!/usr/bin/env python
"""bench.py -- edges*d aggregated / sec for one V->E->V AllSet layer, forward + backward, on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N > 1: launched under
``python -m torch.distributed.run --nproc-per-node N``, one rank per GPU, RCCL -- or bare: without RANK in the environment
``--gpus N`` spawns that launcher itself).  Prints ONE JSON line on rank 0's stdout, always:
  * N > 1 runs its timed regions in the order rows -> (link preflight) -> columns -> primary with the bf16 wire; every region
    runs under a watchdog deadline (``--region-timeout``).  A region that raises is recorded under ``partitions`` and skipped; a
    region that HANGS (a collective that never completes) makes the watchdog print the line assembled from the regions that did
    finish and end the process with exit code 0 -- the first region's result is never lost to a later one.
  * after the first region rank 0 also writes that region's line to STDERR (``[bench] early line ...``), for a log reader;
    stdout carries exactly one line.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
synthetic random hypergraph, |V| = |E| = 1M per GPU, every hyperedge has 16 distinct uniformly drawn members
(nnz = 16M per GPU), d = 128, fp32, AllDeepSets (HalfNLHconv x2 with 2-layer LayerNorm MLPs, aggr = add,
all-ones norm), weights from ``reset_parameters()`` under a fixed seed.  Weak scaling: the job's hypergraph is
N blocks of 1M hyperedges over the global N*1M vertex range (block r is drawn from seed + 1 + r whatever the
partition, so both partitions below run the SAME global hypergraph).

One STEP = zero_grad + full layer forward (dense tail included, dropout active as in train.py) + backward
(gradients w.r.t. the input features and all parameters: four gather passes) + gradient all-reduce + Adam step.
``value`` = N * nnz_per_block * d / (seconds per step), inputs resident in HBM.  Nothing on the path is skipped.

N > 1 -- two partitions of the same job (allset_amd/dist.py):
  rows     hyperedge shards, the partition BASELINE.json's north star names: rank r owns block r; one all-gather +
           one reduce-scatter of the [n_V, d] vertex table per direction;
  columns  column-sharded aggregation: every rank holds the whole incidence and d/N feature columns of every row;
           four all-to-alls per direction-pair, 1/N of the row scheme's bytes at N = 8, overlapped with the dense work.
Both are timed in every N > 1 run, each in its own region of the same K steps (plus the column partition with the chunked
overlapped exchange), and appear under ``partitions``, so a scaling record always carries the north-star partition.
``value`` / ``ms_per_step`` are those of ``--shard`` -- ``rows`` by default since round 5 (the north star's partition); ``columns`` /
``hybrid`` pin another one, ``auto`` reports the FASTEST of the exact (fp32-wire) executions this run timed;
``partitions.fastest_exact`` always names that one.  N = 2 x C >= 4 adds the hybrid partition (2 target groups x C column groups,
``hybrid2xC``, right after ``rows``).  ``preflight`` = the layer's four
collectives timed alone at this job's message sizes (GB/s per rank and per xGMI link).

Extra objects in the JSON line:
  roofline      the dominant gather kernel: algorithmic bytes per launch (SURVEY section 8(d3):
                nnz*(4d+4) + (n_t+1)*4 + n_t*4d) / its mean launch time from HIP events recorded on the launch
                stream INSIDE the timed region; peak = 8 TB/s HBM3E.  ``frac_of_copy_ceiling`` = the same rate over the
                6.3 TB/s streaming-copy ceiling of MI355X_MICROARCH.md; ``traffic`` = HBM bytes per launch from the
                committed rocprofv3 --pmc passes at exactly this shape (``traffic_source`` says which file; null at
                any other shape, and null when the gather kernels' sources / flags differ from the ones the passes
                were taken with -- it is a profile of the same kernel and shape, not a counter of this run);
                ``layer_frac`` = the aggregation's algorithmic bytes per step / the WHOLE step time / peak;
                ``per_kernel`` = the same arithmetic for every timed kernel that states its algorithmic bytes.
  aggregation   the aggregation-only figure (all gather-kernel time per step): the north star's
                "% of HBM roofline on the V->E->V aggregation".
  cpu_baseline  the oracle (a restatement of the reference's CPU torch_scatter path) timed on this box's host cores
                on the SAME hypergraph and features the GPU ran (copied back), median of ``--cpu-iters`` (3) iterations at
                full size after a warm-up on a 1/10 sample (rank 0, N = 1 only).
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

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA16_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 matrix-pipe peak (same guide; never the 2:1-sparsity headline figure)
COPY_CEILING_GBS = 6300.0  # measured streaming-copy ceiling, same guide
AGG_KERNELS = ("segreduce_fwd", "segmax_bwd", "sddmm_rowdot", "pma_fwd", "pma_bwd_stats", "pma_bwd_src")


def parse_args(argv=None):
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
                         "(fp32 accumulation / softmax statistics) and of the dense-tail kernels")
    ap.add_argument("--locality", type=float, default=0.0,
                    help="variant workload (not the headline): each membership of block r's hyperedges is re-drawn with this "
                         "probability from block r's own vertices -- a hypergraph whose hyperedge partitions have few boundary "
                         "vertices (0 = BASELINE's uniform random hypergraph)")
    ap.add_argument("--rows-exchange", default="auto", choices=["auto", "table", "halo"],
                    help="row partition: 'table' = all-gather / reduce-scatter of the whole vertex table, 'halo' = exchange only the "
                         "rows of the vertices a rank's hyperedges touch (allset_amd.dist.Halo); auto = halo iff --locality > 0")
    ap.add_argument("--shard", default="rows", choices=["auto", "rows", "columns", "hybrid"],
                    help="N > 1: which partition is timed as `value` ('auto' = allset_amd.dist.choose_sharding)")
    ap.add_argument("--partitions", default="both", choices=["both", "primary"],
                    help="N > 1: 'both' also times the other partition in a second region and reports it under `partitions`")
    ap.add_argument("--pipeline-chunks", type=int, default=1,
                    help="--shard columns: chunks of owned rows whose all-to-alls overlap the other chunks' dense work "
                         "(1 = off, the default: blocking all-to-alls; 0 = allset_amd.dist.auto_chunks: 4 at 1M rows per GPU, "
                         "1 below 500k).  Off by default since round 2: on one GPU the 4-chunk machinery costs +4.8 ms per step "
                         "(AllDeepSets) and turns the AllSetTransformer step host-bound (profiles/r02_colshard_chunks.txt) against "
                         "<= 6 ms of exchange it can hide at N = 8, and it has never run on more than one rank")
    ap.add_argument("--hip-graph", action="store_true",
                    help="single GPU: capture the step (zero_grad + forward + backward + Adam, dropout seeds from a device counter) "
                         "as one hipGraph and time K replays -- the same kernels on the same data, without the ~100 host launches "
                         "per step that are ~10 %% of a step in the bf16 regime (configs[4] per-GPU shape)")
    ap.add_argument("--chunk-entry", type=int, default=-1,
                    help="N > 1: chunk count of the extra timed region of the column partition with the chunked, overlapped exchange "
                         "(-1 = allset_amd.dist.auto_chunks: 4 at 1M rows per GPU, none below 500k; 0 = skip)")
    ap.add_argument("--no-wire-entry", dest="wire_entry", action="store_false",
                    help="N > 1: skip the extra timed region of the primary partition with the opt-in bf16 wire format")
    ap.add_argument("--self-loops", action="store_true",
                    help="variant (SURVEY 8(d1)): add one singleton hyperedge per vertex as Add_Self_Loops does (single GPU only)")
    ap.add_argument("--model", default="deepsets", choices=["deepsets", "pma"],
                    help="deepsets = AllDeepSets (the headline, BASELINE configs[2]); pma = AllSetTransformer "
                         "(configs[3] per-GPU shape), not the driver's default")
    ap.add_argument("--norm", default="ln", choices=["ln", "bn"],
                    help="normalisation inside the MLPs: ln = LayerNorm (the stock train.py setting and the headline), bn = BatchNorm1d "
                         "with batch statistics (the reference MLP's class default), a variant line")
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--seed", type=int, default=20260928)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--arith", default="auto", choices=["auto", "strict", "bf16x6", "fp16x3"],
                    help="arithmetic of the fused fp32 Linear kernels (allset_amd.dense.set_arithmetic; include/allset_hip_ext.h "
                         "ALLSET_ARITH_*): auto = fp16x3 where built, strict = the exact three-bf16-plane split everywhere")
    ap.add_argument("--no-strict-entry", dest="strict_entry", action="store_false",
                    help="N = 1, fp32: skip the second timed region that repeats the K steps in the strict arithmetic "
                         "(reported as `strict` / dense_tail.strict_ms_per_step, never `value`)")
    ap.add_argument("--cpu-sample-n", type=int, default=0,
                    help="|V| = |E| of the CPU-baseline sample (0 = the GPU workload itself, copied back to the host)")
    ap.add_argument("--cpu-iters", type=int, default=3, help="timed CPU iterations (median reported; BASELINE.md section 3: >= 3)")
    ap.add_argument("--region-timeout", type=float, default=300.0,
                    help="N > 1: seconds a timed region (or the preflight) may take before bench.py's watchdog prints the line from "
                         "the regions that finished and ends the process (the first region gets 3x)")
    ap.add_argument("--preflight", default="auto", choices=["auto", "on", "off"],
                    help="time the layer's four collectives at this job's message sizes after the first region and report GB/s per "
                         "link under `preflight` (auto = when N > 1)")
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="host threads for the CPU baseline (32 is the fastest setting for torch's scatter_add_/"
                         "index_select on the 2x128-thread GPU box: profiles/r01_cpu_threads_probe.txt)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# problem construction (pure index arithmetic + the generator: runs on any device; tests/test_bench_setup.py)
# ---------------------------------------------------------------------------------------------------------------------

def hyperedge_block(args, n_v: int, r: int, device, world: int = 1):
    """Block r of the job's hypergraph: ``n_per_gpu`` hyperedges (local ids 0..n-1) over the GLOBAL vertex range."""
    from allset_amd.synthetic import random_hypergraph
    loc = float(getattr(args, "locality", 0.0))
    return random_hypergraph(n_v, args.n_per_gpu, args.degree, seed=args.seed + 1 + r, device=device, dist=args.degree_dist,
                             locality=loc, home=(r, world) if loc > 0.0 else None)


def rows_halo(args) -> bool:
    mode = getattr(args, "rows_exchange", "auto")
    return mode == "halo" or (mode == "auto" and float(getattr(args, "locality", 0.0)) > 0.0)


def build_problem(args, mode: str, world: int, rank: int, device):
    """One rank's view of the job for partition ``mode``.  Returns (hg, nnz_local, n_e_global): ``hg`` a
    ``ShardedHypergraph`` (rows: this rank's hyperedge block) or ``ColumnShardedHypergraph`` (columns: all blocks, global
    hyperedge id = r * n_per_gpu + local id); ``nnz_local`` = the incidences this rank accounts for in ``value`` --
    its block's under ``rows``, 1/N of all of them under ``columns`` (every rank aggregates every incidence over d/N of
    the columns).  Summed over ranks both give the job's nnz."""
    from allset_amd import dist as adist
    n_loc = args.n_per_gpu
    n_v = n_loc * world
    if mode == "columns" or mode.startswith("hybrid"):
        blocks = [hyperedge_block(args, n_v, r, device, world) for r in range(world)]
        ei = torch.cat([torch.stack([b.edge_index[0], b.edge_index[1] + r * n_loc]) for r, b in enumerate(blocks)], dim=1)
        norm = torch.cat([b.norm for b in blocks])
        if mode.startswith("hybrid"):
            # R target groups x C column groups (allset_amd.dist.ColumnShardedHypergraph(row_groups=R)): every rank aggregates the
            # incidences of its target group (1 / R of them per direction) over d / C columns -- nnz / world incidence-widths, like
            # the column partition
            R = int(mode[len("hybrid"):].split("x")[0])
            cg, gg = _hybrid_groups(world, R, rank)
            hg = adist.ColumnShardedHypergraph(ei, n_v, n_loc * world, world, rank, norm=norm, row_groups=R, col_group=cg, gather_group=gg)
            return hg, ei.shape[1] / world, n_loc * world
        hg = adist.ColumnShardedHypergraph(ei, n_v, n_loc * world, world, rank, norm=norm, chunks=args.pipeline_chunks)
        return hg, ei.shape[1] / world, n_loc * world
    shard = hyperedge_block(args, n_v, rank, device, world)
    n_e_loc = n_loc
    if args.self_loops:
        if world != 1:
            raise SystemExit("--self-loops is a single-GPU variant")
        vs = torch.arange(n_v, device=device, dtype=torch.int64)
        ei = torch.cat([shard.edge_index, torch.stack([vs, n_loc + vs])], dim=1)
        ei = ei[:, torch.argsort(ei[0], stable=True)].contiguous()
        shard.edge_index, shard.nnz, n_e_loc = ei, int(ei.shape[1]), n_loc + n_v
        shard.norm = torch.ones(shard.nnz, dtype=torch.int64, device=device)
    halo = rows_halo(args) and world > 1
    hg = adist.ShardedHypergraph(shard.edge_index, n_v, n_e_loc, world, rank, norm=shard.norm, halo=halo)
    return hg, float(shard.nnz), n_e_loc * world


_HYBRID_GROUPS = {}


def _hybrid_groups(world: int, row_groups: int, rank: int):
    """The hybrid partition's process groups, created once per process (dist.new_group is collective: every rank, same order)."""
    from allset_amd import dist as adist
    key = (world, row_groups)
    if key not in _HYBRID_GROUPS:
        _HYBRID_GROUPS[key] = adist.hybrid_groups(world, row_groups, rank) if dist.is_initialized() else (None, None)
    return _HYBRID_GROUPS[key]


def hybrid_mode(args, world: int):
    """'hybridRxC' if the job has a hybrid partition worth timing (world = 2 x C >= 4, d / C columns = a whole number of heads or
    head fractions that divide, rows of >= 128 bytes), else None."""
    if world < 4 or world % 2 or args.dtype not in ("f32", "bf16"):
        return None
    C = world // 2
    elem = 4 if args.dtype == "f32" else 2
    if args.d % C or (args.d // C) * elem < 128:
        return None
    if args.model == "pma" and (args.heads % C and C % args.heads):
        return None
    return f"hybrid2x{C}"


def job_value(nnz_total: float, d: int, elapsed_s: float, steps: int) -> float:
    """edges*d per second of the whole job: all ranks' incidences x width / (max-over-ranks seconds per step)."""
    return nnz_total * d / (elapsed_s / steps)


def resolve_modes(args, world: int):
    """(primary, other or None) partitions for this run."""
    from allset_amd import dist as adist
    heads = args.heads if args.model == "pma" else None
    auto = adist.choose_sharding(world, args.d, heads)
    primary = args.shard if args.shard != "auto" else auto           # ("hybrid": main() resolves it to hybridRxC)
    forced = os.environ.get("ALLSET_FORCE_COLLECTIVES", "0") == "1"
    if world == 1 and not forced:
        return "rows", None                                    # one rank: the two layouts coincide
    other = None
    if args.partitions == "both":
        other = "rows" if primary in ("columns", "hybrid") else "columns"
        if other == "columns" and (world > 1 and auto != "columns"):
            other = None                                       # the width / head count does not split over this many ranks
    return primary, other


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the only leg of this file that touches oracle/)
# ---------------------------------------------------------------------------------------------------------------------

def cpu_baseline(args, d, degree, edge_index=None, x=None):
    """Time the oracle (oracle/allset_oracle.py: index_select -> mul -> scatter_add_ + autograd, the ops
    torch_scatter 2.0.4 dispatches to for reference layers.py:633-656) on the host cores.  With ``edge_index`` / ``x``
    (host copies of what the GPU just ran) the sample IS the workload; a 1/10-size warm-up pages the operators in."""
    from oracle import allset_oracle as oracle
    from allset_amd.synthetic import random_hypergraph
    from allset_amd.layers import HalfNLHconv
    cores = max(1, min(args.cpu_threads, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    torch.manual_seed(args.seed)
    a = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, attention=False)
    b = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, attention=False)
    sd = {f"V2EConvs.0.{k}": v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in a.state_dict().items()}
    sd.update({f"E2VConvs.0.{k}": v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in b.state_dict().items()})

    def one(ei, norm, xin, iters, warm):
        rev = torch.stack([ei[1], ei[0]])
        agg, full = [], []
        for it in range(iters + warm):
            t0 = time.perf_counter()
            oracle.v2e2v_aggregation_fwd_bwd(xin, ei, norm, "add")                 # (a) aggregation only
            if it >= warm:
                agg.append(time.perf_counter() - t0)
        for it in range(iters + warm):                                             # (b) the full layer (eval mode)
            t0 = time.perf_counter()
            xr = xin.clone().requires_grad_(True)
            e = torch.relu(oracle.halfnlhconv_forward(sd, "V2EConvs.0.", xr, ei, norm, "add", False, 1, "ln"))
            v = torch.relu(oracle.halfnlhconv_forward(sd, "E2VConvs.0.", e, rev, norm, "add", False, 1, "ln"))
            v.backward(torch.ones_like(v))
            for t in sd.values():
                t.grad = None
            if it >= warm:
                full.append(time.perf_counter() - t0)
        return agg, full

    if edge_index is None or args.cpu_sample_n > 0:
        n = args.cpu_sample_n or 100_000
        hg = random_hypergraph(n, n, degree, seed=args.seed + 1, device="cpu", dist=args.degree_dist)
        ei, norm = hg.edge_index, hg.norm                    # int64 all-ones norm: the reference default (Q3)
        xin = torch.randn(n, d, generator=torch.Generator().manual_seed(args.seed))
        agg, full = one(ei, norm, xin, args.cpu_iters, 1)
        what = f"|V|=|E|={n} sample drawn on the host"
    else:
        ei, xin = edge_index, x
        n = xin.shape[0]
        norm = torch.ones(ei.shape[1], dtype=torch.int64)
        k = max(n // 10, 1)
        small = ei[:, (ei[0] < k) & (ei[1] < k)].contiguous()
        if small.shape[1]:
            one(small, torch.ones(small.shape[1], dtype=torch.int64), xin[:k].contiguous(), 1, 0)      # warm-up, 1/10 size
        agg, full = one(ei, norm, xin, args.cpu_iters, 0)
        what = f"the GPU workload itself (|V|=|E|={n}, same incidences and features, copied to the host)"
    unit = int(ei.shape[1]) * d
    return {
        "value": unit / statistics.median(full), "unit": "edges*d/s", "cores": cores, "kind": "port",
        "sample": f"{what}, deg {degree} ({args.degree_dist}), d={d}, nnz={int(ei.shape[1])}, int64 all-ones norm; full "
                  f"AllDeepSets layer fwd+bwd (eval-mode, no dropout), median of {args.cpu_iters} iteration(s) after a warm-up on a "
                  f"1/10-size sample; torch {torch.__version__} CPU, {cores} threads of {os.cpu_count()} logical CPUs",
        "seconds_per_iter": statistics.median(full),
        "aggregation_only": {"value": unit / statistics.median(agg), "seconds_per_iter": statistics.median(agg)},
    }


def hbm_traffic_from_profile(kernel="segreduce_fwd", shape="c3"):
    """(HBM bytes per launch of ``kernel``, source) from the committed rocprofv3 --pmc passes (profiles/), or (None, None).
    ``shape``: "c3" = the headline shape (tools/pmc_probe.py), "c5" = the configs[4] per-GPU shape (tools/pmc_probe_c5.py)."""
    name = "hbm_traffic_c5.json" if shape == "c5" else ("hbm_traffic.json" if kernel == "segreduce_fwd" else "hbm_traffic_pma.json")
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            prof = json.load(open(path))
            val = prof.get(f"{kernel}_bytes_per_launch")
            if prof.get("kernel_source_sha") != kernel_source_sha():
                # the gather kernels (or their flags) changed since the PMC passes were taken: a stale profile is not reported
                return None, (f"profiles/{name} was taken with other kernel sources (its kernel_source_sha "
                              f"{prof.get('kernel_source_sha')!r} != {kernel_source_sha()!r}): re-run tools/pmc_probe"
                              + ("_c5" if shape == "c5" else "") + ".py under rocprofv3 --pmc and tools/pmc_sum.py --stamp")
            if val is not None:
                return val, (f"profiles/{name}: rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes (gfx950 corrections of "
                             "MI355X_MICROARCH.md) of this kernel at exactly this shape, taken with tools/pmc_probe" + ("_c5" if shape == "c5" else "") + ".py; "
                             "not a counter of this run")
        except Exception:
            pass
    return None, None


# ---------------------------------------------------------------------------------------------------------------------
# one timed region
# ---------------------------------------------------------------------------------------------------------------------

def run_partition(args, mode, world, rank, dev, hooks=None):
    """Build partition ``mode``, do W warm-up steps, time exactly K steps between fences, reduce over ranks.
    ``hooks`` (tests only): {"aggregate", "kernels", "incidences"} replace the HIP-backed local aggregation so that the
    N = 2 control flow runs on CPU over gloo; the product never passes them."""
    from allset_amd import dist as adist
    from allset_amd import ops
    from allset_amd.layers import HalfNLHconv
    hooks = hooks or {}
    on_gpu = dev.type == "cuda"
    d = args.d
    hg, nnz_local, n_e_global = build_problem(args, mode, world, rank, dev)
    if "incidences" in hooks:
        hooks["incidences"](hg, mode)
    else:
        hg.build_incidences()

    torch.manual_seed(args.seed)                                       # identical replicated weights on every rank
    attn = args.model == "pma"
    v2e = HalfNLHconv(d, d, d, 2, args.dropout, args.norm, True, heads=args.heads, attention=attn)
    e2v = HalfNLHconv(d, d, d, 2, args.dropout, args.norm, True, heads=args.heads, attention=attn)
    v2e.reset_parameters(); e2v.reset_parameters()
    v2e.to(dev).train(); e2v.to(dev).train()
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if tdt != torch.float32:
        v2e.to(tdt); e2v.to(tdt)
    params = list(v2e.parameters()) + list(e2v.parameters())
    graph_mode = bool(getattr(args, "hip_graph", False)) and on_gpu and world == 1
    # Adam as the product's training driver runs it (allset_amd/train.py: allset_amd.optim.FusedAdam -- torch.optim.Adam's update, one
    # launch for all 24 parameters, step counters on the device, so capturable by construction; tests/test_gpu_dense.py pins its
    # trajectories to torch.optim.Adam's); the host-only plumbing run keeps torch.optim.Adam
    if on_gpu:
        from allset_amd.optim import FusedAdam
        opt = FusedAdam(params, lr=1e-3)
    else:
        opt = torch.optim.Adam(params, lr=1e-3)

    gen = torch.Generator(device=dev).manual_seed(args.seed + 100 + rank)
    rows = hg.v_hi - hg.v_lo
    x = torch.randn(rows, d, device=dev, generator=gen).to(tdt).requires_grad_(True)       # owned vertex block
    G = torch.randn(rows, d, device=dev, generator=gen).to(tdt)
    extra = {}
    if "aggregate" in hooks and not attn:
        extra["aggregate"] = hooks["aggregate"]
    if "kernels" in hooks and attn:
        extra["kernels"] = hooks["kernels"]

    if on_gpu:
        from allset_amd.dense import deferred_param_grads as deferred
    else:
        import contextlib
        deferred = contextlib.nullcontext

    def step():
        opt.zero_grad(set_to_none=True)
        x.grad = None
        if mode == "columns" or mode.startswith("hybrid"):
            out = (adist.colsharded_pma_layer(v2e, e2v, x, hg, dropout=args.dropout, training=True, chunks=args.pipeline_chunks, **extra)
                   if attn else
                   adist.colsharded_deepsets_layer(v2e, e2v, x, hg, aggr="add", dropout=args.dropout, training=True,
                                                   chunks=args.pipeline_chunks, **extra))
        elif attn:
            out = adist.sharded_pma_layer(v2e, e2v, x, hg, dropout=args.dropout, training=True, **extra)
        else:
            out = adist.sharded_deepsets_layer(v2e, e2v, x, hg, aggr="add", dropout=args.dropout, training=True, **extra)
        # as allset_amd/train.py runs its step: every parameter-gradient partial of the backward pass reduced by ONE launch
        with deferred():
            out.backward(G)
        adist.allreduce_grads(params)
        opt.step()

    if graph_mode:            # one hipGraph per step (allset_amd/graphs.py explains what makes the path capturable)
        from allset_amd import dense as _dense
        from allset_amd.graphs import _side_stream_warmup
        counter = torch.zeros(1, dtype=torch.int64, device=dev)
        eager_step = step

        def counted_step():
            counter.add_(1)                        # every replay draws fresh dropout masks: the kernels read the counter
            eager_step()
        with torch.cuda.device(dev), _dense.device_seed_counter(counter):
            _side_stream_warmup(counted_step, 3)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                counted_step()
        step = graph.replay

    def fence():
        if on_gpu:
            torch.cuda.synchronize(dev)
        if dist.is_initialized():
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize(dev)

    # Python's cyclic collector: a generation-2 pass walks every object the process holds (torch alone is ~10^6) and is triggered by
    # allocation counts, i.e. by the autograd nodes of the steps themselves -- tens of milliseconds landing at random inside a timed
    # region (measured on the dataset-scale loop: 4x on an eager step).  Freeze what exists (set-up state never becomes garbage)
    # so the passes that still run only look at the steps' own objects.
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(args.warmup):
        step()
    timer = ops.KernelTimer() if (on_gpu and not graph_mode) else None      # (HIP events per kernel are host calls: eager mode only)
    fence()
    ops.set_kernel_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ops.set_kernel_timer(None)

    stats = torch.tensor([elapsed, float(nnz_local)], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        tmax = adist._all_reduce_(stats.clone(), dist.ReduceOp.MAX)     # (host-staged when the group is gloo, see main())
        tsum = adist._all_reduce_(stats.clone(), dist.ReduceOp.SUM)
        elapsed, nnz_total = float(tmax[0]), float(tsum[1])
    else:
        nnz_total = float(nnz_local)
    res = {"mode": mode, "elapsed": elapsed, "nnz_total": nnz_total, "rows": rows, "n_v": hg.n_v, "n_e": n_e_global,
           "kernels": timer.summary() if timer is not None else {}, "ms_per_step": elapsed / args.steps * 1e3,
           "value": job_value(nnz_total, d, elapsed, args.steps)}
    if hooks.get("keep_inputs") or (world == 1 and rank == 0 and not attn and not args.no_cpu_baseline and on_gpu):
        ei = hg.local_edge_index if hasattr(hg, "local_edge_index") else hg.edge_index
        res["edge_index_cpu"], res["x_cpu"] = ei.detach().cpu(), x.detach().float().cpu()
    return res


def cfg_index(args) -> int:
    """Which BASELINE.json configuration the flags are closest to: [2] AllDeepSets d = 128 fp32 (the headline), [3] AllSetTransformer
    d = 128 fp32, [4] the PMA path on power-law sizes at d = 256 in bf16."""
    if args.model != "pma":
        return 2
    return 4 if (args.dtype == "bf16" and args.degree_dist == "zipf") else 3


def workload_label(args) -> str:
    """'BASELINE configs[i]' only when the flags ARE that configuration (per-GPU shape for the multi-GPU ones); anything else is
    labelled a variant of the closest one, with what differs."""
    i = cfg_index(args)
    diffs = []
    if args.locality > 0:
        diffs.append(f"locality {args.locality:g}")
    if args.self_loops:
        diffs.append("self-loop hyperedges")
    if args.degree != 16:
        diffs.append(f"mean size {args.degree}")
    if i == 2:
        if args.d != 128: diffs.append(f"d = {args.d}")
        if args.dtype != "f32": diffs.append(args.dtype)
        if args.norm != "ln": diffs.append(f"normalization {args.norm}")
        if args.degree_dist != "fixed": diffs.append(f"{args.degree_dist} sizes")
        if args.n_per_gpu != 1_000_000: diffs.append(f"{args.n_per_gpu} rows per GPU")
        if args.dropout != 0.5: diffs.append(f"dropout {args.dropout:g}")
    elif i == 3:
        if args.d != 128: diffs.append(f"d = {args.d}")
        if args.dtype != "f32": diffs.append(args.dtype)
        if args.heads != 4: diffs.append(f"heads {args.heads}")
        if args.degree_dist != "fixed": diffs.append(f"{args.degree_dist} sizes")
        if args.n_per_gpu != 1_000_000: diffs.append(f"{args.n_per_gpu} rows per GPU")
    else:
        if args.d != 256: diffs.append(f"d = {args.d}")
        if args.n_per_gpu != 250_000: diffs.append(f"{args.n_per_gpu} rows per GPU")
    shape = " per-GPU shape" if i != 2 else ""
    if not diffs:
        return f"BASELINE configs[{i}]{shape}"
    return f"variant of BASELINE configs[{i}]{shape} ({', '.join(diffs)})"


def parallelism_label(args, mode, world):
    if world == 1:
        return "single GPU"
    if mode == "rows":
        if rows_halo(args):
            return (f"hyperedge-shard x{world} (boundary-vertex exchange: per direction one all-to-all of the rows of the vertices a rank's "
                    "hyperedges touch + its transpose, allset_amd.dist.Halo)")
        return f"hyperedge-shard x{world} (all-gather + reduce-scatter of the [n_V, d] vertex table per direction)"
    if mode.startswith("hybrid"):
        R = int(mode[len("hybrid"):].split("x")[0])
        C = world // R
        return (f"hybrid {R} target groups x {C} column groups (rank (a, b): the targets of group a, d/{C} columns; per aggregation ONE "
                f"world-wide all-to-all that sends column slice b' of a rank's rows to all {R} ranks (., b') -- the rows -> columns "
                f"change of layout replicated to the {R} target groups, dist._SpreadBlocks --, the local aggregation over rows of "
                f"d/{C} columns, an all-to-all back inside the column group; allset_amd.dist.ColumnShardedHypergraph(row_groups=...))")
    how = f"in {args.pipeline_chunks} overlapped chunks" if args.pipeline_chunks > 1 else "blocking"
    return f"column-shard x{world} (rows for the dense tail, d/{world} columns for the aggregation; all-to-all exchange {how})"


def region_label(args, key, world):
    """parallelism_label for a region key: a partition name plus '+chunksK' / '+bf16wire'."""
    label = parallelism_label(args, key.split("+")[0], world)
    if "+chunks" in key:
        kk = key.split("+chunks")[1].split("+")[0]
        label = label.replace("blocking", f"in {kk} overlapped chunks (the Linear kernels write / read each chunk's exchange buffers, "
                                          "asynchronous all-to-alls)")
    if "+bf16wire" in key:
        label += "; bf16 wire format (opt-in, results within the restated tolerance of tests/test_dist_cpu.py, not bit-comparable)"
    return label


# 16-bit matrix-pipe products per fp32 product in the GEMM kernels: fp16x3 (two scaled fp16 planes, three partial products) under the
# default arithmetic, bf16x6 (three bf16 planes, six) under --arith strict; a bf16 tensor is one product.
GEMM_KERNELS = {"fused_linear_fwd": 1.0, "fused_linear_bwd": 1.0, "wgrad_fused": 1.0, "wgrad": 1.0, "fused_linear_bwd_all": 2.0,
                "gemm_x6": 1.0, "gemm_x6_lnb": 1.0, "linear_bf16_fwd": 1.0, "linear_bf16_bwd": 1.0}


def kernel_entry(v, steps, rows=None, d=None, name=None, products=None):
    """One timed kernel: achieved algorithmic GB/s against the HBM peak and -- for the GEMM kernels -- its matrix-pipe rate against
    the dense 16-bit MFMA peak; `bound` names the roof that is nearer for this kernel's flop / byte (at d = 512 the fp16x3 GEMMs do
    3 x 128 flop per byte: the matrix pipe, not HBM), `roof_frac` the fraction of THAT roof."""
    gbps = (v["algo_bytes"] / (v["avg_ms"] * 1e-3) / 1e9) if v.get("algo_bytes") else None
    out = {"calls_per_step": v["calls"] / steps, "avg_ms": v["avg_ms"], "algo_bytes_per_launch": v.get("algo_bytes") or None,
           "gbps": gbps, "frac": gbps / HBM_PEAK_GBS if gbps else None,
           "frac_of_copy_ceiling": gbps / COPY_CEILING_GBS if gbps else None}
    if rows is not None and name in GEMM_KERNELS:
        flops = GEMM_KERNELS[name] * 2.0 * rows * d * d                    # fp32-equivalent
        out["tflops"] = flops / (v["avg_ms"] * 1e-3) / 1e12
        if products and gbps:
            out["mfma_tflops"] = products * out["tflops"]                     # what the matrix pipe actually executes
            out["mfma_frac"] = out["mfma_tflops"] / MFMA16_PEAK_TFLOPS
            t_hbm, t_mfma = v["algo_bytes"] / (HBM_PEAK_GBS * 1e9), products * flops / (MFMA16_PEAK_TFLOPS * 1e12)
            out["bound"] = "mfma" if t_mfma > t_hbm else "hbm"
            out["roof_frac"] = max(t_hbm, t_mfma) / (v["avg_ms"] * 1e-3)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# launching, hang protection, link preflight
# ---------------------------------------------------------------------------------------------------------------------

def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv=None) -> int:
    """``python bench.py --gpus N`` OUTSIDE ``torch.distributed.run`` (no RANK in the environment): spawn the N ranks here --
    the same launcher command the driver documents, rendezvous on 127.0.0.1 and a free port -- and hand its exit code back.
    Rank 0's stdout (the ONE JSON line) is inherited, so the bare command prints what the launched one prints."""
    import subprocess
    out = []
    for t in list(sys.argv[1:] if argv is None else argv):      # the launcher's own argparse would eat a bare `--d` (see parse_args)
        out.append("--feature-dim" if t == "--d" else ("--feature-dim=" + t[4:] if t.startswith("--d=") else t))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + out
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL between processes needs it on this host driver
    print(f"[bench] --gpus {args.gpus} without a launcher: spawning {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


class Watchdog:
    """One deadline at a time, watched from a daemon thread.  A collective that never completes blocks the main thread inside
    RCCL / a device synchronise (both release the GIL) for ever -- ``try/except`` sees nothing, and the process group's own
    watchdog would abort the process and lose everything measured so far.  When a deadline passes, ``on_expire(label)`` runs on
    this thread; bench.py's handler prints the line assembled from the regions that DID finish and ends the process."""

    def __init__(self, on_expire):
        import threading
        self._cv = threading.Condition()
        self._deadline, self._label, self._on_expire = None, None, on_expire
        threading.Thread(target=self._run, daemon=True, name="bench-watchdog").start()

    def arm(self, seconds: float, label: str) -> None:
        with self._cv:
            self._deadline, self._label = time.monotonic() + seconds, label
            self._cv.notify()

    def disarm(self) -> None:
        with self._cv:
            self._deadline = None
            self._cv.notify()

    def _run(self):
        while True:
            with self._cv:
                if self._deadline is None:
                    self._cv.wait()
                    continue
                left = self._deadline - time.monotonic()
                if left > 0:
                    self._cv.wait(left)
                    continue
                label, self._deadline = self._label, None
            self._on_expire(label)


def _test_hang(label: str, rank: int) -> None:
    """Test hook (tests/test_gpu_two_ranks.py, tests/test_bench_setup.py): ALLSET_BENCH_TEST_HANG="<region>:<rank>" parks that
    rank at the start of that region, so its peers block in the region's first collective exactly as behind a dead link."""
    spec = os.environ.get("ALLSET_BENCH_TEST_HANG", "")
    if spec and spec == f"{label}:{rank}":
        time.sleep(10 ** 6)


def preflight_collectives(args, world: int, rank: int, dev, iters: int = 5):
    """The four collectives the sharded layers issue (allset_amd/dist.py), through the same wrappers and at THIS job's message
    sizes, timed one by one (max over ranks, median of ``iters``): the measured per-link rate that DESIGN.md section 7.3's
    projection needs.  ``sent`` = bytes one rank sends per call; per_link = sent / (N - 1) / time on the fully connected xGMI mesh."""
    from allset_amd import dist as adist
    n, d = args.n_per_gpu, args.d
    on_gpu = dev.type == "cuda"

    def timed(fn):
        ts = []
        for it in range(iters + 1):
            if on_gpu:
                torch.cuda.synchronize(dev)
            dist.barrier()
            t0 = time.perf_counter()
            fn()
            if on_gpu:
                torch.cuda.synchronize(dev)
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            adist._all_reduce_(t, dist.ReduceOp.MAX)
            if it:                                                    # first call: communicator / buffer set-up
                ts.append(float(t[0]))
        return statistics.median(ts)

    f = (world - 1) / world
    out = {}

    def entry(name, sent, sec):
        links = max(world - 1, 1)
        out[name] = {"sent_bytes_per_rank": int(sent), "ms": sec * 1e3, "gbps_per_rank": sent / sec / 1e9,
                     "gbps_per_link": sent / links / sec / 1e9}

    for wire, dt, es in (("f32", torch.float32, 4), ("bf16", torch.bfloat16, 2)):
        own = torch.randn(n, d, device=dev).to(dt)                    # this rank's owned rows, all columns
        part = torch.randn(world * n, d, device=dev).to(dt)           # per-rank partial sums for every vertex
        entry(f"{wire} all_gather [n,d]->[N*n,d] (rows scheme, one per direction)", n * d * es * (world - 1),
              timed(lambda: adist._all_gather_rows(own)))
        if wire == "f32" or dist.get_backend() != "gloo":             # (gloo reduces fp32 only; the bf16 wire never reduce-scatters)
            entry(f"{wire} reduce_scatter [N*n,d]->[n,d] (rows scheme)", world * n * d * es * f,
                  timed(lambda: adist._reduce_scatter_rows(part)))
        if d % world == 0:
            send = own.view(n, world, d // world).transpose(0, 1).contiguous()
            recv = torch.empty_like(send)
            entry(f"{wire} all_to_all [n,d]<->[N*n,d/N] (column scheme, eight per step)", n * d * es * f,
                  timed(lambda: adist._all_to_all_single(recv, send)))
        del own, part
    flat = torch.randn(200_000, device=dev)
    entry("f32 all_reduce of the flat parameter gradient (0.8 MB)", flat.numel() * 4 * 2 * f, timed(lambda: adist._all_reduce_(flat)))
    return {"collectives": out, "iters": iters, "backend": dist.get_backend(),
            "note": "each call timed alone between barriers, max over ranks, median; per_link = sent / (N-1) links / time"}


def kernel_source_sha() -> str:
    """Hash of what determines segreduce_kernel's code object: its source, the shared header and the compiler flags.  Stored
    beside profiles/hbm_traffic*.json when the PMC passes are taken (tools/pmc_sum.py --stamp); ``roofline.traffic`` is null
    when the library in use was built from anything else."""
    import hashlib
    from allset_amd import build as _build
    h = hashlib.sha256()
    for name in ("segreduce.hip", "pma.hip", "common.h"):
        with open(os.path.join(_build.CSRC, name), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(_build.CXXFLAGS).encode())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------------------
# the JSON line
# ---------------------------------------------------------------------------------------------------------------------

def assemble_line(args, world, primary, state, cpu_mode=False, final=True):
    """The ONE JSON line from whatever regions have finished (``state['results']``: partition name -> run_partition result).
    ``value`` is the primary partition's if it finished, otherwise the first finished region's (labelled).  None if nothing has."""
    results, errors = state["results"], state["errors"]
    # `value`: --shard rows / columns = that partition (or, if it did not finish, the first one that did, labelled);
    # --shard auto at N > 1 = the FASTEST of the exact (fp32-wire) executions of the job this run timed -- rows, columns, columns
    # with the chunked overlapped exchange: same global hypergraph, same K steps each, same results up to summation order -- i.e.
    # the partition an autotuning deployment would keep.  The bf16-wire entry changes results and is never `value`.
    exact = [k for k in state["order"] if k in results and "bf16wire" not in k]
    fastest = min(exact, key=lambda k: results[k]["ms_per_step"]) if exact else None
    if args.shard == "auto" and world > 1 and exact:
        value_key = min(exact, key=lambda k: results[k]["ms_per_step"])
    else:
        value_key = primary if primary in results else (exact[0] if exact else None)
    if value_key is None:
        return None
    res = results[value_key]
    d, attn = args.d, args.model == "pma"
    products = 1.0 if args.dtype == "bf16" else (6.0 if args.arith in ("strict", "bf16x6") else 3.0)     # (GEMM_KERNELS)
    ks = res["kernels"]
    agg_ks = {k: v for k, v in ks.items() if k in AGG_KERNELS}                   # HBM-bound gather kernels
    dense_ks = {k: v for k, v in ks.items() if k not in AGG_KERNELS}             # dense tail (MFMA / streaming)
    dom = max(agg_ks, key=lambda k: agg_ks[k]["total_ms"]) if agg_ks else None   # dominant aggregation kernel
    seg = agg_ks.get(dom) if dom else None
    agg_ms = sum(v["total_ms"] for v in agg_ks.values()) / args.steps
    dense_ms = sum(v["total_ms"] for v in dense_ks.values()) / args.steps
    agg_bytes = sum(v["algo_bytes"] * v["calls"] for v in agg_ks.values() if v.get("algo_bytes")) / args.steps
    all_bytes = sum(v["algo_bytes"] * v["calls"] for v in ks.values() if v.get("algo_bytes")) / args.steps
    # the PMC passes were taken at exactly this shape (tools/pmc_probe.py); any other shape reports null
    at_profiled_shape = (world == 1 and args.n_per_gpu == 1_000_000 and d == 128 and args.degree == 16
                         and args.degree_dist == "fixed" and args.dtype == "f32" and not args.self_loops
                         and (not attn or args.heads == 4))
    at_c5_shape = (world == 1 and args.n_per_gpu == 250_000 and d == 256 and args.degree == 16 and args.degree_dist == "zipf"
                   and args.dtype == "bf16" and not args.self_loops and attn and args.heads == 4 and args.seed == 20260928)
    traffic, traffic_source = hbm_traffic_from_profile(dom) if (at_profiled_shape and dom) else (
        hbm_traffic_from_profile(dom, "c5") if (at_c5_shape and dom) else (None, None))
    roofline = None
    if seg:
        achieved = seg["algo_bytes"] / (seg["avg_ms"] * 1e-3) / 1e9
        step_s = res["ms_per_step"] * 1e-3
        roofline = {"bound": "hbm", "kernel": f"allset_{dom}",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_source,
                    "frac_of_copy_ceiling": achieved / COPY_CEILING_GBS,
                    "layer_frac": agg_bytes / step_s / (HBM_PEAK_GBS * 1e9) if world == 1 else None,
                    "layer_frac_all_bytes": all_bytes / step_s / (HBM_PEAK_GBS * 1e9) if world == 1 else None,
                    "layer_note": "layer_frac = the aggregation's algorithmic bytes per step (all gather passes, SURVEY 8(d3)) / the WHOLE "
                                  "step time (dense tail and Adam included) / peak: the north star's '>= 40 % of HBM roofline on the "
                                  "V->E->V aggregation' held against the full step; layer_frac_all_bytes also counts the dense "
                                  "tail's algorithmic activation bytes",
                    "note": "achieved = SURVEY 8(d3) gather-model bytes / HIP-event launch time; the rate can exceed the 6.3 TB/s "
                            "streaming-copy ceiling because each source row is gathered `degree` times and part of the table is "
                            "served by the 256 MiB Infinity Cache (FETCH_SIZE counts at the L2's fabric side, cache hits "
                            "included); DRAM-only bytes are not exposed by rocprofv3 on gfx950",
                    "algo_bytes_per_launch": seg["algo_bytes"], "avg_launch_ms": seg["avg_ms"], "launches": seg["calls"],
                    "per_kernel": {k: kernel_entry(v, args.steps, res["rows"], d, k, products) for k, v in ks.items() if v.get("algo_bytes")}}
    nnz_total = res["nnz_total"]
    line = {
        "metric": "edges*d aggregated / sec (V->E->V layer fwd+bwd)", "value": res["value"], "unit": "edges*d/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "arithmetic": (f"{args.arith}: `value` is measured with fp32 products formed from two scaled fp16 planes (fp16x3) wherever that kernel is "
                       "built; the same K steps on the exact three-plane bf16 split (bf16x6) are timed in the same run: `strict`"
                       if args.dtype == "f32" and args.arith in ("auto", "fp16x3") else args.arith),
        "dtype_note": ("fp32 tensors end to end; aggregation kernels: plain fp32 adds; dense tail: fp32 products formed on the 16-bit "
                       "matrix pipes and accumulated in fp32 -- at 128 x 128 (the forward and the one-pass backward) and in the tiled "
                       "256 / 512-wide GEMMs and weight gradient: every operand scaled by a power of two and split into two fp16, three "
                       "of the four partial products (fp16x3, per-product error <= 2^-21 relative); other shapes, and everything under "
                       "--arith strict: every operand split exactly into three bf16, six of the nine partial products (bf16x6) -- "
                       "measured error vs float64 at the level of "
                       "a native fp32 MFMA / library fp32 GEMM (DESIGN.md section 6.1, tests/test_gpu_dense.py)")
        if args.dtype == "f32" else
        ("bf16 tensors end to end (BASELINE configs[4] regime): bf16 instantiations of the gather kernels with fp32 "
         "accumulation and fp32 softmax statistics; dense tail: this library's bf16 kernels (fp32 arithmetic and "
         "accumulation, bf16 in / out)"),
        "data": "synthetic",
        "config": {"workload": f"{workload_label(args)}: synthetic random hypergraph "
                               f"|V|=|E|={args.n_per_gpu} per GPU, hyperedge size {args.degree} ({args.degree_dist}"
                               + (f", locality {args.locality:g}: VARIANT workload" if args.locality > 0 else "") + "), "
                               f"nnz={int(nnz_total)}, d={d}, " +
                               (f"AllSetTransformer layer (PMA x2, heads={args.heads}, dropout {args.dropout}), " if attn else
                                f"AllDeepSets layer (HalfNLHconv x2, 2-layer {args.norm.upper()} MLPs, aggr=add, dropout {args.dropout}), ") +
                               "fwd+bwd+Adam" + (" + one singleton self-loop hyperedge per vertex" if args.self_loops else ""),
                   "n_v": res["n_v"], "n_e": res["n_e"], "nnz": int(nnz_total), "d": d,
                   "parallelism": region_label(args, value_key, world), "partition": value_key if world > 1 else None,
                   "seed": args.seed, "launch": "one hipGraph replay per step" if (args.hip_graph and world == 1) else "eager launches"},
        "roofline": roofline,
        "aggregation": {"ms_per_step": agg_ms, "value": nnz_total * d / (agg_ms * 1e-3) if (world == 1 and agg_ms > 0) else None,
                        "unit": "edges*d/s", "note": "gather/segment-reduce kernel time per step (HIP events, rank 0): "
                        "the aggregation-only V->E->V fwd+bwd",
                        "kernels": {k: kernel_entry(v, args.steps) for k, v in agg_ks.items()}},
        "dense_tail": {"ms_per_step": dense_ms, "note": ("HIP dense-tail kernels per step (fused norm+Linear forward; ONE backward "
                       "kernel per Linear producing input gradient, LayerNorm parameter gradients and the weight / bias "
                       "gradient from a single read of gy and x). fp32 in, fp32 out, fp32-accurate arithmetic on the 16-bit "
                       "matrix pipes: fp16x3 (2 scaled fp16 planes, 3 of 4 products) in the LayerNorm-prologue forward and the one-pass "
                       "backward, bf16x6 (3 bf16 planes, 6 of 9 products) elsewhere, accumulated in fp32 (error at a library "
                       "fp32 GEMM's level, tests/test_gpu_dense.py); HBM-bound: gbps = algorithmic activation bytes / time")
                       if args.dtype == "f32" else
                       ("HIP dense-tail kernels per step, bf16 in / out with fp32 accumulation: Linear forward / backward-data "
                        "with relu, folded logits, relu mask and gradient-branch sums in the same pass (csrc/fused_bf16.hip), "
                        "full-width weight gradient, add+LayerNorm kernels; gbps = algorithmic activation bytes / time"),
                       "kernels": {k: kernel_entry(v, args.steps, res["rows"], d, k, products) for k, v in dense_ks.items()}},
    }
    if dist.is_initialized():
        line["config"]["collectives"] = ("gloo, device tensors staged through the host, ranks may share a device (test mode)"
                                         if dist.get_backend() == "gloo" and not cpu_mode else
                                         ("gloo (CPU test)" if cpu_mode else "RCCL (torch.distributed nccl backend)"))
    if world > 1 or len(state["order"]) > 1:
        parts = {}
        for key in state["order"]:
            label = region_label(args, key, world)
            if key in results:
                parts[key] = {"ms_per_step": results[key]["ms_per_step"], "value": results[key]["value"], "parallelism": label,
                              "is_value": key == value_key}
            elif key in errors:
                parts[key] = {"error": errors[key], "parallelism": label, "is_value": False}
            elif not final:
                parts[key] = {"pending": True, "parallelism": label, "is_value": False}
        if args.shard == "auto" and world > 1:
            parts["value_note"] = (f"--shard auto: `value` is the fastest exact (fp32-wire) execution this run timed, {value_key!r} "
                                   f"(allset_amd.dist.choose_sharding's a-priori pick was {primary!r}"
                                   + ("" if primary in results else (", which did not finish: see its entry" if final else
                                                                     ", not run yet") ) + ")")
        elif value_key != primary:
            parts["value_note"] = (f"`value` is partition {value_key!r}: the partition this run would report ({primary!r}) did not "
                                   "finish (see its entry)")
        if fastest is not None and world > 1:
            parts["fastest_exact"] = fastest          # (a label: `value` is --shard's partition, `rows` by default -- the north star's)
        parts["note"] = ("`rows` = hyperedge shards, the partition BASELINE.json's north star names; `columns` = column-sharded "
                         "aggregation (DESIGN.md section 7.2). Same global hypergraph, same K steps, separate timed regions (rows first: "
                         "its all-gather / reduce-scatter are the plainest collectives, so a scaling record exists before the "
                         "all-to-all paths run); `value` / `ms_per_step` of the line are those of the entry with is_value = true")
        line["partitions"] = parts
    if state.get("preflight") is not None or "preflight" in errors:
        line["preflight"] = state.get("preflight") or {"error": errors["preflight"]}
    strict = state.get("strict")
    if args.dtype == "f32":
        from allset_amd import dense as _dense
        line["dense_tail"]["arithmetic"] = _dense.get_arithmetic()
    if strict is not None:
        sks = {k: v for k, v in strict["kernels"].items() if k not in AGG_KERNELS}
        strict_dense = sum(v["total_ms"] for v in sks.values()) / args.steps
        line["dense_tail"]["strict_ms_per_step"] = strict_dense
        line["strict"] = {"ms_per_step": strict["ms_per_step"], "value": strict["value"], "dense_tail_ms_per_step": strict_dense,
                          "arithmetic": "bf16x6",
                          "note": "the same K steps in the same run with allset_amd.dense.set_arithmetic('strict') "
                                  "(ALLSET_ARITH_BF16X6: every fused Linear on the exact three-bf16-plane split, no dependence on the "
                                  "data's dynamic range); `value` is the default (AUTO) arithmetic's",
                          "kernels": {k: kernel_entry(v, args.steps, strict["rows"], d, k, 6.0) for k, v in sks.items()}}
    elif "strict" in errors:
        line["strict"] = {"error": errors["strict"]}
    line["cpu_baseline"] = state.get("cpu_baseline")
    return line


def main(argv=None, hooks=None):
    args = parse_args(argv)
    hooks = hooks or {}
    if "RANK" not in os.environ and args.gpus > 1 and hooks.get("device") != "cpu":
        raise SystemExit(self_launch(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    cpu_mode = hooks.get("device") == "cpu"
    if cpu_mode:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU"
        # ALLSET_DIST_BACKEND=gloo: the ranks exchange through host memory and may SHARE a device (local rank modulo the
        # device count) -- how the N = 2 control flow runs the HIP path with a real peer on a 1-GPU box
        # (tests/test_gpu_two_ranks.py); RCCL itself refuses two ranks on one device.  Never the driver's mode.
        backend = os.environ.get("ALLSET_DIST_BACKEND", "nccl")
        if backend == "gloo":
            local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    force = os.environ.get("ALLSET_FORCE_COLLECTIVES", "0") == "1"      # 1-rank RCCL group: API check on a 1-GPU box
    if (world > 1 or (force and "RANK" in os.environ)) and not dist.is_initialized():
        import datetime
        # the group's own watchdog must never fire before bench.py's (it aborts the process and the line with it)
        pg_timeout = datetime.timedelta(seconds=max(3600.0, 4 * args.region_timeout))
        if cpu_mode or backend == "gloo":
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)

    from allset_amd import dist as adist
    if not cpu_mode:
        from allset_amd import _lib
        _lib.load()                                                   # fail loudly if the HIP library is absent
    if args.pipeline_chunks <= 0:
        args.pipeline_chunks = adist.auto_chunks(args.n_per_gpu)
    primary, other = resolve_modes(args, world)

    # Region order.  N > 1 with both partitions: `rows` first whatever the primary is -- the north-star partition, on the
    # plainest collectives (one all-gather, one reduce-scatter) -- so that a scaling record exists before any all-to-all runs.
    order = [primary] + ([other] if other is not None else [])
    if other is not None and "rows" in order:
        order = ["rows"] + [m for m in order if m != "rows"]
    # N = 2 x C >= 4: the hybrid partition (2 target groups x C column groups, 128-byte gather rows at d = 128, C = 4) right after
    # `rows` -- its own `partitions` entry; `value` only under --shard auto (the fastest exact execution) or --shard hybrid
    hyb = hybrid_mode(args, world) if (args.partitions == "both" or args.shard == "hybrid") else None
    if args.shard == "hybrid":
        if hyb is None:
            raise SystemExit(f"--shard hybrid: no hybrid partition for {world} ranks at d = {args.d}")
        primary = hyb
        order = [hyb if m == "hybrid" else m for m in order]
        if "rows" in order:
            order = ["rows"] + [m for m in order if m != "rows"]
    elif hyb is not None:
        order = (order[:1] + [hyb] + order[1:]) if order[0] == "rows" else (order + [hyb])
    wire_key = primary + "+bf16wire"
    want_wire = world > 1 and args.dtype == "f32" and args.wire_entry and args.pipeline_chunks <= 1
    # N > 1: the column partition once more with the chunked, overlapped exchange (off by default: it has never run over xGMI;
    # its own `partitions` entry so that the first multi-GPU lease measures what the overlap is worth -- never `value`)
    kc = adist.auto_chunks(args.n_per_gpu) if args.chunk_entry < 0 else args.chunk_entry
    chunk_key = f"columns+chunks{kc}"
    want_chunks = world > 1 and "columns" in order and args.pipeline_chunks <= 1 and kc > 1
    want_preflight = dist.is_initialized() and (args.preflight == "on" or (args.preflight == "auto" and world > 1))
    state = {"results": {}, "errors": {}, "order": order + ([chunk_key] if want_chunks else []) + ([wire_key] if want_wire else []),
             "preflight": None, "cpu_baseline": None}
    exit_fn = hooks.get("exit", os._exit)

    def on_expire(label):
        state["errors"][label] = (f"timeout: region did not finish within its deadline ({args.region_timeout:.0f} s; first region "
                                  f"{3 * args.region_timeout:.0f} s) -- a collective that never completed; bench.py's watchdog printed the "
                                  "line from the regions that had finished and ended the process")
        print(f"[bench] rank {rank}: region {label!r} timed out; ending the process", file=sys.stderr, flush=True)
        if state.get("printed"):                                      # (teardown: the line is out already)
            exit_fn(0)
        line = assemble_line(args, world, primary, state, cpu_mode) if rank == 0 else None
        if line is not None:
            sys.stdout.write(json.dumps(line) + "\n")
            sys.stdout.flush()
        exit_fn(0 if (rank != 0 or line is not None) else 3)

    dog = Watchdog(on_expire) if dist.is_initialized() else None

    def region(label, fn, first=False):
        """Run ``fn`` under the watchdog.  The first region may raise (nothing to salvage); later ones are recorded and skipped."""
        if dog is not None:
            dog.arm(args.region_timeout * (3 if first else 1), label)
        try:
            _test_hang(label, rank)
            return fn()
        except Exception as exc:                                     # noqa: BLE001 -- reported in the line, earlier results stand
            if first:
                raise
            state["errors"][label] = f"{type(exc).__name__}: {exc}"
            print(f"[bench] region {label!r} failed on rank {rank}: {state['errors'][label]}", file=sys.stderr, flush=True)
            return None
        finally:
            if dog is not None:
                dog.disarm()

    def partition_region(key, mode, first=False):
        if not cpu_mode and not first:
            torch.cuda.empty_cache()
        res = region(key, lambda: run_partition(args, mode, world, rank, dev, hooks), first)
        if res is not None:
            state["results"][key] = res

    if not cpu_mode and args.dtype == "f32":
        from allset_amd import dense as _dense
        _dense.set_arithmetic(args.arith)
    partition_region(order[0], order[0], first=True)
    if rank == 0:                                   # the early line: stderr only -- stdout carries exactly ONE JSON line, the final one
        early = assemble_line(args, world, primary, state, cpu_mode, final=False)
        print("[bench] early line (first region fenced): " + json.dumps(early), file=sys.stderr, flush=True)
    if want_preflight:
        state["preflight"] = region("preflight", lambda: preflight_collectives(args, world, rank, dev))
    for mode in order[1:]:
        partition_region(mode, mode)
    if want_chunks:
        args.pipeline_chunks = kc
        try:
            partition_region(chunk_key, "columns")
        finally:
            args.pipeline_chunks = 1
    # N > 1: the primary partition once more with the opt-in bf16 WIRE format (allset_amd.dist.set_wire_dtype: fp32 tensors and
    # fp32 sums, every exchanged activation rounded once to bf16 -- half the bytes per link, results changed within the tolerance
    # tests/test_dist_cpu.py restates).  Its own `partitions` entry, never `value`.
    if want_wire:
        prev = adist.set_wire_dtype(torch.bfloat16)
        try:
            partition_region(wire_key, primary)
        finally:
            adist.set_wire_dtype(prev)

    # N = 1, fp32: the same K steps once more in the STRICT arithmetic (the exact-split bf16x6 kernels everywhere) -- both numbers in
    # one run, the default's as `value`
    if (world == 1 and not cpu_mode and args.dtype == "f32" and args.strict_entry and args.arith == "auto" and not args.hip_graph):
        from allset_amd import dense as _dense
        prev_arith = _dense.set_arithmetic("strict")
        try:
            torch.cuda.empty_cache()
            sres = region("strict", lambda: run_partition(args, primary, world, rank, dev, hooks))
            if sres is not None:
                sres.pop("edge_index_cpu", None); sres.pop("x_cpu", None)
                state["strict"] = sres
        finally:
            _dense.set_arithmetic(prev_arith)

    line = None
    if rank == 0:
        res = state["results"].get(primary) or next(iter(state["results"].values()))
        if world == 1 and not args.no_cpu_baseline and args.model != "pma" and not cpu_mode:
            state["cpu_baseline"] = cpu_baseline(args, args.d, args.degree, res.get("edge_index_cpu"), res.get("x_cpu"))
        line = assemble_line(args, world, primary, state, cpu_mode)
        print(json.dumps(line), flush=True)
    state["printed"] = True
    if dist.is_initialized() and not hooks.get("keep_group"):
        if dog is not None:
            dog.arm(args.region_timeout, "teardown")                  # a peer that died in a later region must not hang the exit
        dist.destroy_process_group()
        if dog is not None:
            dog.disarm()
    return line


if __name__ == "__main__":
    main()
