#!/usr/bin/env python
"""xGMI / RCCL micro-benchmark for the first multi-GPU lease: the four collectives allset_amd/dist.py issues, at the message
sizes of the bench's N-rank job, so that DESIGN.md section 7.3's assumed link rate (60 GB/s per link and direction) can be
replaced by a measurement.  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/rccl_microbench.py \
        [--rows-per-gpu 1000000] [--d 128] [--iters 20] [--dtypes f32,bf16]

Prints per collective: bytes each rank SENDS per call, median time, per-rank algorithm bandwidth (bytes sent / time) and the
per-link rate it implies on a fully connected xGMI mesh (bytes sent / (N - 1) links / time).  On a 1-rank group (a 1-GPU box)
it only checks that every call runs (ALLSET_FORCE_COLLECTIVES-style)."""
import argparse, os, statistics, sys
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--rows-per-gpu", type=int, default=1_000_000)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dtypes", default="f32,bf16")
args = ap.parse_args()
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
n, d = args.rows_per_gpu, args.d


def timed(fn):
    fn(); torch.cuda.synchronize(); dist.barrier()
    ts = []
    for _ in range(args.iters):
        torch.cuda.synchronize(); dist.barrier()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        t = torch.tensor([s.elapsed_time(e)], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ts.append(float(t))
    return statistics.median(ts)


def report(name, sent_bytes, ms):
    if rank == 0:
        links = max(world - 1, 1)
        print(f"{name:44s} {sent_bytes / 1e6:9.1f} MB sent/rank  {ms:8.3f} ms  {sent_bytes / ms / 1e6:8.1f} GB/s per rank  "
              f"{sent_bytes / links / ms / 1e6:7.1f} GB/s per link", flush=True)


for name in args.dtypes.split(","):
    dt = torch.float32 if name == "f32" else torch.bfloat16
    es = 4 if name == "f32" else 2
    own = torch.randn(n, d, device=dev).to(dt)                              # this rank's rows, all columns
    full = torch.empty(world * n, d, device=dev, dtype=dt)                   # the row scheme's gathered table
    part = torch.randn(world * n, d, device=dev).to(dt)                      # per-rank partial sums for all vertices
    out = torch.empty(n, d, device=dev, dtype=dt)
    f = (world - 1) / world
    report(f"[{name}] all_gather [n/P,d] -> [n,d] (rows, V->E)", n * d * es * (world - 1), timed(lambda: dist.all_gather_into_tensor(full, own)))
    report(f"[{name}] reduce_scatter [n,d] -> [n/P,d] (rows, E->V)", int(world * n * d * es * f), timed(lambda: dist.reduce_scatter_tensor(out, part)))
    send = torch.randn(world, n, d // max(world, 1), device=dev).to(dt) if d % max(world, 1) == 0 else None
    if send is not None:
        recv = torch.empty_like(send)
        report(f"[{name}] all_to_all [n/P,d] <-> [n,d/P] (columns)", int(n * d * es * f), timed(lambda: dist.all_to_all_single(recv, send)))
        outs, ins = list(recv.unbind(0)), list(send.unbind(0))
        report(f"[{name}] list all_to_all, same pieces (chunked path)", int(n * d * es * f), timed(lambda: dist.all_to_all(outs, ins)))
    flat = torch.randn(200_000, device=dev)
    report(f"[{name}] all_reduce of the flat gradient (0.8 MB)", flat.numel() * 4 * 2 * f, timed(lambda: dist.all_reduce(flat)))
dist.destroy_process_group()
