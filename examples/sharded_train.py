#!/usr/bin/env python
"""Multi-GPU training of an AllSet model: one process per GPU, hyperedge shards, the model wrapped once.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 examples/sharded_train.py \
        [--method AllDeepSets|AllSetTransformer] [--partition rows|rows+halo|columns] [--epochs 20] [--eval-mode]

What a maintainer of the reference would write to run `train.py`'s loop body (reference train.py:470-476: forward,
nll_loss(log_softmax) on the train split, backward, optimizer step) on N GPUs.  The reference itself is single-device; this is the
usage of `allset_amd.dist` shown in INTEGRATION.md section 3b, runnable.  Data: the planted-partition task of `allset_amd.train
--dname synthetic`, its vertices renumbered so that a class's vertices are contiguous -- hyperedges then mostly stay inside one
rank's vertex block, the case the boundary-vertex exchange (`--partition rows+halo`) is for.
`--eval-mode`: forward with dropouts off (gradients on): the loss sequence is then a deterministic function of the initial weights
and must not depend on N or on the partition (tests/test_gpu_two_ranks.py runs it with 1 and 2 ranks and compares).
ALLSET_DIST_BACKEND=gloo lets several ranks share one device (tests); the default backend is nccl (= RCCL)."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allset_amd import SetGNN, dist as adist            # noqa: E402
from allset_amd.train import build_parser, synthetic_dataset  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--method", default="AllDeepSets", choices=["AllDeepSets", "AllSetTransformer"])
    ap.add_argument("--partition", default="rows+halo", choices=["rows", "rows+halo", "columns"])
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--eval-mode", action="store_true")
    ap.add_argument("--n-v", type=int, default=4000)
    cli = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    backend = os.environ.get("ALLSET_DIST_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count() if backend == "gloo" else int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    # ---- data (identical on every rank: same seed), vertices sorted by class so that hyperedges have locality
    data = synthetic_dataset(n_v=cli.n_v, n_e=cli.n_v // 2, num_classes=8, num_features=cli.hidden, purity=0.9, seed=0)
    order = torch.argsort(data.y, stable=True)
    new_id = torch.empty_like(order)
    new_id[order] = torch.arange(order.numel())
    n_v = cli.n_v
    ei = data.edge_index[:, data.edge_index[0] < n_v]                      # the V -> E half of the block edge list
    ei = torch.stack([new_id[ei[0]], ei[1] - n_v])
    x, y = data.x[order], data.y[order]
    n_e = int(ei[1].max()) + 1
    g = torch.Generator().manual_seed(1)
    train_mask = torch.rand(n_v, generator=g) < 0.5

    args = build_parser().parse_args(["--method", cli.method, "--MLP_hidden", str(cli.hidden), "--Classifier_hidden", str(cli.hidden),
                                      "--All_num_layers", "1", "--heads", "4"])
    args.PMA = cli.method == "AllSetTransformer"
    args.aggregate = "add"
    args.num_features, args.num_classes = x.shape[1], 8
    torch.manual_seed(0)                                                   # replicated initial weights
    model = SetGNN(args).to(dev)
    model.eval() if cli.eval_mode else model.train()

    # ---- this rank's shard
    if cli.partition == "columns":
        hg = adist.ColumnShardedHypergraph(ei.to(dev), n_v, n_e, world, rank, norm=torch.ones(ei.shape[1], dtype=torch.int64, device=dev))
    else:
        owner = adist.partition_hyperedges(torch.bincount(ei[1], minlength=n_e), world, "lpt")
        loc, gids = adist.local_shard(ei, owner, rank)
        keep = owner[ei[1]] == rank
        hg = adist.ShardedHypergraph(loc.to(dev), n_v, gids.numel(), world, rank, norm=torch.ones(int(keep.sum()), dtype=torch.int64, device=dev),
                                     halo=cli.partition == "rows+halo")
    hg.build_incidences()
    sharded = adist.ShardedSetGNN(model, hg)
    opt = torch.optim.Adam(model.parameters(), lr=cli.lr)

    lo, hi = hg.v_lo, min(hg.v_hi, n_v)
    xp = torch.cat([x, x.new_zeros(hg.n_v_pad - n_v, x.shape[1])])[hg.v_lo:hg.v_hi].to(dev)
    y_own, m_own = y[lo:hi].to(dev), train_mask[lo:hi].to(dev)
    n_train = int(train_mask.sum())
    for epoch in range(cli.epochs):
        opt.zero_grad()
        logits = sharded(xp)[:hi - lo]
        # the global mean over the train split = sum over ranks of (local sum / global count): gradients add up in the all-reduce
        loss = F.nll_loss(F.log_softmax(logits[m_own], dim=1), y_own[m_own], reduction="sum") / n_train
        loss.backward()
        sharded.allreduce_grads()
        opt.step()
        tot = loss.detach().clone()
        if world > 1:
            adist._all_reduce_(tot)
        if rank == 0:
            print(f"epoch {epoch:3d} loss {float(tot):.6f}", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
