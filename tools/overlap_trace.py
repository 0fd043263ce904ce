#!/usr/bin/env python
"""Does chunk k's exchange really run while chunk k+1 is in its MLP?  Timestamps, not assumption.

Two REAL ranks on one GPU (gloo, device tensors staged through the host -- the only two-rank wire a 1-GPU box has; RCCL refuses
two ranks on one device), the product's chunked column-sharded AllDeepSets layer (allset_amd.dist.colsharded_deepsets_layer,
chunks = K, repack-free chunks), forward + backward.  Recorded per rank:
  * every asynchronous all-to-all: host time at issue, host time its future completed (gloo's own thread), host time the
    consumer's wait() returned;
  * every MLP call (`HalfNLHconv._mlp_act`, forward) and every autograd node of an MLP (backward hooks): HIP events at start
    and end, mapped onto the host clock through one synchronised reference event.
Printed: the timeline of step 3 and, per direction, the share of all-to-all time during which an MLP kernel of ANOTHER chunk was
executing on the device.  What this shows is the WIRING (issue order, who waits for what); the wire itself is the host here, not
xGMI -- on RCCL the same calls run on the communicator's stream.

    ALLSET_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29577 tools/overlap_trace.py [--rows 200000] [--chunks 4]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from allset_amd import dist as adist
from allset_amd.layers import HalfNLHconv
from allset_amd.synthetic import random_hypergraph

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=200_000)
ap.add_argument("--chunks", type=int, default=4)
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
n, d, K = args.rows, 128, args.chunks
blocks = [random_hypergraph(n * world, n, 16, seed=5 + r, device=dev) for r in range(world)]
ei = torch.cat([torch.stack([b.edge_index[0], b.edge_index[1] + r * n]) for r, b in enumerate(blocks)], dim=1)
hg = adist.ColumnShardedHypergraph(ei, n * world, n * world, world, rank, norm=torch.cat([b.norm for b in blocks]), chunks=K).build_incidences()
torch.manual_seed(0)
a = HalfNLHconv(d, d, d, 2, 0.5, "ln", True, attention=False).to(dev).train()
b = HalfNLHconv(d, d, d, 2, 0.5, "ln", True, attention=False).to(dev).train()
x = torch.randn(hg.v_hi - hg.v_lo, d, device=dev).requires_grad_(True)
G = torch.randn_like(x)

LOG = []            # (kind, label, t0, t1) on the host clock, seconds
PENDING_EVENTS = []  # (label, start_event, end_event)
_orig_a2a = adist._a2a_async
_seq = [0]


class _TracedWork:
    def __init__(self, work, label, t_issue):
        self.work, self.label, self.t_issue, self.t_done = work, label, t_issue, None
        inner = getattr(work, "work", work)
        try:
            inner.get_future().then(lambda f, s=self: setattr(s, "t_done", time.perf_counter()))
        except Exception:
            pass

    def wait(self):
        self.work.wait()
        t = time.perf_counter()
        LOG.append(("a2a", self.label, self.t_issue, self.t_done if self.t_done is not None else t, t))


def traced_a2a(out_views, in_views, group):
    _seq[0] += 1
    torch.cuda.synchronize(dev)          # the chunk that is sent exists (the host-staged wire would synchronise in .cpu() anyway)
    t = time.perf_counter()
    return _TracedWork(_orig_a2a(out_views, in_views, group), f"a2a#{_seq[0]}", t)


adist._a2a_async = traced_a2a
_orig_act = HalfNLHconv._mlp_act
_calls = [0]


def traced_act(self, mlp, t, p, **kw):
    _calls[0] += 1
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = _orig_act(self, mlp, t, p, **kw)
    e.record()
    label = f"mlp#{_calls[0]} fwd"
    PENDING_EVENTS.append((label, s, e))
    if out.requires_grad:
        bs, be = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        out.register_hook(lambda g, ev=bs: (ev.record(), g)[1])                        # backward of this MLP starts when its output gradient exists
        if t.requires_grad:
            t.register_hook(lambda g, ev=be, lb=label, s0=bs: (ev.record(), PENDING_EVENTS.append((lb.replace("fwd", "bwd"), s0, ev)), g)[2])
    return out


HalfNLHconv._mlp_act = traced_act


def step():
    x.grad = None
    out = adist.colsharded_deepsets_layer(a, b, x, hg, aggr="add", dropout=0.5, training=True, chunks=K)
    out.backward(G)


for it in range(3):
    LOG.clear(); PENDING_EVENTS.clear(); _seq[0] = 0; _calls[0] = 0
    torch.cuda.synchronize(dev); dist.barrier()
    ref = torch.cuda.Event(enable_timing=True); ref.record(); torch.cuda.synchronize(dev); t_ref = time.perf_counter()
    step()
    torch.cuda.synchronize(dev)
for label, s, e in PENDING_EVENTS:
    LOG.append(("mlp", label, t_ref + ref.elapsed_time(s) * 1e-3, t_ref + ref.elapsed_time(e) * 1e-3, None))
LOG.sort(key=lambda r: r[2])
if rank == 0:
    print(f"# tools/overlap_trace.py: {world} ranks on one MI355X over gloo (host-staged), {n} owned rows per rank, d = {d}, {K} chunks; "
          f"rank 0, third step; times in ms from the step's start")
    for kind, label, t0, t1, t2 in LOG:
        extra = f"   consumer's wait() returned at {1e3 * (t2 - t_ref):8.2f}" if kind == "a2a" else ""
        print(f"  {kind:4s} {label:14s} {1e3 * (t0 - t_ref):8.2f} -> {1e3 * (t1 - t_ref):8.2f}{extra}")
    mlps = [(t0, t1) for k, _, t0, t1, _ in LOG if k == "mlp"]
    tot = ov = 0.0
    for k, _, t0, t1, _ in LOG:
        if k != "a2a":
            continue
        tot += t1 - t0
        # device-busy-with-an-MLP time inside [t0, t1]
        segs = sorted((max(t0, m0), min(t1, m1)) for m0, m1 in mlps if min(t1, m1) > max(t0, m0))
        cur = t0
        for s0, s1 in segs:
            s0 = max(s0, cur)
            if s1 > s0:
                ov += s1 - s0
                cur = s1
    a2as = [(t0, t1) for k, _, t0, t1, _ in LOG if k == "a2a"]
    mtot = mov = 0.0
    for m0, m1 in mlps:
        mtot += m1 - m0
        segs = sorted((max(m0, a0), min(m1, a1)) for a0, a1 in a2as if min(m1, a1) > max(m0, a0))
        cur = m0
        for s0, s1 in segs:
            s0 = max(s0, cur)
            if s1 > s0:
                mov += s1 - s0
                cur = s1
    print(f"# MLP kernel time {1e3 * mtot:.2f} ms (sum over {len(mlps)} forward calls / backward nodes); of it {1e3 * mov:.2f} ms = "
          f"{100 * mov / max(mtot, 1e-9):.0f} % ran while at least one all-to-all was in flight (by construction everything but each "
          f"phase's first producer chunk and last consumer chunk)")
    print(f"# all-to-all time in flight {1e3 * tot:.1f} ms (sum over {sum(1 for r in LOG if r[0] == 'a2a')} exchanges); "
          f"of it {1e3 * ov:.1f} ms = {100 * ov / max(tot, 1e-9):.0f} % with an MLP kernel executing on the device at the same time")
dist.destroy_process_group()
