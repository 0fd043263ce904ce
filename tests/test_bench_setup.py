"""bench.py's N > 1 set-up arithmetic and control flow on CPU (world size 2 over gloo, the oracle as the local
aggregation): block construction, both partitions of the same job, nnz accounting, `value`, and the JSON line.  The -m gpu
counterpart (tests/test_gpu_bench_cli.py) launches the real thing under torch.distributed.run on a 1-rank RCCL group."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_both_partitions_describe_the_same_job():
    import bench
    args = bench.parse_args(["--gpus", "4", "--n-per-gpu", "250", "--degree", "5", "--feature-dim", "64"])
    args.pipeline_chunks = 1
    world, n_loc = 4, 250
    cpu = torch.device("cpu")
    rows = [bench.build_problem(args, "rows", world, r, cpu) for r in range(world)]
    cols = [bench.build_problem(args, "columns", world, r, cpu) for r in range(world)]
    # vertex blocks tile the (padded) global range, identically under both partitions
    for r in range(world):
        assert (rows[r][0].v_lo, rows[r][0].v_hi) == (r * n_loc, (r + 1) * n_loc) == (cols[r][0].v_lo, cols[r][0].v_hi)
        assert rows[r][0].n_v == cols[r][0].n_v == world * n_loc and rows[r][2] == cols[r][2] == world * n_loc
    # the column partition's global incidence is the row partition's blocks side by side, hyperedge ids offset per block
    full = cols[0][0].edge_index
    for r in range(1, world):
        assert torch.equal(cols[r][0].edge_index, full)                      # identical on every rank
    want = torch.cat([torch.stack([h.local_edge_index[0], h.local_edge_index[1] + r * n_loc]) for r, (h, _, _) in enumerate(rows)], 1)
    assert torch.equal(full, want)
    assert int(full[0].max()) < world * n_loc and int(full[1].max()) < world * n_loc
    for r, (h, nnz, _) in enumerate(rows):
        assert nnz == h.local_edge_index.shape[1] == n_loc * 5 and int(h.local_edge_index[1].max()) < n_loc
        assert torch.equal(h.local_edge_index[0], h.local_edge_index[0].sort().values)    # sorted by vertex, as preprocessing emits
    # nnz accounting: per-rank shares sum to the job's nnz under both partitions
    assert sum(n for _, n, _ in rows) == sum(n for _, n, _ in cols) == full.shape[1] == world * n_loc * 5
    assert bench.job_value(full.shape[1], 64, 0.5, 10) == full.shape[1] * 64 / 0.05


def test_resolve_modes(monkeypatch):
    import bench
    monkeypatch.delenv("ALLSET_FORCE_COLLECTIVES", raising=False)
    a = bench.parse_args([])                                          # default --shard rows since round 5 (the north star's partition)
    assert bench.resolve_modes(a, 1) == ("rows", None)
    assert bench.resolve_modes(a, 8) == ("rows", "columns")
    assert bench.resolve_modes(bench.parse_args(["--shard", "auto"]), 8) == ("columns", "rows")
    assert bench.resolve_modes(bench.parse_args(["--shard", "hybrid"]), 8) == ("hybrid", "rows")
    assert bench.hybrid_mode(a, 8) == "hybrid2x4" and bench.hybrid_mode(a, 4) == "hybrid2x2" and bench.hybrid_mode(a, 2) is None
    assert bench.hybrid_mode(bench.parse_args(["--feature-dim", "64"]), 8) is None       # 16 columns = 64-byte rows: nothing to gain
    assert bench.hybrid_mode(bench.parse_args(["--model", "pma"]), 8) == "hybrid2x4"
    a = bench.parse_args(["--shard", "rows"])
    assert bench.resolve_modes(a, 4) == ("rows", "columns")
    a = bench.parse_args(["--shard", "rows", "--partitions", "primary"])
    assert bench.resolve_modes(a, 4) == ("rows", None)
    a = bench.parse_args(["--feature-dim", "16"])                     # 16/8 columns per rank: below a 64-byte sector
    assert bench.resolve_modes(a, 8) == ("rows", None)
    monkeypatch.setenv("ALLSET_FORCE_COLLECTIVES", "1")
    assert bench.resolve_modes(bench.parse_args(["--shard", "columns"]), 1) == ("columns", "rows")


def _oracle_aggregate(x, inc, norm, aggr):
    from oracle import allset_oracle as oracle
    ei, n_dst = inc
    out = oracle.deepsets_aggregate(x, ei, norm, aggr)
    if out.shape[0] < n_dst:
        out = torch.cat([out, out.new_zeros(n_dst - out.shape[0], out.shape[1])])
    return out


def _tuple_incidences(hg, mode):
    ei = hg.edge_index if mode == "columns" else hg.local_edge_index
    n_e = hg.n_e_pad if mode == "columns" else hg.n_e_local
    hg.v2e = (ei, n_e)
    hg.e2v = (torch.stack([ei[1], ei[0]]), hg.n_v_pad)
    if getattr(hg, "halo", None) is not None:                  # the row partition's boundary-vertex exchange: compact incidence
        hloc = hg.halo_edge_index()
        hg.halo_v2e = (hloc, n_e)
        hg.halo_e2v = (torch.stack([hloc[1], hloc[0]]), hg.halo.n_needed)


def _bench_worker(rank, world, port, model, q, extra=()):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    from test_dist_cpu import TorchPmaKernels
    hooks = {"device": "cpu", "aggregate": _oracle_aggregate, "kernels": TorchPmaKernels, "incidences": _tuple_incidences}
    argv = ["--gpus", str(world), "--n-per-gpu", "120", "--degree", "4", "--feature-dim", "32", "--steps", "2", "--warmup", "1",
            "--model", model, "--heads", "2", "--dropout", "0.0", "--chunk-entry", "2"] + list(extra)
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        line = bench.main(argv, hooks)
    q.put((rank, line, buf.getvalue()))


@pytest.mark.parametrize("model", ["deepsets", "pma"])
def test_bench_two_ranks_runs_both_partitions_and_reports_one_line(model):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, model, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, line, out0), (r1, line1, out1) = results
    assert line1 is None and out1.strip() == ""                         # only rank 0 prints
    assert out0.count("\n") == 1 and json.loads(out0) == json.loads(json.dumps(line))       # ONE JSON line
    nnz = 2 * 120 * 4
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["nnz"] == nnz and line["config"]["n_v"] == 240 and line["config"]["n_e"] == 240
    parts = line["partitions"]
    assert "hyperedge-shard x2" in parts["rows"]["parallelism"] and "column-shard x2" in parts["columns"]["parallelism"]
    meta = ("note", "value_note", "fastest_exact")
    exact = [k for k in parts if k not in meta and "bf16wire" not in k]
    for name in exact:
        p = parts[name]
        assert p["ms_per_step"] > 0 and abs(p["value"] - nnz * 32 / (p["ms_per_step"] * 1e-3)) <= 1e-6 * p["value"]
    # default --shard rows (round 5): `value` is the north star's partition; the fastest exact execution is a label
    best = min(exact, key=lambda k: parts[k]["ms_per_step"])
    assert [k for k in parts if k not in meta and parts[k]["is_value"]] == ["rows"] and parts["fastest_exact"] == best
    assert line["config"]["partition"] == "rows" and line["value"] == parts["rows"]["value"] and line["ms_per_step"] == parts["rows"]["ms_per_step"]
    assert line["cpu_baseline"] is None and line["vs_baseline"] is None and line["higher_is_better"] is True
    # regions run rows first (the plainest collectives), then the all-to-all partition, then the bf16 wire; the link preflight ran
    assert [k for k in parts if k not in meta] == ["rows", "columns", "columns+chunks2", "rows+bf16wire"]
    assert "2 overlapped chunks" in parts["columns+chunks2"]["parallelism"] and not parts["rows+bf16wire"]["is_value"]
    pf = line["preflight"]["collectives"]
    assert any("all_gather" in k for k in pf) and any("reduce_scatter" in k for k in pf) and any("all_to_all" in k for k in pf)
    assert all(v["ms"] > 0 and v["gbps_per_link"] > 0 for v in pf.values())


def _hang_worker(rank, world, port, label, q):
    """bench.main with rank 1 parked at the start of region ``label``: rank 0 blocks in that region's first collective; the watchdog
    must print the line assembled from the regions that DID finish and end the process (here: hand it to the test, then exit)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      ALLSET_BENCH_TEST_HANG=f"{label}:1")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import contextlib
    import io
    import bench
    from test_dist_cpu import TorchPmaKernels
    buf = io.StringIO()

    def exit_fn(code):
        import time
        q.put((rank, code, buf.getvalue()))
        time.sleep(0.5)                                 # let the queue's feeder thread flush before the hard exit
        os._exit(0)
    hooks = {"device": "cpu", "aggregate": _oracle_aggregate, "kernels": TorchPmaKernels, "incidences": _tuple_incidences, "exit": exit_fn}
    argv = ["--gpus", str(world), "--n-per-gpu", "120", "--degree", "4", "--feature-dim", "32", "--steps", "2", "--warmup", "1",
            "--dropout", "0.0", "--region-timeout", "8", "--shard", "auto"]      # (auto: `value` would have been the fastest partition)
    real_stdout = sys.stdout
    sys.stdout = buf                                   # (the watchdog thread prints through sys.stdout too)
    try:
        bench.main(argv, hooks)
    finally:
        sys.stdout = real_stdout
    q.put((rank, "returned", buf.getvalue()))


@pytest.mark.parametrize("label", ["columns", "preflight"])
def test_bench_hung_later_region_still_yields_the_first_regions_line(label):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hang_worker, args=(r, world, port, label, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        r, code, out = q.get(timeout=300)
        got[r] = (code, out)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    code0, out0 = got[0]
    assert code0 == 0 and got[1][0] == 0 and got[1][1].strip() == ""           # both ended by their watchdogs; rank 1 printed nothing
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                                      # ONE JSON line
    line = json.loads(lines[0])
    parts = line["partitions"]
    assert parts["rows"]["is_value"] and line["value"] == parts["rows"]["value"] and line["config"]["partition"] == "rows"
    assert "timeout" in (parts["columns"]["error"] if label == "columns" else line["preflight"]["error"])
    assert "value_note" in parts and "columns" in parts["value_note"] and "did not finish" in parts["value_note"]


def test_bench_two_ranks_locality_variant_uses_the_boundary_vertex_exchange():
    """--locality 0.9: the variant workload whose hyperedge blocks mostly stay inside their rank's vertex block; the row partition
    then exchanges only the touched rows (dist.Halo) and says so; the line is labelled as a variant."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, "deepsets", q, ("--locality", "0.9"))) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    line = results[0][1]
    assert "locality 0.9: VARIANT workload" in line["config"]["workload"]
    parts = line["partitions"]
    assert "boundary-vertex exchange" in parts["rows"]["parallelism"] and "error" not in parts["rows"] and "error" not in parts["columns"]
