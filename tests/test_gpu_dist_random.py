"""GPU: the column-sharded layers through a 1-rank RCCL group at random shapes -- odd vertex / hyperedge counts (padding of the
owned blocks to the chunk count), widths on and off the fused paths, every aggregation, plain and overlapped (chunked,
asynchronous) exchange -- against the plain module composition.  One process group for the whole module."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu

_N = int(os.environ.get("ALLSET_HYPOTHESIS_EXAMPLES", "25"))
_DERAND = os.environ.get("ALLSET_HYPOTHESIS_RANDOM", "0") != "1"


@pytest.fixture(scope="module")
def rccl_one_rank():
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    old = os.environ.get("ALLSET_FORCE_COLLECTIVES")
    os.environ["ALLSET_FORCE_COLLECTIVES"] = "1"
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    yield
    dist.destroy_process_group()
    if old is None:
        os.environ.pop("ALLSET_FORCE_COLLECTIVES", None)
    else:
        os.environ["ALLSET_FORCE_COLLECTIVES"] = old


@settings(deadline=None, max_examples=_N, derandomize=_DERAND,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(n_v=st.integers(5, 700), n_e=st.integers(3, 500), nnz=st.integers(10, 4000), d=st.sampled_from([32, 64, 128, 256]),
       attention=st.booleans(), heads=st.sampled_from([1, 2, 4]), aggr=st.sampled_from(["add", "mean", "max"]),
       chunks=st.sampled_from([1, 2, 3, 4]), sd=st.integers(0, 10 ** 6))
def test_colsharded_layers_random_shapes(n_v, n_e, nnz, d, attention, heads, aggr, chunks, sd, device, rccl_one_rank):
    from allset_amd import HalfNLHconv, Incidence
    from allset_amd import dist as adist
    rng = np.random.default_rng(sd)
    v = torch.from_numpy(rng.integers(0, n_v, size=nnz)); e = torch.from_numpy(rng.integers(0, n_e, size=nnz))
    ei = torch.unique(torch.stack([v, e]), dim=1).to(device)
    torch.manual_seed(sd)
    a = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=heads, attention=attention).to(device).eval()
    b = HalfNLHconv(d, d, d, 2, 0.0, "ln", True, heads=heads, attention=attention).to(device).eval()
    hg = adist.ColumnShardedHypergraph(ei, n_v, n_e, 1, 0, chunks=chunks).build_incidences()
    assert hg.n_v_pad % chunks == 0 and hg.n_e_pad % chunks == 0 and hg.n_v_pad >= n_v and hg.n_e_pad >= n_e
    x = torch.randn(hg.n_v_pad, d, device=device)
    G = torch.randn(hg.n_v_pad, d, device=device)
    xs = x.clone().requires_grad_(True)
    if attention:
        out = adist.colsharded_pma_layer(a, b, xs, hg, chunks=chunks)
    else:
        out = adist.colsharded_deepsets_layer(a, b, xs, hg, aggr=aggr, chunks=chunks)
    (out * G).sum().backward()
    gs = [p.grad.clone() for p in list(a.parameters()) + list(b.parameters())]
    for p in list(a.parameters()) + list(b.parameters()):
        p.grad = None
    inc = Incidence.from_edge_index(ei, n_src=hg.n_v_pad, n_dst=hg.n_e_pad)
    xr = x.clone().requires_grad_(True)
    ag = "add" if attention else aggr
    ref = F.relu(b(F.relu(a(xr, inc, None, ag)), inc.reversed(n_dst=hg.n_v_pad), None, ag))
    (ref * G).sum().backward()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    if ag != "max":                                        # (arg-extreme ties may route a gradient differently)
        torch.testing.assert_close(xs.grad, xr.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(xr.grad.abs().max())))
        for g_, p in zip(gs, list(a.parameters()) + list(b.parameters())):
            torch.testing.assert_close(g_, p.grad, rtol=1e-3, atol=1e-3 * max(1.0, float(p.grad.abs().max())))
