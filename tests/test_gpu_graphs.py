"""GPU: hipGraph capture (allset_amd/graphs.py).  An eval forward replayed from a graph is bitwise the eager one;
a captured training step draws fresh dropout masks per replay (device seed counter), matches an eager step
bit-for-bit when dropout is off, and actually trains."""
import copy
from types import SimpleNamespace

import pytest
import torch
import torch.nn.functional as F

import cases

pytestmark = pytest.mark.gpu


def _setup(name, device, dropout=None):
    from allset_amd import SetGNN
    case = cases.build_case(name)
    args = case["args"]
    if dropout is not None:
        args.dropout = dropout
    torch.manual_seed(0)
    model = SetGNN(args).to(device)
    model.reset_parameters()
    data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(device),
                           edge_index=torch.from_numpy(case["edge_index"]).to(device),
                           norm=torch.from_numpy(case["norm"]).to(device))
    y = torch.randint(0, args.num_classes, (data.x.shape[0],), generator=torch.Generator().manual_seed(1)).to(device)
    return model, data, y


@pytest.mark.parametrize("name", ["cora_ds_add", "citeseer_pma_h4", "rand50_pma_h4", "wide256_pma_h4"])
def test_graphed_forward_is_bitwise_eager(name, device):
    from allset_amd.graphs import GraphedForward
    model, data, _ = _setup(name, device)
    model.eval()
    with torch.no_grad():
        ref = model(data).clone()
    gf = GraphedForward(model, data)
    out = gf()
    assert torch.equal(out, ref)
    # parameters are read at replay time: perturb one and the graph follows the eager result
    with torch.no_grad():
        next(model.parameters()).mul_(1.5)
        ref2 = model(data).clone()
    assert not torch.equal(ref2, ref)
    assert torch.equal(gf(), ref2)
    # new features for the same hypergraph
    x2 = data.x * 0.5
    with torch.no_grad():
        ref3 = model(SimpleNamespace(x=x2.clone(), edge_index=data.edge_index, norm=data.norm)).clone()
    assert torch.equal(gf(x2), ref3)


@pytest.mark.parametrize("name", ["cora_ds_add", "citeseer_pma_h4", "wide256_ds_add"])
def test_graphed_train_step_matches_eager_without_dropout(name, device):
    """Without dropout (eval-mode step: SetGNN's input dropout p=0.2 is hard-coded, models.py:473) the step is
    deterministic: N graph replays == N eager steps from the same start, bit for bit."""
    from allset_amd.graphs import GraphedTrainStep
    model_a, data_a, y = _setup(name, device, dropout=0.0)
    model_b, data_b, _ = _setup(name, device, dropout=0.0)
    model_b.load_state_dict(model_a.state_dict())
    opt_a = torch.optim.Adam(model_a.parameters(), lr=1e-2, capturable=True)
    opt_b = torch.optim.Adam(model_b.parameters(), lr=1e-2, capturable=True)

    def eager():
        model_b.eval()
        opt_b.zero_grad(set_to_none=True)
        loss = F.nll_loss(F.log_softmax(model_b(data_b), dim=1), y)
        loss.backward()
        opt_b.step()
        return loss

    warm = 3
    step = GraphedTrainStep(model_a, data_a, lambda out: F.nll_loss(F.log_softmax(out, dim=1), y), opt_a, warmup=warm,
                            train_mode=False)
    # warm-up steps ran for real but restore=True put parameters and Adam state back: a is still at step 0
    for pa, pb in zip(model_a.parameters(), model_b.parameters()):
        assert torch.equal(pa, pb)
    assert all(float(st["step"]) == 0.0 for st in opt_a.state.values())
    model_b.load_state_dict(model_a.state_dict())
    opt_b.load_state_dict(copy.deepcopy(opt_a.state_dict()))      # load_state_dict aliases same-device tensors
    for i in range(4):
        la = step().clone()
        lb = eager()
        assert torch.equal(la.detach(), lb.detach()), (i, float(la.detach()), float(lb.detach()))
    for pa, pb in zip(model_a.parameters(), model_b.parameters()):
        assert torch.equal(pa, pb)


@pytest.mark.parametrize("name", ["cora_ds_add", "citeseer_pma_h4"])
def test_graphed_train_step_draws_fresh_masks_and_trains(name, device):
    from allset_amd.graphs import GraphedTrainStep
    model, data, y = _setup(name, device, dropout=0.5)
    opt = torch.optim.Adam(model.parameters(), lr=0.0, capturable=True)          # lr 0: only the masks differ
    step = GraphedTrainStep(model, data, lambda out: F.nll_loss(F.log_softmax(out, dim=1), y), opt)
    losses = [float(step().detach()) for _ in range(6)]
    assert len(set(losses)) == 6, losses                                         # a fresh mask every replay
    # and with a real learning rate the captured loop optimises
    model2, data2, y2 = _setup(name, device, dropout=0.2)
    opt2 = torch.optim.Adam(model2.parameters(), lr=1e-2, capturable=True, fused=True)
    step2 = GraphedTrainStep(model2, data2, lambda out: F.nll_loss(F.log_softmax(out, dim=1), y2), opt2)
    first = sum(float(step2().detach()) for _ in range(3)) / 3
    for _ in range(60):
        step2()
    last = sum(float(step2().detach()) for _ in range(3)) / 3
    assert last < 0.7 * first, (first, last)


def test_device_seed_counter_changes_mask(device):
    """ABI level: same host seed, different device counter -> different mask; same counter -> same mask; the
    backward regenerates the forward's mask from (seed, counter)."""
    from allset_amd import dense
    n, d, p = 2000, 128, 0.4
    x = torch.randn(n, d, device=device)
    gamma, beta = torch.ones(d, device=device), torch.full((d,), 0.5, device=device)
    c = torch.tensor([5], dtype=torch.int64, device=device)
    y1, st = dense.ln_fwd(x, gamma, beta, 1e-5, False, p, 3, c)
    y2, _ = dense.ln_fwd(x, gamma, beta, 1e-5, False, p, 3, c)
    c.add_(1)
    y3, _ = dense.ln_fwd(x, gamma, beta, 1e-5, False, p, 3, c)
    y4, _ = dense.ln_fwd(x, gamma, beta, 1e-5, False, p, 3)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3) and not torch.equal(y1, y4)
    assert abs(float((y3 != 0).float().mean()) - (1 - p)) < 0.01
    G = torch.randn(n, d, device=device)
    gx, _, _ = dense.ln_bwd(G, x, st, gamma, False, p, 3, c)
    xr = x.clone().requires_grad_(True)
    mask = (y3 != 0).float() / (1 - p)
    (F.layer_norm(xr, (d,), gamma, beta, 1e-5) * mask * G).sum().backward()
    torch.testing.assert_close(gx, xr.grad, rtol=1e-4, atol=2e-5)
    # fused linear: the same counter semantics
    W, b = torch.randn(d, d, device=device) * 0.1, torch.zeros(d, device=device)
    f1, _ = dense.fused_linear_fwd(x, W, b, gamma, beta, 1e-5, False, p, 3, True, p, 4, c)
    u = torch.relu(F.linear(y3, W, b))
    kept = f1 != 0
    torch.testing.assert_close(f1[kept], (u / (1 - p))[kept], rtol=1e-4, atol=1e-4)


def test_graphed_forward_with_learnmask_routes_weights_inside_the_graph(device):
    """LearnMask (reference models.py:336-337,451-452): the eval forward multiplies ``Importance * norm`` into a fresh
    temporary every call; routing it to CSR order must neither synchronise with the host (illegal in a capture) nor be
    frozen at capture time -- a replay after an in-place change of ``Importance`` follows the eager result."""
    from allset_amd import SetGNN
    from allset_amd.graphs import GraphedForward
    case = cases.build_case("rand50_ds_add_wnorm")
    args = copy.copy(case["args"])
    args.LearnMask = True
    norm = torch.from_numpy(case["norm"]).float()
    torch.manual_seed(0)
    model = SetGNN(args, norm).to(device)
    model.reset_parameters()
    with torch.no_grad():
        model.Importance.copy_(torch.linspace(0.5, 1.5, norm.numel()))
    data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(device), edge_index=torch.from_numpy(case["edge_index"]).to(device),
                           norm=norm.to(device))
    model.eval()
    with torch.no_grad():
        ref = model(data).clone()
    gf = GraphedForward(model, data)            # raised before round 3: the all-ones probe of the temporary synchronised
    assert torch.equal(gf(), ref)
    with torch.no_grad():
        model.Importance.mul_(0.5).add_(0.1)
        ref2 = model(data).clone()
    assert not torch.equal(ref2, ref)
    assert torch.equal(gf(), ref2)
