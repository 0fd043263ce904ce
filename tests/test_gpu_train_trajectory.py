"""GPU: a short TRAINING TRAJECTORY -- what reference train.py:458-499 does per run: forward, nll_loss(log_softmax) on the train
split, backward, Adam -- on the product (HIP kernels, torch.optim.Adam over its parameters) against the oracle (float64,
torch.optim.Adam over the same state dict): the loss after every step and the parameters after the last one.  Single-step parity
(tests/test_gpu_parity.py, test_gpu_train_parity.py) says each gradient is right; this says the loop around them is the
reference's loop -- parameter layout, optimizer wiring, in-place edge-index re-basing on the first forward, cached incidence
across steps.  Dropouts are off (eval-mode forward, gradients on): the reference's masks come from torch's Philox stream and
cannot be reproduced; the training-mode arithmetic itself is pinned with explicit masks in test_gpu_train_parity.py."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases

pytestmark = pytest.mark.gpu
STEPS = 12


def trajectory_oracle(case, sd, y, train_idx, steps, lr, dtype=torch.float64):
    from oracle import allset_oracle as oracle
    params = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    leaves = [t.requires_grad_(True) for k, t in params.items() if t.is_floating_point() and "running" not in k]
    opt = torch.optim.Adam(leaves, lr=lr, weight_decay=0.0)
    x = torch.from_numpy(case["x"]).to(dtype)
    ei, nrm = torch.from_numpy(case["edge_index"]), torch.from_numpy(case["norm"])
    nrm = nrm.to(dtype) if nrm.is_floating_point() else nrm
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        out = oracle.setgnn_forward(params, case["args"], x, ei, nrm)
        loss = F.nll_loss(F.log_softmax(out, dim=1)[train_idx], y[train_idx])
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses, params


@pytest.mark.parametrize("name,over", [("rand50_ds_add", {}), ("rand50_pma_h4", {}), ("rand50_ds_mean_wnorm", {}), ("cora_ds_add", {}),
                                       ("citeseer_pma_h4", {}), ("rand50_ds_add", dict(All_num_layers=2, GPR=True)),
                                       ("rand50_ds_add_wnorm_mask", {})],
                         ids=lambda v: v if isinstance(v, str) else ("-".join(f"{k}{w}" for k, w in v.items()) or "stock"))
def test_training_trajectory_matches_the_oracle(name, over, device):
    from allset_amd import SetGNN
    case = cases.build_case(name)
    case["args"] = SimpleNamespace(**{**vars(case["args"]), **over})
    args = case["args"]
    torch.manual_seed(case["seed"])
    norm_t = torch.from_numpy(case["norm"])
    model = SetGNN(args, norm=norm_t.to(torch.float32) if args.LearnMask else None)
    model.reset_parameters()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    n = case["x"].shape[0]
    rng = np.random.default_rng(case["seed"])
    y = torch.from_numpy(rng.integers(0, args.num_classes, size=n))
    train_idx = torch.from_numpy(np.sort(rng.choice(n, size=max(n // 2, 4), replace=False)))
    lr = 0.01                                             # (train.py's default lr 0.001; larger here so that twelve steps move the loss)

    model.eval().to(device)                               # dropouts off, gradients on
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0.0)
    data = SimpleNamespace(x=torch.from_numpy(case["x"]).to(device), edge_index=torch.from_numpy(case["edge_index"]).clone().to(device),
                           norm=norm_t.to(device))
    yd, td = y.to(device), train_idx.to(device)
    losses = []
    for _ in range(STEPS):
        opt.zero_grad()
        out = model(data)
        loss = F.nll_loss(F.log_softmax(out, dim=1)[td], yd[td])
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))

    ref_losses, ref_params = trajectory_oracle(case, sd, y, train_idx, STEPS, lr)
    assert ref_losses[-1] < ref_losses[0]                 # the twelve steps do train
    # Adam divides every gradient by its own running magnitude, so a weight whose gradient is rounding noise still steps by ~lr in
    # the direction of that noise.  On the dataset-shaped cases (1433- / 3703-wide input Linear: 10^5 weights, thousands of them
    # with noise-level gradients) ANY two fp32 evaluations leave each other by per cents within a dozen steps -- measured at step 1
    # of cora_ds_add: the product's gradients are within 3e-6 of float64 on every tensor, the fp32 ORACLE's up to 1.4e-2 (it sits on
    # a relu kink there), and still both trajectories are "right".  So: plain parity while nothing has been amplified yet (the
    # first three losses), every step on the small cases, and for the dataset-shaped ones afterwards only that the run trains like
    # the oracle's (5 % on each later loss).
    np.testing.assert_allclose(losses[:3], ref_losses[:3], rtol=2e-4, atol=2e-5)
    # How far may a later loss sit from the float64 one?  As far as rounding alone moves THIS trajectory: the oracle evaluated in
    # float32 (same math, torch CPU kernels) leaves its own float64 run by the amount that Adam's amplification of noise-level
    # gradients and the relu kinks allow at that step -- three times that distance, and never less than the plain tolerance.  (A
    # step where the float32 oracle stays on the float64 curve and the product does not is a product bug; one where both leave it --
    # rand50_ds_add at lr = 0.01 oscillates from step nine on, 1.00 -> 1.18 -> 0.99 -- is the trajectory's conditioning.)
    ref32, _ = trajectory_oracle(case, sd, y, train_idx, STEPS, lr, dtype=torch.float32)
    own = np.abs(np.array(ref32) - np.array(ref_losses)) / np.abs(np.array(ref_losses))
    own = np.maximum.accumulate(own)          # (a trajectory that has left the curve does not return to it: later steps are no better conditioned)
    if case["big"]:
        tol = np.maximum(5e-2, 3.0 * own)
        assert np.all(np.abs(np.array(losses) - np.array(ref_losses)) <= tol * np.abs(np.array(ref_losses))), (losses, ref_losses, ref32)
        assert losses[-1] < losses[0]
        return
    # small cases: six steps at plain parity, the rest within 2 % (one of them -- two layers + GPR at lr = 0.01 -- drives the loss
    # from 1.9 to 0.03 in twelve steps and leaves the float64 trajectory by 0.6 % at step nine; the others hold 2e-4 while the
    # float32 oracle does)
    np.testing.assert_allclose(losses[:6], ref_losses[:6], rtol=2e-4, atol=2e-5)
    tol = np.maximum(2e-2, 3.0 * own)
    assert np.all(np.abs(np.array(losses) - np.array(ref_losses)) <= tol * np.abs(np.array(ref_losses))), (losses, ref_losses, ref32)
    if not np.allclose(losses, ref_losses, rtol=2e-4, atol=2e-5):
        return
    # parameters after the last step, on the scale of what the trajectory moved (STEPS * lr): at most 1 % of a tensor's entries off
    # by more than 5 % of the movement
    moved = STEPS * lr
    for k, p in model.named_parameters():
        diff = (p.detach().cpu().double() - ref_params[k].detach()).abs()
        assert float((diff > 0.05 * moved).double().mean()) <= 0.01, (k, float(diff.max()))
        assert float(diff.max()) <= 2.5 * moved, (k, float(diff.max()))
