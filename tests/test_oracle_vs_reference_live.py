"""CPU, build container only (skipped where /root/reference is absent): the oracle against the LIVE reference
(imported through oracle/ref_shim.py) on randomised SetGNN configurations -- the same generator the GPU-side
random-configuration parity test uses, so that chain reads  reference == oracle (here)  and  oracle == product
(tests/test_gpu_random_shapes.py).  Logits, d(loss)/dx and every parameter gradient."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

import cases
from oracle import allset_oracle as oracle
from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")


@settings(deadline=None, max_examples=int(os.environ.get("ALLSET_HYPOTHESIS_EXAMPLES", "25")),
          derandomize=os.environ.get("ALLSET_HYPOTHESIS_RANDOM", "0") != "1")
@given(pma=st.booleans(), layers=st.integers(1, 3), mlp_layers=st.integers(1, 3), hidden=st.sampled_from([16, 64, 256]),
       heads=st.sampled_from([1, 2, 4]), aggr=st.sampled_from(["add", "mean", "max"]), norm=st.sampled_from(["ln", "bn", "None"]),
       input_norm=st.booleans(), mask=st.booleans(), gpr=st.booleans(), wnorm=st.booleans(), sd=st.integers(0, 10 ** 6))
def test_oracle_equals_live_reference_on_random_configurations(pma, layers, mlp_layers, hidden, heads, aggr, norm, input_norm,
                                                               mask, gpr, wnorm, sd):
    _, ref_models = ref_shim.import_reference()
    rng = np.random.default_rng(sd)
    n_v, n_e, f, k = 40, 17, 12, 5
    ei = cases.random_hypergraph(rng, n_v, n_e, 150, True)
    x = rng.standard_normal((n_v, f)).astype(np.float32)
    args = cases.make_args("pma_h1" if pma else "ds_add", f, hidden, k, All_num_layers=layers, MLP_num_layers=mlp_layers,
                           heads=heads if pma else 1, aggregate=aggr if not pma else "add", normalization=norm,
                           deepset_input_norm=input_norm, LearnMask=mask, GPR=gpr, Classifier_num_layers=2)
    nrm = cases._norm_deg_half_sym(ei) if (wnorm or mask) else np.ones(ei.shape[1], dtype=np.int64)
    norm_t = torch.from_numpy(nrm)
    torch.manual_seed(sd)
    model = ref_models.SetGNN(args, norm=norm_t.to(torch.float32) if mask else None)
    model.reset_parameters()
    model.eval()
    xr = torch.from_numpy(x).clone().requires_grad_(True)
    ref = model(SimpleNamespace(x=xr, edge_index=torch.from_numpy(ei).clone(), norm=norm_t))
    G = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
    (ref * G).sum().backward()
    sdict = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    for kk, v in sdict.items():
        if v.is_floating_point() and "running" not in kk:
            v.requires_grad_(True)
    xo = torch.from_numpy(x).clone().requires_grad_(True)
    out = oracle.setgnn_forward(sdict, args, xo, torch.from_numpy(ei), norm_t)
    (out * G).sum().backward()
    torch.testing.assert_close(out.detach(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xo.grad, xr.grad, rtol=1e-4, atol=1e-5 * max(1.0, float(xr.grad.abs().max())))
    for kk, p in model.named_parameters():
        if p.grad is None or sdict[kk].grad is None:
            assert (p.grad is None or float(p.grad.abs().max()) == 0.0) and (sdict[kk].grad is None or float(sdict[kk].grad.abs().max()) == 0.0), kk
            continue
        torch.testing.assert_close(sdict[kk].grad, p.grad, rtol=1e-4, atol=1e-5 * max(1.0, float(p.grad.abs().max())),
                                   msg=lambda m: f"{kk}: {m}")


class _RecordDropout:
    """Wraps torch's own ``F.dropout`` while the LIVE reference runs in training mode: calls the real function and reads
    the keep mask off its output (``out != 0``; where the input itself is 0 -- behind a relu -- the mask is immaterial
    for the value and the gradient is cut by the relu, so those positions count as kept)."""

    def __init__(self):
        import torch.nn.functional as F
        self.F, self.orig, self.masks = F, F.dropout, []

    def __enter__(self):
        def dropout(x, p=0.5, training=True, inplace=False):
            out = self.orig(x, p=p, training=training, inplace=False)
            if training and p > 0.0:
                self.masks.append(((out != 0) | (x == 0)).detach().clone())
            return out
        self.F.dropout = dropout
        return self

    def __exit__(self, *exc):
        self.F.dropout = self.orig
        return False


@settings(deadline=None, max_examples=int(os.environ.get("ALLSET_HYPOTHESIS_EXAMPLES", "25")),
          derandomize=os.environ.get("ALLSET_HYPOTHESIS_RANDOM", "0") != "1")
@given(pma=st.booleans(), layers=st.integers(1, 2), mlp_layers=st.integers(1, 3), hidden=st.sampled_from([16, 64]),
       heads=st.sampled_from([1, 4]), aggr=st.sampled_from(["add", "mean", "max"]), norm=st.sampled_from(["ln", "None"]),
       gpr=st.booleans(), p=st.sampled_from([0.5, 0.2]), sd=st.integers(0, 10 ** 6))
def test_oracle_training_mode_with_explicit_masks_equals_live_reference(pma, layers, mlp_layers, hidden, heads, aggr, norm, gpr, p, sd):
    """The explicit-mask training mode of the oracle (``setgnn_forward(..., drop=ExplicitDropout(masks))``) against the live
    reference in ``.train()`` mode: same dropout sites in the same order (every recorded mask is consumed, shapes agree),
    same logits and gradients.  Both sides run in float64 (dropout in front of stacked LayerNorms over 16 columns amplifies
    fp32 rounding differences between the two implementations to ~3e-5; in float64 the comparison is about semantics).  This pins the oracle that tests/test_gpu_train_parity.py checks the product's training step
    against."""
    _, ref_models = ref_shim.import_reference()
    rng = np.random.default_rng(sd)
    n_v, n_e, f, k = 40, 17, 12, 5
    ei = cases.random_hypergraph(rng, n_v, n_e, 150, True)
    x = rng.standard_normal((n_v, f)).astype(np.float32)
    args = cases.make_args("pma_h1" if pma else "ds_add", f, hidden, k, All_num_layers=layers, MLP_num_layers=mlp_layers,
                           heads=heads if pma else 1, aggregate=aggr if not pma else "add", normalization=norm, GPR=gpr,
                           Classifier_num_layers=2, dropout=p)
    norm_t = torch.ones(ei.shape[1], dtype=torch.int64)
    torch.manual_seed(sd)
    model = ref_models.SetGNN(args)
    model.reset_parameters()
    model.double().train()
    xr = torch.from_numpy(x).double().requires_grad_(True)
    with _RecordDropout() as rec:
        ref = model(SimpleNamespace(x=xr, edge_index=torch.from_numpy(ei).clone(), norm=norm_t))
    G = torch.from_numpy(rng.standard_normal(tuple(ref.shape)))
    (ref * G).sum().backward()
    sdict = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    for kk, v in sdict.items():
        if v.is_floating_point() and "running" not in kk:
            v.requires_grad_(True)
    xo = torch.from_numpy(x).double().requires_grad_(True)
    drop = oracle.ExplicitDropout(rec.masks)
    out = oracle.setgnn_forward(sdict, args, xo, torch.from_numpy(ei), norm_t, drop=drop)
    assert drop.used == len(rec.masks) and len(rec.masks) >= 2
    (out * G).sum().backward()
    torch.testing.assert_close(out.detach(), ref.detach(), rtol=1e-9, atol=1e-9)
    torch.testing.assert_close(xo.grad, xr.grad, rtol=1e-8, atol=1e-9 * max(1.0, float(xr.grad.abs().max())))
    for kk, pp in model.named_parameters():
        if pp.grad is None or sdict[kk].grad is None:
            assert (pp.grad is None or float(pp.grad.abs().max()) == 0.0) and (sdict[kk].grad is None or float(sdict[kk].grad.abs().max()) == 0.0), kk
            continue
        torch.testing.assert_close(sdict[kk].grad, pp.grad, rtol=1e-8, atol=1e-9 * max(1.0, float(pp.grad.abs().max())),
                                   msg=lambda m: f"{kk}: {m}")


@pytest.mark.parametrize("name", ["rand50_ds_add", "rand50_pma_h4", "rand50_ds_mean_wnorm", "rand50_ds_add_L2_gpr"])
def test_oracle_training_trajectory_equals_live_reference(name):
    """Ten Adam steps of the reference's training loop body (train.py:470-476: forward, nll_loss(log_softmax) on the train split,
    backward, step) on the LIVE reference model and on the oracle, both float64, dropouts off (eval-mode forward): the loss
    sequence and the final parameters agree to rounding.  The GPU side of the chain is tests/test_gpu_train_trajectory.py."""
    import torch.nn.functional as F
    _, ref_models = ref_shim.import_reference()
    case = cases.build_case(name)
    args = case["args"]
    torch.manual_seed(case["seed"])
    norm_t = torch.from_numpy(case["norm"])
    model = ref_models.SetGNN(args, norm=norm_t.to(torch.float32) if args.LearnMask else None)
    model.reset_parameters()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.double().eval()
    n = case["x"].shape[0]
    rng = np.random.default_rng(case["seed"])
    y = torch.from_numpy(rng.integers(0, args.num_classes, size=n))
    train_idx = torch.from_numpy(np.sort(rng.choice(n, size=max(n // 2, 4), replace=False)))
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=0.0)
    data = SimpleNamespace(x=torch.from_numpy(case["x"]).double(), edge_index=torch.from_numpy(case["edge_index"]).clone(),
                           norm=norm_t.double() if norm_t.is_floating_point() else norm_t)
    losses = []
    for _ in range(10):
        opt.zero_grad()
        out = model(data)
        loss = F.nll_loss(F.log_softmax(out, dim=1)[train_idx], y[train_idx])
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_train_trajectory import trajectory_oracle
    ref_losses, ref_params = trajectory_oracle(case, sd, y, train_idx, 10, 0.01)
    np.testing.assert_allclose(ref_losses, losses, rtol=1e-9, atol=1e-11)
    for k, p in model.named_parameters():
        torch.testing.assert_close(ref_params[k].detach(), p.detach(), rtol=1e-7, atol=1e-9, msg=lambda m, k=k: f"{k}: {m}")
