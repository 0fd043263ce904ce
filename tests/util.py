"""Helpers shared by the parity tests."""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Dict

import numpy as np
import torch

import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name: str) -> Dict[str, np.ndarray]:
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_spec(g: Dict[str, np.ndarray]):
    return [(str(k), tuple(json.loads(str(s)))) for k, s in zip(g["spec_keys"], g["spec_shapes"])]


def state_dict_for(case: dict, g: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    sd_np = cases.make_state_dict(golden_spec(g), case["seed"], case.get("kinkfree", False))
    return {k: torch.from_numpy(v) for k, v in sd_np.items()}


class ShapeProbe:
    """A ``drop`` callable for the oracle that only records (shape, p) of every dropout site it meets (values pass through)."""

    def __init__(self):
        self.sites = []

    def __call__(self, x, p):
        if p > 0.0:
            self.sites.append((tuple(x.shape), float(p)))
        return x


def oracle_is_smooth_here(sd: Dict[str, torch.Tensor], args, x_np, ei_np, norm_np, G, masks=None, seed: int = 0,
                          eps: float = 2e-6, tol: float = 3e-4, params: bool = True, eps_bow: float = 5e-7) -> bool:
    """The A-PRIORI acceptance criterion of a parameter draw for an fp32 parity comparison: is the float64 ORACLE's gradient stable
    under a perturbation of ``x`` an order above fp32 rounding?  A relu input within fp32 rounding of zero makes the exact gradient
    itself jump (either side is a correct subgradient), and two correct fp32 evaluations then disagree by 1e-3 .. 2e-2 of a gradient
    tensor's scale; such a draw says nothing about parity.  Evaluates ``d (logits * G).sum()`` w.r.t. ``x`` (and, ``params``, every
    floating-point parameter) at ``x`` and at ``x +- eps * direction`` (same dropout ``masks``, if any) and returns True when no
    gradient moves by more than ``tol`` of its scale.  Bag-of-words features (more than half of ``x`` exactly zero: the Cora- /
    Citeseer-shaped cases) are perturbed RELATIVELY, ``x * (1 +- eps_bow * direction)``: their zeros are exact in every fp32
    evaluation, and an absolute 2e-6 on 1433 zero columns behind a LayerNorm is two orders above fp32 rounding, not one.  With ~1e6
    relu inputs whose pre-activations are sums of ~18 terms of size 20 * |w|, no draw of 64 is stable at a relative 2e-6, 1 in 8 at
    1e-6, 3 in 8 at 5e-7 (= 8 half-ulps of fp32): ``eps_bow`` defaults to 5e-7.  Nothing of the PRODUCT enters: a draw is accepted or refused before the one
    assertion that compares the product with the oracle -- the tests never select a draw on the outcome of that comparison."""
    from oracle import allset_oracle as oracle
    sd64 = {k: (v.detach().double() if v.is_floating_point() else v.detach()) for k, v in sd.items()}
    dirn = torch.from_numpy(np.random.default_rng(seed).standard_normal(x_np.shape))
    if float((x_np == 0).mean()) > 0.5:
        dirn, eps = dirn * torch.from_numpy(x_np).double().abs(), eps_bow
    G64 = G.detach().cpu().double()
    ei, norm = torch.from_numpy(ei_np), torch.from_numpy(norm_np)
    runs = []
    for sgn in (0.0, 1.0, -1.0):
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd64.items()
                  if params and v.is_floating_point() and "running" not in k}
        sdp = {**sd64, **leaves}
        xp = (torch.from_numpy(x_np).double() + sgn * eps * dirn).requires_grad_(True)
        kw = dict(drop=oracle.ExplicitDropout(masks)) if masks is not None else {}
        lp = oracle.setgnn_forward(sdp, args, xp, ei.clone(), norm.double() if norm.is_floating_point() else norm, **kw)
        (lp * G64).sum().backward()
        runs.append({"x": xp.grad, **{k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}})
    base = runs[0]
    gscale = max([float(v.abs().max()) for k, v in base.items() if k != "x"] + [0.0])
    for key, g0 in base.items():
        scale = max(1.0, float(g0.abs().max())) if key == "x" else max(float(g0.abs().max()), 1e-2 * gscale, 1e-30)
        for other in runs[1:]:
            if float((other[key] - g0).abs().max()) > tol * scale:
                return False
    return True


def oracle_relu_margin(sd: Dict[str, torch.Tensor], args, x_np, ei_np, norm_np) -> float:
    """The DIRECT form of the a-priori criterion, for evaluations whose activations hardly depend on ``x`` (bias-dominated stacks: a
    relative 1e-5 on bag-of-words rows arrived as 6e-8 of a row's scale at the third MLP's output -- the perturbation probe of
    ``oracle_is_smooth_here`` then cannot see a pre-activation that sits 9e-8 from its kink, and a fresh-seed run found exactly
    that one): the smallest ``|pre-activation| / (largest entry of its row)`` over every non-zero relu input of the float64 ORACLE
    (exact zeros are structural: relu of a relu).  A caller accepts a draw when this exceeds the arithmetic's rounding by an order
    (fp32 sums of 512 products: 1e-6; two fp16 planes: 5e-6)."""
    import torch.nn.functional as F
    from oracle import allset_oracle as oracle
    sd64 = {k: (v.detach().double() if v.is_floating_point() else v.detach()) for k, v in sd.items()}
    norm = torch.from_numpy(norm_np)
    real, seen = F.relu, []

    def recording(t, *a, **k):
        seen.append(t.detach())
        return real(t, *a, **k)
    F.relu = recording
    try:
        with torch.no_grad():
            oracle.setgnn_forward(sd64, args, torch.from_numpy(x_np).double(), torch.from_numpy(ei_np).clone(),
                                  norm.double() if norm.is_floating_point() else norm)
    finally:
        F.relu = real
    best = float("inf")
    for t in seen:
        t2 = t.reshape(-1, t.shape[-1]).abs()
        rel = t2 / t2.amax(dim=1, keepdim=True).clamp_min(1e-300)
        rel = rel[t2 != 0]
        if rel.numel():
            best = min(best, float(rel.min()))
    return best


def run_oracle(case: dict, sd: Dict[str, torch.Tensor], dtype: torch.dtype = torch.float32):
    """Oracle forward + backward of loss = (logits * G).sum(); returns dict like the fixtures.  ``dtype=torch.float64``: the
    same oracle in double precision (the yardstick where fp32 rounding of the ORACLE itself exceeds the tolerance)."""
    from oracle import allset_oracle as oracle
    sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    x = torch.from_numpy(case["x"]).clone().to(dtype).requires_grad_(True)
    collect = {}
    norm = torch.from_numpy(case["norm"])
    logits = oracle.setgnn_forward(sd, case["args"], x, torch.from_numpy(case["edge_index"]),
                                   norm.to(dtype) if norm.is_floating_point() else norm, collect)
    G = torch.from_numpy(cases.cotangent(case["name"], logits.shape)).to(dtype)
    (logits * G).sum().backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items() if v.requires_grad}
    return dict(logits=logits.detach(), v2e0=collect["v2e0"].detach(), e2v0=collect["e2v0"].detach(),
                grad_x=x.grad.detach(), grads=grads)


def run_product(case: dict, sd: Dict[str, torch.Tensor], device):
    """allset_amd.SetGNN forward + backward on `device` with the same loss."""
    from allset_amd import SetGNN
    args = case["args"]
    norm_t = torch.from_numpy(case["norm"])
    model = SetGNN(args, norm=norm_t.to(torch.float32) if args.LearnMask else None)
    model.load_state_dict(sd)
    model.eval().to(device)
    grabbed = {}
    # the raw conv output as the reference's forward hook sees it: SetGNN fuses its outer relu(+dropout) into
    # the conv's last pass, so for the attention variant hook the PMA module (pre-relu); for Deep Sets the conv
    # output is relu'd in the reference too
    def raw(conv):
        return conv.prop if conv.attention else conv
    raw(model.V2EConvs[0]).register_forward_hook(lambda m, i, o: grabbed.__setitem__("v2e0", o))
    raw(model.E2VConvs[0]).register_forward_hook(lambda m, i, o: grabbed.__setitem__("e2v0", o))
    x = torch.from_numpy(case["x"]).to(device).requires_grad_(True)
    ei = torch.from_numpy(case["edge_index"]).to(device)
    data = SimpleNamespace(x=x, edge_index=ei, norm=norm_t.to(device))
    logits = model(data)
    G = torch.from_numpy(cases.cotangent(case["name"], logits.shape)).to(device)
    (logits * G).sum().backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu()
             for k, p in model.named_parameters()}
    return dict(logits=logits.detach().cpu(), v2e0=grabbed["v2e0"].detach().cpu(), e2v0=grabbed["e2v0"].detach().cpu(),
                grad_x=x.grad.detach().cpu(), grads=grads, model=model, data=data)


def assert_matches_golden(res: dict, g: Dict[str, np.ndarray], big: bool, rtol: float, atol: float) -> None:
    """Compare a result dict (from run_oracle / run_product) with a fixture.  atol is relative to the
    max-abs of the expected tensor (so that it is a statement about significant digits)."""
    # Parameter gradients that are analytically zero (e.g. d/d lin_K.bias with one head: a constant added
    # to every logit of a head cancels in the softmax) are pure rounding noise on both sides; their
    # absolute tolerance is therefore tied to the overall gradient scale of the case, not to their own.
    gscale = max([float(np.abs(g[k]).max()) for k in g if k.startswith(("grad_", "gval_")) and g[k].size] + [0.0])

    def close(got: torch.Tensor, exp: np.ndarray, what: str):
        exp_t = torch.from_numpy(np.asarray(exp))
        scale = float(exp_t.abs().max()) if exp_t.numel() else 0.0
        floor = 1e-2 * gscale if what.startswith(("grad_", "gval_")) else 1e-3
        torch.testing.assert_close(got.to(exp_t.dtype), exp_t, rtol=rtol, atol=atol * max(scale, floor, 1e-3),
                                   msg=lambda m: f"{what}: {m}")

    for k in ("logits", "v2e0", "e2v0"):
        assert res[k].shape[0] == int(g["n_rows_" + k]), (k, res[k].shape)
    if not big:
        for k in ("logits", "v2e0", "e2v0", "grad_x"):
            close(res[k], g["out_" + k], k)
        for key in g:
            if key.startswith("grad_") and key != "grad_x":
                close(res["grads"][key[5:]], g[key], key)
    else:
        for k in ("logits", "v2e0", "e2v0", "grad_x"):
            rows = torch.from_numpy(g["rows_" + k])
            got = res[k][rows]
            exp = g["out_" + k]
            close(got[:, :exp.shape[1]], exp, k + "[rows]")
            s, a = float(res[k].double().sum()), float(res[k].double().abs().sum())
            tol = (rtol * 10) * float(g["abs_" + k]) + 1e-6
            assert abs(s - float(g["sum_" + k])) <= tol, (k, s, float(g["sum_" + k]))
            assert abs(a - float(g["abs_" + k])) <= tol, (k, a, float(g["abs_" + k]))
        for key in g:
            if key.startswith("grad_") and key != "grad_x":
                close(res["grads"][key[5:]], g[key], key)
            elif key.startswith("gidx_"):
                name = key[5:]
                got = res["grads"][name].reshape(-1)
                close(got[torch.from_numpy(g[key])], g["gval_" + name], "gval_" + name)
                tol = (rtol * 10) * float(g["gabs_" + name]) + 1e-6
                assert abs(float(got.double().sum()) - float(g["gsum_" + name])) <= tol, name
                assert abs(float(got.double().abs().sum()) - float(g["gabs_" + name])) <= tol, name
