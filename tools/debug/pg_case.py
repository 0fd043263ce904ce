#!/usr/bin/env python
"""Reproduce one example of tests/test_gpu_random_shapes.py::test_setgnn_random_parameter_gradients_match_oracle and print every
parameter gradient's distance from the float64 oracle, in the default arithmetic, in the strict one, and with the features routed
through the dense path (x.requires_grad).  Run on the GPU box: python tools/debug/pg_case.py pma layers mlp_layers hidden heads norm input_norm bow sd"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from types import SimpleNamespace
import numpy as np, torch
import cases, util
from oracle import allset_oracle as oracle
from allset_amd import SetGNN, dense

a = sys.argv[1:]
pma, layers, mlp_layers, hidden, heads, norm, input_norm, bow, sd = (a[0] == "1", int(a[1]), int(a[2]), int(a[3]), int(a[4]), a[5], a[6] == "1", a[7] == "1", int(a[8]))
device = torch.device("cuda:0")
rng = np.random.default_rng(sd)
n_v, n_e, k = 48, 19, 5
f = 300 if bow else 12
ei = cases.random_hypergraph(rng, n_v, n_e, 170, True)
if bow:
    x = (rng.random((n_v, f)) < 0.03).astype(np.float32)
    x[np.arange(n_v), rng.integers(0, f, n_v)] = 1.0
else:
    x = rng.standard_normal((n_v, f)).astype(np.float32)
args = cases.make_args("pma_h1" if pma else "ds_add", f, hidden, k, All_num_layers=layers, MLP_num_layers=mlp_layers,
                       heads=heads if pma else 1, normalization=norm, deepset_input_norm=input_norm, Classifier_num_layers=2)
nrm = np.ones(ei.shape[1], dtype=np.int64)
norm_t = torch.from_numpy(nrm)
torch.manual_seed(sd)
model = SetGNN(args)
model.reset_parameters()
model.eval()
sdict = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
with torch.no_grad():
    shape = tuple(oracle.setgnn_forward(sdict, args, torch.from_numpy(x), torch.from_numpy(ei), norm_t).shape)
G = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
print("smooth:", util.oracle_is_smooth_here(sdict, args, x, ei, nrm, G, seed=sd))
for eps in (2e-6, 1e-5, 1e-4):
    print("  smooth at eps_bow / eps", eps, util.oracle_is_smooth_here(sdict, args, x, ei, nrm, G, seed=sd, eps=eps, eps_bow=eps))

def oracle_run(dtype):
    sdd = {kk: (v.clone().to(dtype).requires_grad_(True) if v.is_floating_point() and "running" not in kk else v.clone()) for kk, v in sdict.items()}
    xo = torch.from_numpy(x).to(dtype).requires_grad_(True)
    lo = oracle.setgnn_forward(sdd, args, xo, torch.from_numpy(ei), norm_t)
    (lo * G.to(dtype)).sum().backward()
    return lo.detach(), xo.grad, {kk: v.grad for kk, v in sdd.items() if v.is_floating_point() and v.requires_grad}
l32, gx32, g32 = oracle_run(torch.float32)
l64, gx64, g64 = oracle_run(torch.float64)
# every relu input of the float64 oracle: the entries closest to the kink, relative to their row's largest entry
import torch.nn.functional as F_
_relu, calls = F_.relu, []
def rec(t, *a_, **k_):
    calls.append(t.detach().clone())
    return _relu(t, *a_, **k_)
F_.relu = rec
oracle_run(torch.float64)
base_calls = list(calls)
dirn = np.random.default_rng(sd).standard_normal(x.shape)
if float((x == 0).mean()) > 0.5:
    dirn = dirn * np.abs(x)
x_keep = x
for sgn in (1.0, -1.0):
    calls.clear()
    x = (x_keep.astype(np.float64) + sgn * 1e-5 * dirn)
    F_.relu = rec
    sdd = {kk: (v.clone().double() if v.is_floating_point() else v.clone()) for kk, v in sdict.items()}
    oracle.setgnn_forward(sdd, args, torch.from_numpy(x), torch.from_numpy(ei), norm_t)
    F_.relu = _relu
    for ci, (t0, t1) in enumerate(zip(base_calls, calls)):
        if t0.dim() == 2 and t0.shape[1] > 64:
            rel = t0.abs() / t0.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
            i = int(rel.flatten().argmin())
            flips = int(((t0 > 0) != (t1 > 0)).sum())
            print(f"   x {'+' if sgn > 0 else '-'} 1e-5 dir: relu call {ci}: closest entry {float(t0.flatten()[i]):.3e} -> {float(t1.flatten()[i]):.3e}; sign flips in the call: {flips}")
x = x_keep
calls[:] = base_calls
F_.relu = _relu
for ci, t in enumerate(calls):
    if t.dim() != 2:
        continue
    rel = t.abs() / t.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    v, idx = rel.flatten().topk(3, largest=False)
    print(f"relu call {ci} {tuple(t.shape)}: closest to the kink (|pre| / row max): " +
          ", ".join(f"{float(a):.2e} at ({int(i) // t.shape[1]}, {int(i) % t.shape[1]}) pre {float(t.flatten()[i]):.3e}" for a, i in zip(v, idx)))
model.to(device)
gscale = max(float(g.abs().max()) for g in g64.values() if g is not None)
for label, arith, xgrad in (("default", "auto", False), ("strict", "bf16x6", False), ("dense path (x.requires_grad)", "auto", True), ("dense + strict", "bf16x6", True)):
    for p in model.parameters():
        p.grad = None
    xd = torch.from_numpy(x).to(device).requires_grad_(xgrad)
    with dense.arithmetic(arith):
        out = model(SimpleNamespace(x=xd, edge_index=torch.from_numpy(ei).to(device), norm=norm_t.to(device)))
        (out * G.to(device)).sum().backward()
    print(f"== {label}: logits err {float((out.detach().cpu().double() - l64).abs().max()):.3e} (fp32 oracle {float((l32.double() - l64).abs().max()):.3e})")
    named = dict(model.named_parameters())
    for kk, r in g64.items():
        if r is None:
            continue
        s_k = max(float(r.abs().max()), 1e-2 * gscale, 1e-30)
        e = (named[kk].grad.detach().cpu().double() - r).abs()
        err, o32 = float(e.max()), float((g32[kk].double() - r).abs().max())
        flag = " <<<" if err > max(1e-3 * s_k, 3 * o32) else ""
        if flag or "-v" in sys.argv:
            idx = np.unravel_index(int(e.argmax()), e.shape)
            print(f"   {kk:40s} err {err:.3e}  rel {err / s_k:.2e}  fp32-oracle {o32:.2e}  at {tuple(int(i) for i in idx)}  n(>tol/3) {int((e > 3e-4 * s_k).sum())} of {e.numel()}{flag}")
