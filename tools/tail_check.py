"""GPU check of the fused PMA tail (dense.pma_tail: ln0 / ln1 inside the two rFF Linears) and of the row-scaled fp16x3 forward
against float64 torch, several row counts; forward, every gradient.  python tools/tail_check.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allset_amd import dense  # noqa: E402

dev = torch.device("cuda:0")
for n in (1, 33, 4099, 70001):
    g = torch.Generator().manual_seed(n)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev).requires_grad_(True)
    pooled = mk(n, 128)
    att = mk(1, 4, 32, sc=0.5)
    g0, b0, g1, b1n = mk(128, sc=0.2), mk(128, sc=0.3), mk(128, sc=0.2), mk(128, sc=0.3)
    with torch.no_grad():
        g0 += 1; g1 += 1
    w1, w2 = mk(128, 128, sc=128 ** -0.5), mk(128, 128, sc=128 ** -0.5)
    bb1, bb2 = mk(128, sc=0.1), mk(128, sc=0.1)
    G = torch.randn(n, 128, generator=g).to(dev)
    for relu_post in (False, True):
        params = [pooled, att, g0, b0, w1, bb1, w2, bb2, g1, b1n]
        for t in params:
            t.grad = None
        y = dense.pma_tail(pooled, att, g0, b0, 1e-5, w1, bb1, w2, bb2, g1, b1n, 1e-5, relu_post, 0.0)
        (y * G).sum().backward()
        got = [t.grad.clone() for t in params]
        pd = [t.detach().double().requires_grad_(True) for t in params]
        P, A, G0, B0, W1, BB1, W2, BB2, G1, B1 = pd
        out = F.layer_norm(P + A.reshape(1, -1), (128,), G0, B0, 1e-5)
        z = F.relu(F.linear(F.relu(F.linear(out, W1, BB1)), W2, BB2))
        ref = F.layer_norm(out + z, (128,), G1, B1, 1e-5)
        if relu_post:
            ref = F.relu(ref)
        (ref * G.double()).sum().backward()
        e_y = float((y.double() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
        errs = [float((a.double() - b.grad).abs().max()) / max(float(b.grad.abs().max()), 1e-30) for a, b in zip(got, pd)]
        print(f"n={n} relu_post={int(relu_post)}: y {e_y:.1e} grads " + " ".join(f"{e:.1e}" for e in errs), flush=True)
    # the plain forward without a LayerNorm (row-scaled fp16x3) on rows spread over 40 binary orders, vs strict
    x = (torch.randn(n, 128, generator=g) * torch.exp2(torch.randint(-20, 21, (n, 1), generator=g).float())).to(dev)
    ref = x.double() @ w1.detach().double().t() + bb1.detach().double()
    den = x.double().abs() @ w1.detach().double().abs().t() + bb1.detach().double().abs()
    for mode in ("auto", "strict"):
        with dense.arithmetic(mode):
            yy, _ = dense.fused_linear_fwd(x, w1.detach(), bb1.detach())
        print(f"   plain fwd {mode}: {float(((yy.double() - ref).abs() / den).max()):.2e}")
