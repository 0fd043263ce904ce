"""The fp16x3 one-pass backward (csrc/fused_bwd6.hip) against float64 under hostile gradient scales, and its launch time.

    python tools/bwd_f16x3_check.py [--n 1000000] [--iters 30]

Accuracy: gx, gW, gb, dgamma, dbeta against a float64 evaluation of the same math (LayerNorm prologue, no dropout so that the
reference needs no mask logic) for gradients whose rows span 60 binary orders of magnitude, tiny / huge overall scales, zero rows,
and a row that raises the workgroup's running exponent late.  The error measure is the one fp32 itself is held to:
|result - exact| / sum |terms|.
Timing: the light (LayerNorm only) and heavy (LayerNorm + relu + dropout in, mask on gy) variants at [n,128] x [128,128]."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allset_amd import dense  # noqa: E402


def reference(G, W, x, gamma, beta, eps=1e-5):
    Gd, Wd, xd, gd, bd = G.double(), W.double(), x.double(), gamma.double(), beta.double()
    mean = xd.mean(1, keepdim=True)
    var = ((xd - mean) ** 2).mean(1, keepdim=True)
    rstd = (var + eps).rsqrt()
    xh = (xd - mean) * rstd
    u = xh * gd + bd
    gu = Gd @ Wd
    gw = Gd.t() @ u
    gb = Gd.sum(0)
    dgam = (gu * xh).sum(0)
    dbet = gu.sum(0)
    v = gu * gd
    s1 = v.mean(1, keepdim=True)
    s2 = (v * xh).mean(1, keepdim=True)
    gx = rstd * (v - s1 - xh * s2)
    den_w = Gd.abs().t() @ ((xh * gd).abs() + bd.abs())       # (the terms of u = xhat gamma + beta: fp32 rounds u itself)
    den_gu = Gd.abs() @ Wd.abs()
    return gx, gw, gb, dgam, dbet, den_w, den_gu, rstd, gd


def case(name, G, W, x, gamma, beta, dev):
    G, W, x, gamma, beta = (t.to(dev) for t in (G, W, x, gamma, beta))
    b = torch.zeros(W.shape[0], device=dev)
    y, st = dense.fused_linear_fwd(x, W, b, gamma, beta, 1e-5, False, 0.0, 0, False, 0.0, 0, None, None)
    gx, dg, db, gw, gb = dense.fused_linear_bwd_all(G, None, 0.0, W, x, st, gamma, beta, False, 0.0, 0)
    rgx, rgw, rgb, rdg, rdb, den_w, den_gu, rstd, gd = reference(G, W, x, gamma, beta)
    # gx: error relative to the row's sum |terms| of gu carried through the LayerNorm backward (scale rstd * max|gamma|)
    row_den = den_gu.max(1, keepdim=True).values * rstd * gd.abs().max() + 1e-300
    e_gx = float(((gx.double() - rgx).abs() / row_den).max())
    e_gw = float(((gw.double() - rgw).abs() / (den_w + 1e-300)).max())
    e_gb = float(((gb.double() - rgb).abs() / (G.double().abs().sum(0) + 1e-300)).max())
    gu_abs = (G.double().abs() @ W.double().abs())
    e_dg = float(((dg.double() - rdg).abs() / ((gu_abs * 6).sum(0) + 1e-300)).max())
    e_db = float(((db.double() - rdb).abs() / (gu_abs.sum(0) + 1e-300)).max())
    ok = e_gx < 2e-6 and e_gw < 5e-7 and e_gb < 5e-7 and e_dg < 5e-7 and e_db < 5e-7 and bool(torch.isfinite(gw).all())
    print(f"{name:42s} gx {e_gx:.2e}  gW {e_gw:.2e}  gb {e_gb:.2e}  dgamma {e_dg:.2e}  dbeta {e_db:.2e}  {'ok' if ok else 'FAIL'}")
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--no-time", action="store_true")
    ap.add_argument("--time-only", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    ok = True
    for n in (() if a.time_only else (1, 33, 4099, 70001)):
        x = torch.randn(n, 128, generator=g) * 3 + 0.5
        W = torch.randn(128, 128, generator=g) / 128 ** 0.5
        gamma, beta = 1 + 0.2 * torch.randn(128, generator=g), 0.3 * torch.randn(128, generator=g)
        G = torch.randn(n, 128, generator=g)
        ok &= case(f"n={n} plain", G, W, x, gamma, beta, dev)
        ok &= case(f"n={n} gy * 1e-9", G * 1e-9, W, x, gamma, beta, dev)
        ok &= case(f"n={n} gy * 1e+9", G * 1e9, W, x, gamma, beta, dev)
        rs = torch.exp2(torch.randint(-30, 31, (n, 1), generator=g).float())
        ok &= case(f"n={n} row scales 2^-30..2^30", G * rs, W, x, gamma, beta, dev)
        G2 = G * 1e-6
        G2[-1] *= 1e12          # the last row raises the running exponent by 40 binary orders
        G2[n // 2] = 0
        ok &= case(f"n={n} late large row, a zero row", G2, W, x, gamma, beta, dev)
        ok &= case(f"n={n} W * 1e-5, gamma * 1e3", G, W * 1e-5, x, gamma * 1e3, beta, dev)
        ok &= case(f"n={n} gamma * 1e-4, beta * 1e-4", G, W, x, gamma * 1e-4, beta * 1e-4, dev)
        cs = torch.exp2(torch.randint(-12, 13, (1, 128), generator=g).float())
        ok &= case(f"n={n} column scales 2^-12..2^12", G * cs, W, x, gamma, beta, dev)
    print("ACCURACY", "OK" if ok else "FAILED")
    if a.no_time:
        return 0 if ok else 1
    n = a.n
    x = torch.randn(n, 128, device=dev)
    W = (torch.randn(128, 128, device=dev) / 128 ** 0.5)
    b = torch.zeros(128, device=dev)
    gamma, beta = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    G = torch.randn(n, 128, device=dev)
    for name, relu_in, p_in, with_mask in (("light (LayerNorm)", False, 0.0, False), ("heavy (LN + relu + dropout, mask)", True, 0.5, True),
                                           ("light (LayerNorm)", False, 0.0, False), ("heavy (LN + relu + dropout, mask)", True, 0.5, True)):
        mask = torch.empty(dense.activation_mask_words(n, 128), dtype=torch.int32, device=dev) if with_mask else None
        p_out = 0.5 if with_mask else 0.0
        y, st = dense.fused_linear_fwd(x, W, b, gamma, beta, 1e-5, relu_in, p_in, 11, with_mask, p_out, 13, None, mask)
        for _ in range(5):
            dense.fused_linear_bwd_all(G, mask, p_out, W, x, st, gamma, beta, relu_in, p_in, 11)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(a.iters):
            dense.fused_linear_bwd_all(G, mask, p_out, W, x, st, gamma, beta, relu_in, p_in, 11)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / a.iters
        print(f"{name:40s} {ms:.4f} ms per call incl. the partial reduction  ({n * 128 * 12 / ms / 1e9:.2f} TB/s algorithmic)")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
