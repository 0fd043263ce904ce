"""A/B of the two arithmetics of the fused 128 x 128 Linear kernels, every prologue / epilogue variant, against float64.
Usage: python tools/arith_ab.py [n ...]"""
import itertools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allset_amd import dense  # noqa: E402

dev = torch.device("cuda:0")
ns = [int(a) for a in sys.argv[1:]] or [4611, 8710]
for n in ns:
    for has_ln, relu_in, p_in, with_mask in itertools.product([True, False], [False, True], [0.0, 0.5], [False, True]):
        if p_in > 0 and not relu_in:
            continue
        g = torch.Generator().manual_seed(n)
        x = torch.randn(n, 128, generator=g).to(dev)
        W = (torch.randn(128, 128, generator=g) / 128 ** 0.5).to(dev)
        b = torch.randn(128, generator=g).to(dev)
        gamma, beta = (1 + 0.2 * torch.randn(128, generator=g)).to(dev), (0.3 * torch.randn(128, generator=g)).to(dev)
        G = torch.randn(n, 128, generator=g).to(dev)
        ln = (gamma, beta) if has_ln else (None, None)
        p_out, s_in, s_out = (0.5, 4242, 977) if with_mask else (0.0, 4242, 0)
        res = {}
        for mode in ("auto", "strict"):
            with dense.arithmetic(mode):
                mask = torch.empty(dense.activation_mask_words(n, 128), dtype=torch.int32, device=dev) if with_mask else None
                y, st = dense.fused_linear_fwd(x, W, b, ln[0], ln[1], 1e-5, relu_in, p_in, s_in, with_mask, p_out, s_out, None, mask)
                gx, dg, db, gw, gb = dense.fused_linear_bwd_all(G, mask, p_out, W, x, st, ln[0], ln[1], relu_in, p_in, s_in)
                res[mode] = (y, gx, gw, gb, dg, db)
        names = ("y", "gx", "gw", "gb", "dg", "db")
        line = f"n={n} ln={int(has_ln)} relu={int(relu_in)} p_in={p_in} mask={int(with_mask)}:"
        for k, a, s in zip(names, res["auto"], res["strict"]):
            if a is None:
                continue
            d = float((a - s).abs().max()) / max(float(s.abs().max()), 1e-30)
            line += f" {k} {d:.1e}"
        print(line, flush=True)
