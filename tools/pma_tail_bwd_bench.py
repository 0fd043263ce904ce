#!/usr/bin/env python
"""The PMA tail's first rFF Linear, backward, at [1M, 128]: the one-pass kernel with ln0's backward and the pooling statistics inside
(dense.fused_linear_bwd_pma_tail, csrc/fused_bwd6.hip PT) against the two passes it replaces (fused_linear_bwd_all with acc_in +
ln_res_bwd_pma), HIP-event medians on one box.  python tools/pma_tail_bwd_bench.py [heads]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from allset_amd import dense
dev = torch.device("cuda:0")
n, H = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
pooled, cb = torch.randn(n, 128, device=dev), torch.randn(128, device=dev) * 0.5
g0, b0 = torch.ones(128, device=dev), torch.zeros(128, device=dev)
w1 = torch.randn(128, 128, device=dev) / 128 ** 0.5
gh, gs = torch.randn(n, 128, device=dev), torch.randn(n, 128, device=dev)
m, l = torch.randn(n, H, device=dev), torch.rand(n, H, device=dev) + 0.5
x = pooled + cb
stats0 = torch.cat([x.mean(1, keepdim=True), torch.rsqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)], 1).contiguous()
out = F.layer_norm(x, (128,), g0, b0, 1e-5)
del x


def timed(fn, reps=15):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
    return statistics.median(ts)


acc = gs.clone()
t_lin = timed(lambda: dense.fused_linear_bwd_all(gh, None, 0.0, w1, out, None, None, None, False, 0.0, 0, acc_in=acc))
gout = torch.randn(n, 128, device=dev)
t_ln = timed(lambda: dense.ln_res_bwd_pma(gout, pooled, cb, stats0, g0, b0, m, l))
t_pt = timed(lambda: dense.fused_linear_bwd_pma_tail(gh, w1, pooled, cb, stats0, g0, b0, gs, m, l))
t_plain = timed(lambda: dense.fused_linear_bwd_all(gh, None, 0.0, w1, out, None, None, None, False, 0.0, 0))
t_lnv = timed(lambda: dense.fused_linear_bwd_all(gh, None, 0.0, w1, pooled, stats0, g0, b0, False, 0.0, 0))
print(f"heads {H}: two passes {t_lin:.3f} (Linear backward, acc_in) + {t_ln:.3f} (ln_res_bwd_pma) = {t_lin + t_ln:.3f} ms;  one pass {t_pt:.3f} ms")
print(f"   for scale: plain one-pass backward {t_plain:.3f} ms, behind a LayerNorm prologue {t_lnv:.3f} ms  (reduce_partials launches included)")
