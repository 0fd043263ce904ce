#!/usr/bin/env python
"""Weight gradient for bf16 activations (allset_wgrad_bf16) at the configs[4] per-GPU shape: time and achieved GB/s
against the two input streams it has to read once.  usage: wgrad_bf16_bench.py [n] [d]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allset_amd import dense
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
ga = torch.randn(n, d, device=dev).to(torch.bfloat16)
u = torch.randn(n, d, device=dev).to(torch.bfloat16)
for _ in range(3):
    dense.wgrad(ga, u, True)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); dense.wgrad(ga, u, True); e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e))
ms = statistics.median(ts)
print(f"wgrad bf16 n={n} d={d}: {ms*1e3:.0f} us (incl. the partial-sum reduction)  {2*n*d*2/ms/1e6:.0f} GB/s of input streams")
from allset_amd import ops
t = ops.KernelTimer(); ops.set_kernel_timer(t)
for _ in range(10):
    gw, gb = dense.wgrad(ga, u, True)
torch.cuda.synchronize(); ops.set_kernel_timer(None)
for k, v in t.summary().items():
    print(f"  {k}: {v['avg_ms']*1e3:.0f} us x{v['calls']}")
ref = ga.double().t() @ u.double()
print("max rel err vs float64:", float((gw.double() - ref).abs().max() / ref.abs().max()))
