#!/usr/bin/env python
"""The bf16-regime Linear kernels (csrc/fused_bf16.hip) beside the library path they replace, at the configs[4] per-GPU shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from allset_amd import dense
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000


def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for K, N in ((256, 256), (128, 128)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, K, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    aw = torch.randn(4, K, generator=g).to(torch.bfloat16).to(dev); ab = torch.zeros(4, dtype=torch.bfloat16, device=dev)
    gy = torch.randn(n, N, generator=g).to(torch.bfloat16).to(dev)
    y = F.relu(F.linear(x, W, b))
    acc = torch.randn(n, K, generator=g).to(torch.bfloat16).to(dev)
    ga4 = torch.randn(n, 4, generator=g).to(dev)
    floor = n * (K + N) * 2 / 5e12 * 1e6
    print(f"== [{n}, {K}] x [{K}, {N}]  (one read + one write at 5 TB/s: {floor:.0f} us)")
    print(f"fwd  library F.linear                  {timeit(lambda: F.linear(x, W, b)):8.1f} us")
    print(f"fwd  library F.linear + relu           {timeit(lambda: F.relu(F.linear(x, W, b))):8.1f} us")
    print(f"fwd  kernel                            {timeit(lambda: dense.linear_bf16_fwd(x, W, b, False)):8.1f} us")
    print(f"fwd  kernel + relu                     {timeit(lambda: dense.linear_bf16_fwd(x, W, b, True)):8.1f} us")
    print(f"fwd  kernel + 4 logit columns          {timeit(lambda: dense.linear_bf16_fwd(x, W, b, False, aw, ab)):8.1f} us")
    print(f"fwd  library F.linear + skinny logits  {timeit(lambda: (F.linear(x, W, b), F.linear(x, aw, ab))):8.1f} us")
    print(f"bwd  library gy @ W                    {timeit(lambda: gy @ W):8.1f} us")
    print(f"bwd  library relu-bwd + gy @ W + add   {timeit(lambda: (torch.where(y > 0, gy, torch.zeros_like(gy)) @ W) + acc):8.1f} us")
    print(f"bwd  kernel                            {timeit(lambda: dense.linear_bf16_bwd(gy, W)):8.1f} us")
    print(f"bwd  kernel + mask                     {timeit(lambda: dense.linear_bf16_bwd(gy, W, y, want_ga=False)):8.1f} us")
    print(f"bwd  kernel + mask + ga out            {timeit(lambda: dense.linear_bf16_bwd(gy, W, y, want_ga=True)):8.1f} us")
    print(f"bwd  kernel + mask + ga out + acc_in   {timeit(lambda: dense.linear_bf16_bwd(gy, W, y, want_ga=True, acc_in=acc)):8.1f} us")
    print(f"bwd  kernel + logits' rank-4 term      {timeit(lambda: dense.linear_bf16_bwd(gy, W, galpha=ga4, aux_w=aw)):8.1f} us")
