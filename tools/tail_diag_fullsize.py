import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from allset_amd import dense
device = torch.device("cuda:0")
n = 1_000_003
g = torch.Generator(device=device).manual_seed(21)
mk = lambda *s, sc=1.0, off=0.0: (torch.randn(*s, device=device, generator=g) * sc + off).requires_grad_(True)
pooled = mk(n, 128, sc=2.0); att = mk(1, 4, 32, sc=0.5)
g0, b0, g1, b1 = mk(128, sc=0.2, off=1.0), mk(128, sc=0.3), mk(128, sc=0.2, off=1.0), mk(128, sc=0.3)
w1, w2 = mk(128, 128, sc=128 ** -0.5), mk(128, 128, sc=128 ** -0.5)
bb1, bb2 = mk(128, sc=0.1), mk(128, sc=0.1)
sign = torch.where(torch.arange(128, device=device) % 2 == 0, 5.0, -5.0)
with torch.no_grad():
    bb1 += sign; bb2 += 4 * sign
G = torch.randn(n, 128, device=device, generator=g)
params = [pooled, att, g0, b0, w1, bb1, w2, bb2, g1, b1]
names = ["pooled", "att_r", "ln0.w", "ln0.b", "w1", "b1", "w2", "b2", "ln1.w", "ln1.b"]
res = {}
for mode in ("auto", "strict"):
    for t in params: t.grad = None
    with dense.arithmetic(mode):
        if mode == "auto":
            y = dense.pma_tail(pooled, att, g0, b0, 1e-5, w1, bb1, w2, bb2, g1, b1, 1e-5, False, 0.0)
        else:
            o2 = dense.layer_norm_res(pooled, att.reshape(-1), None, g0, b0, 1e-5)
            y = dense.pma_residual_ff(o2, w1, bb1, w2, bb2, g1, b1, 1e-5, False, 0.0)
        (y * G).sum().backward()
    res[mode] = [t.grad.clone() for t in params]
pd = [t.detach().double().requires_grad_(True) for t in params]
P, A, G0, B0, W1, BB1, W2, BB2, G1, B1 = pd
out = F.layer_norm(P + A.reshape(1, -1), (128,), G0, B0, 1e-5)
h = F.relu(F.linear(out, W1, BB1))
ref = F.layer_norm(out + F.relu(F.linear(h, W2, BB2)), (128,), G1, B1, 1e-5)
(ref * G.double()).sum().backward()
for i, nm in enumerate(names):
    r = pd[i].grad
    sc = float(r.abs().max())
    print(f"{nm:8s} scale {sc:10.3e}  auto {float((res['auto'][i].double() - r).abs().max()) / sc:.2e}  strict {float((res['strict'][i].double() - r).abs().max()) / sc:.2e}")
