"""GPU: training-mode BatchNorm1d of the reference MLP (`Normalization='bn'`, the class default: layers.py:499-517; applied as
``lin(dropout(bn(relu(x))))``, layers.py:571-579) on the HIP kernels of csrc/batchnorm.hip + the column-affine prologue of the fused
Linear, against float64 torch (batch statistics, biased variance to normalise, unbiased into running_var)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d", [(2, 64), (37, 128), (1031, 64), (70001, 128), (513, 1024), (300, 12)])
@pytest.mark.parametrize("relu_in", [False, True])
def test_col_moments_and_affine_add(n, d, relu_in, device):
    from allset_amd import dense
    g = torch.Generator().manual_seed(n * 7 + d)
    x = (torch.randn(n, d, generator=g) * 3 + 5 * torch.randn(d, generator=g)).to(device)        # column means far from 0
    f = F.relu(x.double()) if relu_in else x.double()
    s1 = dense.col_moments(x, relu_in)
    torch.testing.assert_close(s1.double(), f.sum(0), rtol=1e-5, atol=1e-5 * float(f.abs().sum(0).max()))
    mean = (f.sum(0) / n).float()
    s2 = dense.col_moments(x, relu_in, mean)
    ref2 = ((f - mean.double()) ** 2).sum(0)
    torch.testing.assert_close(s2.double(), ref2, rtol=2e-5, atol=1e-6 * float(ref2.max()))
    m1, v1 = dense.col_mean_var(x, relu_in)                      # one read, fp64 raw moments
    torch.testing.assert_close(m1.double(), f.mean(0), rtol=1e-6, atol=1e-6 * float(f.abs().max()))
    torch.testing.assert_close(v1.double(), f.var(0, unbiased=False), rtol=1e-5, atol=1e-7 * float(f.var(0, unbiased=False).max()))
    gx = torch.randn(n, d, generator=g).to(device)
    s, t = torch.randn(d, generator=g).to(device), torch.randn(d, generator=g).to(device)
    want = gx.double() + ((x > 0).double() if relu_in else 1.0) * (f * s.double() + t.double())
    got = dense.col_affine_add_(gx.clone(), x, s, t, relu_in)
    torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))


def test_col_moments_rejects_what_it_cannot_do(device):
    from allset_amd import dense, _lib
    with pytest.raises(_lib.AllSetHipError):
        dense.col_moments(torch.randn(10, 6, device=device))                        # width not a multiple of 4
    with pytest.raises(_lib.AllSetHipError):
        dense.col_moments(torch.randn(10, 8))                                       # CPU tensor


@pytest.mark.parametrize("K,N", [(128, 128), (64, 128), (128, 64), (64, 64)])
@pytest.mark.parametrize("relu_in,p_in,relu_out,p_out", [(False, 0.0, False, 0.0), (True, 0.0, True, 0.0), (True, 0.3, True, 0.4), (True, 0.5, False, 0.0)])
@pytest.mark.parametrize("n", [2, 33, 4099])          # (n = 2: two samples per column -- rstd up to 1e2, its cube in dvar: 20x the tolerance)
def test_bn_linear_matches_float64_torch(K, N, relu_in, p_in, relu_out, p_out, n, device, monkeypatch):
    """One BatchNorm -> dropout -> Linear (+ epilogue) step, training mode: output, running statistics, input gradient (direct +
    through the batch statistics), BatchNorm and Linear parameter gradients.  The dropout masks are the product's own (read back by
    running the same kernels on ones), applied explicitly on the float64 side."""
    from allset_amd import dense
    g = torch.Generator().manual_seed(K * 1000 + N * 10 + n)
    x = (torch.randn(n, K, generator=g) + 0.5 * torch.randn(K, generator=g)).to(device)
    bn = nn.BatchNorm1d(K).to(device).train()
    lin = nn.Linear(K, N).to(device)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * torch.randn(K, generator=g)); bn.bias.copy_(0.2 * torch.randn(K, generator=g))
    assert dense.bn_linear_supported(bn, lin, x)
    xg = x.clone().requires_grad_(True)
    seeds, draw = [], dense._draw_seed
    monkeypatch.setattr(dense, "_draw_seed", lambda: (seeds.append(draw()), seeds[-1])[1])
    torch.manual_seed(99)
    y = dense.bn_linear(bn, lin, xg, relu_in, p_in, relu_out, p_out)
    G = torch.randn(n, N, generator=g).to(device)
    (y * G).sum().backward()

    # float64 reference with the SAME masks: the sites' counter hash re-evaluated by the stand-alone relu-dropout kernel on ones
    # (tests/test_gpu_train_parity.py); seeds in the order the forward drew them (input site first)
    from test_gpu_train_parity import _keep_mask
    assert len(seeds) == int(p_in > 0) + int(p_out > 0)
    it = iter(seeds)
    keep_in = (_keep_mask(next(it), (n, K), p_in, device).to(device).double() / (1 - p_in)) if p_in > 0 else torch.ones(n, K, dtype=torch.float64, device=device)
    keep_out = (_keep_mask(next(it), (n, N), p_out, device).to(device).double() / (1 - p_out)) if p_out > 0 else torch.ones(n, N, dtype=torch.float64, device=device)
    xr = x.double().clone().requires_grad_(True)
    w, b = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    W, B = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    f = F.relu(xr) if relu_in else xr
    mean, var = f.mean(0), f.var(0, unbiased=False)
    u = ((f - mean) * torch.rsqrt(var + bn.eps) * w + b) * keep_in
    yr = u @ W.t() + B
    if relu_out:
        yr = F.relu(yr)
    yr = yr * keep_out
    (yr * G.double()).sum().backward()
    sc = lambda t: max(float(t.detach().abs().max()), 1e-3)
    slack = 20.0 if n == 2 else 1.0
    torch.testing.assert_close(y.detach().double(), yr.detach(), rtol=1e-4 * slack, atol=1e-4 * slack * sc(yr))
    torch.testing.assert_close(xg.grad.double(), xr.grad, rtol=2e-4 * slack, atol=2e-4 * slack * sc(xr.grad))
    for got, ref in ((bn.weight.grad, w.grad), (bn.bias.grad, b.grad), (lin.weight.grad, W.grad), (lin.bias.grad, B.grad)):
        torch.testing.assert_close(got.double(), ref, rtol=2e-4 * slack, atol=2e-4 * slack * sc(ref))
    # running statistics: torch's update rule (momentum 0.1, unbiased variance)
    torch.testing.assert_close(bn.running_mean.double(), 0.1 * mean.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bn.running_var.double(), 0.9 + 0.1 * f.detach().var(0, unbiased=True), rtol=1e-4, atol=1e-5)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("layers", [1, 2, 3])
def test_bn_mlp_training_mode_equals_the_torch_composition(d, layers, device):
    """The reference MLP with Normalization='bn', InputNorm=True in TRAINING mode without dropout: the HIP path
    (`MLP._bn_trainable`) against the same module evaluated by torch's own BatchNorm1d / Linear in float64 -- logits, every
    gradient, running statistics after the step."""
    import copy
    from allset_amd.layers import MLP
    torch.manual_seed(d + layers)
    m = MLP(d, d, d, layers, 0.0, "bn", True).to(device).train()
    ref = copy.deepcopy(m).double()
    x = torch.randn(777, d, device=device)
    assert m._bn_trainable(x)
    xg = x.clone().requires_grad_(True)
    y = m(xg)
    G = torch.randn_like(y)
    (y * G).sum().backward()
    xr = x.double().clone().requires_grad_(True)
    h = ref.normalizations[0](xr)
    for i, lin in enumerate(ref.lins[:-1]):
        h = ref.normalizations[i + 1](F.relu(lin(h)))
    yr = ref.lins[-1](h)
    (yr * G.double()).sum().backward()
    sc = lambda t: max(float(t.detach().abs().max()), 1e-3)
    torch.testing.assert_close(y.detach().double(), yr.detach(), rtol=1e-4, atol=1e-4 * sc(yr))
    torch.testing.assert_close(xg.grad.double(), xr.grad, rtol=3e-4, atol=3e-4 * sc(xr.grad))
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad.double(), q.grad, rtol=3e-4, atol=3e-4 * sc(q.grad), msg=lambda s, k=k: f"{k}: {s}")
    for (k, a), (_, b) in zip(m.named_buffers(), ref.named_buffers()):
        torch.testing.assert_close(a.double(), b.double(), rtol=1e-4, atol=1e-5, msg=lambda s, k=k: f"{k}: {s}")
