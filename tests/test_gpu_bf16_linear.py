"""GPU: the bf16-regime Linear kernels (csrc/fused_bf16.hip) against a torch fp32 reference of the same arithmetic on the
same bf16 inputs.  The kernels accumulate in fp32 and round ONCE to bf16, so the bound is one bf16 rounding of the fp32
reference (2^-8 relative, half an ulp plus slack for the summation order) -- stated in `close_bf16`."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(256, 256), (128, 128), (256, 128), (128, 256)]
ROWS = [1, 15, 16, 17, 129, 1000, 4099]


def close_bf16(got, ref32, what=""):
    got = got.float().cpu()
    ref32 = ref32.float().cpu()
    tol = ref32.abs() * 2.0 ** -8 + 1e-3 * float(ref32.abs().max()) * 2.0 ** -8 + 1e-30
    bad = (got - ref32).abs() > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} beyond one bf16 rounding; worst {float((got - ref32).abs().max())}"


def bf(t, device):
    return t.to(torch.bfloat16).to(device)


@pytest.mark.parametrize("K,N", SHAPES)
@pytest.mark.parametrize("n", ROWS)
@pytest.mark.parametrize("relu", [False, True])
def test_forward_matches_fp32_reference(K, N, n, relu, device):
    from allset_amd import dense
    g = torch.Generator().manual_seed(n * 7 + K + 3 * N + int(relu))
    x, W, b = bf(torch.randn(n, K, generator=g), device), bf(torch.randn(N, K, generator=g) / K ** 0.5, device), bf(torch.randn(N, generator=g), device)
    aw, ab = bf(torch.randn(4, K, generator=g) / K ** 0.5, device), bf(torch.randn(4, generator=g), device)
    y, aux = dense.linear_bf16_fwd(x, W, b, relu, aw, ab)
    ref = x.float() @ W.float().t() + b.float()
    close_bf16(y, F.relu(ref) if relu else ref, "y")
    torch.testing.assert_close(aux.cpu(), (x.float() @ aw.float().t() + ab.float()).cpu(), rtol=1e-5, atol=1e-5)
    y2, none = dense.linear_bf16_fwd(x, W, None, relu)
    assert none is None
    ref2 = x.float() @ W.float().t()
    close_bf16(y2, F.relu(ref2) if relu else ref2, "y without bias")


@pytest.mark.parametrize("O,I", SHAPES)
@pytest.mark.parametrize("n", ROWS)
@pytest.mark.parametrize("mask,acc,aux", [(False, False, False), (True, False, False), (True, True, False), (False, False, True),
                                          (False, True, True), (True, True, True)])
def test_backward_data_matches_fp32_reference(O, I, n, mask, acc, aux, device):
    from allset_amd import dense
    g = torch.Generator().manual_seed(n * 11 + O + 5 * I + 4 * mask + 2 * acc + aux)
    gy, W = bf(torch.randn(n, O, generator=g), device), bf(torch.randn(O, I, generator=g) / O ** 0.5, device)
    y = bf(F.relu(torch.randn(n, O, generator=g)), device) if mask else None
    if mask and n > 1:
        y[1, 3] = -0.0                                     # a negative zero is not "> 0"
    a = bf(torch.randn(n, I, generator=g), device) if acc else None
    ga4 = torch.randn(n, 4, generator=g).to(device) if aux else None
    aw = bf(torch.randn(4, I, generator=g), device) if aux else None
    gx, ga = dense.linear_bf16_bwd(gy, W, y, want_ga=mask, acc_in=a, galpha=ga4, aux_w=aw)
    ga_ref = torch.where(y.float() > 0, gy.float(), torch.zeros((), device=device)) if mask else gy.float()
    assert torch.equal(ga.float(), ga_ref), "the masked gradient is a selection, exact"
    ref = ga_ref @ W.float()
    if acc:
        ref = ref + a.float()
    if aux:
        ref = ref + ga4 @ aw.float()
    close_bf16(gx, ref, "gx")


def test_rows_with_a_leading_dimension(device):
    """Row-strided views (columns of a wider buffer) are taken without a copy."""
    from allset_amd import dense
    g = torch.Generator().manual_seed(5)
    big = bf(torch.randn(700, 512, generator=g), device)
    x = big[:, 256:]
    W, b = bf(torch.randn(256, 256, generator=g) / 16, device), bf(torch.randn(256, generator=g), device)
    y, _ = dense.linear_bf16_fwd(x, W, b, True)
    close_bf16(y, F.relu(x.float() @ W.float().t() + b.float()))
    gx, _ = dense.linear_bf16_bwd(x, W, big[:, :256], want_ga=False)
    close_bf16(gx, torch.where(big[:, :256].float() > 0, x.float(), torch.zeros((), device=device)) @ W.float())


@pytest.mark.parametrize("relu", [False, True])
def test_autograd_node_against_torch_fp32(relu, device):
    from allset_amd import dense
    g = torch.Generator().manual_seed(17 + relu)
    n, K, N = 3000, 256, 256
    x, W, b = bf(torch.randn(n, K, generator=g), device), bf(torch.randn(N, K, generator=g) / 16, device), bf(torch.randn(N, generator=g), device)
    G = bf(torch.randn(n, N, generator=g), device)
    xs, Ws, bs = (t.clone().requires_grad_(True) for t in (x, W, b))
    y = dense.linear_bf16(xs, Ws, bs, relu)
    y.backward(G)
    xr, Wr, br = (t.float().requires_grad_(True) for t in (x, W, b))
    yr = xr @ Wr.t() + br
    yr = F.relu(yr) if relu else yr
    # the reference backward uses the mask of the bf16-rounded output, as the kernel (and torch's bf16 relu) does
    gr = torch.where(y.detach().float() > 0, G.float(), torch.zeros((), device=device)) if relu else G.float()
    yr.backward(gr if not relu else torch.where(yr.detach() > 0, gr, gr))
    close_bf16(y, yr.detach(), "y")
    close_bf16(xs.grad, gr @ W.float(), "gx")
    gw_ref, gb_ref = gr.t() @ x.float(), gr.sum(0)
    torch.testing.assert_close(Ws.grad.float(), gw_ref, rtol=2e-2, atol=2e-2 * float(gw_ref.abs().max()))
    torch.testing.assert_close(bs.grad.float(), gb_ref, rtol=2e-2, atol=2e-2 * float(gb_ref.abs().max()))


def test_unsupported_widths_fail_loudly(device):
    from allset_amd import dense, _lib
    x = torch.zeros(8, 64, dtype=torch.bfloat16, device=device)
    W = torch.zeros(64, 64, dtype=torch.bfloat16, device=device)
    assert not dense.linear_bf16_supported(x, W)
    with pytest.raises(_lib.AllSetHipError):
        dense.linear_bf16_fwd(x, W, None)
