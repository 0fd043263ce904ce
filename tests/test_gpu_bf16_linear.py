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


# ---- round 6: the relu mask as one bit per element (allset_linear_bf16_fwd_mask / _bwd_bits / allset_wgrad_bf16_ex2) ----------
def decode_bits(bits, N):
    """uint8 [n, N / 8] in the kernels' private order -> bool [n, N]: a row is four words of N / 32 bytes (little-endian); bit
    16 hb + j of word sq is column 64 hb + 16 sq + j (include/allset_hip_ext.h)."""
    b = bits.cpu().numpy()
    import numpy as np
    n = b.shape[0]
    ws = N // 32
    flat = np.unpackbits(b, axis=1, bitorder="little").reshape(n, 4, ws * 8)          # [row, sq, bit]
    out = np.zeros((n, N), dtype=bool)
    for sq in range(4):
        for hb in range(N // 64):
            out[:, 64 * hb + 16 * sq:64 * hb + 16 * sq + 16] = flat[:, sq, 16 * hb:16 * hb + 16].astype(bool)
    return torch.from_numpy(out)


@pytest.mark.parametrize("K,N", SHAPES)
@pytest.mark.parametrize("n", ROWS)
def test_forward_with_bit_mask_is_the_plain_forward_plus_its_mask(K, N, n, device):
    from allset_amd import dense
    g = torch.Generator().manual_seed(n * 13 + K + 3 * N)
    x, W, b = bf(torch.randn(n, K, generator=g), device), bf(torch.randn(N, K, generator=g) / K ** 0.5, device), bf(torch.randn(N, generator=g), device)
    y0, _ = dense.linear_bf16_fwd(x, W, b, True)
    y, bits = dense.linear_bf16_fwd_mask(x, W, b)
    assert torch.equal(y, y0), "the mask output must not change y"
    assert bits.shape == (n, N // 8) and bits.dtype == torch.uint8
    assert torch.equal(decode_bits(bits, N), (y.float() > 0).cpu()), "bit = (y > 0) in the documented order"


@pytest.mark.parametrize("O,I", SHAPES)
@pytest.mark.parametrize("n", ROWS)
@pytest.mark.parametrize("acc", [False, True])
def test_backward_data_from_bits_equals_the_activation_masked_form(O, I, n, acc, device):
    """Same kernel, same arithmetic, only the mask's encoding differs: bit-identical gx."""
    from allset_amd import dense
    g = torch.Generator().manual_seed(n * 17 + O + 5 * I + acc)
    u, Wf, bfw = bf(torch.randn(n, I, generator=g), device), bf(torch.randn(O, I, generator=g) / I ** 0.5, device), bf(torch.randn(O, generator=g), device)
    y, bits = dense.linear_bf16_fwd_mask(u, Wf, bfw)                      # a real forward: the mask of y = relu(u Wf^T + b)
    gy = bf(torch.randn(n, O, generator=g), device)
    a = bf(torch.randn(n, I, generator=g), device) if acc else None
    gx0, ga0 = dense.linear_bf16_bwd(gy, Wf, y, want_ga=True, acc_in=a)
    gx = dense.linear_bf16_bwd_bits(gy, Wf, bits, acc_in=a)
    assert torch.equal(gx, gx0)
    # and the weight gradient with the mask applied in its staging equals the one computed from the stored masked gradient
    gw0, gb0 = dense.wgrad(ga0, u, want_bias=True)
    gw, gb = dense.wgrad_bf16_ex2(gy, u, bits=bits, want_bias=True)
    assert torch.equal(gw, gw0) and torch.equal(gb, gb0)
    gw_nb, none = dense.wgrad_bf16_ex2(gy, u, bits=bits, want_bias=False)
    assert none is None and torch.equal(gw_nb, gw0)


@pytest.mark.parametrize("n", [1, 31, 32, 33, 257, 4099, 70001])
@pytest.mark.parametrize("with_bits,bias", [(False, True), (False, False), (True, True)])
def test_weight_gradient_with_the_auxiliary_logit_rows(n, with_bits, bias, device):
    """PMA's four folded-logit rows inside the value projection's weight-gradient pass: the main block is bit-identical to the
    pass without them, the four rows match float64 on the bf16-rounded g4."""
    from allset_amd import dense
    g = torch.Generator().manual_seed(n * 19 + with_bits + 2 * bias)
    O = I = 256
    u, gy = bf(torch.randn(n, I, generator=g), device), bf(torch.randn(n, O, generator=g), device)
    g4 = torch.randn(n, 4, generator=g).to(device)
    bits = None
    if with_bits:
        Wf = bf(torch.randn(O, I, generator=g) / 16, device)
        _, bits = dense.linear_bf16_fwd_mask(u, Wf, None)
    gw0, gb0 = dense.wgrad_bf16_ex2(gy, u, bits=bits, want_bias=bias)
    gw, gb, gwa, gba = dense.wgrad_bf16_ex2(gy, u, bits=bits, g4=g4, want_bias=bias)
    assert torch.equal(gw, gw0) and (gb is None if not bias else torch.equal(gb, gb0))
    g16 = g4.to(torch.bfloat16).double()
    ref_w, ref_b = g16.t() @ u.double(), g16.sum(0)
    close_bf16(gwa, ref_w.float(), "gWa")
    close_bf16(gba, ref_b.float(), "gba")


@pytest.mark.parametrize("H,C,K,bias", [(4, 64, 256, True), (4, 32, 128, True), (1, 128, 128, False), (2, 64, 256, True), (8, 16, 64, True)])
def test_pma_fold_bf16_against_float64(H, C, K, bias, device):
    from allset_amd import dense
    g = torch.Generator().manual_seed(H * 100 + C + K)
    Wk, bk, att = bf(torch.randn(H * C, K, generator=g) / K ** 0.5, device), (bf(torch.randn(H * C, generator=g), device) if bias else None), \
        bf(torch.randn(1, H, C, generator=g), device)
    Wk_, att_ = Wk.clone().requires_grad_(True), att.clone().requires_grad_(True)
    bk_ = bk.clone().requires_grad_(True) if bias else None
    w, b = dense.pma_fold(Wk_, bk_, att_)
    assert w.dtype == b.dtype == torch.bfloat16
    Gw, Gb = bf(torch.randn(H, K, generator=g), device), bf(torch.randn(H, generator=g), device)
    ((w.float() * Gw.float()).sum() + (b.float() * Gb.float()).sum()).backward()
    Wd, ad = Wk.double().requires_grad_(True), att.double().requires_grad_(True)
    bd = bk.double().requires_grad_(True) if bias else None
    wr = (Wd.view(H, C, K) * ad.view(H, C, 1)).sum(1)
    br = (bd.view(H, C) * ad.view(H, C)).sum(1) if bias else torch.zeros(H, dtype=torch.float64, device=device)
    ((wr * Gw.double()).sum() + (br * Gb.double()).sum()).backward()
    close_bf16(w, wr.detach().float(), "w")
    close_bf16(b, br.detach().float(), "b")
    close_bf16(Wk_.grad, Wd.grad.float(), "gWk")
    close_bf16(att_.grad, ad.grad.float(), "gatt")
    if bias:
        close_bf16(bk_.grad, bd.grad.float(), "gbk")


def test_residual_block_node_with_bit_masks_equals_the_activation_masked_node(device, monkeypatch):
    """The rFF + ln1 node of PMA's tail (bf16 regime): with the bit masks every output and gradient is bit-identical to the round-5
    form (masks from the saved activations, masked gradient written and re-read)."""
    from allset_amd import dense
    g = torch.Generator().manual_seed(77)
    n, d = 5000, 256
    mk = lambda *s, sc=1.0: bf(torch.randn(*s, generator=g) * sc, device)
    out, w1, b1, w2, b2, gam, bet = mk(n, d), mk(d, d, sc=1 / 16), mk(d), mk(d, d, sc=1 / 16), mk(d), mk(d), mk(d)
    G = mk(n, d)

    def run():
        ts = [t.clone().requires_grad_(True) for t in (out, w1, b1, w2, b2, gam, bet)]
        y = dense.pma_residual_ff_bf16(*ts, 1e-5, True, 0.0)
        y.backward(G)
        return [y.detach()] + [t.grad for t in ts]
    new = run()
    monkeypatch.setattr(dense, "wgrad_bf16_ex2_supported", lambda *a: False)
    old = run()
    for a, b_ in zip(new, old):
        assert torch.equal(a, b_)
