"""-m gpu: the convolution kernels of the convolutional VAE (BASELINE config 5; no reference counterpart -- DESIGN.md 3.4) against
the float64 oracle's layer functions (oracle/conv_vae_oracle.py) on float32-rounded inputs.  bf16 matrix-core products with
float32 accumulation: the bf16 Dense path's envelope (1e-2 of the output's max-abs; measured ~3e-3), not the 1e-5 contract."""
import math

import numpy as np
import pytest
import torch

from oracle import conv_vae_oracle as CO

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda().contiguous()


@pytest.mark.parametrize("B,S,cin,cout,relu", [
    (8, 64, 1, 32, True),          # the four encoder layers of config 5
    (8, 32, 32, 64, True),
    (8, 16, 64, 128, True),
    (8, 8, 128, 256, True),
    (3, 16, 4, 5, False),          # ragged everything: rows, columns, a k-range that is exactly two k-tiles
    (2, 6, 3, 7, True),            # channel count off the 4-grid: the scalar gather; non-square-power sizes
    (1, 2, 8, 130, False),         # one output pixel per image; more than one column tile
    (37, 8, 12, 16, True),         # 592 output pixels: a partial last row tile
    (3, 10, 8, 32, True),          # the LDS-DMA form (power-of-two c_in, c_out % 32 == 0): fewer rows than one tile, 32-column tiles
    (5, 6, 16, 96, False),         # three 32-column tiles
    (9, 12, 32, 192, True),        # 64-column tiles, a partial last row tile
])
def test_conv2d_forward_matches_the_oracle(B, S, cin, cout, relu):
    from vae_training_amd.conv import conv2d_forward
    rng = np.random.default_rng(B * 1000 + S + cin)
    r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    x = r32(rng.standard_normal((B, S, S, cin)))
    w = r32(rng.standard_normal((4, 4, cin, cout)) / np.sqrt(16 * cin))
    b = r32(0.1 * rng.standard_normal(cout))
    want = CO.conv_fwd(x, w, b)
    if relu:
        want = np.maximum(want, 0.0)
    got = conv2d_forward(_dev(x), _dev(w), _dev(b), relu).cpu().numpy().astype(np.float64)
    assert got.shape == want.shape
    err = np.max(np.abs(got - want)) / np.max(np.abs(want))
    assert err <= 1e-2, err
    # the products are bf16: the same call is NOT float32-exact, but it is repeatable bit for bit
    again = conv2d_forward(_dev(x), _dev(w), _dev(b), relu).cpu().numpy().astype(np.float64)
    assert np.array_equal(got, again)


def test_lds_dma_and_register_staged_forms_agree():
    """Same bf16 products, float32 accumulation in a different order: 1e-5 of the result's scale."""
    from vae_training_amd.conv import conv2d_forward, conv2d_transpose_forward
    torch.manual_seed(1)
    x = torch.randn(6, 16, 16, 32, device="cuda"); K = torch.randn(4, 4, 32, 64, device="cuda") / 22.6; b = torch.randn(64, device="cuda")
    a, c = conv2d_forward(x, K, b, True), conv2d_forward(x, K, b, True, fast=False)
    assert float((a - c).abs().max()) <= 1e-5 * float(c.abs().max())
    y = torch.randn(6, 8, 8, 64, device="cuda"); m = torch.randn(6, 16, 16, 32, device="cuda")
    a, c = conv2d_transpose_forward(y, K, None, False, m), conv2d_transpose_forward(y, K, None, False, m, fast=False)
    assert float((a - c).abs().max()) <= 1e-5 * float(c.abs().max())


def test_bf16_copies_in_and_out_change_nothing():
    """x16 / want16 (conv.py): with the operand copy handed in the result is bitwise the same; the copy handed out is the bf16
    rounding of the float32 result in every form (epilogue or conversion pass)."""
    from vae_training_amd.conv import conv2d_forward, conv2d_transpose_forward, conv2d_weight_grad, to_bf16
    torch.manual_seed(2)
    for cin, cout in ((32, 64), (1, 32), (12, 16)):                     # LDS-DMA form, streaming form, register-staged form
        x = torch.randn(5, 16, 16, cin, device="cuda"); K = torch.randn(4, 4, cin, cout, device="cuda") / 8; dy = torch.randn(5, 8, 8, cout, device="cuda")
        x16 = to_bf16(x)
        assert torch.equal(x16, x.to(torch.bfloat16))
        y0 = conv2d_forward(x, K, None, True)
        y1, y16 = conv2d_forward(x, K, None, True, x16=x16, want16=True)
        assert torch.equal(y0, y1) and torch.equal(y16, y1.to(torch.bfloat16))
        a, _ = conv2d_weight_grad(x, dy)
        b, _ = conv2d_weight_grad(x, dy, x16=x16, dy16=to_bf16(dy))
        assert torch.equal(a, b)
        t0 = conv2d_transpose_forward(dy, K)
        t1, t16 = conv2d_transpose_forward(dy, K, y16=to_bf16(dy), want16=True)
        assert torch.equal(t0, t1) and torch.equal(t16, t1.to(torch.bfloat16))


def test_conv2d_forward_rejects_odd_sizes():
    from vae_training_amd._lib import VaekError
    from vae_training_amd.conv import conv2d_forward
    with pytest.raises(VaekError):
        conv2d_forward(torch.zeros(1, 5, 5, 4, device="cuda"), torch.zeros(4, 4, 4, 8, device="cuda"))


@pytest.mark.parametrize("B,h,cin,cout,relu,masked", [
    (8, 4, 256, 128, True, False),     # the four decoder layers of config 5
    (8, 8, 128, 64, True, False),
    (8, 16, 64, 32, True, False),
    (8, 32, 32, 1, False, False),
    (3, 5, 4, 5, False, True),         # odd sizes, ragged rows and columns, the input-gradient form with a relu mask
    (2, 3, 3, 7, True, False),         # channel count off the 4-grid: the scalar gathers
    (1, 1, 8, 130, False, True),       # one input pixel; more than one column tile
    (3, 5, 16, 32, False, True),       # the LDS-DMA form: odd sizes, ragged rows, mask
    (2, 3, 32, 96, True, False),
    (1, 1, 64, 64, True, True),        # one input pixel: every neighbour tap off the image
])
def test_conv2d_transpose_forward_matches_the_oracle(B, h, cin, cout, relu, masked):
    from vae_training_amd.conv import conv2d_transpose_forward
    rng = np.random.default_rng(B * 1000 + h + cin)
    r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    y = r32(rng.standard_normal((B, h, h, cin)))
    w = r32(rng.standard_normal((4, 4, cout, cin)) / np.sqrt(4 * cin))
    b = r32(0.1 * rng.standard_normal(cout))
    want = CO.conv_t_fwd(y, w, b)
    if relu:
        want = np.maximum(want, 0.0)
    mask = r32(rng.standard_normal(want.shape)) if masked else None
    if masked:
        want = want * (mask > 0.0)
    got = conv2d_transpose_forward(_dev(y), _dev(w), _dev(b), relu, None if mask is None else _dev(mask)).cpu().numpy().astype(np.float64)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) <= 1e-2


def test_transposed_kernel_is_the_adjoint_of_the_forward_kernel():
    """<conv(x), y> == <x, conv_t(y)> for the two HIP kernels themselves (bf16 products: to 1e-2 of the inner product's scale)."""
    from vae_training_amd.conv import conv2d_forward, conv2d_transpose_forward
    torch.manual_seed(0)
    x = torch.randn(4, 16, 16, 8, device="cuda"); K = torch.randn(4, 4, 8, 12, device="cuda") / 11.3
    y = torch.randn(4, 8, 8, 12, device="cuda")
    lhs = float((conv2d_forward(x, K).double() * y.double()).sum())
    rhs = float((x.double() * conv2d_transpose_forward(y, K).double()).sum())
    scale = float(conv2d_forward(x, K).double().norm() * y.double().norm())
    assert abs(lhs - rhs) <= 1e-2 * scale, (lhs, rhs, scale)


@pytest.mark.parametrize("B,S,cin,cout", [
    (8, 64, 1, 32), (8, 32, 32, 64), (8, 16, 64, 128), (8, 8, 128, 256),       # the four encoder layers of config 5
    (3, 16, 4, 5), (2, 6, 3, 7), (1, 2, 8, 130), (37, 8, 12, 16),
])
def test_conv2d_weight_grad_matches_the_oracle(B, S, cin, cout):
    from vae_training_amd.conv import conv2d_weight_grad
    rng = np.random.default_rng(B * 1000 + S + cin + 7)
    r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    x = r32(rng.standard_normal((B, S, S, cin)))
    dy = r32(rng.standard_normal((B, S // 2, S // 2, cout)))
    _, want_w, want_b = CO.conv_bwd(x, np.zeros((4, 4, cin, cout)), dy)
    dw, db = conv2d_weight_grad(_dev(x), _dev(dy))
    dw2, db2 = conv2d_weight_grad(_dev(x), _dev(dy))
    assert torch.equal(dw, dw2) and torch.equal(db, db2)                     # slabs + fixed-order sum: bitwise repeatable
    dw, db = dw.cpu().numpy().astype(np.float64), db.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(dw - want_w)) <= 1e-2 * np.max(np.abs(want_w))
    assert np.max(np.abs(db - want_b)) <= 1e-2 * np.max(np.abs(want_b))


@pytest.mark.parametrize("B,S,c", [(8, 64, 32), (3, 10, 8), (5, 6, 64), (2, 4, 256), (1, 2, 1), (3, 16, 16), (1, 8, 4), (2, 24, 128)])
def test_one_channel_layers_stream_in_exact_float32(B, S, c):
    """A 1 <-> c channel layer (config 5's first convolution, last transposed convolution, and their gradients) runs in streaming
    f32 kernels, not bf16 GEMMs: 1e-5 of the result's scale against the oracle, masks and relu included, ragged sizes included."""
    from vae_training_amd.conv import conv2d_forward, conv2d_transpose_forward, conv2d_weight_grad
    rng = np.random.default_rng(B * 100 + S + c)
    r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    close = lambda got, want: np.max(np.abs(got.cpu().numpy().astype(np.float64) - want)) <= 1e-5 * max(np.max(np.abs(want)), 1e-30)
    x = r32(rng.standard_normal((B, S, S, 1)))
    w = r32(rng.standard_normal((4, 4, 1, c)) / 4)
    b = r32(0.1 * rng.standard_normal(c))
    mask = r32(rng.standard_normal((B, S // 2, S // 2, c)))
    want = np.maximum(CO.conv_fwd(x, w, b), 0.0)
    assert close(conv2d_forward(_dev(x), _dev(w), _dev(b), True), want)
    assert close(conv2d_forward(_dev(x), _dev(w), None, False, _dev(mask)), CO.conv_fwd(x, w, np.zeros(c)) * (mask > 0))
    dy = r32(rng.standard_normal((B, S // 2, S // 2, c)))
    _, want_w, want_b = CO.conv_bwd(x, np.zeros((4, 4, 1, c)), dy)
    dw, db = conv2d_weight_grad(_dev(x), _dev(dy))
    assert close(dw, want_w) and close(db, want_b)
    if c % 4 == 0:
        h = S // 2
        y = r32(rng.standard_normal((B, h, h, c)))
        wt = r32(rng.standard_normal((4, 4, 1, c)) / np.sqrt(4 * c))
        b1 = r32(0.1 * rng.standard_normal(1))
        want = CO.conv_t_fwd(y, wt, b1)
        assert close(conv2d_transpose_forward(_dev(y), _dev(wt), _dev(b1), False), want)
        m2 = r32(rng.standard_normal(want.shape))
        assert close(conv2d_transpose_forward(_dev(y), _dev(wt), _dev(b1), True, _dev(m2)), np.maximum(want, 0.0) * (m2 > 0))


@pytest.mark.parametrize("pixels,c", [(8 * 64 * 64, 1), (8 * 32 * 32, 32), (1001, 4), (37, 2), (50, 12), (9, 1024), (3, 1), (70000, 64)])
def test_conv2d_bias_grad_sums_the_pixels(pixels, c):
    from vae_training_amd.conv import conv2d_bias_grad
    rng = np.random.default_rng(pixels + c)
    dy = np.asarray(rng.standard_normal((pixels, c)), np.float32)
    got = conv2d_bias_grad(_dev(dy)).cpu().numpy().astype(np.float64)
    want = dy.astype(np.float64).sum(axis=0)
    assert np.max(np.abs(got - want)) <= 1e-5 * np.sqrt(pixels) * max(1.0, np.max(np.abs(want)) / np.sqrt(pixels))
    assert torch.equal(conv2d_bias_grad(_dev(dy)), conv2d_bias_grad(_dev(dy)))            # fixed-order sums


@pytest.mark.parametrize("pixels,c", [(8 * 32 * 32, 32), (70000, 64), (16 * 16 * 3, 128), (1000, 8), (2048, 2048)])
def test_conv2d_bias_grad_of_a_bf16_tensor(pixels, c):
    """vaek_conv2d_bias_grad_bf16: float32 sums of the bf16 VALUES (the lean conv VAE's transposed-layer bias gradients), fixed order."""
    from vae_training_amd.conv import conv2d_bias_grad
    rng = np.random.default_rng(pixels + c)
    dy16 = torch.from_numpy(np.asarray(rng.standard_normal((pixels, c)), np.float32)).cuda().to(torch.bfloat16).contiguous()
    got = conv2d_bias_grad(dy16).cpu().numpy().astype(np.float64)
    want = dy16.float().cpu().numpy().astype(np.float64).sum(axis=0)
    assert np.max(np.abs(got - want)) <= 1e-5 * np.sqrt(pixels) * max(1.0, np.max(np.abs(want)) / np.sqrt(pixels))
    assert torch.equal(conv2d_bias_grad(dy16), conv2d_bias_grad(dy16))


def test_lean_forms_of_the_conv_entry_points():
    """The bf16-only forms (NULL float32 tensor beside its bf16 copy, relu mask as a bf16 copy) give bitwise the bf16 results of the
    float32-twin calls, and are refused (VAEK_ERR_INVALID) for shapes only the register-staged kernels cover."""
    from vae_training_amd.conv import conv2d_forward, conv2d_transpose_forward, conv2d_weight_grad, to_bf16
    g = torch.Generator(device="cpu").manual_seed(3)
    B, H, Cin, Cout = 4, 16, 32, 64
    x = torch.randn(B, H, H, Cin, generator=g).cuda(); K = (torch.randn(4, 4, Cin, Cout, generator=g) * 0.05).cuda()
    m = torch.randn(B, H // 2, H // 2, Cout, generator=g).cuda()
    x16, m16 = to_bf16(x), to_bf16(m)
    y, y16 = conv2d_forward(x, K, None, True, mask=m, x16=x16, want16=True)
    none, z16 = conv2d_forward(None, K, None, True, mask16=m16, x16=x16, want16=True, want32=False)
    assert none is None and torch.equal(z16, y16)
    t, t16 = conv2d_transpose_forward(y, K, None, False, mask=x, y16=y16, want16=True)
    none, u16 = conv2d_transpose_forward(None, K, None, False, mask16=x16, y16=y16, want16=True, want32=False)
    assert none is None and torch.equal(u16, t16)
    dw, db = conv2d_weight_grad(x, y, x16=x16, dy16=y16)
    dw2, db2 = conv2d_weight_grad(None, None, x16=x16, dy16=y16)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    xs = torch.randn(2, 6, 6, 3, generator=g).cuda(); Ks = torch.randn(4, 4, 3, 5, generator=g).cuda()      # register-staged shapes only
    with pytest.raises(Exception):
        conv2d_forward(None, Ks, None, False, x16=to_bf16(xs), want16=True, want32=False)


def test_transposed_layer_backward_from_the_same_three_kernels():
    """conv_t_bwd of the oracle (d input, d kernel, d bias of the transposed layer) assembled from the HIP kernels:
    d input = conv2d_forward(d out, K), d kernel = conv2d_weight_grad(x := d out, dy := input) in the [4, 4, C_out, C_in] layout."""
    from vae_training_amd.conv import conv2d_forward, conv2d_weight_grad
    rng = np.random.default_rng(5)
    r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    B, h, cin, cout = 4, 8, 16, 12
    y = r32(rng.standard_normal((B, h, h, cin))); K = r32(rng.standard_normal((4, 4, cout, cin)) / 8.0)
    dout = r32(rng.standard_normal((B, 2 * h, 2 * h, cout)))
    want_dy, want_dK, want_db = CO.conv_t_bwd(y, K, dout)
    got_dy = conv2d_forward(_dev(dout), _dev(K)).cpu().numpy().astype(np.float64)
    got_dK = conv2d_weight_grad(_dev(dout), _dev(y), want_bias=False)[0].cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got_dy - want_dy)) <= 1e-2 * np.max(np.abs(want_dy))
    assert got_dK.shape == want_dK.shape and np.max(np.abs(got_dK - want_dK)) <= 1e-2 * np.max(np.abs(want_dK))


def _conv_problem(cfg, B, seed=0):
    rng = np.random.default_rng(seed)
    r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    p = {k: r32(v) for k, v in CO.init_params(cfg, seed=seed + 1).items()}
    for k in p:
        if not k.endswith("kernel"):
            p[k] = r32(p[k] + 0.1 * rng.standard_normal(p[k].shape))
    return p, r32(rng.random((B, cfg.S, cfg.S, 1))), r32(rng.standard_normal((B, cfg.L))), r32(rng.standard_normal((B, cfg.S, cfg.S, 1)))


def _bf16_emulation_grads(cfg, p, x, z1, z2):
    """The same network in torch float64 on the CPU with every convolution product's OPERANDS rounded to bf16 where the HIP kernels
    round them (forward, input gradient and kernel gradient of both layer kinds; relu masks, biases, Dense layers and the ELBO
    unrounded): what the kernels compute up to the order of the float32 accumulation."""
    import math
    import torch.nn.functional as F
    rb = lambda t: t.to(torch.bfloat16).to(torch.float64)
    ident = lambda t: t

    class Conv(torch.autograd.Function):
        @staticmethod
        def forward(ctx, h, w, b):                      # h NCHW, w OIHW
            ctx.save_for_backward(h, w)
            r = ident if h.shape[1] == 1 and 256 % w.shape[0] == 0 else rb          # one input channel: the exact-f32 streaming kernels
            return F.conv2d(r(h), r(w), b, stride=2, padding=1)

        @staticmethod
        def backward(ctx, dy):
            h, w = ctx.saved_tensors
            r = ident if h.shape[1] == 1 and 256 % w.shape[0] == 0 else rb
            dh = F.conv_transpose2d(rb(dy), rb(w), None, stride=2, padding=1)
            dw = torch.nn.grad.conv2d_weight(r(h), w.shape, r(dy), stride=2, padding=1)
            return dh, dw, dy.sum(dim=(0, 2, 3))

    class ConvT(torch.autograd.Function):
        @staticmethod
        def forward(ctx, h, w, b):                      # w [C_in, C_out, 4, 4] (torch's transposed layout)
            ctx.save_for_backward(h, w)
            r = ident if w.shape[1] == 1 and w.shape[0] % 4 == 0 and w.shape[0] <= 256 else rb   # one output channel: exact f32
            return F.conv_transpose2d(r(h), r(w), b, stride=2, padding=1)

        @staticmethod
        def backward(ctx, dout):
            h, w = ctx.saved_tensors
            r = ident if w.shape[1] == 1 and 256 % w.shape[0] == 0 else rb
            dh = F.conv2d(r(dout), r(w), None, stride=2, padding=1)
            dw = torch.nn.grad.conv2d_weight(r(dout), w.shape, r(h), stride=2, padding=1)
            return dh, dw, dout.sum(dim=(0, 2, 3))

    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    xt, z1t, z2t = (torch.tensor(a, dtype=torch.float64) for a in (x, z1, z2))
    B = x.shape[0]
    h = xt.permute(0, 3, 1, 2)
    for i in range(4):
        h = torch.relu(Conv.apply(h, tp[f"Encoder/Conv{i}/kernel"].permute(3, 2, 0, 1), tp[f"Encoder/Conv{i}/bias"]))
    flat = h.permute(0, 2, 3, 1).reshape(B, -1)
    mu = flat @ tp["Encoder/FC/kernel"] + tp["Encoder/FC/bias"]
    lv = tp["epsilon_p"]
    samples = mu + torch.exp(lv / 2) * z1t
    d = torch.relu(samples @ tp["Decoder/FC/kernel"] + tp["Decoder/FC/bias"])
    h = d.reshape(B, cfg.S // 16, cfg.S // 16, cfg.widths[3]).permute(0, 3, 1, 2)
    for i in range(4):
        h = ConvT.apply(h, tp[f"Decoder/ConvT{i}/kernel"].permute(3, 2, 0, 1), tp[f"Decoder/ConvT{i}/bias"])
        if i < 3:
            h = torch.relu(h)
    eps = tp["epsilon"][0] * cfg.epsilon if cfg.tdv else torch.tensor(cfg.epsilon, dtype=torch.float64)
    x_hat = h.permute(0, 2, 3, 1) + z2t * torch.exp(eps / 2)
    dkl = -0.5 * torch.sum(1 + lv - torch.exp(lv) - mu ** 2, dim=-1)
    mse = (0.5 * ((x_hat - xt) ** 2).reshape(B, -1) / torch.exp(eps) + 0.5 * (math.log(2 * math.pi) + eps)).sum(dim=-1)
    loss = (dkl + mse).mean()
    loss.backward()
    return loss.item(), {k: t.grad.numpy() for k, t in tp.items()}


def _checked_conv_calls(monkeypatch, log):
    """Wrap the three convolution entry points conv_vae.py calls: every call is compared, on ITS OWN inputs, with torch float64 with
    the operands rounded to bf16 where the kernel rounds them (not at all for the one-channel streaming kernels)."""
    import torch.nn.functional as F
    import vae_training_amd.conv_vae as CV
    rb = lambda t: t.to(torch.bfloat16).to(torch.float64)
    ident = lambda t: t
    nchw = lambda t: t.double().cpu().permute(0, 3, 1, 2)
    oihw = lambda t: t.double().cpu().permute(3, 2, 0, 1)
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30))
    o_f, o_t, o_w = CV.conv2d_forward, CV.conv2d_transpose_forward, CV.conv2d_weight_grad

    def finish(ref, relu, mask):
        ref = ref.permute(0, 2, 3, 1)
        ref = torch.relu(ref) if relu else ref
        return ref if mask is None else ref * (mask.cpu() > 0)

    def same16(t, t16):              # a bf16 copy handed along must BE the bf16 rounding of the float32 tensor
        return t16 is None or torch.equal(t16, t.to(torch.bfloat16))

    def fwd(x, w, bias=None, relu=False, mask=None, out=None, **kw):
        res = o_f(x, w, bias, relu, mask, out, **kw)
        y, y16 = res if isinstance(res, tuple) else (res, None)
        r = ident if x.shape[3] == 1 and 256 % w.shape[3] == 0 else rb
        ref = F.conv2d(r(nchw(x)), r(oihw(w)), None if bias is None else bias.double().cpu(), stride=2, padding=1)
        log.append(("forward", tuple(x.shape), rel(y, finish(ref, relu, mask))))
        assert same16(x, kw.get("x16")) and same16(y, y16) and (y16 is not None) == bool(kw.get("want16"))
        return res

    def tfwd(y, w, bias=None, relu=False, mask=None, **kw):
        res = o_t(y, w, bias, relu, mask, **kw)
        out, out16 = res if isinstance(res, tuple) else (res, None)
        r = ident if w.shape[2] == 1 and w.shape[3] % 4 == 0 and w.shape[3] <= 256 else rb
        ref = F.conv_transpose2d(r(nchw(y)), r(oihw(w)), None if bias is None else bias.double().cpu(), stride=2, padding=1)
        log.append(("transposed", tuple(y.shape), rel(out, finish(ref, relu, mask))))
        assert same16(y, kw.get("y16")) and same16(out, out16) and (out16 is not None) == bool(kw.get("want16"))
        return res

    def wgrad(x, dy, want_bias=True, dw=None, db=None, **kw):
        dw, db = o_w(x, dy, want_bias, dw, db, **kw)
        r = ident if x.shape[3] == 1 and 256 % dy.shape[3] == 0 else rb
        ref = torch.nn.grad.conv2d_weight(r(nchw(x)), (dy.shape[3], x.shape[3], 4, 4), r(nchw(dy)), stride=2, padding=1).permute(2, 3, 1, 0)
        log.append(("kernel gradient", tuple(x.shape), rel(dw.reshape(ref.shape), ref)))
        if db is not None:
            log.append(("bias gradient", tuple(x.shape), rel(db, r(dy.double().cpu()).sum(dim=(0, 1, 2)))))      # a ones row of the same bf16 product
        assert same16(x, kw.get("x16")) and same16(dy, kw.get("dy16"))
        return dw, db

    monkeypatch.setattr(CV, "conv2d_forward", fwd)
    monkeypatch.setattr(CV, "conv2d_transpose_forward", tfwd)
    monkeypatch.setattr(CV, "conv2d_weight_grad", wgrad)


@pytest.mark.parametrize("size,widths,L,B,tdv", [(16, (4, 8, 8, 16), 5, 6, True), (32, (8, 16, 16, 32), 7, 4, False), (64, (4, 8, 8, 16), 8, 3, True),
                                                 (64, (32, 64, 128, 256), 32, 2, True),         # BASELINE config 5's own widths
                                                 (48, (8, 16, 16, 32), 6, 2, True)])            # 48 x 48: no power-of-two sizes -- the general forms
def test_conv_vae_loss_and_every_gradient_leaf(size, widths, L, B, tdv, monkeypatch):
    """The whole convolutional VAE -- forward, ELBO, backward assembled from the library's blocks.
    (i) Every one of the 23 convolution calls of the step, on the inputs it actually received, against torch float64 with the
    operands rounded to bf16 where that kernel rounds them: 2e-6 of the result's scale (this is the tight check: the kernels
    compute THAT up to the order of the float32 accumulation).
    (ii) The leaves against the same emulation run end to end: loose (0.1), because a 1e-7 difference in a layer's output flips
    bf16 roundings in the next one and the flips grow from layer to layer (perturbing the first kernel by 1e-7 moves the
    emulation's own first-layer gradient by 2e-2 at config 5's widths) -- and against the float64 oracle (the bf16 envelope)."""
    from vae_training_amd.conv_vae import ConvVAE
    cfg = CO.ConvConfig(size, widths, L, -1.5, tdv)
    p, x, z1, z2 = _conv_problem(cfg, B)
    loss, g = CO.loss_and_grad(cfg, p, x, z1, z2)
    eloss, eg = _bf16_emulation_grads(cfg, p, x, z1, z2)
    net = ConvVAE(B, size, widths, L, -1.5, tdv, lean=False)      # (the per-call emulation below reads the float32 twins of the hidden tensors)
    assert [n for n, _ in net.leaf_shapes()] == [n for n, _ in cfg.leaves()] and net.n_params == cfg.n_params()
    params, grads = net.new_flat(), net.new_flat()
    for name in net.leaves:
        net.view(params, name).copy_(_dev(p[name]))
    calls = []
    _checked_conv_calls(monkeypatch, calls)
    out4 = net.loss_and_grad(params, grads, _dev(x), _dev(z1), _dev(z2)).cpu().numpy().astype(np.float64)
    assert len([c for c in calls if c[0] != "bias gradient"]) == 23 and max(c[2] for c in calls) <= 2e-6, calls
    assert abs(out4[0] - eloss) <= 5e-5 * abs(eloss), (out4[0], eloss)
    assert abs(out4[0] - loss) <= 2e-3 * abs(loss), (out4[0], loss)
    for name in net.leaves:
        got = net.view(grads, name).cpu().numpy().astype(np.float64)
        assert got.shape == g[name].shape, name
        emu, want = eg[name], g[name]
        assert np.max(np.abs(got - emu)) <= 0.1 * (np.max(np.abs(emu)) + 1e-30), (name, np.max(np.abs(got - emu)), np.max(np.abs(emu)))
        assert np.sqrt(np.mean((got - want) ** 2)) <= 0.3 * (np.sqrt(np.mean(want ** 2)) + 1e-30), name


@pytest.mark.parametrize("size,widths,L,B", [(16, (8, 16, 16, 32), 6, 16), (32, (32, 64, 128, 256), 8, 32)])
def test_conv_vae_graph_replay_is_the_eager_step(size, widths, L, B):
    """ConvVAE.capture: the step as a hipGraph (the tunable eps read on the device, no host access) -- replays walk the same
    trajectory as eager steps, bit for bit.  The second case has config 5's widths: the lean model (bf16-only hidden tensors)."""
    from vae_training_amd.conv_vae import ConvVAE
    cfg = CO.ConvConfig(size, widths, L, -1.5, True)
    lr = 2e-3
    p, x, z1, z2 = _conv_problem(cfg, B, seed=5)
    runs = []
    for graph in (False, True):
        net = ConvVAE(B, size, widths, L, -1.5, True)
        assert net.lean == (widths[0] >= 32)
        params, grads, m, v = net.new_flat(), net.new_flat(), net.new_flat(), net.new_flat()
        for name in net.leaves:
            net.view(params, name).copy_(_dev(p[name]))
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        xs, z1s, z2s = _dev(x), _dev(z1), _dev(z2)
        losses = []
        if graph:
            replay, out4 = net.capture(params, grads, m, v, step, xs, z1s, z2s, lr, warmup=2)
            losses += [None, None]
            for _ in range(3):
                replay()
                losses.append(float(out4[0]))
        else:
            for _ in range(5):
                losses.append(float(net.train_step(params, grads, m, v, step, xs, z1s, z2s, lr)[0]))
        runs.append((losses, params.clone(), int(step[0])))
    (le, pe, se), (lg, pg, sg) = runs
    assert se == sg == 5 and le[2:] == lg[2:] and torch.equal(pe, pg)


def test_conv_vae_train_steps_reduce_the_loss_like_the_oracle():
    from oracle import elbo_oracle as O
    from vae_training_amd.conv_vae import ConvVAE
    cfg = CO.ConvConfig(16, (4, 8, 8, 16), 5, -1.5, True)
    B, lr = 8, 2e-3
    p, x, z1, z2 = _conv_problem(cfg, B, seed=3)
    net = ConvVAE(B, 16, (4, 8, 8, 16), 5, -1.5, True)
    params, grads, m, v = net.new_flat(), net.new_flat(), net.new_flat(), net.new_flat()
    for name in net.leaves:
        net.view(params, name).copy_(_dev(p[name]))
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    st = O.adam_init(p)
    got, want = [], []
    for k in range(6):
        loss, g = CO.loss_and_grad(cfg, p, x, z1, z2)
        p, st = O.adam_update(p, g, st, lr)
        want.append(loss)
        got.append(float(net.train_step(params, grads, m, v, step, _dev(x), _dev(z1), _dev(z2), lr)[0]))
    got, want = np.array(got), np.array(want)
    assert int(step.item()) == 6 and got[-1] < got[0]
    assert np.max(np.abs(got - want) / np.abs(want)) <= 1e-2, (got, want)


def test_conv_vae_at_the_benched_size_is_repeatable_and_shard_additive():
    """B = 4 096 rows per GPU with config 5's widths -- the size bench.py --workload C5 times; far beyond what the float64 oracle
    finishes in seconds, so the checks are size-independent properties: the gradient is bitwise repeatable, and two half-batch
    shards evaluated with the GLOBAL divisor (world = 2) sum to the full-batch gradient within 1e-4 of each leaf's max-abs
    (networks.py:97-98: the loss is a batch mean, so the gradient is additive over row shards)."""
    from vae_training_amd.conv_vae import ConvVAE
    B, S, widths, L = 4096, 64, (32, 64, 128, 256), 32
    g = torch.Generator(device="cpu").manual_seed(3)
    full = ConvVAE(B, S, widths, L, -3.0, True)
    params = full.new_flat()
    for name, (off, shape) in full.leaves.items():
        if name.endswith("kernel"):
            fan = 16 * shape[2] if "Conv" in name and "ConvT" not in name else (4 * shape[3] if "ConvT" in name else shape[0])
            full.view(params, name).copy_((torch.randn(*shape, generator=g) / math.sqrt(fan)).cuda())
        elif name == "epsilon":
            full.view(params, name).fill_(1.0)
    x = torch.rand(B, S, S, 1, generator=g).cuda(); z1 = torch.randn(B, L, generator=g).cuda(); z2 = torch.randn(B, S, S, 1, generator=g).cuda()
    g1, g2 = full.new_flat(), full.new_flat()
    o1 = full.loss_and_grad(params, g1, x, z1, z2).clone()
    o2 = full.loss_and_grad(params, g2, x, z1, z2).clone()
    torch.cuda.synchronize()
    assert torch.equal(g1, g2) and torch.equal(o1, o2) and bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    half = ConvVAE(B // 2, S, widths, L, -3.0, True, world=2)
    acc = torch.zeros_like(g1, dtype=torch.float64)
    loss = 0.0
    for w in range(2):
        s = slice(w * B // 2, (w + 1) * B // 2)
        gs = half.new_flat()
        o = half.loss_and_grad(params, gs, x[s].contiguous(), z1[s].contiguous(), z2[s].contiguous())
        acc += gs.double(); loss += float(o[0])
    assert abs(loss - float(o1[0])) <= 1e-5 * abs(float(o1[0]))
    worst = {}
    for name, (off, shape) in full.leaves.items():
        k = math.prod(shape)
        a, b = acc[off:off + k], g1[off:off + k].double()
        worst[name] = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    assert max(worst.values()) <= 1e-4, sorted(worst.items(), key=lambda kv: -kv[1])[:3]


def test_lean_conv_vae_is_the_float32_twin_model_bit_for_bit():
    """ConvVAE(lean=1) keeps the hidden activations and their gradients as bf16 ONLY (no float32 twin written, relu masks from the
    bf16 copies: vaek_conv2d_forward's lean forms).  Every product reads the same bf16 operands as with lean=False, so the loss and
    every gradient leaf must be BITWISE equal -- except the transposed layers' bias gradients, which become column sums of the bf16
    gradient instead of the float32 one (2^-9 rounding per term, random sign: 2e-3 of the leaf's max-abs).
    lean=2 (the default) also hands the one-channel ends' 32-channel images to the streaming kernels as bf16: a different rounding
    point, so that model is held to the bf16 envelope against lean=1 (loss 1e-3, every leaf 2e-2 of its max-abs)."""
    from vae_training_amd.conv_vae import ConvVAE
    B, S, widths, L = 64, 64, (32, 64, 128, 256), 32
    a, b = ConvVAE(B, S, widths, L, -3.0, True, lean=1), ConvVAE(B, S, widths, L, -3.0, True, lean=False)
    assert a.lean and not a.lean2 and not b.lean
    g = torch.Generator(device="cpu").manual_seed(5)
    params = a.new_flat()
    params.copy_((torch.randn(a.P, generator=g) * 0.05).cuda())
    for name, (off, shape) in a.leaves.items():                      # (padding floats stay zero)
        pass
    x = torch.rand(B, S, S, 1, generator=g).cuda(); z1 = torch.randn(B, L, generator=g).cuda(); z2 = torch.randn(B, S, S, 1, generator=g).cuda()
    ga, gb = a.new_flat(), b.new_flat()
    oa = a.loss_and_grad(params, ga, x, z1, z2).clone(); ob = b.loss_and_grad(params, gb, x, z1, z2).clone()
    torch.cuda.synchronize()
    assert torch.equal(oa[:3], ob[:3]), (oa, ob)
    for name in a.leaves:
        va, vb = a.view(ga, name), b.view(gb, name)
        if name.startswith("Decoder/ConvT") and name.endswith("/bias") and not name.startswith("Decoder/ConvT3"):
            assert float((va - vb).abs().max()) <= 2e-3 * float(vb.abs().max()) + 1e-30, name
        else:
            assert torch.equal(va, vb), name
    c = ConvVAE(B, S, widths, L, -3.0, True)                          # the default: lean level 2
    assert c.lean and c.lean2
    gc = c.new_flat()
    oc = c.loss_and_grad(params, gc, x, z1, z2).clone()
    torch.cuda.synchronize()
    assert abs(float(oc[0]) - float(oa[0])) <= 1e-3 * abs(float(oa[0])), (oc, oa)
    for name in a.leaves:
        va, vc = a.view(ga, name), c.view(gc, name)
        assert float((va - vc).abs().max()) <= 2e-2 * float(va.abs().max()) + 1e-30, (name, float((va - vc).abs().max()), float(va.abs().max()))
    assert torch.equal(oc, c.loss_and_grad(params, gc, x, z1, z2))    # and it is repeatable bit for bit
    # a lean model with widths the LDS-DMA kernels do not cover falls back to the float32 twins by itself
    assert not ConvVAE(8, 16, (4, 8, 8, 16), 5, -1.5, True).lean
