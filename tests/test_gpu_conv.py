"""-m gpu: the convolution kernels of the convolutional VAE (BASELINE config 5; no reference counterpart -- DESIGN.md 3.4) against
the float64 oracle's layer functions (oracle/conv_vae_oracle.py) on float32-rounded inputs.  bf16 matrix-core products with
float32 accumulation: the bf16 Dense path's envelope (1e-2 of the output's max-abs; measured ~3e-3), not the 1e-5 contract."""
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


def test_conv2d_forward_rejects_odd_sizes():
    from vae_training_amd._lib import VaekError
    from vae_training_amd.conv import conv2d_forward
    with pytest.raises(VaekError):
        conv2d_forward(torch.zeros(1, 5, 5, 4, device="cuda"), torch.zeros(4, 4, 4, 8, device="cuda"))
