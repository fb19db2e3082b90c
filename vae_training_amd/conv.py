"""Convolution layers of the convolutional VAE (BASELINE config 5; no reference counterpart -- DESIGN.md 3.4).  Thin wrappers over
the C ABI; torch only holds the device memory."""
import ctypes as C

import torch

from . import _lib


def conv2d_forward(x, w, bias=None, relu=False):
    """4 x 4 / stride 2 / pad 1 convolution: x [B, H, W, Cin] float32 NHWC, w [4, 4, Cin, Cout] HWIO -> [B, H/2, W/2, Cout]
    (vaek_conv2d_forward: implicit GEMM on the bf16 matrix cores, float32 accumulation)."""
    lib = _lib.load()
    assert x.is_cuda and w.is_cuda and x.dtype == w.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous()
    B, H, W, Cin = x.shape
    assert tuple(w.shape[:3]) == (4, 4, Cin)
    Cout = w.shape[3]
    y = torch.empty(B, H // 2, W // 2, Cout, dtype=torch.float32, device=x.device)
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    _lib.check(lib.vaek_conv2d_forward(p(x), p(w), p(bias), p(y), B, H, W, Cin, Cout, int(bool(relu)),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y
