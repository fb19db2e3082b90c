"""Convolution layers of the convolutional VAE (BASELINE config 5; no reference counterpart -- DESIGN.md 3.4).  Thin wrappers over
the C ABI; torch only holds the device memory."""
import ctypes as C

import torch

from . import _lib


def _forward_workspace(lib, B, H, W, Cin, Cout, transposed, device, fast):
    if not fast:
        return None
    nbytes = C.c_size_t()
    _lib.check(lib.vaek_conv2d_forward_workspace(B, H, W, Cin, Cout, int(transposed), C.byref(nbytes)))
    return torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=device) if nbytes.value else None


def conv2d_forward(x, w, bias=None, relu=False, mask=None, out=None, fast=True):
    """4 x 4 / stride 2 / pad 1 convolution: x [B, H, W, Cin] float32 NHWC, w [4, 4, Cin, Cout] HWIO -> [B, H/2, W/2, Cout]
    (vaek_conv2d_forward: implicit GEMM on the bf16 matrix cores, float32 accumulation)."""
    lib = _lib.load()
    assert x.is_cuda and w.is_cuda and x.dtype == w.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous()
    B, H, W, Cin = x.shape
    assert tuple(w.shape[:3]) == (4, 4, Cin)
    Cout = w.shape[3]
    y = torch.empty(B, H // 2, W // 2, Cout, dtype=torch.float32, device=x.device) if out is None else out
    assert mask is None or (mask.shape == y.shape and mask.is_contiguous())
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    ws = _forward_workspace(lib, B, H, W, Cin, Cout, False, x.device, fast)      # fast=False: the register-staged kernel
    _lib.check(lib.vaek_conv2d_forward(p(x), p(w), p(bias), p(mask), p(y), B, H, W, Cin, Cout, int(bool(relu)), p(ws),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y


def conv2d_transpose_forward(y, w, bias=None, relu=False, mask=None, fast=True):
    """The adjoint of conv2d_forward with the same kernel array: y [B, h, w, Cin], w [4, 4, Cout, Cin] -> [B, 2 h, 2 w, Cout]
    (vaek_conv2d_transpose_forward).  With bias=None it is the convolution's input gradient; `mask` applies the relu of the layer
    below ([mask > 0])."""
    lib = _lib.load()
    assert y.is_cuda and w.is_cuda and y.dtype == w.dtype == torch.float32 and y.is_contiguous() and w.is_contiguous()
    B, h, wd, Cin = y.shape
    assert tuple(w.shape[:2]) == (4, 4) and w.shape[3] == Cin
    Cout = w.shape[2]
    out = torch.empty(B, 2 * h, 2 * wd, Cout, dtype=torch.float32, device=y.device)
    assert mask is None or (mask.shape == out.shape and mask.is_contiguous())
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    ws = _forward_workspace(lib, B, h, wd, Cin, Cout, True, y.device, fast)
    _lib.check(lib.vaek_conv2d_transpose_forward(p(y), p(w), p(bias), p(mask), p(out), B, h, wd, Cin, Cout, int(bool(relu)), p(ws),
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def conv2d_weight_grad(x, dy, want_bias=True, dw=None, db=None):
    """Kernel (and bias) gradient of conv2d_forward: x [B, H, W, Cin], dy [B, H/2, W/2, Cout] -> (dw [4, 4, Cin, Cout], db [Cout] or None)
    (vaek_conv2d_weight_grad: batch-split implicit GEMM + fixed-order slab sum)."""
    lib = _lib.load()
    assert x.is_cuda and dy.is_cuda and x.dtype == dy.dtype == torch.float32 and x.is_contiguous() and dy.is_contiguous()
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    assert tuple(dy.shape[:3]) == (B, H // 2, W // 2)
    nbytes = C.c_size_t()
    _lib.check(lib.vaek_conv2d_weight_grad_workspace(B, H, W, Cin, Cout, C.byref(nbytes)))
    ws = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=x.device)
    dw = torch.empty(4, 4, Cin, Cout, dtype=torch.float32, device=x.device) if dw is None else dw
    if want_bias and db is None:
        db = torch.empty(Cout, dtype=torch.float32, device=x.device)
    assert dw.is_contiguous() and dw.numel() == 16 * Cin * Cout
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    _lib.check(lib.vaek_conv2d_weight_grad(p(x), p(dy), p(dw), p(db), p(ws), B, H, W, Cin, Cout,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return dw, db


def conv2d_bias_grad(dy, db=None):
    """db[c] = sum over the pixels of dy [..., c] (vaek_conv2d_bias_grad: the bias gradient of a transposed layer)."""
    lib = _lib.load()
    assert dy.is_cuda and dy.dtype == torch.float32 and dy.is_contiguous()
    Cc = dy.shape[-1]
    pixels = dy.numel() // Cc
    ws = torch.empty(512 * Cc, dtype=torch.float32, device=dy.device)
    db = torch.empty(Cc, dtype=torch.float32, device=dy.device) if db is None else db
    _lib.check(lib.vaek_conv2d_bias_grad(C.c_void_p(dy.data_ptr()), C.c_void_p(db.data_ptr()), C.c_void_p(ws.data_ptr()), pixels, Cc,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return db
