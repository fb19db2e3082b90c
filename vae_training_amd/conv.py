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


def _bf16_like(t):
    return torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)


def to_bf16(t):
    """bf16 copy of a contiguous float32 device tensor (vaek_to_bf16) -- the operand copy the LDS-DMA kernels read."""
    lib = _lib.load()
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    out = _bf16_like(t)
    _lib.check(lib.vaek_to_bf16(C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), t.numel(),
                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def _mask_arg(mask, mask16, shape):
    """(pointer source, relu-flag bit) of a relu mask given as the float32 tensor or as its bf16 copy (the lean form)."""
    assert mask is None or mask16 is None
    m = mask if mask16 is None else mask16
    assert m is None or (tuple(m.shape) == tuple(shape) and m.is_contiguous())
    assert mask16 is None or mask16.dtype == torch.bfloat16
    return m, (2 if mask16 is not None else 0)


def conv2d_forward(x, w, bias=None, relu=False, mask=None, out=None, fast=True, x16=None, want16=False, mask16=None, want32=True):
    """4 x 4 / stride 2 / pad 1 convolution: x [B, H, W, Cin] float32 NHWC, w [4, 4, Cin, Cout] HWIO -> [B, H/2, W/2, Cout]
    (vaek_conv2d_forward: implicit GEMM on the bf16 matrix cores, float32 accumulation; a streaming f32 kernel for Cin = 1).
    x16: a bf16 copy of x from an earlier call (skips the conversion pass); want16: also return the bf16 copy of the result.
    Lean forms (the LDS-DMA shapes): x=None beside x16; mask16 = the bf16 copy of the mask source instead of mask; want32=False:
    no float32 result (the first element of the returned pair is None)."""
    lib = _lib.load()
    src = x if x is not None else x16
    assert src.is_cuda and w.is_cuda and w.dtype == torch.float32 and src.is_contiguous() and w.is_contiguous()
    assert x is None or x.dtype == torch.float32
    B, H, W, Cin = src.shape
    assert tuple(w.shape[:3]) == (4, 4, Cin)
    assert x16 is None or (x16.dtype == torch.bfloat16 and tuple(x16.shape) == (B, H, W, Cin) and x16.is_contiguous())
    assert want32 or want16
    Cout = w.shape[3]
    y = (torch.empty(B, H // 2, W // 2, Cout, dtype=torch.float32, device=src.device) if out is None else out) if want32 else None
    y16 = torch.empty(B, H // 2, W // 2, Cout, dtype=torch.bfloat16, device=src.device) if want16 else None
    m, mbit = _mask_arg(mask, mask16, (B, H // 2, W // 2, Cout))
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    ws = _forward_workspace(lib, B, H, W, Cin, Cout, False, src.device, fast)      # fast=False: the register-staged kernel
    _lib.check(lib.vaek_conv2d_forward(p(x), p(w), p(bias), p(m), p(y), B, H, W, Cin, Cout, int(bool(relu)) | mbit, p(ws), p(x16), p(y16),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return (y, y16) if want16 else y


def conv2d_transpose_forward(y, w, bias=None, relu=False, mask=None, fast=True, y16=None, want16=False, mask16=None, want32=True):
    """The adjoint of conv2d_forward with the same kernel array: y [B, h, w, Cin], w [4, 4, Cout, Cin] -> [B, 2 h, 2 w, Cout]
    (vaek_conv2d_transpose_forward).  With bias=None it is the convolution's input gradient; `mask` applies the relu of the layer
    below ([mask > 0]).  y16 / want16 / mask16 / want32 / y=None: as in conv2d_forward."""
    lib = _lib.load()
    src = y if y is not None else y16
    assert src.is_cuda and w.is_cuda and w.dtype == torch.float32 and src.is_contiguous() and w.is_contiguous()
    assert y is None or y.dtype == torch.float32
    B, h, wd, Cin = src.shape
    assert tuple(w.shape[:2]) == (4, 4) and w.shape[3] == Cin
    assert y16 is None or (y16.dtype == torch.bfloat16 and tuple(y16.shape) == (B, h, wd, Cin) and y16.is_contiguous())
    assert want32 or want16
    Cout = w.shape[2]
    out = torch.empty(B, 2 * h, 2 * wd, Cout, dtype=torch.float32, device=src.device) if want32 else None
    out16 = torch.empty(B, 2 * h, 2 * wd, Cout, dtype=torch.bfloat16, device=src.device) if want16 else None
    m, mbit = _mask_arg(mask, mask16, (B, 2 * h, 2 * wd, Cout))
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    ws = _forward_workspace(lib, B, h, wd, Cin, Cout, True, src.device, fast)
    _lib.check(lib.vaek_conv2d_transpose_forward(p(y), p(w), p(bias), p(m), p(out), B, h, wd, Cin, Cout, int(bool(relu)) | mbit, p(ws), p(y16),
                                                 p(out16), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return (out, out16) if want16 else out


def conv2d_weight_grad(x, dy, want_bias=True, dw=None, db=None, x16=None, dy16=None):
    """Kernel (and bias) gradient of conv2d_forward: x [B, H, W, Cin], dy [B, H/2, W/2, Cout] -> (dw [4, 4, Cin, Cout], db [Cout] or None)
    (vaek_conv2d_weight_grad: batch-split implicit GEMM + fixed-order slab sum).  x16 / dy16: bf16 copies from earlier calls."""
    lib = _lib.load()
    xs, dys = (x if x is not None else x16), (dy if dy is not None else dy16)      # (lean form: x / dy None beside their bf16 copies)
    assert xs.is_cuda and dys.is_cuda and xs.is_contiguous() and dys.is_contiguous()
    assert (x is None or x.dtype == torch.float32) and (dy is None or dy.dtype == torch.float32)
    B, H, W, Cin = xs.shape
    Cout = dys.shape[3]
    assert tuple(dys.shape[:3]) == (B, H // 2, W // 2)
    assert x16 is None or (x16.dtype == torch.bfloat16 and tuple(x16.shape) == tuple(xs.shape) and x16.is_contiguous())
    assert dy16 is None or (dy16.dtype == torch.bfloat16 and tuple(dy16.shape) == tuple(dys.shape) and dy16.is_contiguous())
    nbytes = C.c_size_t()
    _lib.check(lib.vaek_conv2d_weight_grad_workspace(B, H, W, Cin, Cout, C.byref(nbytes)))
    ws = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=xs.device)
    dw = torch.empty(4, 4, Cin, Cout, dtype=torch.float32, device=xs.device) if dw is None else dw
    if want_bias and db is None:
        db = torch.empty(Cout, dtype=torch.float32, device=xs.device)
    assert dw.is_contiguous() and dw.numel() == 16 * Cin * Cout
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    _lib.check(lib.vaek_conv2d_weight_grad(p(x), p(dy), p(dw), p(db), p(ws), B, H, W, Cin, Cout, p(x16), p(dy16),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return dw, db


def conv2d_bias_grad(dy, db=None):
    """db[c] = sum over the pixels of dy [..., c] (vaek_conv2d_bias_grad: the bias gradient of a transposed layer)."""
    lib = _lib.load()
    assert dy.is_cuda and dy.dtype in (torch.float32, torch.bfloat16) and dy.is_contiguous()
    Cc = dy.shape[-1]
    pixels = dy.numel() // Cc
    ws = torch.empty(512 * Cc, dtype=torch.float32, device=dy.device)
    db = torch.empty(Cc, dtype=torch.float32, device=dy.device) if db is None else db
    fn = lib.vaek_conv2d_bias_grad if dy.dtype == torch.float32 else lib.vaek_conv2d_bias_grad_bf16      # (bf16: the sums of the copy's values)
    _lib.check(fn(C.c_void_p(dy.data_ptr()), C.c_void_p(db.data_ptr()), C.c_void_p(ws.data_ptr()), pixels, Cc,
                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return db
