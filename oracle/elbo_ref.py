"""ctypes loader of oracle/elbo_ref.c (CPU port, C + OpenMP) -- TEST / BASELINE INFRASTRUCTURE.
Only tests/ and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libelbo_ref.so")


class Cfg(C.Structure):
    _fields_ = [("D", C.c_int), ("L", C.c_int), ("n_enc", C.c_int), ("enc", C.c_int * 8), ("n_dec", C.c_int), ("dec", C.c_int * 8),
                ("sigmoid", C.c_int), ("tdv", C.c_int), ("eps_cli", C.c_float)]


def load(native=False):
    """native=True: rebuild with -march=native into a temp dir on THIS host (bench.py on the GPU box)."""
    path = _LIB
    if native:
        import tempfile
        path = os.path.join(tempfile.gettempdir(), f"libelbo_ref_native_{os.getpid()}.so")
        try:
            subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", os.path.join(_HERE, "elbo_ref.c"), "-o", path, "-lm"],
                           check=True, capture_output=True)
        except Exception:
            path = _LIB
    if not os.path.exists(path):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    lib = C.CDLL(path)
    lib.elbo_ref_param_count.restype = C.c_long
    lib.elbo_ref_param_count.argtypes = [C.POINTER(Cfg)]
    lib.elbo_ref_step.restype = C.c_float
    fp = C.POINTER(C.c_float)
    lib.elbo_ref_step.argtypes = [C.POINTER(Cfg), fp, fp, fp, C.c_int, fp, fp, fp, C.c_int, C.c_long, C.c_float, fp, C.c_int, C.c_int]
    return lib


def make_cfg(D, L, enc=(), dec=(), eps=0.0, tdv=False, sigmoid=False):
    c = Cfg()
    c.D, c.L, c.n_enc, c.n_dec, c.sigmoid, c.tdv, c.eps_cli = D, L, len(enc), len(dec), int(sigmoid), int(tdv), eps
    for i, h in enumerate(enc):
        c.enc[i] = h
    for i, h in enumerate(dec):
        c.dec[i] = h
    return c


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def step(lib, cfg, params, m, v, t, x, z1, z2, lr, grads=None, apply=True, batch_total=0, nthreads=0):
    """params/m/v/x/z1/z2 (and grads, P+4): contiguous float32 NumPy arrays, updated in place."""
    return float(lib.elbo_ref_step(C.byref(cfg), _p(params), _p(m), _p(v), int(t), _p(x), _p(z1), _p(z2), x.shape[0], int(batch_total),
                                   float(lr), _p(grads) if grads is not None else None, int(apply), int(nthreads)))
