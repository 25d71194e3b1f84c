"""ctypes wrapper over ``oracle/_build/liboracle_native.so`` (C restatement of the
reference's two CUDA kernels).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_native.so")
_lib = None


def build() -> str:
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "native_ops.c")):
        subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        f = ctypes.c_float
        i64 = ctypes.c_int64
        p = ctypes.c_void_p
        _lib.oracle_fused_bias_act.argtypes = [p, p, p, p, ctypes.c_int, ctypes.c_int, f, f, i64, i64, i64]
        _lib.oracle_fused_bias_act.restype = None
        _lib.oracle_upfirdn2d.argtypes = [p, p, p] + [ctypes.c_int] * 13
        _lib.oracle_upfirdn2d.restype = None
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def fused_bias_act(x: np.ndarray, b, ref, act: int, grad: int, alpha: float, scale: float) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    step_b = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    b = None if b is None or b.size == 0 else np.ascontiguousarray(b, dtype=np.float32)
    ref = None if ref is None or ref.size == 0 else np.ascontiguousarray(ref, dtype=np.float32)
    lib().oracle_fused_bias_act(_ptr(out), _ptr(x), _ptr(b), _ptr(ref), act, grad, alpha, scale,
                                x.size, step_b, 0 if b is None else b.size)
    return out


def upfirdn2d(x: np.ndarray, k: np.ndarray, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0)) -> np.ndarray:
    """x [N,C,H,W]; up/down = (x, y); pad = (x0, x1, y0, y1)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    k = np.ascontiguousarray(k, dtype=np.float32)
    n, c, h, w = x.shape
    kh, kw = k.shape
    oh = (h * up[1] + pad[2] + pad[3] - kh) // down[1] + 1
    ow = (w * up[0] + pad[0] + pad[1] - kw) // down[0] + 1
    out = np.empty((n, c, oh, ow), dtype=np.float32)
    lib().oracle_upfirdn2d(_ptr(out), _ptr(x), _ptr(k), n * c, h, w, kh, kw, up[0], up[1], down[0], down[1], *pad)
    return out
