"""CPU oracle package: TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

`conv_oracle.c` restates the convolution passes in plain C (double accumulation); this module builds it with
gcc if the shared object is missing and exposes numpy-facing wrappers.  `ref_models.py` restates the
reference's model graphs and step functions on stock torch CPU ops.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "conv_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-shared", "-o", so, src])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def conv_out_size(H, k, s, p, d=1):
    return (H + 2 * p - d * (k - 1) - 1) // s + 1


def conv2d_fprop(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    x, w = _f32(x), _f32(w)
    bias = _f32(bias) if bias is not None else None
    N, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    Ho, Wo = conv_out_size(H, kh, stride, padding, dilation), conv_out_size(W, kw, stride, padding, dilation)
    y = np.empty((N, Cout, Ho, Wo), np.float32)
    lib().oracle_conv2d_fprop(_p(x), _p(w), _p(bias), _p(y), N, Cin, H, W, Cout, kh, kw, stride, stride, padding, padding,
                              dilation, dilation, groups, Ho, Wo)
    return y


def conv2d_dgrad(gy, w, x_shape, stride=1, padding=0, dilation=1, groups=1):
    gy, w = _f32(gy), _f32(w)
    N, Cin, H, W = x_shape
    Cout, _, kh, kw = w.shape
    Ho, Wo = gy.shape[2], gy.shape[3]
    gx = np.empty((N, Cin, H, W), np.float32)
    lib().oracle_conv2d_dgrad(_p(gy), _p(w), _p(gx), N, Cin, H, W, Cout, kh, kw, stride, stride, padding, padding,
                              dilation, dilation, groups, Ho, Wo)
    return gx


def conv2d_wgrad(x, gy, w_shape, stride=1, padding=0, dilation=1, groups=1, with_bias=False):
    x, gy = _f32(x), _f32(gy)
    N, Cin, H, W = x.shape
    Cout, Cg, kh, kw = w_shape
    Ho, Wo = gy.shape[2], gy.shape[3]
    gw = np.empty((Cout, Cg, kh, kw), np.float32)
    gb = np.empty((Cout,), np.float32) if with_bias else None
    lib().oracle_conv2d_wgrad(_p(x), _p(gy), _p(gw), _p(gb), N, Cin, H, W, Cout, kh, kw, stride, stride, padding, padding,
                              dilation, dilation, groups, Ho, Wo)
    return (gw, gb) if with_bias else gw


def conv_transpose2d_fprop(x, w, bias=None, stride=1, padding=0, output_padding=0, dilation=1):
    x, w = _f32(x), _f32(w)
    bias = _f32(bias) if bias is not None else None
    N, Cin, H, W = x.shape
    _, Cout, kh, kw = w.shape
    Ho = (H - 1) * stride - 2 * padding + dilation * (kh - 1) + output_padding + 1
    Wo = (W - 1) * stride - 2 * padding + dilation * (kw - 1) + output_padding + 1
    y = np.empty((N, Cout, Ho, Wo), np.float32)
    lib().oracle_conv_transpose2d_fprop(_p(x), _p(w), _p(bias), _p(y), N, Cin, H, W, Cout, kh, kw, stride, stride,
                                        padding, padding, dilation, dilation, Ho, Wo)
    return y


def dwblur3x3(x):
    x = _f32(x)
    N, C, H, W = x.shape
    y = np.empty_like(x)
    lib().oracle_dwblur3x3(_p(x), _p(y), N, C, H, W)
    return y


def round_bf16(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().oracle_round_bf16(_p(x), _p(y), ctypes.c_size_t(x.size))
    return y
