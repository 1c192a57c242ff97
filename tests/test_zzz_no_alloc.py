"""gc_conv2d_wgrad_noalloc (the entry point that allocates nothing: split-K partial sums in a scratch array that is part
of the library's device image) against gc_conv2d_wgrad with a caller-provided workspace, through the C ABI: bit-identical
weight and bias gradients on two calls in a row, the second with a smaller input (the scratch keeps the first call's
contents), for the LDS-DMA and the register-staged weight-gradient kernels."""
import ctypes

import pytest
import torch

from gan_amd import _lib

pytestmark = pytest.mark.gpu


def _desc(N, H, Cin, Cout, k, s, p, dt):
    d = _lib.ConvDesc()
    Ho = (H + 2 * p - k) // s + 1
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, H, Cin, Ho, Ho, Cout
    d.kh = d.kw = k
    d.stride_h = d.stride_w = s
    d.pad_h = d.pad_w = p
    d.x_dtype = d.y_dtype = dt
    d.precision = _lib.GC_PREC_BF16
    return d, Ho


def _both(N, H, Cin, Cout, k, s, p, dtype, seed):
    L = _lib.lib()
    torch.manual_seed(seed)
    dt = _lib.GC_BF16 if dtype == torch.bfloat16 else _lib.GC_F32
    d, Ho = _desc(N, H, Cin, Cout, k, s, p, dt)
    x = torch.randn(N, H, H, Cin, device="cuda").to(dtype)            # NHWC
    gy = torch.randn(N, Ho, Ho, Cout, device="cuda").to(dtype)
    strides = (ctypes.c_int64 * 4)(Cin * k * k, k * k, k, 1)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    need = L.gc_wgrad_workspace_bytes(ctypes.byref(d))
    assert 0 < need <= (32 << 20)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    out = []
    for noalloc in (False, True):
        gw = torch.full((Cout, Cin, k, k), float("nan"), device="cuda")
        gb = torch.full((Cout,), float("nan"), device="cuda")
        args = [ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(gy.data_ptr()), ctypes.c_void_p(gw.data_ptr()), strides,
                0.5, 0, ctypes.c_void_p(gb.data_ptr())]
        if noalloc:
            rc = L.gc_conv2d_wgrad_noalloc(*args, stream)
        else:
            rc = L.gc_conv2d_wgrad(*args, ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream)
        assert rc == 0, L.gc_last_error()
        torch.cuda.synchronize()
        out.append((gw, gb))
    (gw_a, gb_a), (gw_b, gb_b) = out
    assert torch.isfinite(gw_a).all() and torch.isfinite(gb_a).all()
    assert torch.equal(gw_a, gw_b)                      # same kernels, fixed summation order: bit for bit
    assert torch.allclose(gb_a, gb_b, rtol=1e-5, atol=2e-3)     # the bias sums are fp32 atomic adds of a few thousand N(0, 1) values: the order varies
    return float(gw_a.abs().sum())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16_lds_dma", "f32_register_staged"])
def test_no_alloc_weight_gradient_matches_workspace_form(dtype):
    big = _both(16, 32, 64, 128, 4, 2, 1, dtype, 1)      # several splits: partial sums of a few MiB in the scratch
    small = _both(3, 12, 40, 24, 3, 1, 1, dtype, 2)      # right behind it, a smaller problem on the same scratch
    again = _both(16, 32, 64, 128, 4, 2, 1, dtype, 1)
    assert big == again and small > 0


def test_no_alloc_accumulates_like_the_workspace_form():
    L = _lib.lib()
    torch.manual_seed(3)
    d, Ho = _desc(4, 16, 128, 64, 4, 2, 1, _lib.GC_BF16)
    x = torch.randn(4, 16, 16, 128, device="cuda").bfloat16()
    gy = torch.randn(4, Ho, Ho, 64, device="cuda").bfloat16()
    strides = (ctypes.c_int64 * 4)(128 * 16, 16, 4, 1)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    base = torch.randn(64, 128, 4, 4, device="cuda")
    ws = torch.empty(L.gc_wgrad_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device="cuda")
    a, b = base.clone(), base.clone()
    assert L.gc_conv2d_wgrad(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(gy.data_ptr()), ctypes.c_void_p(a.data_ptr()),
                             strides, 1.0, 1, None, ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream) == 0
    assert L.gc_conv2d_wgrad_noalloc(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(gy.data_ptr()),
                                     ctypes.c_void_p(b.data_ptr()), strides, 1.0, 1, None, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b) and not torch.equal(a, base)
