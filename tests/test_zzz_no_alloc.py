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


def test_no_alloc_call_writes_the_scratch_array_and_nothing_past_its_need():
    """The array gc_noalloc_scratch names is the one the call uses: filled with a sentinel, its head holds the call's
    partial sums afterwards (and reproduces the weight gradient when summed over the splits) while the bytes past the
    call's need keep the sentinel."""
    L = _lib.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    ptr, size = ctypes.c_void_p(), ctypes.c_size_t()
    assert L.gc_noalloc_scratch(ctypes.byref(ptr), ctypes.byref(size)) == 0 and ptr.value and size.value == (32 << 20)
    torch.manual_seed(5)
    d, Ho = _desc(8, 16, 64, 64, 4, 2, 1, _lib.GC_BF16)
    need = L.gc_wgrad_workspace_bytes(ctypes.byref(d))
    assert 0 < need < size.value
    x = torch.randn(8, 16, 16, 64, device="cuda").bfloat16()
    gy = torch.randn(8, Ho, Ho, 64, device="cuda").bfloat16()
    gw = torch.empty(64, 64, 4, 4, device="cuda")
    strides = (ctypes.c_int64 * 4)(64 * 16, 16, 4, 1)
    torch.cuda.synchronize()
    assert hip.hipMemset(ptr, 0x5A, size.value) == 0
    assert L.gc_conv2d_wgrad_noalloc(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(gy.data_ptr()), ctypes.c_void_p(gw.data_ptr()),
                                     strides, 1.0, 0, None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    head = torch.empty(need, dtype=torch.uint8, device="cuda")
    tail = torch.empty(4096, dtype=torch.uint8, device="cuda")
    assert hip.hipMemcpy(ctypes.c_void_p(head.data_ptr()), ptr, need, 3) == 0                                  # device to device
    assert hip.hipMemcpy(ctypes.c_void_p(tail.data_ptr()), ctypes.c_void_p(ptr.value + size.value - 4096), 4096, 3) == 0
    torch.cuda.synchronize()
    assert bool((tail == 0x5A).all())
    # the need: weight partials [split][tap][Cout][Cin], then room for one bias partial row [Cout] per split (not written by
    # this call, which asks for no bias gradient: those bytes keep the sentinel)
    words = head.view(torch.float32)
    splits = words.numel() // (16 * 64 * 64 + 64)
    assert splits >= 1 and words.numel() == splits * (16 * 64 * 64 + 64)
    parts = words[:splits * 16 * 64 * 64].view(splits, 16, 64, 64)
    assert bool((head[splits * 16 * 64 * 64 * 4:] == 0x5A).all())
    assert bool(torch.isfinite(parts).all())
    assert torch.equal(parts.sum(0).permute(1, 2, 0).reshape(64, 64, 4, 4), gw) or torch.allclose(
        parts.sum(0).permute(1, 2, 0).reshape(64, 64, 4, 4), gw, rtol=1e-5, atol=1e-4)


def test_fp32_image_beside_bf16_gradient_is_refused_outside_the_image_kernel():
    """An fp32 x with a bf16 gy is converted by the image-layer kernel only.  With a gy that is not 16-byte aligned, or a gy
    channel slice, that kernel does not apply: the call must fail instead of reading the fp32 image as bf16."""
    L = _lib.lib()
    N, Cin, Cout = 4, 3, 32
    d = _lib.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, 64, 64, Cin, 32, 32, Cout
    d.kh = d.kw = 4; d.stride_h = d.stride_w = 2; d.pad_h = d.pad_w = 1
    d.x_dtype, d.y_dtype, d.precision = _lib.GC_F32, _lib.GC_BF16, _lib.GC_PREC_BF16
    x = torch.randn(N, 64, 64, Cin, device="cuda")
    gy_store = torch.randn(N * 32 * 32 * Cout + 64, device="cuda").bfloat16()
    gw = torch.empty(Cout, Cin, 4, 4, device="cuda")
    strides = (ctypes.c_int64 * 4)(Cin * 16, 16, 4, 1)
    ws = torch.empty(max(L.gc_wgrad_workspace_bytes(ctypes.byref(d)), 16), dtype=torch.uint8, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(gy_ptr):
        return L.gc_conv2d_wgrad(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(gy_ptr), ctypes.c_void_p(gw.data_ptr()), strides,
                                 1.0, 0, None, ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream)
    assert call(gy_store.data_ptr()) == 0, L.gc_last_error()             # aligned: the image-layer kernel takes it
    torch.cuda.synchronize()
    ref = torch.zeros(Cout, Cin, 4, 4, requires_grad=True)
    y = torch.nn.functional.conv2d(x.cpu().bfloat16().float().permute(0, 3, 1, 2), ref, None, 2, 1)
    want, = torch.autograd.grad(y, ref, gy_store[:N * 32 * 32 * Cout].float().cpu().view(N, 32, 32, Cout).permute(0, 3, 1, 2))
    assert torch.allclose(gw.cpu(), want, rtol=1e-4, atol=1e-3)
    assert call(gy_store.data_ptr() + 8) != 0                              # gy offset by 8 bytes
    assert b"fp32 x" in L.gc_last_error() or b"dtype" in L.gc_last_error()
    d.y_cstride, d.y_coff = Cout + 8, 4                                    # a gy channel slice at an unaligned offset
    assert call(gy_store.data_ptr()) != 0
    torch.cuda.synchronize()
