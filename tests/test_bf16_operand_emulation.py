"""oracle/bf16_operands.py (the checker's restatement of the library's bf16 mode: operands of every convolution kernel
rounded to bf16 at every derivative order) against the explicit formulas on the CPU."""
import torch
import torch.nn.functional as F

from oracle import bf16_operands as E


def r(t):
    return t.to(torch.bfloat16).to(t.dtype)


def test_first_order_matches_explicit_rounding():
    torch.manual_seed(3)
    x = torch.randn(2, 5, 9, 7, requires_grad=True)
    w = torch.randn(6, 5, 3, 3, requires_grad=True)
    b = torch.randn(6, requires_grad=True)
    gy = torch.randn(2, 6, 5, 4)
    with E.conv_operands_rounded():
        y = F.conv2d(x, w, b, 2, 1)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), gy)
    assert torch.equal(y, F.conv2d(r(x), r(w), None, 2, 1) + b.view(1, -1, 1, 1))
    assert torch.equal(gx, torch.nn.grad.conv2d_input(x.shape, r(w), r(gy), 2, 1))
    assert torch.equal(gw, torch.nn.grad.conv2d_weight(r(x), w.shape, r(gy), 2, 1))
    assert torch.equal(gb, gy.sum((0, 2, 3)))          # the bias gradient sums the unrounded gradient, in fp32


def test_transposed_and_module_forms_are_patched_and_restored():
    torch.manual_seed(4)
    m = torch.nn.ConvTranspose2d(4, 3, 4, 2, 1)
    c = torch.nn.Conv2d(3, 8, 3, padding=1, groups=1)
    dw = torch.nn.Conv2d(8, 8, 3, padding=1, groups=8, bias=False)       # depthwise: left alone
    x = torch.randn(2, 4, 5, 5)
    with E.conv_operands_rounded():
        y = dw(c(m(x)))
    t = F.conv_transpose2d(r(x), r(m.weight), None, 2, 1) + m.bias.view(1, -1, 1, 1)
    t = F.conv2d(r(t), r(c.weight), None, 1, 1) + c.bias.view(1, -1, 1, 1)
    t = F.conv2d(t, dw.weight, None, 1, 1, 1, 8)
    assert torch.equal(y, t)
    assert torch.equal(c(m(x)), F.conv2d(F.conv_transpose2d(x, m.weight, m.bias, 2, 1), c.weight, c.bias, 1, 1))   # restored


def test_second_order_rounds_the_incoming_vector():
    """grad of <dgrad(gy, w), u> with respect to gy is fprop(u, w): the kernels round u like any other operand."""
    torch.manual_seed(5)
    x = torch.randn(1, 3, 6, 6, requires_grad=True)
    w = torch.randn(4, 3, 3, 3, requires_grad=True)
    with E.conv_operands_rounded():
        y = F.conv2d(x, w, None, 1, 1)
    gy = torch.randn_like(y).requires_grad_(True)
    gx, = torch.autograd.grad(y, x, gy, create_graph=True)
    u = torch.randn_like(gx)
    g_gy, g_w = torch.autograd.grad(gx, (gy, w), u)
    # The host library may add the products of these two in another order than the explicit formula does (one fp32
    # rounding step, 1e-6); an operand left unrounded would show at 1e-2.
    torch.testing.assert_close(g_gy, F.conv2d(r(u), r(w), None, 1, 1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g_w, torch.nn.grad.conv2d_weight(r(u), w.shape, r(gy.detach()), 1, 1),
                               rtol=1e-5, atol=1e-5)
    unrounded = torch.nn.grad.conv2d_weight(u, w.shape, gy.detach(), 1, 1)
    assert (g_w - unrounded).abs().max() > 1e-3
