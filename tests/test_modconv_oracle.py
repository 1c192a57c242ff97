"""CPU checks of the modulated-convolution restatement (oracle/ref_modconv.py): an extension of the reference's operator
set, parity unpinned (the reference has no such layer) -- so the oracle is checked against a second formulation of the
published formula and against the algebraic form the kernels use."""
import torch

from oracle import ref_modconv


def _case(seed, N=3, Cin=8, Cout=12, H=9, W=7, k=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64)
    s = torch.randn(N, Cin, generator=g, dtype=torch.float64) + 1.0
    return x, w, s


def test_grouped_form_equals_per_sample_loops():
    for seed, (k, stride, pad, demod) in enumerate([(3, 1, 1, True), (3, 1, 1, False), (1, 1, 0, True), (4, 2, 1, True)]):
        x, w, s = _case(seed, k=k)
        a = ref_modconv.modulated_conv2d(x, w, s, stride, pad, demod, weight_scale=0.37)
        b = ref_modconv.modulated_conv2d_loops(x, w, s, stride, pad, demod, weight_scale=0.37)
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)


def test_scaled_input_and_output_form_equals_per_sample_weights():
    """conv(x * s) * d with d[n, o] = rsqrt(sum_i s[n, i]^2 sum_k w[o, i, k]^2 + eps): what the kernels compute."""
    x, w, s = _case(7)
    eps = 1e-8
    y = torch.nn.functional.conv2d(x * s.view(3, 8, 1, 1), w, None, 1, 1)
    d = torch.rsqrt(s.square() @ w.square().sum((2, 3)).t() + eps)
    assert torch.allclose(y * d.view(3, 12, 1, 1), ref_modconv.modulated_conv2d(x, w, s, 1, 1, True, eps), rtol=1e-11, atol=1e-11)


def test_demodulated_weights_have_unit_norm():
    x, w, s = _case(11)
    wm = w.unsqueeze(0) * s.view(3, 1, 8, 1, 1)
    wm = wm * torch.rsqrt(wm.square().sum((2, 3, 4), keepdim=True) + 1e-8)
    assert torch.allclose(wm.square().sum((2, 3, 4)), torch.ones(3, 12, dtype=torch.float64), atol=1e-6)
