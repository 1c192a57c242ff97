"""Every convolution launch of a whole training step, checked on its own operands.

The end-to-end trace tests compare gradients that have passed through many ReLU / LeakyReLU kinks, where one activation
on the other side of zero moves a whole tensor by percents, so their bounds are loose.  Here the library's convolution
operators are wrapped for the duration of one training step of each model family (reduced width, the step logic and every
layer geometry / epilogue / channel slice / deferral of the full-size steps): each call -- forward, data gradient (plain
and with the activation derivative in its output stage), weight gradient (with and without the bias sum), in the first- and
second-order passes alike -- is recomputed on the CPU in float64 from the SAME operands the kernel read (rounded to bf16
where the kernel rounds them) and compared at the accuracy of fp32 accumulation.  A wrong tile, tap, mask or scale in any
single launch fails, whatever the rest of the network does with it."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_F32_OUT = 3e-5        # fp32 result from bf16-rounded operands: accumulation order only
TOL_BF16_OUT = 3.5e-3     # the same, stored in bf16 (half an ulp = 2^-9 per element)
TOL_X3 = 2e-4             # three-term split of fp32 operands


@pytest.fixture(scope="module")
def gan():
    import gan_amd
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return gan_amd


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


class Audit:
    """Wraps gan_amd.ops' convolution operators; `worst` collects the largest error per operator."""

    NAMES = ("conv_fprop", "conv_dgrad", "conv_dgrad_masked", "conv_wgrad", "conv_wgrad_bias")

    def __init__(self, gan):
        self.gan, self.ops, self.L = gan, gan.ops, gan._lib
        self.saved, self.calls, self.worst, self.failures = {}, {n: 0 for n in self.NAMES}, {n: 0.0 for n in self.NAMES}, []

    def __enter__(self):
        for n in self.NAMES:
            self.saved[n] = getattr(self.ops, n)
            setattr(self.ops, n, getattr(self, "_" + n))
        return self

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(self.ops, n, f)

    # ---- operand values as the kernels see them
    def _operand(self, t, precision):
        t = t.detach()
        if precision == self.L.GC_PREC_BF16:
            t = t.to(torch.bfloat16)
        return t.double().cpu()

    def _weight(self, w, alpha, precision, wkey=0):
        ptr = w.data_ptr()
        w = w.detach().float() * float(alpha)          # the pack kernel's product, in fp32
        dv = self.ops._divisors.get(wkey) if wkey else None
        if dv is not None and dv[0] == ptr:            # a spectrally normalised layer: the image holds weight / sigma
            w = w / dv[1][dv[2]]
        if precision == self.L.GC_PREC_BF16:
            w = w.to(torch.bfloat16)
        return w.double().cpu()

    @staticmethod
    def _act(y, act, slope):
        if act == 1:
            return torch.relu(y)
        if act == 2:
            return torch.nn.functional.leaky_relu(y, slope)
        if act == 3:
            return torch.tanh(y)
        if act == 4:
            return torch.sigmoid(y)
        return y

    def _judge(self, name, got, want, precision, what):
        tol = TOL_X3 if precision == self.L.GC_PREC_F32X3 else (TOL_BF16_OUT if got.dtype == torch.bfloat16 else TOL_F32_OUT)
        if want.numel() == 0:
            return
        scale = float(want.norm())
        err = _rel(got, want) if scale > 0 else float(got.double().abs().max())
        self.calls[name] += 1
        self.worst[name] = max(self.worst[name], err)
        if not err < tol:
            self.failures.append(f"{name} {what}: relative L2 error {err:.2e} (limit {tol:.0e})")

    def _sample(self, N, flops_per_sample):
        """Samples to recompute: all of them, or the first and the last when the launch is large."""
        return list(range(N)) if N * flops_per_sample < 6e9 else sorted({0, N - 1})

    def _conv_fprop(self, x, w, bias, stride, padding, precision, wkey, act=0, slope=0.0, out_dtype=-1, alpha=1.0, defer_act_grad=False,
                    x_act_slope=-1.0):
        y = self.saved["conv_fprop"](x, w, bias, stride, padding, precision, wkey, act, slope, out_dtype, alpha, defer_act_grad, x_act_slope)
        idx = self._sample(x.shape[0], 2.0 * y[0].numel() * w[0].numel())
        xr, wr = self._operand(x[idx], precision), self._weight(w, alpha, precision, wkey)
        ref = torch.nn.functional.conv2d(xr, wr, None if bias is None else bias.detach().double().cpu(), stride, padding)
        self._judge("conv_fprop", y[idx], self._act(ref, act, slope), precision, f"x {tuple(x.shape)} w {tuple(w.shape)} s{stride} p{padding} act {act}")
        return y

    def _dgrad_ref(self, gy, w, bias, H, W, stride, padding, precision, alpha, idx, wkey=0):
        kh = w.shape[2]
        op_h = H - ((gy.shape[2] - 1) * stride - 2 * padding + kh)
        op_w = W - ((gy.shape[3] - 1) * stride - 2 * padding + w.shape[3])
        gr, wr = self._operand(gy[idx], precision), self._weight(w, alpha, precision, wkey)
        return torch.nn.functional.conv_transpose2d(gr, wr, None if bias is None else bias.detach().double().cpu(), stride, padding,
                                                    output_padding=(op_h, op_w))

    def _conv_dgrad(self, gy, w, bias, H, W, stride, padding, precision, wkey, act=0, slope=0.0, out_dtype=-1, alpha=1.0):
        gx = self.saved["conv_dgrad"](gy, w, bias, H, W, stride, padding, precision, wkey, act, slope, out_dtype, alpha)
        idx = self._sample(gy.shape[0], 2.0 * gy[0].numel() * w[0].numel())
        ref = self._act(self._dgrad_ref(gy, w, bias, H, W, stride, padding, precision, alpha, idx, wkey), act, slope)
        self._judge("conv_dgrad", gx[idx], ref, precision, f"gy {tuple(gy.shape)} w {tuple(w.shape)} s{stride} p{padding} act {act}")
        return gx

    def _conv_dgrad_masked(self, gy, w, x, x_act_slope, stride, padding, precision, wkey, out_dtype=-1, alpha=1.0):
        gx = self.saved["conv_dgrad_masked"](gy, w, x, x_act_slope, stride, padding, precision, wkey, out_dtype, alpha)
        idx = self._sample(gy.shape[0], 2.0 * gy[0].numel() * w[0].numel())
        ref = self._dgrad_ref(gy, w, None, x.shape[2], x.shape[3], stride, padding, precision, alpha, idx, wkey)
        ref = ref * torch.where(x[idx].detach().double().cpu() > 0, 1.0, float(x_act_slope))
        self._judge("conv_dgrad_masked", gx[idx], ref, precision, f"gy {tuple(gy.shape)} w {tuple(w.shape)} s{stride} p{padding}")
        return gx

    def _wgrad_ref(self, x, gy, kh, kw, stride, padding, precision, alpha):
        xr, gr = self._operand(x, precision), self._operand(gy, precision)
        return float(alpha) * torch.nn.grad.conv2d_weight(xr, (gy.shape[1], x.shape[1], kh, kw), gr, stride=stride, padding=padding)

    def _conv_wgrad(self, x, gy, kh, kw, stride, padding, precision, alpha=1.0):
        gw = self.saved["conv_wgrad"](x, gy, kh, kw, stride, padding, precision, alpha)
        if 2.0 * gy.numel() * x.shape[1] * kh * kw < 4e10:
            self._judge("conv_wgrad", gw, self._wgrad_ref(x, gy, kh, kw, stride, padding, precision, alpha), precision,
                        f"x {tuple(x.shape)} gy {tuple(gy.shape)} k{kh} s{stride} p{padding}")
        return gw

    def _conv_wgrad_bias(self, x, gy, kh, kw, stride, padding, precision, alpha=1.0):
        gw, gb = self.saved["conv_wgrad_bias"](x, gy, kh, kw, stride, padding, precision, alpha)
        if 2.0 * gy.numel() * x.shape[1] * kh * kw < 4e10:
            self._judge("conv_wgrad_bias", gw, self._wgrad_ref(x, gy, kh, kw, stride, padding, precision, alpha), precision,
                        f"x {tuple(x.shape)} gy {tuple(gy.shape)} k{kh} s{stride} p{padding}")
        # a sum of signed terms: the error of fp32 accumulation scales with the sum of magnitudes, not with the (cancelling) total
        g64 = gy.detach().double().cpu()
        want, mag = g64.sum((0, 2, 3)), g64.abs().sum((0, 2, 3)).clamp_min(1e-30)
        err = float(((gb.detach().double().cpu() - want).abs() / mag).max())
        if not err < 2e-6:
            self.failures.append(f"conv_wgrad_bias bias sum gy {tuple(gy.shape)}: error {err:.2e} of the sum of magnitudes")
        return gw, gb


def _steps(gan):
    from gan_amd.harness import steps as S
    return {
        "dcgan": lambda: S.dcgan("cuda:0", batch=8),
        "pix2pix": lambda: S.pix2pix_unet("cuda:0", batch=2, ch=16, size=256),
        "stylegan": lambda: S.stylegan_v1("cuda:0", batch=2, pg=2),
        "biggan": lambda: S.biggan_128("cuda:0", batch=4, ch=8, n_classes=4),
    }, S


@pytest.mark.parametrize("family", ["dcgan", "pix2pix", "stylegan", "biggan"])
def test_every_convolution_launch_of_a_training_step(gan, family):
    """bf16 operands, bf16 activation storage (the benchmarked mode): one training step (both phases, optimizer steps)."""
    gan.set_precision("bf16")
    makers, S = _steps(gan)
    try:
        step = makers[family]()
        with Audit(gan) as audit:
            step()
            torch.cuda.synchronize()
    finally:
        S.reset_options()
    assert not audit.failures, "\n".join(audit.failures[:12]) + f"\n({len(audit.failures)} failing launches)"
    assert audit.calls["conv_fprop"] >= 8 and audit.calls["conv_dgrad"] + audit.calls["conv_dgrad_masked"] >= 4, audit.calls
    assert audit.calls["conv_wgrad"] + audit.calls["conv_wgrad_bias"] >= 4, audit.calls
    assert bool(torch.isfinite(step.losses).all())


def test_every_convolution_launch_of_a_wgan_gp_critic_iteration(gan):
    """The gradient penalty's double backward on the DCGAN critic (three-term fp32 arithmetic and bf16 arithmetic): the
    second-order launches (forward-like passes over gradients, weight gradients from two gradient tensors) are audited like
    the first-order ones."""
    from gan_amd.harness import basic_gans as B
    for prec in ("fp32x3", "bf16"):
        gan.set_precision(prec)
        torch.manual_seed(7)
        D = B.Discriminator(3, 16, 64, is_wgan=True).cuda()
        D.apply(B.weights_init)
        real = (torch.rand(6, 3, 64, 64, device="cuda") * 2 - 1)
        fake = (torch.rand(6, 3, 64, 64, device="cuda") * 2 - 1)
        eps = torch.rand(6, 1, 1, 1, device="cuda")
        with Audit(gan) as audit:
            mixed = (real * eps + fake * (1 - eps)).requires_grad_(True)
            score = D(mixed)
            grad, = torch.autograd.grad(score.sum(), mixed, create_graph=True)
            pen = 10.0 * ((grad.flatten(1).norm(2, dim=1) - 1) ** 2).mean()
            (D(real).mean() - D(fake).mean() + pen).backward()
            torch.cuda.synchronize()
        assert not audit.failures, prec + "\n" + "\n".join(audit.failures[:12])
        assert audit.calls["conv_fprop"] >= 15 and audit.calls["conv_wgrad"] + audit.calls["conv_wgrad_bias"] >= 10, audit.calls
    gan.set_precision("bf16")
