"""DCGAN / WGAN-GP generator, discriminator and training phases of the reference's BasicGANs family on the
gan_amd convolution modules.

Graphs and parameter names follow BasicGANs/models.py:6-86 (`main.{i}.0.weight` ...; each BatchNorm2d + activation
pair is one fused module that keeps BatchNorm2d's parameter and buffer names; the LeakyReLU / Tanh / Sigmoid that
directly follows a convolution runs in that convolution's epilogue), the phases follow
BasicGANs/loss_functions.py:4-76 and the step order BasicGANs/DCGAN.py:165-181 / BasicGANs/train.py:95-119.
Differences that do not change results: no host synchronisation inside a phase (losses are returned as
tensors, never .item()), and the WGAN-GP interpolation factor epsilon is an argument instead of a draw inside
the loss, so that a step can be replayed exactly (tests) and captured into a HIP graph (bench).
"""
import math

import torch

from .. import nn as gnn
from .. import ops as gops
from .. import optim as goptim


def _seq(*m):
    return torch.nn.Sequential(*m)


class Generator(torch.nn.Module):
    """activation_dtype: storage type of the activations between the first and the last layer (torch.float32, or
    torch.bfloat16 for half the memory traffic; the input noise and the image stay float32 either way)."""

    def __init__(self, z_dim, hidden_dim, image_dim=3, resolution=128, activation_dtype=torch.float32):
        super().__init__()
        n_layers = int(math.log2(resolution)) - 1
        width = 8 * hidden_dim
        layers = []
        for i in range(n_layers):
            first, last = i == 0, i == n_layers - 1
            cin = z_dim if first else int(width / 2 ** (i - 1))
            cout = image_dim if last else int(width / 2 ** i)
            conv = gnn.ConvTranspose2d(cin, cout, 4, 1 if first else 2, 0 if first else 1, bias=False,
                                       activation="tanh" if last else None,
                                       out_dtype=torch.float32 if last else (activation_dtype if first else None))
            layers.append(_seq(conv) if last else _seq(conv, gnn.BatchNormAct2d(cout, activation="relu")))
        self.main = _seq(*layers)
        if fused_batchnorm_statistics:
            gnn.fuse_batchnorm_statistics(self)      # the BatchNorm layers take their batch statistics from the convolutions' epilogues

    def forward(self, noise_vector):
        return self.main(noise_vector)


class Discriminator(torch.nn.Module):
    def __init__(self, image_dim, hidden_dim, resolution=128, is_wgan=False, activation_dtype=torch.float32, fuse_activation_grad=True):
        super().__init__()
        n_layers = int(math.log2(resolution)) - 1
        layers = []
        for i in range(n_layers):
            first, last = i == 0, i == n_layers - 1
            cin = image_dim if first else int(hidden_dim * 2 ** (i - 1))
            cout = 1 if last else int(hidden_dim * 2 ** i)
            fused = "leaky_relu" if first else ("sigmoid" if last and not is_wgan else None)
            # the first LeakyReLU's derivative is applied by the second convolution's data-gradient kernel
            ops = [gnn.Conv2d(cin, cout, 4, 1 if last else 2, 0 if last else 1, bias=False, activation=fused, negative_slope=0.2,
                              out_dtype=torch.float32 if last else (activation_dtype if first else None),
                              defer_activation_grad=first and fuse_activation_grad,
                              input_activation_slope=0.2 if (i == 1 and fuse_activation_grad) else None)]
            if first:
                pass
            elif not last and is_wgan:   # InstanceNorm2d + LeakyReLU(0.2) of the reference as one fused op (no parameters)
                ops.append(gnn.InstanceNormAct2d(cout, negative_slope=0.2))
            elif not last:
                ops.append(gnn.BatchNormAct2d(cout, activation="leaky_relu", negative_slope=0.2))
            layers.append(_seq(*ops))
        self.main = _seq(*layers)
        if fused_batchnorm_statistics:
            gnn.fuse_batchnorm_statistics(self)

    def forward(self, x):
        return self.main(x)


def weights_init(m):
    """BasicGANs/utils.py:31-36."""
    if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
        torch.nn.init.normal_(m.weight, 0.0, 0.02)
    if isinstance(m, (torch.nn.BatchNorm2d, gnn.BatchNormAct2d)):
        torch.nn.init.normal_(m.weight, 0.0, 0.02)
        torch.nn.init.constant_(m.bias, 0)


# ---- WGAN-GP phases (BasicGANs/loss_functions.py:4-38) --------------------------------------------------------
_side_streams = {}


fused_batchnorm_statistics = True     # networks built while this is set: see gan_amd.nn.fuse_batchnorm_statistics
early_weight_gradients = 1   # > 0: a layer's weight-gradient launch starts on one of this many streams as soon as its last contribution exists
flush_lanes = 2              # side streams that share the layers' weight-gradient launches with the calling stream
serialize_chains = False     # True: the chains run one after the other on the calling stream (same kernels; for per-kernel timing)


def _side_stream(device, i):
    if serialize_chains:
        return torch.cuda.current_stream(device)
    st = _side_streams.get((device, i))
    if st is None:
        st = _side_streams[(device, i)] = torch.cuda.Stream(device)
    return st


# Which scalar parts of a critic iteration run as single launches of the library on a GPU (the others, and everything
# on the CPU, as the reference's tensor expressions; narrowing the set is an A/B aid).
fused_scalars = {"mix", "ones", "penalty", "loss"}
# The streamed critic iteration collects its weight gradients through gan_amd.ops.weight_gradient_sink and reduces them
# with one launch per layer after the chains have joined; False: autograd returns and accumulates them (A/B aid).
weight_gradient_sink = True
# Measured on the recorded benchmark cycle (tools/ab_bench.py --what flag): which chain stays on the calling stream is
# neutral; recording the forked chains after the calling stream's chain costs 0.3 ms (the graph runtime's
# branch-to-queue assignment follows recording order); the penalty chain's weight-gradient kernels on a stream of
# their own (ops.weight_gradient_sink(fork=True)) cost 0.4 ms, a fork and an event wait per launch outweigh the overlap.
side_chains_first = True            # launch (and record) the forked chains before the calling stream's own chain
penalty_on_calling_stream = False   # True: the penalty chain on the calling stream, the real / fake chains fork
_ones = {}


def _ones_like(t):
    """A cached tensor of ones (the seed of d scores.sum() / d scores): no fill launch per iteration."""
    key = (tuple(t.shape), t.dtype, t.device)
    o = _ones.get(key)
    if o is None:
        o = _ones[key] = torch.ones_like(t)
    return o


def _penalty(net_D, real, fake, epsilon, c_lambda):
    """c_lambda * mean((||d D(mixed) / d mixed||_2 - 1)^2) with the graph of the double backward attached
    (BasicGANs/loss_functions.py:18-27).  Nothing reads the gradient with respect to epsilon that the reference's
    requires_grad=True asks for, so the fused interpolation does not provide it."""
    on = fused_scalars if real.is_cuda else ()
    if "mix" in on:
        mixed = gops.wgan_mix(real, fake, epsilon).requires_grad_(True)
    else:
        eps = epsilon.detach().clone().requires_grad_(True)
        mixed = real.detach() * eps + fake.detach() * (1 - eps)
    scores = net_D(mixed)
    with gops.input_gradients_only():       # the weight gradients of this pass would be thrown away
        if "ones" in on:
            gradient = torch.autograd.grad(scores, mixed, grad_outputs=_ones_like(scores), create_graph=True)[0]
        else:
            gradient = torch.autograd.grad(inputs=mixed, outputs=scores.sum(), create_graph=True)[0]
    if "penalty" in on:
        return gops.gradient_penalty(gradient, c_lambda)[0]
    return c_lambda * ((gradient.reshape(gradient.shape[0], -1).norm(2, dim=1) - 1) ** 2).mean()


def _real_fake_loss(d_both, B):
    """For the scores of B real samples followed by B generated ones: (D_x with its drift term, D_G_z1, the tensor to
    differentiate, its seed gradient or None): the two backward passes of BasicGANs/loss_functions.py:7-15 as one."""
    if d_both.is_cuda and "loss" in fused_scalars:
        stats, seed = gops.wgan_critic_loss(d_both.detach(), 0.001)
        return stats[0], stats[1], d_both, seed
    flat = d_both.view(-1)
    d_real = flat[:B].mean() - 0.001 * (flat[:B] ** 2).mean()
    d_fake = flat[B:].mean()
    return d_real.detach(), d_fake.detach(), d_fake - d_real, None


def _critic_streams(net_D, real, fake, epsilon, c_lambda, split_real_fake, pair=None):
    """The independent parts of a critic iteration on streams of their own: the gradient-penalty pass beside the real +
    fake pass (or beside the real pass and the fake pass, `split_real_fake`).  The mid-size layers of these networks
    leave a good part of the GPU idle on their own; two chains in flight at once measured 11 % less time per training
    cycle.  Each chain returns its parameter gradients; they are added on the main stream after the join, so nothing
    accumulates into one tensor from two streams."""
    params = [p for p in net_D.parameters() if p.requires_grad]
    main = torch.cuda.current_stream(real.device)
    B = real.shape[0]

    def grads_of(out, seed):
        """d out / d params as autograd returns them; inside the sink the convolution weights come back as None (their
        gradients are collected by the sink)."""
        if seed is None and out.is_cuda:            # (a cached one instead of the fill launch autograd would make per call)
            seed = _ones_like(out)
        with gops.parameter_gradients_only():      # (no data gradient for the mixed images, a leaf that requires grad)
            return torch.autograd.grad(out, params, grad_outputs=seed, allow_unused=weight_gradient_sink)

    def penalty_chain():
        penalty = _penalty(net_D, real, fake, epsilon, c_lambda)
        return grads_of(penalty, None), penalty.detach()

    def real_chain():
        d_real = net_D(real).view(-1)
        d_real = d_real.mean() - 0.001 * (d_real ** 2).mean()
        return grads_of(-d_real, None), d_real.detach()

    def fake_chain():
        d_fake = net_D(fake).view(-1).mean()
        return grads_of(d_fake, None), d_fake.detach()

    def both_chain():
        # pair: real || generated as one tensor that already exists (the data side wrote `real` into its first half, the
        # generator its images into the second): no concatenation launch
        d_real, d_fake, out, seed = _real_fake_loss(net_D(pair if pair is not None else torch.cat([real, fake])), B)
        return grads_of(out, seed), (d_real, d_fake)

    rf_chains = [real_chain, fake_chain] if split_real_fake else [both_chain]
    side_chains = rf_chains if penalty_on_calling_stream else [penalty_chain] + rf_chains[1:]
    results = []
    fork = torch.cuda.Event()
    fork.record(main)

    def run_side_chains():
        with gops.suspended_backward_overlap():
            for i, chain in enumerate(side_chains):
                st = _side_stream(real.device, i)
                st.wait_event(fork)
                with torch.cuda.stream(st):
                    results.append(chain())

    def run_chains():
        if side_chains_first:
            run_side_chains()
        if penalty_on_calling_stream:
            head = penalty_chain()
        else:
            with gops.suspended_backward_overlap():
                head = rf_chains[0]()
        if not side_chains_first:
            run_side_chains()
        results.insert(0 if penalty_on_calling_stream else 1, head)
        for i in range(len(side_chains)):      # (each fork waits for the main stream again, so a side stream does not
            main.wait_stream(_side_stream(real.device, i))     # reuse its blocks before the additions below have run)

    # results: penalty chain, then the real (+ fake) chain, then the fake chain if it is separate
    if weight_gradient_sink:
        # Every weight-gradient contribution of the iteration (per layer: the real + fake pass and the two of the
        # penalty's double backward) is collected and, after the chains have joined, reduced by ONE launch per layer
        # over the pixels of all of them: a third of the weight-gradient launches and split reductions, no additions.
        bufs = [torch.empty_like(p, dtype=torch.float32, memory_format=torch.contiguous_format) for p in params]
        lanes = [] if serialize_chains else [_side_stream(real.device, i) for i in range(flush_lanes)]
        early = [] if (serialize_chains or not early_weight_gradients) else [_side_stream(real.device, 8 + i) for i in range(early_weight_gradients)]
        with gops.weight_gradient_sink(params, bufs, defer=True, flush_streams=lanes, expected=len(rf_chains) + 2,
                                       early_streams=early) as sink:
            run_chains()
        g_main, later = [], []      # later[k]: (sums, addends) of the parameters with more than k + 1 contributions
        for i, (p, b) in enumerate(zip(params, bufs)):
            rest = [r[0][i] for r in results if r[0][i] is not None]
            if p.data_ptr() in sink.touched:
                g = b
            elif rest:
                g = rest.pop(0)          # (autograd.grad's own result tensor: nobody else holds it)
            else:
                raise RuntimeError("a critic parameter received no gradient")
            for k, r in enumerate(rest):
                while len(later) <= k:
                    later.append(([], []))
                later[k][0].append(g)
                later[k][1].append(r)
            g_main.append(g)
        for sums, addends in later:      # the normalisation layers' gradients of all chains: one launch per chain, not per tensor
            torch._foreach_add_(sums, addends)
    else:
        run_chains()
        g_main = list(results[0][0])
        for r in results[1:]:
            torch._foreach_add_(g_main, list(r[0]))
    penalty = results[0][1]
    if split_real_fake:
        d_real, d_fake = results[1][1], results[2][1]
    else:
        d_real, d_fake = results[1][1]
    if all(p.grad is None for p in params):
        for p, g in zip(params, g_main):
            p.grad = g
    else:
        for p, g in zip(params, g_main):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        torch._foreach_add_([p.grad for p in params], g_main)
    return d_real, d_fake, penalty


def wgan_critic_phase(net_G, net_D, real, z, epsilon, c_lambda=10.0, on_last_backward=None, batch_real_fake=False,
                      two_streams=False, fake=None, pair=None):
    """Backward passes accumulate into net_D's gradients: real batch (with the 0.001 drift term), fake batch,
    gradient penalty (double backward).  `on_last_backward` is called right before the last one (a data-parallel
    gradient reducer arms its hooks there).  Returns (D_x, D_G_z1, penalty) as tensors.

    batch_real_fake: run the real and the fake batch through the critic as one batch of 2B samples and backpropagate
    the sum of the two losses once.  The critic has no cross-sample coupling (convolutions, InstanceNorm, LeakyReLU),
    so scores and accumulated gradients are those of the two separate passes up to summation order; each
    convolution then runs once at twice the size instead of twice."""
    if fake is None:         # (a caller may have run the generator ahead of time: its weights do not move in this phase)
        fake = net_G(z)
    fake = fake.detach()
    if two_streams and real.is_cuda:
        if on_last_backward is not None:
            raise RuntimeError("two_streams computes the gradients explicitly: no hook-driven reducer")
        return _critic_streams(net_D, real, fake, epsilon, c_lambda, not batch_real_fake, pair)
    if batch_real_fake:
        d_real, d_fake, out, seed = _real_fake_loss(net_D(pair if pair is not None else torch.cat([real, fake])), real.shape[0])
        torch.autograd.backward(out, seed)
    else:
        d_real = net_D(real).view(-1)
        d_real = d_real.mean() - 0.001 * (d_real ** 2).mean()
        (-d_real).backward()
        d_fake = net_D(fake).view(-1).mean()
        d_fake.backward()
    penalty = _penalty(net_D, real, fake, epsilon, c_lambda)
    if on_last_backward is not None:
        on_last_backward()
    with gops.parameter_gradients_only():      # (the mixed images are leaves that require grad; nobody reads their .grad)
        penalty.backward()
    return d_real.detach(), d_fake.detach(), penalty.detach()


def wgan_generator_phase(net_G, net_D, z, on_last_backward=None, fake=None):
    if fake is None:         # (a caller may have run the generator's forward pass ahead of time, graph attached)
        fake = net_G(z)
    d_fake = net_D(fake).view(-1)
    g_error = -d_fake.mean()
    if on_last_backward is not None:
        on_last_backward()
    g_error.backward(_ones_like(g_error) if g_error.is_cuda else None)
    return g_error.detach()


# ---- DCGAN phases (BasicGANs/loss_functions.py:41-76, step order BasicGANs/DCGAN.py:165-181) ---------------------
def dcgan_discriminator_phase(net_G, net_D, real, z, on_last_backward=None):
    bce = torch.nn.functional.binary_cross_entropy
    out_real = net_D(real).view(-1)
    err_real = bce(out_real, torch.ones_like(out_real))
    err_real.backward()
    fake = net_G(z)
    out_fake = net_D(fake.detach()).view(-1)
    err_fake = bce(out_fake, torch.zeros_like(out_fake))
    if on_last_backward is not None:
        on_last_backward()
    err_fake.backward()
    return (err_real + err_fake).detach(), out_fake.mean().detach(), out_real.mean().detach(), fake


def dcgan_generator_phase(net_D, fake, on_last_backward=None, freeze_critic=False):
    """`fake` still carries the generator graph built in the discriminator phase (the reference re-uses it).
    freeze_critic: the critic's parameters take no gradient in this phase.  The reference lets them accumulate one
    (BasicGANs/DCGAN.py:177-181 never freezes net_D; :168 discards it) and throws it away at the next net_D.zero_grad(); skipping it changes
    no parameter of either network."""
    bce = torch.nn.functional.binary_cross_entropy
    frozen = [p for p in net_D.parameters() if p.requires_grad] if freeze_critic else []
    for p in frozen:
        p.requires_grad_(False)
    try:
        out = net_D(fake).view(-1)
    finally:
        for p in frozen:
            p.requires_grad_(True)
    g_error = bce(out, torch.ones_like(out))
    if on_last_backward is not None:
        on_last_backward()
    g_error.backward()
    return g_error.detach(), out.mean().detach()


def synth_inputs(gen, B, nz, res, n_z):
    """Synthetic real batch U(-1,1) (stands in for Normalize(0.5, 0.5) images) and n_z latent batches N(0,1)
    drawn on the CPU from `gen`, so the same numbers exist on every machine."""
    real = torch.rand(B, 3, res, res, generator=gen) * 2 - 1
    zs = [torch.randn(B, nz, 1, 1, generator=gen) for _ in range(n_z)]
    return real, zs


class WganGpTrainer:
    """WGAN-GP 64x64 training cycle of BasicGANs/train.py: `disc_repeats` critic iterations, one generator
    iteration, Adam(lr 1e-4, betas (0, 0.9)) for both nets.  Inputs live in static device buffers so that the
    cycle can be captured into one HIP graph (no host work, no allocation, no synchronisation per cycle)."""

    def __init__(self, batch, hidden_dim=64, z_dim=100, resolution=64, device="cuda", lr=1e-4, betas=(0.0, 0.9),
                 c_lambda=10.0, disc_repeats=5, seed=5000, reducer_factory=None, capturable=True, batch_real_fake=False, fused_adam=False,
                 activation_dtype=torch.float32, two_streams=False):
        torch.manual_seed(seed)
        self.B, self.nz, self.res, self.reps, self.c_lambda = batch, z_dim, resolution, disc_repeats, c_lambda
        self.batch_real_fake = batch_real_fake
        self.two_streams = two_streams      # critic iteration: penalty chain and real + fake chain on two streams
        self.gen_ahead = two_streams        # with two_streams: the generator phase's forward pass runs ahead as well (-1.6 %)
        self.freeze_critic_in_generator_phase = False   # True: skip the critic's (discarded) weight gradients of that phase
        self.device = torch.device(device)
        self.net_G = Generator(z_dim, hidden_dim, 3, resolution, activation_dtype=activation_dtype)
        self.net_D = Discriminator(3, hidden_dim, resolution, is_wgan=True, activation_dtype=activation_dtype)
        self.net_G.apply(weights_init)
        self.net_D.apply(weights_init)
        self.net_G.to(self.device)
        self.net_D.to(self.device)
        if fused_adam and self.device.type == "cuda":     # one launch per step, device-side step counter (capturable)
            self.opt_D = goptim.Adam(self.net_D.parameters(), lr=lr, betas=betas).repack_after_step(self.net_D)
            self.opt_G = goptim.Adam(self.net_G.parameters(), lr=lr, betas=betas).repack_after_step(self.net_G)
        else:
            cap = capturable and self.device.type == "cuda"
            self.opt_D = torch.optim.Adam(self.net_D.parameters(), lr=lr, betas=betas, capturable=cap, foreach=True)
            self.opt_G = torch.optim.Adam(self.net_G.parameters(), lr=lr, betas=betas, capturable=cap, foreach=True)
        cl = torch.channels_last
        self.real = torch.zeros(batch, 3, resolution, resolution, device=self.device).contiguous(memory_format=cl)
        self.z = torch.zeros(disc_repeats + 1, batch, z_dim, 1, 1, device=self.device)
        self.eps = torch.zeros(disc_repeats, batch, 1, 1, 1, device=self.device)
        # real || generated batches of the critic iterations: `real` sits in the first half (set_inputs), the generator's
        # forward passes that run ahead write their images into the second half (ops.output_into)
        self.pairs = None
        if batch_real_fake and two_streams and self.device.type == "cuda":
            self.pairs = [torch.zeros(2 * batch, 3, resolution, resolution, device=self.device).contiguous(memory_format=cl)
                          for _ in range(disc_repeats)]
        self.red_D = reducer_factory(self.net_D) if reducer_factory else None
        self.red_G = reducer_factory(self.net_G) if reducer_factory else None
        self.graph = None
        self.last = None

    def set_inputs(self, real, zs, eps):
        self.real.copy_(real)
        for p in self.pairs or []:
            p[:self.B].copy_(self.real)
        self.z.copy_(torch.stack(list(zs)))
        self.eps.copy_(eps)

    def _zero(self, net, red):
        if red is not None:
            red.zero_grad()
        else:
            for p in net.parameters():
                p.grad = None

    def _fakes_ahead(self, gen_too=False):
        """The generator's forward passes of all critic iterations of a cycle, on a stream of their own beside the first
        critic iterations (the generator does not change before its own step at the end of the cycle).  Returns one
        (images, event) pair per iteration."""
        main = torch.cuda.current_stream(self.device)
        st = _side_stream(self.device, "generator")
        st.wait_stream(main)
        fakes = []
        with torch.cuda.stream(st):
            for it in range(self.reps):
                if self.pairs is not None:
                    with gops.output_into(self.pairs[it][self.B:]):
                        f = self.net_G(self.z[it]).detach()
                else:
                    f = self.net_G(self.z[it]).detach()
                ev = torch.cuda.Event()
                ev.record(st)
                fakes.append((f, ev))
            if gen_too:             # the generator phase's own forward pass too (graph kept for its backward pass)
                f = self.net_G(self.z[self.reps])
                ev = torch.cuda.Event()
                ev.record(st)
                fakes.append((f, ev))
        return fakes

    def _fake(self, fakes, it, wait=True):
        if fakes is None:
            return None
        f, ev = fakes[it]
        fakes[it] = None
        if wait:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return f

    def _pair(self, it, fake):
        """The real || generated tensor of critic iteration `it` if the generator's images landed in its second half."""
        if self.pairs is None or fake is None or fake.data_ptr() != self.pairs[it][self.B:].data_ptr():
            return None
        return self.pairs[it]

    def _generator_phase(self, arm, fake=None):
        """The generator phase of the cycle.  The critic's weight gradients of this pass -- added to the critic's
        gradients as the reference adds them (BasicGANs/loss_functions.py:31-36), read by nobody -- are no link of the
        backward chain: each is launched on a stream of its own as soon as its operands exist."""
        weights = [p for p in self.net_D.parameters()
                   if p.requires_grad and p.dim() == 4 and p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous()]
        if self.device.type != "cuda" or not weights or not weight_gradient_sink:
            return wgan_generator_phase(self.net_G, self.net_D, self.z[self.reps], arm, fake)
        early = [] if (serialize_chains or not early_weight_gradients) else [_side_stream(self.device, 8)]
        with gops.weight_gradient_sink(weights, [p.grad for p in weights], defer=True, expected=1, early_streams=early) as sink:
            sink.touched.update(p.data_ptr() for p in weights)      # the launches add to the gradients that are there
            return wgan_generator_phase(self.net_G, self.net_D, self.z[self.reps], arm, fake)

    def cycle_eager(self):
        out = []
        fakes = self._fakes_ahead(self.gen_ahead) if self.two_streams and self.device.type == "cuda" else None
        for it in range(self.reps):
            self._zero(self.net_D, self.red_D)
            arm = self.red_D.arm if self.red_D is not None else None
            fake = self._fake(fakes, it)
            d_x, d_g_z1, gp = wgan_critic_phase(self.net_G, self.net_D, self.real, self.z[it], self.eps[it], self.c_lambda,
                                                None if self.two_streams else arm, self.batch_real_fake, self.two_streams,
                                                fake, self._pair(it, fake))
            if self.red_D is not None:
                self.red_D.finish()
            self.opt_D.step()
            out += [d_x, d_g_z1, gp]
        self._zero(self.net_G, self.red_G)
        # The reference leaves the critic trainable in the generator phase (BasicGANs/loss_functions.py:31-36): its weight
        # gradients of this pass are computed and then thrown away by the next optimizer_D.zero_grad() (BasicGANs/train.py:96).
        # The measured cycle computes them as well; freeze_critic_in_generator_phase skips them (no parameter changes).
        frozen = list(self.net_D.parameters()) if self.freeze_critic_in_generator_phase else []
        for p in frozen:
            p.requires_grad_(False)
        arm = self.red_G.arm if self.red_G is not None else None
        if fakes is not None and self.gen_ahead:
            # the generator's backward pass runs on the stream of its forward pass, a forked one: no further fork from there
            with gops.suspended_backward_overlap():
                g_err = self._generator_phase(arm, self._fake(fakes, self.reps))
        else:
            g_err = self._generator_phase(arm)
        for p in frozen:
            p.requires_grad_(True)
        if self.red_G is not None:
            self.red_G.finish()
        self.opt_G.step()
        out.append(g_err)
        gops.clear_cast_cache()
        self.last = torch.stack([o.reshape(()) for o in out])
        return self.last

    def capture(self, warmup=3):
        """Warm up on a side stream, then record the cycle into HIP graphs.  Without gradient reducers: one graph for
        the whole cycle.  With reducers (data parallel): one graph per forward / backward phase, each opening with the
        averaging + Adam step of the phase before it; the gradient buckets are all-reduced between two replays, so no
        collective is ever captured."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.cycle_eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # images packed during the warm-up must not be hit inside the capture: their pack kernels would be missing
        # from the graph and replays would keep reading the weights of the moment of recording
        gops.clear_weight_cache()
        gops.clear_cast_cache()
        if self.red_D is None:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph), gops.capturing_on():
                self.static_out = self.cycle_eager()
            return self.graph
        self.graphs, outs, pool = [], [], None

        def record(fn):
            nonlocal pool
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool), gops.capturing_on():
                out = fn()
            pool = pool or g.pool()
            return g, out

        # Graph boundaries only where a collective goes: the optimizer step behind an all-reduce opens the graph of the next
        # forward / backward (7 replays per cycle instead of 12).
        fakes = None
        step_D = lambda: (self.red_D.average_(), self.opt_D.step(), gops.clear_cast_cache())
        for it in range(self.reps):
            def crit(it=it):
                nonlocal fakes
                if it > 0:
                    step_D()
                self.red_D.zero_grad()
                if it == 0 and self.two_streams:        # (the images stay allocated in the graphs' pool until their graph is recorded)
                    fakes = self._fakes_ahead()
                # the first graph holds all generator passes and ends after them: later graphs find the images ready
                fake = self._fake(fakes, it, wait=(it == 0))
                out = wgan_critic_phase(self.net_G, self.net_D, self.real, self.z[it], self.eps[it], self.c_lambda, None,
                                        self.batch_real_fake, self.two_streams, fake, self._pair(it, fake))
                if it == 0 and self.two_streams:
                    torch.cuda.current_stream(self.device).wait_stream(_side_stream(self.device, "generator"))
                return out
            g_fb, out = record(crit)
            self.graphs.append((g_fb, self.red_D))
            outs += list(out)

        def gen():
            step_D()
            self.red_G.zero_grad()
            # (the critic's parameters stay trainable, as in the reference: its gradients of this phase land in its buckets
            #  and are zeroed at the start of the next critic iteration)
            return self._generator_phase(None)
        g_fb, out = record(gen)
        g_opt, _ = record(lambda: (self.red_G.average_(), self.opt_G.step(), gops.clear_cast_cache()))
        self.graphs += [(g_fb, self.red_G), (g_opt, None)]
        outs.append(out)
        self.static_out = torch.stack([o.reshape(()) for o in outs])   # (a view-free copy made once; refreshed in cycle())
        self._outs = outs
        self.graph = self.graphs[0][0]
        return self.graphs

    def cycle(self):
        if getattr(self, "graphs", None):
            for g, red in self.graphs:
                g.replay()
                if red is not None:
                    red.all_reduce()
            self.static_out = torch.stack([o.reshape(()) for o in self._outs])
            return self.static_out
        if self.graph is not None:
            self.graph.replay()
            return self.static_out
        return self.cycle_eager()
