"""torch.optim.Adam as one fused launch (plus a one-wavefront counter tick) per step.

Drop-in for the reference's `torch.optim.Adam(net.parameters(), lr=..., betas=...)` (BasicGANs/train.py:79-80,
Pix2Pix/train.py, StyleGAN/train.py, BigGAN/train.py): same constructor arguments for the features the reference uses
(lr, betas, eps; weight_decay / amsgrad / maximize must stay at their defaults), same arithmetic in the same order,
same state layout as torch.optim.Adam(capturable=True): per parameter `step` (a float32 scalar on the parameter's
device), `exp_avg`, `exp_avg_sq`.  A parameter that gets its first gradient late (the higher-resolution blocks of a
progressively grown StyleGAN) starts its bias correction at step 1, as in torch.  The step counts live on the device,
so step() can be recorded into a HIP graph.  state_dict() / load_state_dict() interchange with torch.optim.Adam
(StyleGAN/train.py:150-151 restores both optimizers from a checkpoint): a loaded `step` may be a Python number, a CPU
tensor (torch's default) or a device tensor.

`Adam.average_into(averaged_net, net, weight)` folds the reference's `moving_average(averaged_G, net_G, weight)`
(StyleGAN/train.py:71-76, called after every generator step) into the step's launch; `moving_average` below is the
stand-alone form with the reference's signature (two multi-tensor launches)."""
import ctypes

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, maximize=False, **unused):
        if weight_decay != 0 or amsgrad or maximize:
            raise ValueError("gan_amd.optim.Adam supports weight_decay=0, amsgrad=False, maximize=False only")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameter")
        # capturable=True: torch's state_dict machinery then keeps `step` as a float32 tensor on the parameter's device
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, capturable=True))
        self._tables = {}
        self._nets = []
        self._ema = {}          # id(parameter) -> its moving average (a tensor of another module)
        self._ema_weight = 0.999

    def average_into(self, averaged, net, weight=0.999):
        """Every later step() also updates `averaged`'s parameters as weight * averaged + (1 - weight) * updated parameter
        of `net`, matched by name (parameters of `net` that this optimizer does not step, or that got no gradient in a
        step, are not averaged by it: use moving_average for those)."""
        if not 0.0 <= weight <= 1.0:
            raise ValueError("weight must lie in [0, 1]")
        avg = dict(averaged.named_parameters())
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        for k, p in net.named_parameters():
            if id(p) not in mine:
                continue
            e = avg[k]
            if e.shape != p.shape or e.dtype != torch.float32 or e.device != p.device or not e.is_contiguous():
                raise RuntimeError(f"gan_amd.optim.Adam.average_into: {k} of the averaged module does not match the parameter")
            self._ema[id(p)] = e
        self._ema_weight = float(weight)
        self._tables.clear()
        return self

    def repack_after_step(self, *nets):
        """After every step, refresh the packed weight images of the gan_amd convolution modules of `nets` in one launch
        (otherwise each convolution re-packs its own weight when it next runs: two small launches per layer and step)."""
        self._nets = list(nets)
        return self

    def _init_state(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        step = st.get("step")
        if not (torch.is_tensor(step) and step.device == p.device and step.dtype == torch.float32 and step.numel() == 1):
            # absent, a Python number, or a tensor somewhere else (a checkpoint of torch.optim.Adam read to the CPU)
            value = 0.0 if step is None else float(step)
            st["step"] = torch.full((), value, dtype=torch.float32, device=p.device)
        return st

    def _table(self, gi, params):
        """Host arrays of device pointers for the parameters that have a gradient; rebuilt when anything moved."""
        ema = [self._ema.get(id(p)) for p in params]
        st = [self.state[p] for p in params]
        sig = tuple((p.data_ptr(), p.grad.data_ptr(), p.numel(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(), s["step"].data_ptr(),
                     0 if e is None else e.data_ptr()) for p, s, e in zip(params, st, ema))
        ent = self._tables.get(gi)
        if ent is not None and ent[0] == sig:
            return ent[1]
        n = len(params)
        vp, i64 = ctypes.c_void_p * n, ctypes.c_int64 * n
        arrays = (n, vp(*[p.data_ptr() for p in params]), vp(*[p.grad.data_ptr() for p in params]),
                  vp(*[s["exp_avg"].data_ptr() for s in st]), vp(*[s["exp_avg_sq"].data_ptr() for s in st]),
                  i64(*[p.numel() for p in params]), vp(*[s["step"].data_ptr() for s in st]),
                  vp(*[None if e is None else e.data_ptr() for e in ema]) if any(e is not None for e in ema) else None)
        self._tables[gi] = (sig, arrays)
        return arrays

    def load_state_dict(self, state_dict):
        """torch.optim.Optimizer.load_state_dict, then the state put where the kernel needs it: moments on the parameter's
        device in float32, one float32 device scalar `step` per parameter (whatever form the checkpoint held it in), and
        the cached pointer tables dropped (the load replaces every state tensor)."""
        super().load_state_dict(state_dict)
        self._tables.clear()
        for group in self.param_groups:
            group["capturable"] = True
            legacy = group.pop("step", None)          # one counter per group: the form an earlier version of this class wrote
            for p in group["params"]:
                st = self.state.get(p)
                if not st:
                    continue
                if "step" not in st and legacy is not None:
                    st["step"] = float(legacy)
                for k in ("exp_avg", "exp_avg_sq"):
                    if k in st and (st[k].device != p.device or st[k].dtype != torch.float32 or not st[k].is_contiguous()):
                        st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
                self._init_state(p)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            for p in params:
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                    raise RuntimeError("gan_amd.optim.Adam: parameters and gradients must be dense float32 tensors on the GPU")
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("gan_amd.optim.Adam: parameters and gradients must be contiguous")
                self._init_state(p)
            n, pp, gp, mp, vp, ne, sp, ep = self._table(gi, params)
            b1, b2 = group["betas"]
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            if ep is None:
                _lib.check(_lib.lib().gc_adam_step(n, pp, gp, mp, vp, ne, sp, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                   stream), "gc_adam_step")
            else:
                _lib.check(_lib.lib().gc_adam_ema_step(n, pp, gp, mp, vp, ep, self._ema_weight, ne, sp, float(group["lr"]), float(b1),
                                                       float(b2), float(group["eps"]), stream), "gc_adam_ema_step")
                for p in params:
                    e = self._ema.get(id(p))
                    if e is not None:
                        torch.autograd.graph.increment_version(e)
            for p in params:     # the kernel wrote through raw pointers: tell autograd (and every cache keyed on the
                torch.autograd.graph.increment_version(p)      # version, e.g. packed weight images) that p changed
        if self._nets:
            keys = {m.__dict__["_ganconv_wkey"] for net in self._nets for m in net.modules() if "_ganconv_wkey" in m.__dict__}
            if keys:
                from . import ops
                ops.repack(keys)
        return loss


@torch.no_grad()
def moving_average(model_1, model_2, weight=0.999):
    """model_1's parameters <- weight * model_1's + (1 - weight) * model_2's, matched by name (StyleGAN/train.py:71-76) as
    two multi-tensor launches instead of two per parameter."""
    p1, p2 = dict(model_1.named_parameters()), dict(model_2.named_parameters())
    a = [p1[k].data for k in p1]
    b = [p2[k].data for k in p1]
    if a:
        torch._foreach_mul_(a, weight)
        torch._foreach_add_(a, b, alpha=1 - weight)
