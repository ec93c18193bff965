"""Flat-buffer parameters, fused clip+AdamW+EMA, and the EMA object of the reference API.

Reference: losses.py:30-62 (get_optimizer / optimization_manager), models/ema.py:10-97.
MI355X design: all trainable parameters of BOTH networks are views into one flat fp32 buffer and
all their gradients are views into one flat gradient buffer, so
  * the data-parallel exchange is ONE RCCL all-reduce over xGMI per step (parallel.py),
  * zero_grad is one memset; the gradients autograd hands out are gathered into the flat buffer by ONE
    multi-tensor copy (indm_multi_copy_f32) instead of one accumulate kernel per parameter,
  * clip_grad_norm_ + AdamW + EMA is one read-modify-write pass per network
    (indm_adamw_ema_f32; the global norm comes from indm_sqnorm_f32 and stays on the device).
Parameters that never receive a gradient in the reference (iResBlock.lamb / geom_p, which are only
read through .item(); GaussianFourierProjection.W has requires_grad=False) are kept out of the flat
buffers: torch's AdamW skips them because their .grad is None.
"""
import torch

from . import hip

_NO_GRAD_SUFFIXES = (".lamb", ".geom_p")


def _receives_grad(name, p):
    return p.requires_grad and not name.endswith(_NO_GRAD_SUFFIXES)


class FlatParams:
    """Re-homes a module's trainable parameters (and grads) into slices of flat buffers."""

    def __init__(self, module, flat_p=None, flat_g=None, offset=0):
        self.module = module
        self.named = [(n, p) for n, p in module.named_parameters() if _receives_grad(n, p)]
        self.other = [(n, p) for n, p in module.named_parameters() if p.requires_grad and not _receives_grad(n, p)]
        self.numel = sum(_pad4(p.numel()) for _, p in self.named)
        dev = self.named[0][1].device
        own = flat_p is None
        if own:
            flat_p = torch.zeros(self.numel, dtype=torch.float32, device=dev)
            flat_g = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.p = flat_p[offset:offset + self.numel]
        self.g = flat_g[offset:offset + self.numel]
        o = 0
        self.gslices = []   # (parameter, its slice of the flat gradient buffer, that slice's address)
        for _, p in self.named:
            n = p.numel()
            self.p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.p[o:o + n].view(p.shape)
            p.grad = self.g[o:o + n].view(p.shape)
            self.gslices.append((p, self.g[o:o + n], self.g[o:o + n].data_ptr()))
            o += _pad4(n)

    def zero_grad(self):
        """One memset of the flat buffer; the parameters' .grad go back to None so that autograd ASSIGNS the first
        gradient of each parameter (no accumulate kernel per parameter); gather_grads() collects them."""
        self.g.zero_()
        for _, p in self.named:
            p.grad = None
        self.gathered = False

    def gather_grads(self):
        """Copy every parameter's .grad into its slice of the flat gradient buffer: ONE multi-tensor launch.
        Idempotent until the next zero_grad() (the all-reduce and the optimiser then work on the flat buffer)."""
        if getattr(self, "gathered", False):
            return
        srcs, dsts = [], []
        for p, dst, addr in self.gslices:
            g = p.grad
            if g is not None and g.data_ptr() != addr:
                srcs.append(g if g.is_contiguous() else g.contiguous())
                dsts.append(dst)
        hip.multi_copy(srcs, dsts)
        self.gathered = True

    def adopt_flat_grads(self):
        """After gather_grads(): point EVERY parameter's .grad at its slice of the flat buffer, so that a FURTHER
        backward pass accumulates into the flat buffer (training.st's second phase adds to re-scaled first-phase
        gradients, losses.py:389-397).  A parameter that has received no gradient so far gets its zero-filled slice
        (zero_grad() cleared the buffer): a gradient that first appears in the later pass then lands in the flat buffer
        as well, where the per-micro-batch re-scaling (one mul_ on the flat buffer) reaches it like the reference's loop
        over every non-None p.grad does."""
        for p, dst, _ in self.gslices:
            p.grad = dst.view(p.shape)
        self.gathered = False   # gradients that appear (or get re-allocated) from here on are collected by the next gather

    def parameters(self):
        return [p for _, p in self.named]


def _pad4(n):
    return (n + 3) // 4 * 4


def flatten_models(modules):
    """One joint flat parameter buffer and one joint gradient buffer for several modules."""
    sizes = []
    for m in modules:
        sizes.append(sum(_pad4(p.numel()) for n, p in m.named_parameters() if _receives_grad(n, p)))
    dev = next(modules[0].parameters()).device
    flat_p = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    flat_g = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    out, off = [], 0
    for m, s in zip(modules, sizes):
        out.append(FlatParams(m, flat_p, flat_g, off))
        off += s
    return out, flat_p, flat_g


class FusedAdamW:
    """torch.optim.AdamW(lr, betas=(beta1, 0.99), eps, weight_decay) semantics (losses.py:40) on a FlatParams."""

    def __init__(self, flat, lr, beta1=0.9, beta2=0.99, eps=1e-8, weight_decay=0.01, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad=True is not used by any reference config")
        self.flat = flat
        self.param_groups = [dict(lr=lr, betas=(beta1, beta2), eps=eps, weight_decay=weight_decay)]
        self.m = torch.zeros_like(flat.p)
        self.v = torch.zeros_like(flat.p)
        self.t = 0

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def step(self, grad_clip=-1.0, ema=None, grad_scale=1.0):
        """clip_grad_norm_(grad_clip) -> AdamW -> (optional) EMA update, one fused pass.
        grad_scale multiplies the raw gradient first (1/world_size after a SUM all-reduce)."""
        g = self.param_groups[0]
        self.t += 1
        self.flat.gather_grads()
        gn = hip.sqnorm(self.flat.g) if grad_clip >= 0 else None
        omd = 0.0
        shadow = None
        if ema is not None:
            omd = ema.advance()
            shadow = ema.shadow
        hip.adamw_ema(self.flat.p, self.flat.g, self.m, self.v, shadow, g["lr"], g["betas"][0], g["betas"][1],
                      g["eps"], g["weight_decay"], self.t, float(grad_clip), gn, float(grad_scale), float(omd))
        self.last_grad_sqnorm = gn

    def _views(self, flat):
        out, o = {}, 0
        for n, p in self.flat.named:
            out[n] = flat[o:o + p.numel()].view(p.shape)
            o += _pad4(p.numel())
        return out

    def state_dict(self):
        """torch.optim.AdamW.state_dict() layout (what the reference's save_checkpoint stores, utils.py:36-48):
        `state[i]` for the i-th entry of model.parameters() (only parameters that have been stepped), moments in
        the REFERENCE tensor layout."""
        model = self.flat.module
        m, v = self._views(self.m), self._views(self.v)
        names = [n for n, _ in model.named_parameters()]
        state = {}
        if self.t > 0:
            for i, n in enumerate(names):
                if n in m:
                    state[i] = dict(step=torch.tensor(float(self.t)),
                                    exp_avg=model.to_reference_layout(n, m[n]).clone(),
                                    exp_avg_sq=model.to_reference_layout(n, v[n]).clone())
        g = self.param_groups[0]
        group = dict(lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"], amsgrad=False,
                     foreach=None, maximize=False, capturable=False, differentiable=False, fused=None,
                     decoupled_weight_decay=True, params=list(range(len(names))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        model = self.flat.module
        m, v = self._views(self.m), self._views(self.v)
        names = [n for n, _ in model.named_parameters()]
        steps = set()
        for i, st in sd["state"].items():
            n = names[int(i)]
            if n not in m:
                continue
            m[n].copy_(model.from_reference_layout(n, st["exp_avg"]).to(m[n].device))
            v[n].copy_(model.from_reference_layout(n, st["exp_avg_sq"]).to(v[n].device))
            steps.add(int(float(st["step"])))
        self.t = max(steps) if steps else 0
        g = sd["param_groups"][0]
        self.param_groups = [dict(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"])]


class ExponentialMovingAverage:
    """models/ema.py:10-97 on the flat buffer.  `shadow_params` (state_dict) keeps the reference's list order
    (= module.parameters() order of requires_grad params, including lamb / geom_p)."""

    def __init__(self, flat, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.flat = flat
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow = flat.p.clone()
        self.other_shadow = [p.detach().clone() for _, p in flat.other]
        self.collected = None

    def advance(self):
        """Increment num_updates and return (1 - decay) with the reference's warm-up (ema.py:43-48)."""
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        return 1.0 - decay

    def update(self, parameters=None):
        omd = self.advance()
        hip.ema_update(self.shadow, self.flat.p, omd)
        # lamb / geom_p never change; their shadow stays equal

    def copy_to(self, parameters=None):
        self.flat.p.copy_(self.shadow)
        for (_, p), s in zip(self.flat.other, self.other_shadow):
            p.data.copy_(s)

    def store(self, parameters=None):
        self.collected = (self.flat.p.clone(), [p.detach().clone() for _, p in self.flat.other])

    def restore(self, parameters=None):
        self.flat.p.copy_(self.collected[0])
        for (_, p), s in zip(self.flat.other, self.collected[1]):
            p.data.copy_(s)

    def _shadow_views(self):
        views, o = {}, 0
        for n, p in self.flat.named:
            views[n] = self.shadow[o:o + p.numel()].view(p.shape)
            o += _pad4(p.numel())
        for (n, _), s in zip(self.flat.other, self.other_shadow):
            views[n] = s
        return views

    def shadow_params(self):
        v = self._shadow_views()
        return [v[n] for n, p in self.flat.module.named_parameters() if p.requires_grad]

    def state_dict(self):
        """models/ema.py:91-92: shadow_params in module.parameters() order, REFERENCE tensor layout."""
        model = self.flat.module
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        return dict(decay=self.decay, num_updates=self.num_updates,
                    shadow_params=[model.to_reference_layout(n, s).clone() for n, s in zip(names, self.shadow_params())])

    def load_state_dict(self, sd):
        model = self.flat.module
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        self.decay = sd["decay"]
        self.num_updates = sd["num_updates"]
        for n, dst, src in zip(names, self.shadow_params(), sd["shadow_params"]):
            dst.copy_(model.from_reference_layout(n, src).to(dst.device))
