"""Loss / step-function factories with the reference's API (losses.py:30-420, models/utils.py:96-197).

    get_optimizer, optimization_manager, get_sde_loss_fn, get_step_fn, get_score_fn, get_model_fn
keep the reference signatures (plus an optional `rng=` noise source for parity tests).  The work inside a
step is HIP kernels end to end: flow forward + explicit-VJP log-det, fused SDE perturbation, NCSN++ on MFMA
implicit GEMMs, fused DSM loss, ordinary first-order autograd backward through the same kernels, then one
flat-buffer all-reduce (if distributed) and one fused clip+AdamW+EMA pass per network.
"""
import numpy as np
import torch

from . import graphed, hip, ops, sde_lib
from .flow import flow_forward
from .optim import FusedAdamW
from .rng import DeviceRng, default_rng, mix64
from .sde_lib import VESDE, VPSDE


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def get_optimizer(config, flat, lr=None, beta1=None, eps=None, weight_decay=None):
    """losses.py:30-45 (takes the FlatParams of a model instead of an iterator of parameters)."""
    lr = config.optim.lr if lr is None else lr
    beta1 = config.optim.beta1 if beta1 is None else beta1
    eps = config.optim.eps if eps is None else eps
    weight_decay = config.optim.weight_decay if weight_decay is None else weight_decay
    if config.optim.optimizer == "AdamW":
        return FusedAdamW(flat, lr=lr, beta1=beta1, beta2=0.99, eps=eps, weight_decay=weight_decay,
                          amsgrad=config.optim.amsgrad)
    raise NotImplementedError(f"Optimizer {config.optim.optimizer} not supported yet!")


def optimization_manager(config):
    """losses.py:48-62: warm-up, clip_grad_norm_, step -- clip and step are one fused kernel here."""

    def optimize_fn(optimizer, params, step, lr=config.optim.lr, warmup=config.optim.warmup,
                    grad_clip=config.optim.grad_clip, ema=None, grad_scale=1.0):
        if warmup > 0:
            for g in optimizer.param_groups:
                g["lr"] = lr * np.minimum(step / warmup, 1.0)
        optimizer.step(grad_clip=grad_clip, ema=ema, grad_scale=grad_scale)

    return optimize_fn


# ----------------------------------------------------------------------------- model / score wrappers
def set_mode(model, train):
    """model.train() / model.eval() as the reference calls them on EVERY use (models/utils.py:118-123), without
    re-assigning 780 unchanged `training` attributes through nn.Module.__setattr__ (3 ms of host time per step)."""
    train = bool(train)
    for m in model.modules():
        if m.training != train:
            m.training = train
    return model


def get_model_fn(model, train=False):
    """models/utils.py:96-125"""

    def model_fn(x, labels, **kw):
        set_mode(model, train)
        return model(x, labels, **kw)

    return model_fn


def get_score_fn(config, sde, model, gamma_t=None, train=False, continuous=False, rng=None):
    """models/utils.py:140-197 (continuous=True branches; every shipped config trains continuously)."""
    model_fn = get_model_fn(model, train=train)
    if not continuous:
        raise NotImplementedError("discrete-time score functions are not used by the INDM configs")
    if isinstance(sde, VPSDE):
        def score_fn(x, t):
            th = t.detach().cpu()
            out = model_fn(x, hip.to_device(th * 999, x.device), rng=rng)
            if config.training.ddpm_score:
                std = sde.marginal_coeffs(th)[1]
                out = ops.scale_rows(out, hip.to_device(-1.0 / std, x.device))
            return out
    elif isinstance(sde, VESDE):
        def score_fn(x, t):
            labels = sde.marginal_coeffs(t.detach().cpu())[1]
            return model_fn(x, hip.to_device(labels, x.device), rng=rng)
    else:
        raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
    return score_fn


def _host_schedule(sde, t):
    """[B] fp32 host vectors the fused kernels need for a batch of times t."""
    mc, std = sde.marginal_coeffs(t)
    return mc, std, sde.model_labels(t)


def get_sde_loss_fn(config, sde, train, variance="scoreflow"):
    """losses.py:65-144.  loss_fn(model, batch_nchw) -> per-image losses [B] (device)."""
    reduce_mean = bool(config.training.reduce_mean)
    kind = 0 if (isinstance(sde, VPSDE) and config.training.ddpm_score) else 1

    def loss_fn(model, batch, st=False, recon_loss=None, importance_sampling=None, rng=None):
        if recon_loss is None:
            recon_loss = config.training.reconstruction_loss
        if importance_sampling is None:
            importance_sampling = config.training.importance_sampling
        rng = rng or default_rng(batch.device)
        dev = batch.device
        B, C, H, W = batch.shape
        t_min = sde.get_t_min(config, st, rng)
        t, Z = sde.get_diffusion_time(config, B, "cpu", t_min, importance_sampling=importance_sampling, rng=rng)
        mc, std, labels = _host_schedule(sde, t)
        z = rng.randn_image(B, C, H, W)                                    # losses.py:103 (NHWC here)
        net = _unwrap(model)
        set_mode(net, train)
        x_nhwc = ops.to_nhwc(batch)
        xt = ops.sde_perturb(x_nhwc, z, hip.to_device(mc, dev), hip.to_device(std, dev))
        if (train and not recon_loss and config.optim.num_micro_batch == 1
                and (config.training.likelihood_weighting or not config.training.st)):
            # the step's training calls of the score network: two captured hipGraphs (graphed.py; eager when disabled).  The
            # captured backward OVERWRITES the score slice of the flat gradient buffer, so the two cases in which gradients of
            # several backward passes add up stay eager: micro-batches, and the two-phase fid step with training.st (its second
            # phase accumulates onto the re-scaled first-phase gradients, losses.py:378-397)
            out = graphed.score_call(net, net.__dict__.get("_flat"), xt, hip.to_device(labels, dev), rng)
        else:
            out = net(xt, hip.to_device(labels, dev), rng=rng, nhwc=True)
        if importance_sampling or not config.training.likelihood_weighting:
            w = (0.5 * Z) * torch.ones(B)
            mode = 0
        else:
            g2 = sde.diffusion(t) ** 2
            w = 0.5 * Z * g2
            mode = 1
        losses = ops.dsm_loss(out, z, hip.to_device(std, dev), hip.to_device(w.float(), dev), kind, mode, reduce_mean)
        if recon_loss:
            # losses.py:120-140: a second network pass at t_min.  With perturbed = alpha x + beta z and
            # q_mean = (perturbed + beta^2 score) / alpha:   x - q_mean = -(beta / alpha) (beta score + z),   so the
            # squared-error sum IS the denoising sum of this pass and the whole term is one more fused DSM launch:
            #   q_recon - p_entropy = n/2 (2 log q_std - 2 log beta - 1) + (beta / alpha)^2 / (2 q_std^2) * sum (beta score + z)^2
            eps_vec = torch.ones(B) * t_min
            alpha, beta, labels2 = _host_schedule(sde, eps_vec)
            z2 = rng.randn_image(B, C, H, W)                               # losses.py:123
            xt2 = ops.sde_perturb(x_nhwc, z2, hip.to_device(alpha, dev), hip.to_device(beta, dev))
            out2 = net(xt2, hip.to_device(labels2, dev), rng=rng, nhwc=True)
            a64, b64 = alpha.double(), beta.double()
            q_std = b64 if variance == "ddpm" else b64 / a64
            n_dim = float(C * H * W)
            w2 = 0.5 * (b64 / a64) ** 2 / q_std ** 2
            const = n_dim / 2.0 * (2.0 * torch.log(q_std) - 2.0 * torch.log(b64) - 1.0)
            recon = ops.dsm_loss(out2, z2, hip.to_device(beta, dev), hip.to_device(w2.float(), dev), kind, 0, False)
            recon = recon + hip.to_device(const.float(), dev)
            if reduce_mean:
                recon = recon / n_dim
            losses = losses + recon
        return losses

    return loss_fn


# ----------------------------------------------------------------------------- step functions
def get_step_fn(config, sde, train, optimize_fn=None, scaler=None, world_size=1, allreduce=None):
    """losses.py:194-420.  Returns step_fn(state, flow_state, batch[, rng]) -> 4 per-image loss vectors on
    the CPU, exactly like the reference.  `state` / `flow_state` are dicts(optimizer, model, ema, step)."""
    if not config.training.continuous:
        raise NotImplementedError("discrete SMLD/DDPM losses are not used by the INDM configs")
    from . import hip
    hip.set_compute_dtype(config.get("compute_dtype", "f32") if hasattr(config, "get") else "f32")
    loss_fn = get_sde_loss_fn(config, sde, train)
    if optimize_fn is None:
        optimize_fn = optimization_manager(config)

    def calculate_logp(batch, rng):  # losses.py:219-225
        B, C, H, W = batch.shape
        Ts = torch.ones(B) * sde.T
        mc, std = sde.marginal_coeffs(Ts)
        z = rng.randn_image(B, C, H, W)
        yT = ops.sde_perturb(ops.to_nhwc(batch), z, hip.to_device(mc, batch.device), hip.to_device(std, batch.device))
        return sde_lib.prior_logp(sde, yT)

    def _zero(state):
        state["optimizer"].zero_grad()

    _rngs = {}

    def _step_rng(device, step):
        """default noise source of a step: per-rank noise, rank-independent series lengths (see DeviceRng).  ONE
        resident source per device, re-keyed per step (no per-step generator objects or synchronous state upload)."""
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        seed, pseed = mix64(config.get("seed", 0), step, rank), mix64(config.get("seed", 0), step) % (1 << 32)
        key = (device.type, device.index)
        r = _rngs.get(key)
        if r is None:
            r = _rngs[key] = DeviceRng(device, seed=seed, poisson_seed=pseed)
            return r
        return r.reseed(seed, pseed)

    class _HostLosses:
        """The per-image loss vectors the step returns on the CPU (reference API: losses.py:281-284,320).  Their
        device->host copy is enqueued right after the FORWARD pass (pinned buffer, non-blocking) and only that copy's
        event is awaited when the step returns: the backward pass and the optimiser are still running on the GPU
        then, and the host-bound preamble of the NEXT step (encoder, prior flow, per-call constants: ~7.6 ms with an
        idle GPU when the step ended with a full synchronisation) hides behind them."""

        def __init__(self, nrows):
            self.parts, self.nrows = [], nrows

        def push(self, rows):
            """rows: {row index: [b] device tensor} of one micro-batch"""
            idx = sorted(rows)
            dev = torch.stack([rows[i].detach() for i in idx])
            if dev.is_cuda:
                host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
                host.copy_(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            else:
                host, ev = dev.clone(), None
            self.parts.append((idx, host, ev))

        def result(self, B):
            outs = [[] for _ in range(self.nrows)]
            for idx, host, ev in self.parts:
                if ev is not None:
                    ev.synchronize()
                for j, i in enumerate(idx):
                    outs[i].append(host[j])
            return tuple(torch.cat(o) if o else torch.zeros(B) for o in outs)

    def _reduce_and_step(states):
        """gather -> (data-parallel) all-reduce of exactly the slices being stepped -> fused clip+AdamW+EMA.
        The reference ramps BOTH optimisers from the constant config.optim.lr (optimize_fn's default lr,
        losses.py:51-57 -- the flow's own lr is overwritten as soon as optim.warmup > 0); with warmup = 0 (every
        shipped config) each optimiser keeps the lr it was built with."""
        for s in states:
            s["optimizer"].flat.gather_grads()   # autograd's per-parameter grads -> the flat buffer (one launch)
        if allreduce is not None:
            allreduce([s["optimizer"].flat for s in states])
        for s in states:
            optimize_fn(s["optimizer"], None, step=s["step"], ema=s["ema"], grad_scale=1.0 / world_size)

    def _eval_ema(state, flow_state):
        """train=False in flow_step_fn_nll: the reference still runs ema.update on both networks (losses.py:314-318);
        in a training step that update is part of the fused optimiser kernel."""
        state["ema"].update()
        flow_state["ema"].update()

    def _nll_losses(state, flow_state, mini_batch, rng, importance_sampling=None, st=False, keep=None):
        flow_model = flow_state["model"]
        _unwrap(flow_model).train()
        z, losses_flow = flow_forward(config, flow_model, mini_batch, reverse=False, rng=rng)
        losses_score = loss_fn(state["model"], z, st=st, importance_sampling=importance_sampling, rng=rng)
        if keep is not None:   # training.st: flow_step_fn_fid's second phase re-uses these (losses.py:383-389)
            keep["z"], keep["losses_score"] = z.detach(), losses_score.detach()
        losses_logp = calculate_logp(z, rng)
        D = float(np.prod(mini_batch.shape[1:]))
        if config.training.reduce_mean:
            losses_flow, losses_logp = -losses_flow / D, -losses_logp / D
        else:
            losses_flow, losses_logp = -losses_flow, -losses_logp
        assert losses_score.shape == losses_flow.shape == losses_logp.shape == torch.Size([z.shape[0]])
        return losses_score, losses_flow, losses_logp

    def flow_step_fn_nll(state, flow_state, batch, rng=None):
        """losses.py:258-320"""
        rng = rng or _step_rng(batch.device, state["step"])
        B = batch.shape[0]
        nmb = config.optim.num_micro_batch
        outs = _HostLosses(4)
        _zero(state)
        _zero(flow_state)
        if hasattr(allreduce, "begin_step"):
            allreduce.begin_step()
        if train:
            for k in range(nmb):
                mb = batch[B // nmb * k: B // nmb * (k + 1)]
                ls, lf, lp = _nll_losses(state, flow_state, mb, rng, st=config.training.st)   # losses.py:294
                losses = ls + lf + lp
                outs.push({0: losses, 1: ls, 2: lf, 3: lp})   # D2H enqueued before the backward pass
                if k == nmb - 1 and hasattr(allreduce, "arm"):
                    # data parallel: the score slice's all-reduce starts inside this backward pass, as soon as the
                    # score network's last gradient exists, and overlaps the flow's backward (parallel.GradReducer)
                    allreduce.arm(state["optimizer"].flat)
                torch.mean(losses).backward()
            _reduce_and_step([state, flow_state])
        else:
            _eval_ema(state, flow_state)
        state["step"] += 1
        flow_state["step"] += 1
        return outs.result(B)

    def flow_step_fn_fid(state, flow_state, batch, rng=None):
        """losses.py:322-406, training.st included (:378-395)"""
        st = bool(config.training.st)
        rng = rng or _step_rng(batch.device, state["step"])
        B = batch.shape[0]
        nmb = config.optim.num_micro_batch
        outs = _HostLosses(4)
        _zero(state)
        _zero(flow_state)
        if hasattr(allreduce, "begin_step"):
            allreduce.begin_step()
        if train:
            keep = {} if st else None
            for k in range(nmb):
                mb = batch[B // nmb * k: B // nmb * (k + 1)]
                ls, lf, lp = _nll_losses(state, flow_state, mb, rng, importance_sampling=True, keep=keep)
                losses = ls + lf + lp
                outs.push({0: losses, 2: lf, 3: lp})
                torch.mean(losses).backward()
            _reduce_and_step([flow_state])
            sflat = state["optimizer"].flat
            if st:
                # the first phase's score gradients stay (no zero_grad, losses.py:379-380): collect them into the flat
                # buffer and let the second phase's backward accumulate into its slices
                sflat.gather_grads()
                sflat.adopt_flat_grads()
            else:
                _zero(state)
            for k in range(nmb):
                mb = batch[B // nmb * k: B // nmb * (k + 1)]
                if not st:
                    with torch.no_grad():
                        z, _ = flow_forward(config, flow_state["model"], mb, log_det=None, reverse=False, rng=rng)
                else:
                    z = keep["z"]   # the LAST micro-batch's latent of the first phase, whatever `mb` is (losses.py:383-386)
                la = loss_fn(state["model"], z.detach(), st=st, recon_loss=False, rng=rng)
                if st:   # p.grad <- const_adj * p.grad for every score parameter (losses.py:389-395): one launch on the
                    # flat buffer, the factor stays on the device.  Data parallel: the reference's single-process
                    # DataParallel forms the two means over the WHOLE batch, so the per-rank means are all-reduced
                    # first (equal shards: the ratio of summed means is the ratio of global means) -- every rank then
                    # scales its un-reduced phase-1 gradients by the same factor, which commutes with the later SUM.
                    means = torch.stack([la.mean(), keep["losses_score"].mean()]).detach()
                    if world_size > 1 and torch.distributed.is_available() and torch.distributed.is_initialized():
                        torch.distributed.all_reduce(means, op=torch.distributed.ReduceOp.SUM)
                    sflat.g.mul_(means[0] / means[1])
                outs.push({1: la})
                torch.mean(la).backward()
            _reduce_and_step([state])
            state["step"] += 1
            flow_state["step"] += 1
        return outs.result(B)   # train=False: the reference's fid step does nothing at all (losses.py:353,406)

    if config.flow.model == "identity":
        raise NotImplementedError("flow.model='identity' (score-only training) is outside the INDM hot path")
    if not config.training.likelihood_weighting:
        return flow_step_fn_fid
    return flow_step_fn_nll
