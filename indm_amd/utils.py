"""Boundary helpers with the reference's names (utils.py:61-154, datasets.py:56-71, models/utils.py:88-94).

    create_model(config)            -> NCSNpp on config.device
    load_model / load_flow_model    -> state dicts `dict(optimizer, model, ema, step)` (no checkpoint I/O)
    get_loss_fns(config, sde, inverse_scaler, train=True, scaler=None)
                                    -> (train_step_fn, nll_fn, nelbo_fn, sampling_fn)
"""
import torch

from . import likelihood, losses, sampling
from .flow import create_flow_model
from .ncsnpp import NCSNpp
from .optim import ExponentialMovingAverage, flatten_models


def get_data_scaler(config):
    """datasets.py:56-62"""
    return (lambda x: x * 2.0 - 1.0) if config.data.centered else (lambda x: x)


def get_data_inverse_scaler(config):
    """datasets.py:65-71"""
    return (lambda x: (x + 1.0) / 2.0) if config.data.centered else (lambda x: x)


def create_model(config, device=None):
    """models/utils.py:88-94 (no DataParallel wrapper: one process per GPU, see parallel.py)."""
    if config.model.name != "ncsnpp":
        raise NotImplementedError(config.model.name)
    return NCSNpp(config).to(device or config.device)


def build_states(config, score_model=None, flow_model=None, device=None):
    """utils.load_model + utils.load_flow_model (utils.py:61-145) without the checkpoint I/O: builds both
    networks (unless given), re-homes their parameters into ONE flat parameter / gradient buffer and returns
    (state, flow_state, flat_grad) with the reference's dict layout."""
    device = device or config.device
    if score_model is None:
        score_model = create_model(config, device)
    if flow_model is None:
        flow_model = create_flow_model(config).to(device)
    (fs, ff), flat_p, flat_g = flatten_models([score_model, flow_model])
    score_model.__dict__["_flat"] = fs      # graphed.score_call delivers the captured backward's gradients into its flat buffer
    opt = losses.get_optimizer(config, fs)
    fopt = losses.get_optimizer(config, ff, lr=config.flow.lr)
    state = dict(optimizer=opt, model=score_model, ema=ExponentialMovingAverage(fs, config.model.ema_rate), step=0)
    flow_state = dict(optimizer=fopt, model=flow_model, ema=ExponentialMovingAverage(ff, config.flow.ema_rate),
                      step=0)
    return state, flow_state, flat_g


def save_checkpoint(config, ckpt_path, state):
    """utils.save_checkpoint (utils.py:36-48): the file is readable by the REFERENCE's restore_checkpoint --
    'model' carries the `module.`-prefixed OIHW state dict, 'ema' the shadow_params list in parameters() order,
    'optimizer' a torch.optim.AdamW state dict."""
    model = state["model"]
    saved = {"optimizer": state["optimizer"].state_dict(),
             "model": {k: v.cpu() for k, v in model.reference_state_dict(prefix="module.").items()},
             "ema": {k: ([t.cpu() for t in v] if isinstance(v, list) else v) for k, v in state["ema"].state_dict().items()},
             "step": state["step"]}
    for st in saved["optimizer"]["state"].values():
        st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].cpu(), st["exp_avg_sq"].cpu()
    torch.save(saved, ckpt_path)


def restore_checkpoint(config, ckpt_path, state, device=None):
    """utils.restore_checkpoint (utils.py:14-33) for checkpoints written by this repo OR by the reference."""
    import os
    if not os.path.exists(ckpt_path):
        return state
    loaded = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    if config.training.sde != "vesde":   # utils.py:23-24: VE runs do not restore the optimiser
        state["optimizer"].load_state_dict(loaded["optimizer"])
    state["model"].load_reference_state_dict(loaded["model"])
    state["ema"].load_state_dict(loaded["ema"])
    state["step"] = loaded["step"]
    return state


def get_loss_fns(config, sde, inverse_scaler, train=True, scaler=None, world_size=1, allreduce=None):
    """utils.py:147-154"""
    optimize_fn = losses.optimization_manager(config)
    train_step_fn = losses.get_step_fn(config, sde, train=train, optimize_fn=optimize_fn, scaler=scaler,
                                       world_size=world_size, allreduce=allreduce)
    nll_fn = likelihood.get_likelihood_fn(config, sde, inverse_scaler, rtol=config.eval.rtol, atol=config.eval.atol)
    nelbo_fn = likelihood.get_elbo_fn(config, sde, inverse_scaler=inverse_scaler)
    shape = (config.sampling.batch_size, config.data.num_channels, config.data.image_size, config.data.image_size)
    sampling_fn = sampling.get_sampling_fn(config, sde, shape, inverse_scaler, config.sampling.truncation_time)
    return train_step_fn, nll_fn, nelbo_fn, sampling_fn
