"""Sample generation with the reference's on-disk cache: sampling_lib.get_samples (sampling_lib.py:31-175).

    get_samples(config, score_model, flow_model, sampling_fn, step, r, sample_dir, temperature=1.,
                inverse_scaler=None, this_sample_dir=None, scaler=None, data_mean=None)
        -> (samples_before_flow, samples_after_flow)      [N,C,H,W] in [0,1]

File names, npz keys, dtypes and layouts are the reference's, so caches written by either side can be resumed by the
other:  <sample_dir>/samples_{r}_before_flow.npz            float, NHWC * 255 (latent-space samples, unclipped)
        <sample_dir>/samples_{r}_before_flow_for_search.npz  (written by pc_sampler one step before the end, :436-445)
        <sample_dir>/samples_{r}_before_flow_denoise_{t}.npz / ..._more_step.npz
        <this_sample_dir>/samples_{r}.npz                    uint8 NHWC, clipped (+ sample_{r}.png grid)
        <this_sample_dir>/samples_{r}_denoise_{t}.npz / samples_{r}_more_step.npz
The arithmetic (sampler, flow inverse in chunks of 16) runs on the HIP path; this module is host-side file plumbing.
get_latents / save_statistics (Inception features for FID) are out of scope.
"""
import logging
import os

import numpy as np
import torch

from .flow import flow_forward


def _save_npz(path, samples):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        np.savez_compressed(f, samples=samples)


def _to_nhwc255(config, x):
    s = x.permute(0, 2, 3, 1).cpu().numpy() * 255.0
    s = s.reshape((-1, config.data.image_size, config.data.image_size, config.data.num_channels))
    assert s.shape == (s.shape[0], config.data.image_size, config.data.image_size, config.data.num_channels)
    return s


def _to_uint8(config, x):
    return np.clip(_to_nhwc255(config, x), 0.0, 255.0).astype(np.uint8)


def _load(path):
    return torch.tensor(np.load(path)["samples"]).permute(0, 3, 1, 2) / 255.0


def _flow_inverse_chunks(config, flow_model, before, scaler, inverse_scaler, temperature):
    """sampling_lib.py:86-96 / 145-152: the flow inverse of cached latent samples, 16 at a time."""
    dev = config.device
    with torch.no_grad():
        if config.flow.model != "identity":
            assert before.shape[0] % 16 == 0
            after = torch.zeros_like(before, device=dev)
            for k in range(before.shape[0] // 16):
                after[16 * k:16 * (k + 1)] = flow_forward(
                    config, flow_model, scaler(before[16 * k:16 * (k + 1)]).to(dev) * temperature, log_det=None,
                    reverse=True)[0]
        else:
            after = before.to(dev)
        return inverse_scaler(after)


def _save_grid(path, samples_u8):
    """make_grid(nrow = sqrt(N), padding 2) + save_image of sampling_lib.py:162-166; skipped when PIL is absent."""
    try:
        from PIL import Image
    except ImportError:  # pragma: no cover
        return
    n, h, w, c = samples_u8.shape
    nrow = max(1, int(np.sqrt(n)))           # images per row, as torchvision.make_grid(tensor, nrow)
    ncol = int(np.ceil(n / nrow))
    pad = 2
    grid = np.zeros((ncol * (h + pad) + pad, nrow * (w + pad) + pad, c), dtype=np.uint8)
    for i in range(n):
        y, x = pad + (i // nrow) * (h + pad), pad + (i % nrow) * (w + pad)
        grid[y:y + h, x:x + w] = samples_u8[i]
    Image.fromarray(grid[..., 0] if c == 1 else grid).save(path)


def get_samples(config, score_model, flow_model, sampling_fn, step, r, sample_dir, temperature=1.0,
                inverse_scaler=None, this_sample_dir=None, scaler=None, data_mean=None):
    logging.info("sampling -- ckpt step: %d, round: %d" % (step, r))
    os.makedirs(sample_dir, exist_ok=True)
    os.makedirs(this_sample_dir, exist_ok=True)
    first = os.path.join(sample_dir, f"samples_{r}_before_flow.npz")
    if not os.path.exists(first):
        before, after, n = sampling_fn(score_model, flow_model, temperature, data_mean, sample_dir=sample_dir, r=r)
        logging.info(f"nfe: {n}")
        _save_npz(first, _to_nhwc255(config, before))
        _save_npz(os.path.join(this_sample_dir, f"samples_{r}.npz"), _to_uint8(config, after))
        return before, after
    before = _load(first)
    t = config.sampling.pc_denoise_time
    if config.sampling.pc_denoise:
        out = os.path.join(this_sample_dir, f"samples_{r}_denoise_{t}.npz")
        if not os.path.exists(out):
            mid = os.path.join(sample_dir, f"samples_{r}_before_flow_denoise_{t}.npz")
            if not os.path.exists(mid):
                logging.info(f"denoise for pc with round {r} and final time {t}")
                src = f"samples_{r}_before_flow_for_search.npz" if config.training.sde == "vesde" \
                    else f"samples_{r}_before_flow.npz"
                before = _load(os.path.join(sample_dir, src))
                before, after, n = sampling_fn(score_model, flow_model, temperature, data_mean, final_time=t,
                                               before_data=scaler(before))
                _save_npz(mid, _to_nhwc255(config, before))
            else:
                before = _load(mid)
                after = _flow_inverse_chunks(config, flow_model, before, scaler, inverse_scaler, temperature)
            _save_npz(out, _to_uint8(config, after))
        else:
            after = _load(out)
    elif config.sampling.more_step:
        out = os.path.join(this_sample_dir, f"samples_{r}_more_step.npz")
        if not os.path.exists(out):
            logging.info(f"more step with round {r}")
            before, after, n = sampling_fn(score_model, flow_model, temperature, data_mean, before_data=scaler(before))
            _save_npz(os.path.join(sample_dir, f"samples_{r}_before_flow_more_step.npz"), _to_nhwc255(config, before))
            _save_npz(out, _to_uint8(config, after))
        else:
            after = _load(out)
    else:
        out = os.path.join(this_sample_dir, f"samples_{r}.npz")
        if not os.path.exists(out):
            after = _flow_inverse_chunks(config, flow_model, before, scaler, inverse_scaler, temperature)
            u8 = _to_uint8(config, after)
            _save_npz(out, u8)
            _save_grid(os.path.join(this_sample_dir, f"sample_{r}.png"), u8)
        else:
            after = _load(out)
    return before, after
