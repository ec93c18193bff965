"""The bits/dim evaluation protocol of the reference: evaluation.get_bpd (evaluation.py:388-494).

    get_bpd(config, eval_ds, scaler, nelbo_fn, nll_fn, score_model, flow_model=None, step=0, eval=False)

`config.eval.num_nelbo` NELBO passes over the evaluation set, then the PF-ODE NLL "wrong" (residual=False) and
"correct" (residual=True) at eps_bpd (1e-5, or config.eval.truncation_time), then once more at the training
truncation time when that differs from 1e-5.  Every batch is uniformly dequantised ((255 x + U)/256) and scaled
exactly as the reference does; nelbo_fn / nll_fn are the functions utils.get_loss_fns returns (likelihood.py), i.e.
all arithmetic runs on the HIP path.  The reference only LOGS the means; this mirror logs the same lines and also
returns them.

eval_ds is any re-iterable of batches: [B,C,H,W] float tensors in [0,1] (what datasets.get_batch_ hands on,
datasets.py:115-128) or TFDS-style dicts {'image': NHWC array}.  The TFDS readers themselves are out of scope
(SURVEY section 8f rank 4).
"""
import logging

import numpy as np
import torch

from .rng import default_rng


def _to_batch(config, item):
    if isinstance(item, dict):  # datasets.get_batch_ :124-125
        img = item["image"]
        img = img._numpy() if hasattr(img, "_numpy") else np.asarray(img)
        item = torch.from_numpy(np.asarray(img)).float().permute(0, 3, 1, 2)
    elif isinstance(item, (tuple, list)):
        item = item[0]
    b = torch.as_tensor(item).float()
    assert b.shape == (b.shape[0], config.data.num_channels, config.data.image_size, config.data.image_size)
    return b.to(config.device)


def get_batch(config, data_iter, data):
    """datasets.get_batch datasets.py:106-113: restart the iterator when it is exhausted."""
    try:
        batch = _to_batch(config, next(data_iter))
    except StopIteration:
        logging.info("New Epoch Start")
        data_iter = iter(data)
        batch = _to_batch(config, next(data_iter))
    return batch, data_iter


def get_bpd(config, eval_ds, scaler, nelbo_fn, nll_fn, score_model, flow_model=None, step=0, eval=False, rng=None):
    results = {}
    with torch.no_grad():
        if config.flow.model != "identity":
            flow_model.eval()
        num_data = config.eval.num_test_data if eval else 10000
        rng = rng or default_rng(config.device)
        kw = dict(rng=rng)

        def batches(n):
            it = iter(eval_ds)
            for _ in range((n - 1) // config.eval.batch_size + 1):
                b, it = get_batch(config, it, eval_ds)
                yield scaler((255.0 * b + rng.rand(*b.shape)) / 256.0)

        full, full_r = [], []
        for _ in range(config.eval.num_nelbo):
            bpds, bpds_r = [], []
            for b in batches(num_data):
                nelbo_bpd, nelbo_residual_bpd = nelbo_fn(score_model, flow_model, b, None, **kw)
                bpds.extend(nelbo_bpd.detach().cpu().numpy().reshape(-1))
                bpds_r.extend(nelbo_residual_bpd.detach().cpu().numpy().reshape(-1))
            full.append(np.mean(bpds))
            full_r.append(np.mean(bpds_r))
            logging.info("step: %d, num samples: %d, mean nelbo bpd: %.5e, std nelbo bpd: %.5e" % (
                step, len(bpds), np.mean(bpds), np.std(bpds)))
            logging.info("step: %d, num samples: %d, mean nelbo_residual bpd: %.5e, std nelbo_residual bpd: %.5e" % (
                step, len(bpds_r), np.mean(bpds_r), np.std(bpds_r)))
        logging.info("step: %d, average nelbo bpd out of %d evaluations: %.5e" % (step, len(full), np.mean(full)))
        logging.info("step: %d, average nelbo bpd out of %d evaluations: %.5e" % (step, len(full_r), np.mean(full_r)))
        results["nelbo_bpd_per_eval"], results["nelbo_residual_bpd_per_eval"] = full, full_r
        results["nelbo_bpd"], results["nelbo_residual_bpd"] = float(np.mean(full)), float(np.mean(full_r))

        if not eval:
            num_data = num_data // 10
        eps_bpd = 1e-5 if config.eval.truncation_time == -1.0 else config.eval.truncation_time

        def nll_pass(tag, residual, e):
            v = []
            for b in batches(num_data):
                bpd = nll_fn(score_model, flow_model, b, None, residual=residual, eps_bpd=e, **kw)[0]
                v.extend(bpd.detach().cpu().numpy().reshape(-1))
                if eval:
                    logging.info("step: %d, [%s] num samples: %d, mean nll bpd: %.5e, std nll bpd: %.5e" % (
                        step, tag, len(v), np.mean(v), np.std(v)))
            logging.info("step: %d, [%s] num samples: %d, mean nll bpd: %.5e, std nll bpd: %.5e" % (
                step, tag, len(v), np.mean(v), np.std(v)))
            return float(np.mean(v))

        if not config.eval.skip_nll_wrong:
            results["nll_wrong_bpd"] = nll_pass("NLL WRONG w/ eps=%.1e" % eps_bpd, False, eps_bpd)
        results["nll_correct_bpd"] = nll_pass("NLL CORRECT w/ eps=%.1e" % eps_bpd, True, eps_bpd)
        if config.training.truncation_time != 1e-5:
            results["nll_correct_train_eps_bpd"] = nll_pass("NLL CORRECT w/ eps=eps", True,
                                                            config.training.truncation_time)
        if config.flow.model != "identity":
            flow_model.train()
    return results
