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


# ----------------------------------------------------------------------------- FID / IS statistics
# The reference feeds samples through an Inception-v3 it downloads at run time (TF-Hub / clean-fid weights:
# evaluation.py:100-158, cleanfid/features.py) and reduces the pooled features / logits below.  The network is an
# external asset that is not part of the reference tree (and there is no network here): callers pass the features.
def feature_statistics(pool3):
    """mean [D] and covariance [D, D] (rowvar=False, unbiased) of pooled features [N, D] (cleanfid/fid.py:101-103)"""
    f = np.asarray(pool3, dtype=np.float64)
    return f.mean(axis=0), np.cov(f, rowvar=False)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """||mu1 - mu2||^2 + Tr(S1 + S2 - 2 (S1 S2)^(1/2)) with the singular-product fallback of cleanfid/fid.py:27-80"""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    if mu1.shape != mu2.shape or sigma1.shape != sigma2.shape:
        raise ValueError("feature statistics of different dimensions")
    diff = mu1 - mu2
    covmean = linalg.sqrtm(sigma1.dot(sigma2))
    if not np.isfinite(covmean).all():
        off = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + off).dot(sigma2 + off))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError(f"Imaginary component {np.max(np.abs(covmean.imag))}")
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2.0 * np.trace(covmean))


def fid_from_features(pool3_samples, pool3_data):
    return frechet_distance(*feature_statistics(pool3_samples), *feature_statistics(pool3_data))


def inception_score_from_logits(logits):
    """exp(E_x KL(p(y|x) || p(y)))  (tfgan.eval.classifier_score_from_logits, evaluation.py:365)"""
    z = np.asarray(logits, dtype=np.float64)
    logp = z - z.max(axis=1, keepdims=True)
    logp = logp - np.log(np.exp(logp).sum(axis=1, keepdims=True))
    p = np.exp(logp)
    log_py = np.log(p.mean(axis=0, keepdims=True))
    return float(np.exp((p * (logp - log_py)).sum(axis=1).mean()))


def inception_scores_10_splits(all_logits, split=5000):
    """evaluation.py:352-374: the score of each of the 10 consecutive 5000-sample splits"""
    out = []
    for k in range(10):
        part = all_logits[k * split:(k + 1) * split]
        if len(part) != split:
            raise ValueError(f"split {k} holds {len(part)} logits, expected {split}")
        out.append(inception_score_from_logits(part))
    return out
