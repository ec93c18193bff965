"""The reference's experiment configurations, restated as data.

Values follow configs/default_cifar10_configs.py:5-133, configs/default_celeba_configs.py,
configs/{vp,ve}/{CIFAR10,CELEBA}/*.py and the wolf JSON
flow_models/wolf/wolf_configs/cifar10/glow/resflow-gaussian-uni.json:1-34 (CelebA:
wolf_configs/imagenet/64x64/glow/resflow-gaussian-uni.json).  `flow.wolf` carries the JSON's
content so no file path has to be resolved at run time.
"""
import copy


class ConfigDict(dict):
    """Attribute-access dict (stand-in for ml_collections.ConfigDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wolf(in_planes):
    return ConfigDict(
        encoder=ConfigDict(levels=3, in_planes=in_planes, hidden_planes=[48, 96, 96], out_planes=8, activation="elu"),
        in_dim=128, dim=64,
        prior=ConfigDict(num_steps=2, in_features=64, hidden_features=256, activation="elu", transform="affine",
                         alpha=1.0, coupling_type="mlp"))


def _defaults(dataset):
    c = ConfigDict()
    c.training = ConfigDict(batch_size=128, n_iters=13000001, snapshot_freq=10000, log_freq=100, eval_freq=100,
                            snapshot_freq_for_preemption=10000, snapshot_sampling=True, likelihood_weighting=True,
                            continuous=True, reduce_mean=False, importance_sampling=True,
                            unbounded_parametrization=False, ddpm_score=True, st=False, k=1.2, truncation_time=1e-5,
                            num_train_data=50000, reconstruction_loss=False)
    c.sampling = ConfigDict(n_steps_each=1, noise_removal=True, probability_flow=False,
                            snr=0.16 if dataset == "CIFAR10" else 0.15, batch_size=1024, truncation_time=1e-5,
                            temperature=1.0, need_sample=True, idx_rand=True, pc_denoise=False, pc_denoise_time=0.0,
                            more_step=False, num_scales=1000, pc_ratio=1.0, begin_snr=0.16, end_snr=0.16,
                            snr_scheduling="none")
    c.eval = ConfigDict(batch_size=200, enable_sampling=True, num_samples=50000, enable_loss=True, enable_bpd=True,
                        bpd_dataset="test", num_test_data=10000 if dataset == "CIFAR10" else 19962, residual=False,
                        score_ema=True, flow_ema=False, num_nelbo=3, rtol=1e-5, atol=1e-5, gap_diff=False,
                        target_ckpt=-1, truncation_time=-1.0, data_mean=False, skip_nll_wrong=False)
    c.data = ConfigDict(dataset=dataset, image_size=32 if dataset == "CIFAR10" else 64, random_flip=True,
                        centered=False, num_channels=3)
    c.model = ConfigDict(sigma_min=0.01, sigma_max=50.0 if dataset == "CIFAR10" else 90.0, num_scales=1000,
                         beta_min=0.1, beta_max=20.0, dropout=0.1, embedding_type="fourier", auxiliary_resblock=True,
                         attention=True, fourier_feature=False)
    c.optim = ConfigDict(optimizer="AdamW", weight_decay=0.01, lr=2e-4, beta1=0.9, eps=1e-8, warmup=0, grad_clip=1.0,
                         num_micro_batch=1, reset=True, amsgrad=False)
    c.flow = ConfigDict(model="identity", lr=1e-3, ema_rate=0.999, optim_reset=False, nblocks="16-16",
                        intermediate_dim=512, resblock_type="resflow", squeeze=dataset != "CIFAR10", actnorm=False,
                        grad_in_forward=False, act_fn="sin")
    c.seed = 42
    # not a reference key: 'f32' (exact fp32 MFMA, the parity gate) or 'bf16' (MFMA inputs rounded to bf16, fp32
    # accumulate: the "bf16 compute / fp32 accum" of BASELINE configs #3-#5); applied by losses.get_step_fn & friends
    c.compute_dtype = "f32"
    return c


def _ncsnpp(c):
    m = c.model
    m.update(name="ncsnpp", normalization="GroupNorm", nonlinearity="swish", nf=128, ch_mult=(1, 2, 2, 2),
             num_res_blocks=4, attn_resolutions=(16,), resamp_with_conv=True, conditional=True,
             fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type="biggan", progressive="none",
             progressive_combine="sum", attention_type="ddpm", init_scale=0.0, fourier_scale=16, conv_size=3)
    f = c.flow
    f.update(model="wolf", lr=1e-3, ema_rate=0.999, optim_reset=False, nblocks="16-16", intermediate_dim=512,
             resblock_type="resflow", n_bits=8, train_k=1,
             wolf=_wolf(3 if not f.squeeze else 12))
    return c


def vp_cifar10_indm_nll():
    """configs/vp/CIFAR10/indm_nll.py"""
    c = _ncsnpp(_defaults("CIFAR10"))
    c.training.update(sde="vpsde", continuous=True, reduce_mean=True)
    c.sampling.update(method="ode", predictor="euler_maruyama", corrector="none")
    c.data.centered = True
    c.model.update(scale_by_sigma=False, ema_rate=0.9999, fir=False, progressive_input="none",
                   embedding_type="positional")
    return c


def vp_cifar10_indm_fid():
    """configs/vp/CIFAR10/indm_fid.py"""
    c = vp_cifar10_indm_nll()
    c.training.update(likelihood_weighting=False, importance_sampling=False)
    return c


def ve_cifar10_indm():
    """configs/ve/CIFAR10/indm.py"""
    c = _ncsnpp(_defaults("CIFAR10"))
    c.training.update(sde="vesde", continuous=True, likelihood_weighting=True, importance_sampling=True)
    c.sampling.update(method="pc", predictor="reverse_diffusion", corrector="langevin")
    c.model.update(scale_by_sigma=True, ema_rate=0.999, fir=True, progressive_input="residual")
    return c


def vp_celeba_indm_fid():
    """configs/vp/CELEBA/indm_fid.py"""
    c = _ncsnpp(_defaults("CELEBA"))
    c.training.update(sde="vpsde", continuous=True, reduce_mean=True, likelihood_weighting=False,
                      importance_sampling=False)
    c.sampling.update(method="ode", predictor="euler_maruyama", corrector="none")
    c.data.centered = True
    c.model.update(scale_by_sigma=False, ema_rate=0.9999, fir=False, progressive_input="none",
                   embedding_type="positional")
    return c


def ve_celeba_indm():
    """configs/ve/CELEBA/indm.py"""
    c = _ncsnpp(_defaults("CELEBA"))
    c.training.update(sde="vesde", continuous=True, likelihood_weighting=True, importance_sampling=True)
    c.sampling.update(method="pc", predictor="reverse_diffusion", corrector="langevin")
    c.model.update(scale_by_sigma=True, ema_rate=0.999, fir=True, progressive_input="residual")
    return c


REGISTRY = {
    "vp/CIFAR10/indm_nll": vp_cifar10_indm_nll,
    "vp/CIFAR10/indm_fid": vp_cifar10_indm_fid,
    "ve/CIFAR10/indm": ve_cifar10_indm,
    "vp/CELEBA/indm_fid": vp_celeba_indm_fid,
    "ve/CELEBA/indm": ve_celeba_indm,
}


def get_config(name):
    return REGISTRY[name]()


def mini(c, batch=2):
    """The shrunken configuration the golden fixtures were captured on (tests/golden/make_golden.py)."""
    c.model.update(nf=16, ch_mult=(1, 2), num_res_blocks=1, dropout=0.0)
    c.flow.update(nblocks="2-2", intermediate_dim=32)
    c.flow.wolf.encoder.hidden_planes = [8, 16, 16]
    c.flow.wolf.prior.hidden_features = 32
    c.training.batch_size = batch
    c.sampling.batch_size = batch
    c.sampling.num_scales = 3
    c.eval.batch_size = batch
    return c
