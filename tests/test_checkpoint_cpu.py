"""Checkpoint boundary: a checkpoint written by the product after one training step has the
REFERENCE's format -- the oracle (which uses torch.optim.AdamW like the reference) can load it and continue
with the same second step, and the product can resume from it bit-for-bit."""
import torch

import golden_utils as gu
import indm_oracle as O
import parity_cases as pc
from indm_amd import losses, sde_lib, utils
from indm_amd.rng import ReplayRng


def test_checkpoint_roundtrip_and_reference_format(emulate_hip, tmp_path):
    fx = "mini_vp_nll"
    d, cfg, score, fl = pc.build(fx, "cpu")
    sde = sde_lib.get_sde(cfg)
    state, fstate, _ = utils.build_states(cfg, score, fl, "cpu")
    step_fn = losses.get_step_fn(cfg, sde, train=True)
    score.train(); fl.train()
    step_fn(state, fstate, torch.from_numpy(d["step0/x"]), rng=ReplayRng(gu.events(d, "step0"), "cpu"))
    utils.save_checkpoint(cfg, str(tmp_path / "checkpoint.pth"), state)
    utils.save_checkpoint(cfg, str(tmp_path / "flow_checkpoint.pth"), fstate)

    ck = torch.load(str(tmp_path / "checkpoint.pth"), weights_only=False)
    assert set(ck) == {"optimizer", "model", "ema", "step"} and ck["step"] == 1
    assert all(k.startswith("module.") for k in ck["model"])
    assert ck["model"]["module.all_modules.2.weight"].shape == (16, 3, 3, 3)          # OIHW like the reference
    assert len(ck["ema"]["shadow_params"]) == len(d["score_param_order"])

    # (1) product resumes from its own checkpoint and reproduces step 2 of the golden run
    _, _, score2, fl2 = pc.build(fx, "cpu")
    st2, fst2, _ = utils.build_states(cfg, score2, fl2, "cpu")
    utils.restore_checkpoint(cfg, str(tmp_path / "checkpoint.pth"), st2)
    utils.restore_checkpoint(cfg, str(tmp_path / "flow_checkpoint.pth"), fst2)
    score2.train(); fl2.train()
    out = step_fn(st2, fst2, torch.from_numpy(d["step1/x"]), rng=ReplayRng(gu.events(d, "step1"), "cpu"))
    for nm, v in zip(("loss", "loss_score", "loss_flow", "loss_logp"), out):
        gu.assert_close(v, d[f"step1/{nm}"], 5e-4, 2e-5, "resumed " + nm)
    assert st2["step"] == 2 and st2["ema"].num_updates == 2

    # (2) the oracle (reference-style torch.optim.AdamW / EMA lists) loads the same files and continues
    fck = torch.load(str(tmp_path / "flow_checkpoint.pth"), weights_only=False)
    ost = O.OracleState({k[len("module."):]: v for k, v in ck["model"].items()}, cfg.optim.lr, cfg.model.ema_rate, cfg)
    ofst = O.OracleState({k[len("module."):]: v for k, v in fck["model"].items()}, cfg.flow.lr, cfg.flow.ema_rate, cfg)
    for o, c in ((ost, ck), (ofst, fck)):
        # torch's AdamW indexes params of ITS param list; the oracle's list = trainable tensors that receive grads
        # plus lamb/geom_p -> same order as model.parameters() restricted to requires_grad
        o.opt.load_state_dict(c["optimizer"])
        o.ema = [t.clone() for t in c["ema"]["shadow_params"]]
        o.num_updates = c["ema"]["num_updates"]
        o.step = c["step"]
    want = O.flow_step_fn_nll(cfg, O.get_sde(cfg), ost, ofst, torch.from_numpy(d["step1/x"]),
                              O.ReplayRng(gu.events(d, "step1")))
    for nm, a, b in zip(("loss", "loss_score", "loss_flow", "loss_logp"), out, want):
        gu.assert_close(a, b, 5e-4, 2e-5, "oracle-resumed " + nm)
    k = "all_modules.2.weight"
    gu.assert_close(score2.reference_state_dict(prefix="")[k], ost.P[k], 1e-4, 2e-6, "weights after resumed step")
