"""Data-parallel path on CPU: world_size-2 gloo job (kernel layer emulated) must reproduce the 1-rank step on
the concatenated batch: same averaged gradients -> same parameters after the fused optimiser step."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _patch_emu():
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import _emu
    from indm_amd import hip
    for name in dir(_emu):
        if not name.startswith("_") and callable(getattr(_emu, name)) and hasattr(hip, name):
            setattr(hip, name, getattr(_emu, name))


def _run_steps(rank, world, port, out_path, fx="mini_vp_nll"):
    _patch_emu()
    import golden_utils as gu
    from indm_amd import flow as iflow
    from indm_amd import losses, ncsnpp, parallel, sde_lib, utils
    from indm_amd.rng import ReplayRng
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_distributed(backend="gloo")
    d, cfg = gu.load(fx), gu.mini_config(fx)
    score = ncsnpp.NCSNpp(cfg); score.load_reference_state_dict(gu.state_dict(d, "score"))
    fl = iflow.create_flow_model(cfg); fl.load_reference_state_dict(gu.state_dict(d, "flow"))
    state, fstate, flat_g = utils.build_states(cfg, score, fl, "cpu")
    reducer = parallel.make_allreduce(flat_g, world)
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True, world_size=world, allreduce=reducer)
    score.train(); fl.train()
    # each rank trains on ONE image of the golden batch with that image's slice of the recorded noise
    ev = []
    for k, v in gu.events(d, "step0"):
        ev.append((k, v if k == "poisson" else v[rank:rank + 1] if world > 1 else v))
    x = torch.from_numpy(d["step0/x"])
    xb = parallel.shard_batch(x, rank, world)
    out = step_fn(state, fstate, xb, rng=ReplayRng(ev, "cpu"))
    # score-only step on a FIXED latent batch: no per-rank BatchNorm in the way, so the 2-rank result must equal
    # the 1-rank result on the full batch (mean of per-rank means == global mean for equal shards)
    score2 = ncsnpp.NCSNpp(cfg); score2.load_reference_state_dict(gu.state_dict(d, "score"))
    (f2,), _, g2 = __import__("indm_amd.optim", fromlist=["x"]).flatten_models([score2])
    opt2 = losses.get_optimizer(cfg, f2)
    loss_fn = losses.get_sde_loss_fn(cfg, sde_lib.get_sde(cfg), train=True)
    zfix = torch.from_numpy(d["flow_eval/z"])
    u = torch.tensor([0.3, 0.7])
    g = torch.Generator().manual_seed(7)
    noise = torch.randn(2, 3, 32, 32, generator=g).numpy()
    sl = slice(rank, rank + 1) if world > 1 else slice(0, 2)
    opt2.zero_grad()
    ls = loss_fn(score2, parallel.shard_batch(zfix, rank, world) if world > 1 else zfix,
                 rng=ReplayRng([("rand", u[sl].numpy()), ("randn", noise[sl])], "cpu"))
    ls.mean().backward()
    f2.gather_grads()            # autograd's per-parameter grads -> the flat buffer, before anything reads it
    if world > 1:
        parallel.make_allreduce(g2, world)()
    opt2.step(grad_clip=cfg.optim.grad_clip, grad_scale=1.0 / world)
    torch.save({"loss": out[0], "p": state["optimizer"].flat.p.clone(), "pf": fstate["optimizer"].flat.p.clone(),
                "p2": f2.p.clone(), "g2": g2.clone() / world, "calls": list(reducer.calls) if reducer else [],
                "sizes": (state["optimizer"].flat.g.numel(), fstate["optimizer"].flat.g.numel())},
               out_path.format(rank=rank))
    if world > 1:
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_step_matches_single_rank(tmp_path):
    port = 29500 + os.getpid() % 1000
    path = str(tmp_path / "w{rank}.pt")
    mp.spawn(_run_steps, args=(2, port, path), nprocs=2, join=True)
    a, b = torch.load(path.format(rank=0)), torch.load(path.format(rank=1))
    assert torch.equal(a["p"], b["p"]) and torch.equal(a["pf"], b["pf"])  # replicas stay in lock-step
    # flow_step_fn_nll: the score slice is reduced from INSIDE the backward pass (overlapping the flow's backward), the
    # flow slice after it -- each gradient element crosses the wire exactly once
    ns, nf = a["sizes"]
    assert a["calls"] == [ns, nf] == b["calls"], (a["calls"], ns, nf)
    # single-process reference on the full batch (BatchNorm statistics are per-rank by design, so compare the
    # score network, whose loss has no cross-image coupling given the same z; the flow differs through BN only)
    single = str(tmp_path / "s{rank}.pt")
    _run_steps(0, 1, port + 1, single)
    s = torch.load(single.format(rank=0))
    assert torch.isfinite(a["loss"]).all() and torch.isfinite(s["loss"]).all()
    assert torch.equal(a["p2"], b["p2"])
    gerr = (a["g2"] - s["g2"]).abs().max().item() / s["g2"].abs().max().item()
    assert gerr < 1e-5, gerr
    perr = (a["p2"] - s["p2"]).abs().max().item()
    assert perr < 1e-6, perr


@pytest.mark.timeout(600)
def test_two_rank_fid_step_reduces_only_the_stepped_slice(tmp_path):
    """flow_step_fn_fid on 2 ranks: phase 1 all-reduces the flow slice only, phase 2 the score slice only (SURVEY 8e:
    52.7 MB then 247 MB at full size -- not the joint 300 MB buffer twice); replicas end bit-identical."""
    port = 29500 + (os.getpid() + 7) % 1000
    path = str(tmp_path / "f{rank}.pt")
    mp.spawn(_run_steps, args=(2, port, path, "mini_vp_fid"), nprocs=2, join=True)
    a, b = torch.load(path.format(rank=0)), torch.load(path.format(rank=1))
    assert torch.equal(a["p"], b["p"]) and torch.equal(a["pf"], b["pf"])
    ns, nf = a["sizes"]
    assert a["calls"] == [nf, ns] == b["calls"], (a["calls"], ns, nf)
    assert torch.isfinite(a["loss"]).all()


class _RecOracleRng:
    """oracle noise source that records every draw (kind, array) so that the ranks can replay their slices"""

    def __init__(self, base):
        self.base, self.events = base, []

    def _rec(self, kind, v):
        import numpy as np
        self.events.append((kind, np.asarray(v.numpy() if torch.is_tensor(v) else v).copy()))
        return v

    def randn(self, *shape):
        return self._rec("randn", self.base.randn(*shape))

    def randn_like(self, x):
        return self._rec("randn", self.base.randn_like(x))

    def rand(self, *shape):
        return self._rec("rand", self.base.rand(*shape))

    def poisson(self, lam):
        return int(self._rec("poisson", self.base.poisson(lam)))

    def np_rand(self):
        return float(self._rec("np_rand", self.base.np_rand()))


def _st_config():
    import golden_utils as gu
    cfg = gu.mini_config("mini_vp_fid")
    cfg.training.st = True
    cfg.optim.grad_clip = -1.0      # keep the raw gradients comparable (no in-place clip)
    return cfg


def _run_st(rank, world, port, blob_path, out_path):
    _patch_emu()
    import golden_utils as gu
    from indm_amd import flow as iflow
    from indm_amd import losses, ncsnpp, parallel, sde_lib, utils
    from indm_amd.rng import ReplayRng
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    parallel.init_distributed(backend="gloo")
    blob = torch.load(blob_path, weights_only=False)
    d, cfg = gu.load("mini_vp_fid"), _st_config()
    score = ncsnpp.NCSNpp(cfg); score.load_reference_state_dict(gu.state_dict(d, "score"))
    fl = iflow.create_flow_model(cfg); fl.load_reference_state_dict(gu.state_dict(d, "flow"))
    state, fstate, flat_g = utils.build_states(cfg, score, fl, "cpu")
    step_fn = losses.get_step_fn(cfg, sde_lib.get_sde(cfg), train=True, world_size=world,
                                 allreduce=parallel.make_allreduce(flat_g, world))
    score.train(); fl.train()
    x = blob["x"]
    per = x.shape[0] // world
    ev = [(k, v[rank * per:(rank + 1) * per] if (getattr(v, "ndim", 0) >= 1 and v.shape[0] == x.shape[0]) else v)
          for k, v in blob["events"]]
    out = step_fn(state, fstate, x[rank * per:(rank + 1) * per], rng=ReplayRng(ev, "cpu"))
    flat = state["optimizer"].flat
    grads, o = {}, 0
    for n, p in flat.named:
        grads[n] = score.to_reference_layout(n, flat.g[o:o + p.numel()].view(p.shape)).clone() / world
        o += (p.numel() + 3) // 4 * 4
    torch.save({"losses": out, "grads": grads, "p": flat.p.clone()}, out_path.format(rank=rank))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_st_fid_step_uses_the_global_batch_means(tmp_path):
    """training.st's re-scaling factor mean(la) / mean(losses_score) (losses.py:389-395) under data parallelism: the
    reference's single-process DataParallel forms both means over the WHOLE batch.  A world-size-2 gloo job on the st_fid
    golden's network must reproduce the single-process ORACLE step on the same global batch: score gradients equal, replicas
    bit-identical.  The global batch is [x0, x1, x0, x1] with DIFFERENT noise per image, so that per-rank BatchNorm
    statistics (the one place images couple in the flow) equal whole-batch statistics while the per-rank loss means differ."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import numpy as np
    import golden_utils as gu
    import indm_oracle as O
    d, d3, cfg = gu.load("mini_vp_fid"), gu.load("mini_r3"), _st_config()
    x2 = torch.from_numpy(d3["st_fid/x"])[:2]
    x = torch.cat([x2, x2])
    torch.manual_seed(1234); np.random.seed(99)
    ost = O.OracleState(gu.state_dict(d, "score"), cfg.optim.lr, cfg.model.ema_rate, cfg)
    ofst = O.OracleState(gu.state_dict(d, "flow"), cfg.flow.lr, cfg.flow.ema_rate, cfg)
    rec = _RecOracleRng(O.Rng())
    want = O.flow_step_fn_fid(cfg, O.get_sde(cfg), ost, ofst, x, rec)
    ograds = {k: ost.P[k].grad.clone() for k in ost.trainable if ost.P[k].grad is not None}
    blob = str(tmp_path / "blob.pt")
    torch.save({"x": x, "events": rec.events}, blob)
    port = 29500 + (os.getpid() + 13) % 1000
    path = str(tmp_path / "st{rank}.pt")
    mp.spawn(_run_st, args=(2, port, blob, path), nprocs=2, join=True)
    a, b = torch.load(path.format(rank=0), weights_only=False), torch.load(path.format(rank=1), weights_only=False)
    assert torch.equal(a["p"], b["p"])                                   # replicas stay in lock-step
    got = [torch.cat([a["losses"][i], b["losses"][i]]) for i in range(4)]
    for i, nm in enumerate(("loss", "score", "flow", "logp")):
        gu.assert_close(got[i], want[i], 2e-4, 1e-5, f"st 2-rank {nm}")
    # the test discriminates: the per-rank factors differ from the global one by far more than the tolerance below
    la, ls = want[1], want[0] - want[2] - want[3]
    glob = (la.mean() / ls.mean()).item()
    per_rank = [(la[r * 2:r * 2 + 2].mean() / ls[r * 2:r * 2 + 2].mean()).item() for r in range(2)]
    assert max(abs(f - glob) for f in per_rank) > 1e-3 * abs(glob), (glob, per_rank)
    checked = 0
    for n, g in a["grads"].items():
        ref = ograds.get(n)
        if ref is None:
            continue
        err = (g - ref).abs().max().item()
        assert err <= 1e-4 * ref.abs().max().item() + 1e-7, (n, err, ref.abs().max().item())
        checked += 1
    assert checked > 50
