"""The reference's training loop (run_lib.py:44-121) on this library, end to end: dataset files -> batches ->
dequantise + scale -> train_step_fn -> reference-format checkpoints (resume included).  Not a product component (the
control plane stays the user's run_lib.py); it shows the switch-over of INTEGRATION.md in ~60 lines and is what
tests/test_train_example_gpu.py runs.

  python tools/train_example.py --config vp/CIFAR10/indm_nll --data /path/with/cifar-10-batches-py --workdir /tmp/run \
      --steps 100 --batch 128 [--dtype bf16]
  N GPUs:  python -m torch.distributed.run --nproc-per-node N tools/train_example.py ...   (batch = global batch)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from indm_amd import configs, datasets, parallel, rng as irng, sde_lib, utils


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vp/CIFAR10/indm_nll")
    ap.add_argument("--data", required=True)
    ap.add_argument("--workdir", required=True)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--batch", type=int, default=128, help="global batch")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--snapshot-freq", type=int, default=50)
    ap.add_argument("--log-freq", type=int, default=10)
    args = ap.parse_args()

    rank, world, local = parallel.init_distributed()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    cfg = configs.get_config(args.config)
    cfg.device, cfg.compute_dtype = dev, args.dtype
    cfg.data.root = args.data
    cfg.training.batch_size = args.batch
    cfg.training.n_iters = args.steps
    torch.manual_seed(cfg.seed)                       # run_lib.py seeds once
    irng.manual_seed(cfg.seed + rank)

    state, flow_state, flat_g = utils.build_states(cfg, device=dev)
    ck = os.path.join(args.workdir, "checkpoints-meta")
    os.makedirs(ck, exist_ok=True)
    state = utils.restore_checkpoint(cfg, os.path.join(ck, "checkpoint.pth"), state, dev)          # run_lib.py:55
    flow_state = utils.restore_checkpoint(cfg, os.path.join(ck, "checkpoint_flow.pth"), flow_state, dev)
    if world > 1:
        parallel.broadcast_parameters(state["optimizer"].flat.p)
        parallel.broadcast_parameters(flow_state["optimizer"].flat.p)

    train_ds, _ = datasets.get_dataset(cfg, rank=rank, world=world)                                # run_lib.py:67
    train_iter = iter(train_ds)
    sde = sde_lib.get_sde(cfg)
    train_step_fn, _, _, _ = utils.get_loss_fns(cfg, sde, utils.get_data_inverse_scaler(cfg),
                                                scaler=utils.get_data_scaler(cfg), world_size=world,
                                                allreduce=parallel.make_allreduce(flat_g, world))
    state["model"].train(); flow_state["model"].train()
    noise = irng.default_rng(dev)
    for step in range(int(state["step"]), args.steps):
        batch, train_iter = datasets.get_batch(cfg, train_iter, train_ds)                          # run_lib.py:85
        batch = datasets.dequantize_and_scale(cfg, batch, noise)                                   # run_lib.py:86-87
        losses_, ls, lf, lp = train_step_fn(state, flow_state, batch)
        if rank == 0 and (step % args.log_freq == 0 or step == args.steps - 1):
            print("step: %d, loss mean: %.5e, score loss mean: %.5e, flow loss mean: %.5e, logp mean: %.5e"
                  % (step, losses_.mean(), ls.mean(), lf.mean(), lp.mean()), flush=True)
        if rank == 0 and ((step + 1) % args.snapshot_freq == 0 or step == args.steps - 1):
            utils.save_checkpoint(cfg, os.path.join(ck, "checkpoint.pth"), state)                  # run_lib.py:104-106
            utils.save_checkpoint(cfg, os.path.join(ck, "checkpoint_flow.pth"), flow_state)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
