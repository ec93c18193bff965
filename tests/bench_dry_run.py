"""CPU dry run of bench.py's multi-rank path (TEST INFRASTRUCTURE: launched by tests/test_bench_cpu.py under
torch.distributed.run with two gloo ranks).  The kernel layer is emulated (tests/_emu.py), the configuration is the mini
golden config, CUDA events / synchronisation are stand-ins -- what runs for real is everything an unattended 8-GPU run of
`bench.py --gpus N [--batch B | --global-batch G]` depends on besides the kernels: the rank / world bookkeeping, the per-rank
batch of weak and strong scaling, the gradient reducer, the barrier-bracketed timing with the MAX over ranks, the replay
without the collective behind `allreduce_exposed_ms`, and the fields of the N > 1 line.  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


class _Event:
    def __init__(self, enable_timing=True):
        self.t = None

    def record(self, *a):
        self.t = time.time()

    def elapsed_time(self, other):
        return 1e3 * (other.t - self.t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, required=True)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--global-batch", type=int, default=0)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()
    import _emu
    import golden_utils as gu
    from indm_amd import configs, hip, parallel
    for name in dir(_emu):
        if not name.startswith("_") and callable(getattr(_emu, name)) and hasattr(hip, name):
            setattr(hip, name, getattr(_emu, name))
    torch.cuda.Event = _Event
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda: None
    real_get_config = configs.get_config
    configs.get_config = lambda name: configs.mini(real_get_config(name))   # the mini golden configuration of the same file
    import bench
    rank, world, _ = parallel.init_distributed(backend="gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}")
    batch, strong = bench.per_rank_batch(args.batch, args.global_batch, world)
    r = bench.run_training(args, "vp/CIFAR10/indm_nll", batch, "f32", args.steps, args.warmup, rank, world, torch.device("cpu"),
                           kernel_roofline=False, exposed=True)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "scaling": "strong" if strong else "weak", "per_rank_batch": batch,
                          "global_batch": batch * world, "value": r["images_per_s"], "ms_per_step": r["ms_per_step"],
                          "allreduce_exposed_ms": r["allreduce_exposed_ms"], "allreduce_exposed_raw_ms": r["allreduce_exposed_raw_ms"],
                          "allreduce_exposed_detail": r["allreduce_exposed_detail"],
                          "ms_per_step_without_collective": r["ms_per_step_without_collective"],
                          "collective_elements_per_step": r["collectives_per_step"], "n_bar": r["n_bar"],
                          "backend": torch.distributed.get_backend()}))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
