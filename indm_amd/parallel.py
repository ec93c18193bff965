"""Data parallelism: one process per GPU, ONE RCCL all-reduce of the flat gradient buffer per step.

The reference uses single-process nn.DataParallel (models/utils.py:93, flow_models/flow_model.py:109):
per-step parameter broadcast + scatter/gather + gradient reduce onto device 0.  Here every rank owns full
replicas whose gradients live in one flat fp32 buffer (optim.flatten_models), so a step needs a single
`all_reduce(SUM)` over xGMI; the 1/world_size averaging is folded into the fused clip+AdamW kernel.
Images are independent units: the batch is sharded evenly, BatchNorm statistics stay per-rank (which is
what DataParallel computes per replica) and the global-norm clip runs after the reduction.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("INDM_FORCE_DIST") == "1"  # exercise the RCCL path with a single rank (tests)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" IS RCCL on ROCm
        kw = {}
        if backend == "nccl":
            # bind this rank to its GPU BEFORE the communicator exists: barrier() and the first collective then
            # run on the right device instead of guessing it from the rank
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def make_allreduce(flat_grad, world_size):
    """Closure performing the step's single collective on the joint gradient buffer."""
    if world_size <= 1 and not (dist.is_initialized() and os.environ.get("INDM_FORCE_DIST") == "1"):
        return None

    def allreduce():
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)

    return allreduce


def shard_batch(batch, rank, world_size):
    """Even split of the global batch across ranks (independent images -> no data-path collective)."""
    per = batch.shape[0] // world_size
    return batch[rank * per:(rank + 1) * per]


def broadcast_parameters(flat_params, src=0):
    if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("INDM_FORCE_DIST") == "1"):
        dist.broadcast(flat_params, src=src)
