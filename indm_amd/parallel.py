"""Data parallelism: one process per GPU, the flat gradient buffer all-reduced over RCCL/xGMI once per step (each
element crosses the wire once: the score slice from inside the backward pass, the flow slice after it -- GradReducer).

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


def _active(world_size):
    return world_size > 1 or (dist.is_initialized() and os.environ.get("INDM_FORCE_DIST") == "1")


class GradReducer:
    """The step's gradient exchange.  `reducer(flats)` all-reduces (SUM) exactly the gradient slices of the
    FlatParams being stepped: the joint 300 MB buffer in ONE collective when both networks step together
    (flow_step_fn_nll), the 52.7 MB flow slice and then the 247 MB score slice in flow_step_fn_fid.

    Overlap (flow_step_fn_nll): backward reaches the score network first, so `arm(flat)` + the per-parameter
    post-accumulate hooks launch the score slice's all-reduce asynchronously the moment its last gradient exists,
    while the flow's backward (~40 % of the step) is still running; the call in `reducer(...)` then only reduces the
    flow slice and waits for the early one.  RCCL collectives run on their own stream; `work.wait()` orders the
    optimiser kernels after them without blocking the host."""

    def __init__(self, flat_grad, world_size):
        self.flat_grad, self.world_size = flat_grad, world_size
        self.pending = {}      # id(flat) -> (flat, work) launched early; entries live only within one step
        self.armed = {}        # id(flat) -> [flat, remaining parameter count]; entries live only within one step
        self.calls = []        # element counts of the collectives of the CURRENT step (reset by begin_step())

    def begin_step(self):
        """called by the step functions before the first backward pass of a step"""
        self.calls = []
        self.armed.clear()

    def _launch(self, t, async_op=False):
        self.calls.append(t.numel())
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)

    def arm(self, flat):
        """Start flat's all-reduce from inside the backward pass as soon as every parameter has its gradient."""
        key = id(flat)
        # the hook handles live ON the FlatParams object (not in a set of ids a later object could re-use)
        if getattr(flat, "_reducer_hooks", None) is None or flat._reducer_hooks[0] is not self:
            for h in (getattr(flat, "_reducer_hooks", None) or (None, []))[1]:
                h.remove()
            flat._reducer_hooks = (self, [p.register_post_accumulate_grad_hook(lambda _p, f=flat: self._on_grad(id(f)))
                                          for _, p in flat.named])
        self.armed[key] = [flat, len(flat.named)]

    def _on_grad(self, key):
        a = self.armed.get(key)
        if a is None:
            return
        a[1] -= 1
        if a[1] == 0:
            del self.armed[key]
            flat = a[0]
            flat.gather_grads()
            self.pending[key] = (flat, self._launch(flat.g, async_op=True))

    def grads_ready(self, flat):
        """every gradient of `flat` is in its slice of the flat buffer already (the captured backward graph of the score
        network, graphed.py, delivers them with one copy instead of one AccumulateGrad per parameter): what the last
        per-parameter hook would have done"""
        a = self.armed.pop(id(flat), None)
        if a is not None:
            flat.gather_grads()
            self.pending[id(flat)] = (flat, self._launch(flat.g, async_op=True))

    def __call__(self, flats=None):
        self.armed.clear()
        if flats is None:
            self._launch(self.flat_grad)
            return
        todo = [f for f in flats if id(f) not in self.pending]
        if len(todo) > 1 and sum(f.g.numel() for f in todo) == self.flat_grad.numel():
            self._launch(self.flat_grad)             # everything at once: the single collective of the step
        else:
            for f in todo:
                self._launch(f.g)
        for f in flats:
            pend = self.pending.pop(id(f), None)
            if pend is not None:
                pend[1].wait()


def make_allreduce(flat_grad, world_size):
    """The step's collective on the joint gradient buffer (None on a single rank)."""
    if not _active(world_size):
        return None
    return GradReducer(flat_grad, world_size)


def shard_batch(batch, rank, world_size):
    """Even split of the global batch across ranks (independent images -> no data-path collective)."""
    per = batch.shape[0] // world_size
    return batch[rank * per:(rank + 1) * per]


def broadcast_parameters(flat_params, src=0):
    if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("INDM_FORCE_DIST") == "1"):
        dist.broadcast(flat_params, src=src)
