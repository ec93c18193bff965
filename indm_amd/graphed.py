"""The score network's training call (forward + backward) as two captured hipGraphs.

At the per-rank shards of every multi-GPU BASELINE configuration (16 / 32 / 64 images) the training step is bound by the
HOST: ~4 800 launches per step at 9 us of Python each (tools/host_profile.py), of which the NCSN++ forward and backward are
about half.  The network's launch sequence does not depend on the data -- same shapes, same kernels, same order every step --
so it is captured ONCE per (batch shape, mode) and replayed: one launch for the forward, one for the backward
(`hipGraphLaunch`), the way models/utils.py:96-125's `model_fn` is called by losses.py:103-118 every step.

What makes the capture valid:
  * inputs live in static buffers (x, the time labels); the per-call dropout seed lives in DEVICE memory and the GroupNorm
    kernels read it when they run (indm_groupnorm_*: `seed_dev`), so every replay draws a fresh mask;
  * every intermediate the backward needs sits in the graphs' private memory pool at a fixed address (torch.cuda.graph);
  * weights are read through the parameters' own storage (the flat buffer of optim.FlatParams): an optimiser step between
    two replays is seen by the next replay; the bf16 weight packs of the H16 path are re-packed inside the forward graph;
  * the parameter gradients the backward graph leaves in static buffers are copied into the flat gradient buffer by ONE
    multi-tensor launch (persistent table) and the parameters' .grad point at their flat slices.
The eager path stays the reference for parity: tests/test_graphed_gpu.py compares the two bit for bit (dropout 0).
Disable with INDM_SCORE_GRAPH=0.
"""
import os

import torch
from torch.autograd import Function

from . import hip

ENABLED = {"on": os.environ.get("INDM_SCORE_GRAPH", "1") != "0"}


def enabled(on=None):
    if on is not None:
        ENABLED["on"] = bool(on)
    return ENABLED["on"]


class _SeedRng:
    """what the network sees as `rng` while it is captured: consecutive constants per dropout site; the kernels add the
    device-resident per-call seed (hip.SEED_DEV) to them"""

    def __init__(self):
        self.i = 0

    def dropout_seed(self):
        self.i += 1
        return self.i * 0x9E3779B1


class _Replay(Function):
    @staticmethod
    def forward(ctx, g, x, labels, anchor):
        # (anchor: a leaf that requires grad -- the parameters are not inputs of this node, and x may be detached (the fid
        # step's second phase): the node must be differentiable regardless, its backward delivers the parameter gradients)
        g.sx.copy_(x)
        g.sl.copy_(labels)
        g.fwd.replay()
        ctx.g = g
        return g.sout.detach()

    @staticmethod
    def backward(ctx, gout):
        g = ctx.g
        g.sgout.copy_(gout)
        g.bwd.replay()
        for k, v in g.h16_delta.items():
            hip._H16_STATS[k] += v
        g.deliver_param_grads()
        return None, g.sgx.detach(), None, None


class GraphedScore:
    """One captured (forward, backward) pair of `net(x, labels, rng=..., nhwc=True)` in training mode for one input shape."""

    WARMUP = 2

    def __init__(self, net, flat, x, labels):
        self.net, self.flat = net, flat
        dev = x.device
        self.key = self.signature(net, x, labels)
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.anchor = torch.zeros(1, device=dev, requires_grad=True)
        self.sx = x.detach().clone().requires_grad_(True)
        self.sl = labels.detach().clone()
        params = [p for _, p in flat.named]
        saved_grads = [p.grad for p in params]
        # The capture differentiates w.r.t. fresh leaf ALIASES of the parameters (same storage), not the parameters themselves:
        # a parameter's AccumulateGrad node may be kept alive from an earlier eager pass (someone still holds that pass's output);
        # it remembers the stream it was created on, and autograd would then order the capturing stream after the default
        # stream -- an illegal dependency inside a capture (it crashed the process, tests/parity_cases.py::case_celeba_squeeze).
        aliases = {n: p.detach().requires_grad_(True) for n, p in flat.named}
        leaves = list(aliases.values())
        for a in leaves:
            a._indm_param_alias = True      # hip._packed_weight must not cache re-laid-out copies of these (see there)

        def run():
            return torch.func.functional_call(net, aliases, (self.sx, self.sl), {"rng": _SeedRng(), "nhwc": True})

        hip.SEED_DEV[0] = self.seed_dev
        try:
            # eager warm-up on a side stream: lazy initialisation (LDS opt-ins, workspaces, tables, packers) happens here,
            # not inside the capture
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(self.WARMUP):
                    out = run()
                    grads = torch.autograd.grad(out, [self.sx] + leaves, torch.ones_like(out), allow_unused=True)
                    del out, grads
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            pool = torch.cuda.graph_pool_handle()
            before = dict(hip._H16_STATS)
            self.fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.fwd, pool=pool):
                self.sout = run()
            self.sgout = torch.zeros_like(self.sout)
            self.bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.bwd, pool=pool):
                grads = torch.autograd.grad(self.sout, [self.sx] + leaves, self.sgout, allow_unused=True)
            self.sgx = grads[0]
            # what one (forward, backward) pair launches on the H16 kernels: replays add it to hip.h16_stats() -- the coverage
            # counters keep meaning "launched" when the launches come out of a graph
            self.h16_delta = {k: hip._H16_STATS[k] - before[k] for k in before}
        finally:
            hip.SEED_DEV[0] = None
            for p, g in zip(params, saved_grads):
                p.grad = g
        # static parameter gradients -> their slices of the flat gradient buffer: one persistent copy table
        srcs, dsts, self.views = [], [], []
        for (p, dst, _), g in zip(flat.gslices, grads[1:]):
            if g is None:
                continue
            if not g.is_contiguous():   # (a copy made here would be made once, not on every replay)
                raise RuntimeError("graphed: a parameter gradient left the captured backward non-contiguous")
            srcs.append(g)
            dsts.append(dst)
            self.views.append((p, dst.view(p.shape)))
        self._static_param_grads = srcs     # keep the graph-pool tensors referenced
        self._aliases = aliases
        # the captured kernels hold the raw address of the library's scratch buffer: keep THAT tensor alive even if a later,
        # larger request replaces the cached one (hip._ws grows by reallocation)
        self._ws_ref = list(hip._ws_cache.values())
        self.copy_table = hip.multi_copy_table(srcs, dsts)

    @staticmethod
    def signature(net, x, labels):
        return (tuple(x.shape), x.dtype, tuple(labels.shape), labels.dtype, hip.compute_dtype(), hip.h16(),
                tuple(p.data_ptr() for p in list(net.parameters())[:4]))

    def deliver_param_grads(self):
        hip.multi_copy_run(self.copy_table)
        for p, v in self.views:
            p.grad = v
        red = getattr(self.flat, "_reducer_hooks", None)
        if red is not None and hasattr(red[0], "grads_ready"):
            red[0].grads_ready(self.flat)      # data parallel: the score slice's all-reduce starts here

    def __call__(self, x, labels, seed):
        if seed is not None:
            self.seed_dev.fill_(int(seed) & ((1 << 62) - 1))
        return _Replay.apply(self, x, labels, self.anchor)


_cache = {}


def _profiling_the_score_net():
    """a launch profile is being recorded that could contain kernels of the score network (hip.profile_start() without a class,
    or with a class other than the flow's chain kernel -- bench.py's timed region brackets only that one)"""
    P = hip._Profile
    if not P.enabled:
        return False
    return P.only is None or not P.only.split(",", 2)[-1].startswith("ires_chain")


def score_call(net, flat, x, labels, rng):
    """net(x, labels, rng=rng, nhwc=True) in training mode through the captured graphs when that is possible (GPU tensors,
    graphs enabled, gradients on, the network's FlatParams known, a noise source that can seed dropout), else eagerly."""
    drop = float(net.config.model.dropout) if net.training else 0.0
    if not (ENABLED["on"] and x.is_cuda and net.training and torch.is_grad_enabled() and flat is not None
            and not _profiling_the_score_net()                # per-launch HIP-event timing (bench / tests) needs the launches
            and not torch.cuda.is_current_stream_capturing()):
        return net(x, labels, rng=rng, nhwc=True)
    seed = None
    if drop > 0:
        if rng is None:
            from .rng import default_rng
            rng = default_rng(x.device)
        seed = rng.dropout_seed()
        if seed is None:
            raise RuntimeError("training with dropout > 0 on a replayed noise source: the parity fixtures are recorded with "
                               "model.dropout = 0")
    key = (id(net), GraphedScore.signature(net, x, labels))
    g = _cache.get(key)
    if g is None:
        if len(_cache) >= 4:        # (a changing batch shape must not pin one memory pool per shape)
            _cache.pop(next(iter(_cache)))
        g = _cache[key] = GraphedScore(net, flat, x, labels)
    return g(x, labels, seed)


def clear():
    _cache.clear()
