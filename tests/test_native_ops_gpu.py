"""The reference's two native ops through their drop-in mirrors (indm_amd/op.py) against the golden vectors the
reference's own CPU fallbacks produced (tests/golden/native_ops.npz): forward, gradient and double gradient."""
import pytest
import torch

import golden_utils as gu
from indm_amd import op

pytestmark = pytest.mark.gpu


def test_upfirdn2d_golden_fwd_grad():
    d = gu.load("native_ops")
    for i, (c, h, w, up, down, p0, p1) in enumerate(d["cases"]):
        for kn in ("fir", "rand"):
            g = gu.sub(d, f"ufd{i}_{kn}")
            x = g["x"].cuda().requires_grad_(True)
            y = op.upfirdn2d(x, g["k"].cuda(), up=int(up), down=int(down), pad=(int(p0), int(p1)))
            gu.assert_close(y, g["y"], 1e-5, 1e-6, f"ufd{i} {kn}")
            (gx,) = torch.autograd.grad(y, x, g["gy"].cuda(), create_graph=True)
            gu.assert_close(gx, g["gx"], 1e-5, 1e-6, f"ufd{i} {kn} grad")
            # double backward: d/d(gy) <gx, v> = upfirdn2d(v) with the forward parameters (op/upfirdn2d.py:64-85)
            gy = g["gy"].cuda().requires_grad_(True)
            (gx2,) = torch.autograd.grad(y, x, gy, create_graph=True)
            v = torch.linspace(-1, 1, gx2.numel(), device="cuda").reshape(gx2.shape)
            (ggy,) = torch.autograd.grad(gx2, gy, v)
            ref = op.upfirdn2d(v, g["k"].cuda(), up=int(up), down=int(down), pad=(int(p0), int(p1)))
            gu.assert_close(ggy, ref, 1e-5, 1e-6, f"ufd{i} {kn} double-backward")


def test_fused_leaky_relu_golden_and_grads():
    d = gu.load("native_ops")
    g = gu.sub(d, "flr")
    x = g["x"].cuda().requires_grad_(True)
    b = g["b"].cuda().requires_grad_(True)
    y = op.fused_leaky_relu(x, b)
    gu.assert_close(y, g["y"], 1e-6, 1e-6, "fused_leaky_relu")
    gy = torch.linspace(-1, 1, y.numel(), device="cuda").reshape(y.shape)
    gx, gb = torch.autograd.grad(y, [x, b], gy)
    xr, br = g["x"].double().requires_grad_(True), g["b"].double().requires_grad_(True)
    yr = torch.nn.functional.leaky_relu(xr + br.view(1, -1, 1, 1), 0.2) * 2 ** 0.5
    gxr, gbr = torch.autograd.grad(yr, [xr, br], gy.cpu().double())
    gu.assert_close(gx, gxr, 1e-5, 1e-6, "flr gx")
    gu.assert_close(gb, gbr, 1e-5, 1e-5, "flr gb")
    m = op.FusedLeakyReLU(8, negative_slope=0.1).cuda()
    out = m(g["x"].cuda())
    ref = torch.nn.functional.leaky_relu(g["x"], 0.1) * 2 ** 0.5
    gu.assert_close(out, ref, 1e-6, 1e-6, "FusedLeakyReLU slope 0.1 (CUDA-kernel semantics)")


def test_bench_rccl_path_single_rank():
    """bench.py under the driver's launcher with ONE rank and the RCCL path forced on: init_process_group(nccl,
    device_id), parameter broadcast, the flat-gradient all_reduce, barrier + MAX-reduce of the timing."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, INDM_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                        "--gpus", "1", "--steps", "1", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["scaling"] == "weak"


def test_upfirdn2d_bf16_matches_the_fp32_kernel_on_the_same_values():
    """indm_upfirdn2d_bf16 (the reference kernel's half instantiation, op/upfirdn2d_kernel.cu:311): bf16 in / out, fp32 taps and
    accumulation -- equal to rounding the fp32 kernel's result on the bf16-rounded input, on the tile and on the generic path"""
    import torch
    from indm_amd import hip
    g = torch.Generator().manual_seed(5)
    k = torch.tensor([1., 3., 3., 1.])
    k = (torch.outer(k, k) / 64).cuda()
    for shape, cfg in (((2, 16, 16, 32), (2, 2, 1, 1, 2, 1, 2, 1)), ((3, 16, 16, 64), (1, 1, 2, 2, 1, 1, 1, 1)),
                       ((2, 9, 9, 6), (1, 1, 1, 1, 2, 1, 2, 1)), ((1, 8, 8, 128), (2, 2, 1, 1, 2, 1, 2, 1))):
        x = (torch.rand(*shape, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
        y16 = hip.upfirdn2d(x, k * (cfg[0] * cfg[1]), *cfg, 1)
        y32 = hip.upfirdn2d(x.float(), k * (cfg[0] * cfg[1]), *cfg, 1)
        assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.to(torch.bfloat16)), shape
    assert hip.version().endswith("gfx950")
