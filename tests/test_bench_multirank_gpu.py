"""bench.py's multi-process path end to end on a single-GPU box: two ranks launched with torch.distributed.run exactly
as the driver launches N ranks, sharing GPU 0 and exchanging gradients over gloo (INDM_BENCH_ONE_GPU=1; the real run is
one rank per GPU over RCCL).  Checks the contract line rank 0 prints, for weak and for strong scaling -- and the bare
`python bench.py --gpus 2` form, where bench.py has to start the ranks itself."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port, self_launch=False, gpus=2):
    env = dict(os.environ, INDM_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    launcher = [] if self_launch else ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                       "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd = [sys.executable] + launcher + [os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2",
                                         "--warmup", "1", "--no-cpu-baseline", "--no-sampling"] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout           # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_two_rank_bench_weak_and_strong():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    port = 29600 + os.getpid() % 300
    weak = _run(["--batch", "8"], port)
    assert weak["n_gpus"] == 2 and weak["steps"] == 2 and weak["warmup"] == 1 and weak["scaling"] == "weak"
    assert weak["config"]["global_batch"] == 16 and weak["config"]["parallelism"] == "dp2"
    assert weak["value"] > 0 and abs(weak["value"] - 16 * 1000.0 / weak["ms_per_step"]) < 1e-6 * weak["value"] + 1e-3
    assert weak["roofline"]["bound"] == "mfma" and weak["roofline"]["achieved"] > 0
    # BASELINE.md section 3's whole-step roofline at the realised mean series length
    st = weak["roofline"]["step"]
    assert st["n_bar"] >= 0 and abs(st["gflop_per_image"] - (12.73 * (st["n_bar"] + 8) + 65.08)) < 1e-6
    assert 0 < st["frac"] < 1 and abs(st["achieved_tflops"] - st["gflop_per_image"] * weak["value"] / 2 / 1e3) < 1e-6 * st["achieved_tflops"] + 1e-9
    assert weak["config"]["n_bar"] == st["n_bar"]
    # who ran: two ranks, ONE device here (both ranks share GPU 0 in this test), exchanged over gloo; and what the
    # collective costs on the critical path
    assert weak["ranks_seen"] == 2 and weak["distinct_devices"] == 1 and weak["backend"] == "gloo"
    assert len(weak["devices"]) == 2 and all("name" in d for d in weak["devices"])
    assert isinstance(weak["allreduce_exposed_ms"], float) and weak["ms_per_step_without_collective"] > 0
    # clamped at 0; the raw difference and the per-step spread of the replay (same Poisson stream, same step count) are there
    assert weak["allreduce_exposed_ms"] == max(0.0, weak["allreduce_exposed_raw_ms"])
    assert weak["allreduce_exposed_detail"]["same_poisson_stream"] and weak["allreduce_exposed_detail"]["replayed_steps"] == 2
    assert weak["collective_elements_per_step"] and len(weak["collective_elements_per_step"]) == 2
    assert weak["cpu_baseline"] is None            # rank 0 at N = 1 only
    strong = _run(["--global-batch", "16"], port + 1)
    assert strong["scaling"] == "strong" and strong["config"]["global_batch"] == 16 and strong["n_gpus"] == 2
    assert strong["value"] > 0


@pytest.mark.timeout(900)
def test_bare_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun around it (the shape of the driver's command): bench.py spawns the two
    rank processes itself and relays rank 0's line."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    line = _run(["--batch", "8"], None, self_launch=True)
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 16
    assert line["scaling"] == "weak" and line["value"] > 0


@pytest.mark.timeout(600)
def test_world_size_must_match_gpus():
    """two torchrun ranks but --gpus 4: refuse instead of printing a line with the wrong n_gpus"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, INDM_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(29950 + os.getpid() % 40), os.path.join(ROOT, "bench.py"), "--gpus", "4",
           "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-sampling", "--batch", "8"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(900)
def test_single_gpu_line_carries_the_bf16_and_shard_legs():
    """the default N = 1 line: the whole-step roofline at the realised series length, the bf16 (F16 storage) sub-leg with its own
    roofline, and the training step at the per-rank shards of an 8-GPU job (16 / 32 / 64 images, fp32 and bf16)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "INDM_BENCH_ONE_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "2", "--batch", "16", "--no-cpu-baseline",
           "--no-sampling"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["distinct_devices"] == 1 and line["backend"] is None
    assert "extra_legs_error" not in line, line.get("extra_legs_error")
    assert line["roofline"]["step"]["n_bar"] == line["config"]["n_bar"]
    b = line["bf16"]
    # (which class is dominant at batch 16 depends on launch times, i.e. on who else uses the GPU: only its form is checked)
    assert b["value"] > 0 and b["roofline"]["kernel"].startswith("igemm<") and 0 < b["roofline"]["step"]["frac"] < 1
    assert "bf16" in b["dtype"]
    sh = line["train_per_rank_shard_of_8"]
    for key in ("batch16_f32", "batch16_bf16", "batch32_f32", "batch32_bf16", "batch64_f32", "batch64_bf16"):
        assert sh[key]["ms_per_step"] > 0 and sh[key]["images_per_s"] > 0 and sh[key]["ms_per_step_median"] > 0
    # (no comparison of the legs' times here: under `pytest -n 3` other workers share the GPU and a 5-step leg can be twice as
    # slow as its neighbour -- 4 of 5 whole-suite xdist runs failed on exactly such an assertion before it was removed)
