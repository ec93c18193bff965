"""bench.py's host-side bookkeeping (no GPU): the whole-step roofline of BASELINE.md section 3 at the REALISED series length,
the Poisson counter that supplies it, the traffic-file lookup, the synthetic batch of SURVEY section 8(d)."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_step_flops_follow_the_measured_closed_form():
    # BASELINE.md section 2 / SURVEY 8(d): 12.73 (n + 8) + 65.08 GFLOP per image; 192.4 at E[n] = 2, 101.0 + 65.08 at n = 0
    assert bench.step_gflop_per_image(2.0) == pytest.approx(192.38, abs=0.01)
    assert bench.step_gflop_per_image(0.0) == pytest.approx(12.73 * 8 + 65.08)
    # linear in n-bar: one more series term per block costs one g-pass (12.73 GFLOP)
    assert bench.step_gflop_per_image(3.5) - bench.step_gflop_per_image(2.5) == pytest.approx(12.73)


def test_step_roofline_prices_against_the_peak_of_the_arithmetic_type():
    r = bench.step_roofline(2.0, 500.0, "f32")
    assert r["peak_tflops"] == bench.PEAK_F32_MFMA_TFLOPS == 157.3
    assert r["achieved_tflops"] == pytest.approx(192.38 * 500.0 / 1e3, rel=1e-6)
    assert r["frac"] == pytest.approx(r["achieved_tflops"] / 157.3)
    b = bench.step_roofline(2.0, 500.0, "bf16")
    assert b["peak_tflops"] == bench.PEAK_BF16_MFMA_TFLOPS == 2500.0          # dense peak, never the 2:1-sparsity figure
    assert b["achieved_tflops"] == r["achieved_tflops"] and b["frac"] < r["frac"]
    # an explicit per-image figure (NELBO / sampling legs) replaces the training closed form
    assert bench.step_roofline(None, 10.0, "f32", gflop_per_image=376.0)["achieved_tflops"] == pytest.approx(3.76)


def test_poisson_counter_sees_every_draw_of_the_wrapped_source():
    class Src:
        def __init__(self):
            self.k = 0

        def poisson(self, lam):
            self.k += 1
            return self.k % 4

    src = Src()
    c = bench._CountingPoisson(src)
    assert c.mean() is None
    got = [src.poisson(2.0) for _ in range(8)]        # callers keep calling rng.poisson: the wrapper sits in its place
    assert got == [1, 2, 3, 0, 1, 2, 3, 0] and c.draws == got and c.mean() == pytest.approx(1.5)
    c.reset()
    src.poisson(2.0)
    assert c.draws == [1]


def test_traffic_lookup_takes_the_newest_file_of_the_same_kernel():
    key = "igemm<A_ROW,B_ROW,ires_chain64x512x32,gated>"
    val, path = bench.pmc_traffic(key)
    assert path is not None and os.path.exists(os.path.join(ROOT, path))
    assert json.load(open(os.path.join(ROOT, path)))["bench_key"] == key
    # newest round first: the lookup walks profiles/*traffic*.json in reverse name order
    assert os.path.basename(path) >= "r04_final_traffic_dominant_kernel.json"
    assert 0.8e9 < val < 1.3e9                       # 821 MB algorithmic; the PMC passes see 1.19 x that
    v16, p16 = bench.pmc_traffic(key[:-1] + ",bf16>")
    assert p16 is not None and 2.8e8 < v16 < 3.6e8    # the F16 chain kernel: 283 MB algorithmic
    assert bench.pmc_traffic("igemm<no such class>") == (None, None)


def test_synthetic_batch_is_the_dequantised_uniform_image_of_the_reference_pipeline():
    x = bench.synthetic_batch(4, 32, 0, torch.device("cpu"))
    assert x.shape == (4, 3, 32, 32) and x.dtype == torch.float32
    assert float(x.min()) >= -1.0 and float(x.max()) < 1.0          # (255 u + U[0,1)) / 256 * 2 - 1
    assert torch.equal(x, bench.synthetic_batch(4, 32, 0, torch.device("cpu")))      # seeded: the same batch again
    assert not torch.equal(x, bench.synthetic_batch(4, 32, 1, torch.device("cpu")))


def test_per_rank_batch_of_weak_and_strong_scaling():
    assert bench.per_rank_batch(128, 0, 8) == (128, False)          # --batch: per-GPU batch fixed
    assert bench.per_rank_batch(128, 128, 8) == (16, True)          # --global-batch: configs #2 / #3 / #4-#5 on 8 ranks
    assert bench.per_rank_batch(128, 512, 8) == (64, True) and bench.per_rank_batch(7, 256, 8) == (32, True)
    with pytest.raises(SystemExit):
        bench.per_rank_batch(128, 100, 8)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("extra,scaling,per_rank", [(["--batch", "2"], "weak", 2), (["--global-batch", "4"], "strong", 2)])
def test_multi_rank_path_dry_run_on_cpu(extra, scaling, per_rank):
    """What an unattended `bench.py --gpus N` run depends on besides the kernels, with two gloo ranks on CPU (kernel layer
    emulated, mini configuration, tests/bench_dry_run.py): rank bookkeeping, weak / strong batch partitioning, the gradient
    reducer's two collectives per step, barrier-bracketed timing, and the no-collective replay behind `allreduce_exposed_ms`
    (same Poisson stream and step count as the timed run, clamped at 0, spread reported)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    port = 29700 + (os.getpid() + len(extra[0])) % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "bench_dry_run.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1"] + extra
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["per_rank_batch"] == per_rank and d["global_batch"] == 2 * per_rank
    assert d["value"] > 0 and abs(d["value"] - d["global_batch"] * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"] + 1e-6
    assert d["backend"] == "gloo" and len(d["collective_elements_per_step"]) == 2 and min(d["collective_elements_per_step"]) > 0
    assert d["allreduce_exposed_ms"] >= 0.0 and d["allreduce_exposed_ms"] == max(0.0, d["allreduce_exposed_raw_ms"])
    det = d["allreduce_exposed_detail"]
    assert det["same_poisson_stream"] and det["replayed_steps"] == 2
    s = det["per_step_difference_ms_rank0"]
    assert s["min"] <= s["mean"] <= s["max"] and s["std"] >= 0
    assert d["ms_per_step_without_collective"] > 0
