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
