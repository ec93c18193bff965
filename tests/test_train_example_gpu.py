"""tools/train_example.py end to end on synthetic CIFAR-10 files: dataset -> dequantise -> step -> reference-format
checkpoints, then a second invocation RESUMES from them (step counter, optimiser, EMA) and trains on."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_cifar(root, n_per=24):
    rs = np.random.RandomState(0)
    d = os.path.join(root, "cifar-10-batches-py")
    os.makedirs(d)
    for name in [f"data_batch_{i}" for i in range(1, 6)] + ["test_batch"]:
        with open(os.path.join(d, name), "wb") as f:
            pickle.dump({b"data": rs.randint(0, 256, (n_per, 3072)).astype(np.uint8),
                         b"labels": rs.randint(0, 10, n_per).tolist()}, f)


def _run(data, work, steps):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train_example.py"), "--data", data, "--workdir", work,
           "--steps", str(steps), "--batch", "8", "--snapshot-freq", "2", "--log-freq", "1"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return [ln for ln in out.stdout.splitlines() if ln.startswith("step:")]


@pytest.mark.timeout(1200)
def test_train_example_trains_checkpoints_and_resumes(tmp_path):
    data, work = str(tmp_path / "data"), str(tmp_path / "run")
    os.makedirs(data)
    _write_cifar(data)
    first = _run(data, work, 3)
    assert [ln.split(",")[0] for ln in first] == ["step: 0", "step: 1", "step: 2"]
    assert all(np.isfinite(float(ln.split("loss mean:")[1].split(",")[0])) for ln in first)
    ck = os.path.join(work, "checkpoints-meta", "checkpoint.pth")
    saved = torch.load(ck, map_location="cpu", weights_only=False)
    assert saved["step"] == 3 and all(k.startswith("module.") for k in saved["model"])      # reference format
    second = _run(data, work, 5)                                                             # resumes at step 3
    assert [ln.split(",")[0] for ln in second] == ["step: 3", "step: 4"]
    assert torch.load(ck, map_location="cpu", weights_only=False)["step"] == 5
    fl = torch.load(os.path.join(work, "checkpoints-meta", "checkpoint_flow.pth"), map_location="cpu", weights_only=False)
    assert fl["step"] == 5
