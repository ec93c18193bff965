"""Input pipeline (indm_amd/datasets.py; reference datasets.py:74-288, run_lib.py:84-88) on synthetic files in the
published CIFAR-10 / CelebA layouts.  TensorFlow is absent, so the resize is pinned by properties only (see the
module's parity note); everything else is exact integer / permutation logic."""
import os
import pickle

import numpy as np
import pytest
import torch

from indm_amd import configs, datasets
from indm_amd.rng import DeviceRng


def _write_cifar(root, fmt, n_per=40, seed=0):
    rs = np.random.RandomState(seed)
    allx, ally = {}, {}
    d = os.path.join(root, "cifar-10-batches-" + fmt)
    os.makedirs(d)
    for name in [f"data_batch_{i}" for i in range(1, 6)] + ["test_batch"]:
        x = rs.randint(0, 256, (n_per, 3072)).astype(np.uint8)
        y = rs.randint(0, 10, n_per)
        allx[name], ally[name] = x, y
        if fmt == "py":
            with open(os.path.join(d, name), "wb") as f:
                pickle.dump({b"data": x, b"labels": y.tolist()}, f)
        else:
            np.concatenate([y[:, None].astype(np.uint8), x], axis=1).tofile(os.path.join(d, name + ".bin"))
    return allx, ally


def _cfg(root, bs=16):
    cfg = configs.get_config("vp/CIFAR10/indm_nll")
    cfg.data.root = root
    cfg.training.batch_size = bs
    cfg.eval.batch_size = bs
    cfg.device = torch.device("cpu")
    return cfg


@pytest.mark.parametrize("fmt", ["py", "bin"])
def test_cifar10_readers_agree_with_the_file_format(tmp_path, fmt):
    allx, ally = _write_cifar(str(tmp_path), fmt)
    x, y = datasets.load_cifar10(str(tmp_path), train=True)
    want = np.concatenate([allx[f"data_batch_{i}"] for i in range(1, 6)]).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)
    assert x.dtype == np.uint8 and np.array_equal(x, want)
    assert np.array_equal(y, np.concatenate([ally[f"data_batch_{i}"] for i in range(1, 6)]))
    xt, _ = datasets.load_cifar10(str(tmp_path), train=False)
    assert np.array_equal(xt, allx["test_batch"].reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))
    with pytest.raises(FileNotFoundError):
        datasets.load_cifar10(str(tmp_path / "nowhere"), train=True)


def test_eval_stream_is_one_pass_without_flip_and_drops_the_remainder(tmp_path):
    allx, _ = _write_cifar(str(tmp_path), "py", n_per=40)
    cfg = _cfg(str(tmp_path), bs=16)
    _, eval_ds = datasets.get_dataset(cfg)
    batches = list(eval_ds)
    assert len(batches) == 40 // 16                      # drop_remainder
    assert all(b["image"].shape == (16, 32, 32, 3) and b["image"].dtype == np.float32 for b in batches)
    # every emitted image is an (unflipped) record of the test file, each at most once
    recs = {r.tobytes() for r in (allx["test_batch"].reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))}
    seen = [np.round(img * 255).astype(np.uint8).tobytes() for b in batches for img in b["image"]]
    assert all(s in recs for s in seen) and len(set(seen)) == len(seen)
    assert list(eval_ds)[0]["image"].shape == (16, 32, 32, 3)   # re-iterable (get_batch restarts it)


def test_train_stream_repeats_shuffles_and_flips(tmp_path):
    allx, _ = _write_cifar(str(tmp_path), "bin", n_per=8)       # 40 training records
    cfg = _cfg(str(tmp_path), bs=8)
    train_ds, _ = datasets.get_dataset(cfg)
    it = iter(train_ds)
    got = [next(it)["image"] for _ in range(12)]                # 96 images > 2 epochs: the stream repeats
    recs = np.concatenate([allx[f"data_batch_{i}"] for i in range(1, 6)]).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)
    plain = {r.tobytes() for r in recs}
    flipped = {np.ascontiguousarray(r[:, ::-1]).tobytes() for r in recs}
    keys = [np.round(img * 255).astype(np.uint8).tobytes() for b in got for img in b]
    assert all(k in plain or k in flipped for k in keys)
    nflip = sum(k in flipped and k not in plain for k in keys)
    assert 20 < nflip < 76                                      # ~half of 96 are mirrored
    first_epoch = keys[:40]
    assert first_epoch != [r.tobytes() for r in recs]           # not the file order
    # a second dataset object with the same seed replays the same stream; another rank reads other records
    again = [next(iter(datasets.get_dataset(cfg)[0]))["image"]]
    assert np.array_equal(again[0], got[0])
    cfg.training.batch_size = 8
    r0 = datasets.get_dataset_split(cfg, False, rank=0, world=2)
    r1 = datasets.get_dataset_split(cfg, False, rank=1, world=2)
    assert r0.batch_size == 4 and set(r0.ids) & set(r1.ids) == set() and len(r0.ids) + len(r1.ids) == 40
    with pytest.raises(ValueError):
        datasets.get_dataset_split(cfg, False, rank=0, world=3)


def test_get_batch_and_dequantisation(tmp_path):
    _write_cifar(str(tmp_path), "py", n_per=16)
    cfg = _cfg(str(tmp_path), bs=16)
    _, eval_ds = datasets.get_dataset(cfg)
    it = iter(eval_ds)
    batch, it = datasets.get_batch(cfg, it, eval_ds)
    assert batch.shape == (16, 3, 32, 32) and batch.dtype == torch.float32
    batch2, it = datasets.get_batch(cfg, it, eval_ds)           # the single eval batch is exhausted: restart
    assert torch.equal(batch, batch2)
    # run_lib.py:86-87 as the reference writes it
    rng = DeviceRng(torch.device("cpu"), seed=3)
    u = DeviceRng(torch.device("cpu"), seed=3).rand(*batch.shape)
    want = ((255.0 * batch + u) / 256.0) * 2.0 - 1.0
    # ops.axpby needs the kernel layer; restate its arithmetic here (a*x + b*y + c) to check the folded constants
    a, b, c = 2.0 * 255.0 / 256.0, 2.0 / 256.0, -1.0
    assert torch.allclose(a * batch + b * u + c, want, atol=1e-6)
    q = torch.floor((want + 1.0) / 2.0 * 256.0)
    assert torch.equal(q, torch.round(batch * 255.0))           # the dequantised value stays inside its own bin


def test_resize_properties():
    rs = np.random.RandomState(0)
    img = rs.rand(140, 140, 3).astype(np.float32)
    assert datasets.resize_antialias(img, 140, 140) is img                     # scale 1: identity (CIFAR-10 path)
    W = datasets._resize_weights(140, 64)
    assert W.shape == (64, 140) and np.allclose(W.sum(1), 1.0, atol=1e-6) and (W >= 0).all()
    assert np.allclose(datasets.resize_antialias(np.full((140, 140, 3), 0.37, np.float32), 64, 64), 0.37, atol=1e-6)
    # exact 2x reduction: triangle of half-width 2 sampled at +-0.5, +-1.5 -> weights (1, 3, 3, 1)/8
    W2 = datasets._resize_weights(8, 4)
    assert np.allclose(W2[1, 1:5], np.array([1, 3, 3, 1]) / 8.0)
    # up-sampling keeps the plain bilinear kernel (kernel scale 1)
    W3 = datasets._resize_weights(2, 4)
    assert np.allclose(W3, [[1, 0], [0.75, 0.25], [0.25, 0.75], [0, 1]])
    out = datasets.resize_small(datasets.central_crop(rs.rand(218, 178, 3).astype(np.float32), 140), 64)
    assert out.shape == (64, 64, 3) and 0.0 <= out.min() and out.max() <= 1.0


def test_celeba_folder(tmp_path):
    from PIL import Image
    folder = tmp_path / "img_align_celeba"
    folder.mkdir()
    rs = np.random.RandomState(1)
    for i in range(6):
        Image.fromarray(rs.randint(0, 256, (218, 178, 3)).astype(np.uint8)).save(folder / f"{i:06d}.png")
    (tmp_path / "list_eval_partition.txt").write_text("".join(f"{i:06d}.png {0 if i < 4 else 2}\n" for i in range(6)))
    cfg = configs.get_config("vp/CELEBA/indm_fid")
    cfg.data.root, cfg.training.batch_size, cfg.eval.batch_size = str(tmp_path), 2, 2
    train_ds, eval_ds = datasets.get_dataset(cfg)
    assert len(train_ds) == 4 and len(eval_ds) == 2
    b = next(iter(eval_ds))
    assert b["image"].shape == (2, 64, 64, 3) and 0.0 <= b["image"].min() and b["image"].max() <= 1.0


def test_evaluation_get_batch_consumes_the_dataset_stream(tmp_path):
    """evaluation.get_bpd's batch helper (evaluation.py:388-494 protocol) reads the dict batches of datasets.Dataset"""
    from indm_amd import evaluation
    _write_cifar(str(tmp_path), "py", n_per=16)
    cfg = _cfg(str(tmp_path), bs=16)
    _, eval_ds = datasets.get_dataset(cfg)
    it = iter(eval_ds)
    b, it = evaluation.get_batch(cfg, it, eval_ds)
    assert b.shape == (16, 3, 32, 32) and b.dtype == torch.float32 and 0.0 <= float(b.min()) and float(b.max()) <= 1.0
    b2, it = evaluation.get_batch(cfg, it, eval_ds)             # exhausted -> restarted
    assert torch.equal(b, b2)
