"""Input side of the training / evaluation step: the reference's `datasets.py` without TensorFlow.

What the reference does (datasets.py:140-288, run_lib.py:84-88):
  tfds record -> float32 [0,1] HWC -> resize_op (CIFAR-10: identity; CelebA: central crop 140 + area-aware bilinear
  resize to 64) -> random horizontal flip (training) -> repeat -> shuffle(buffer 10000) -> batch(drop_remainder) ->
  `get_batch` (NHWC numpy -> NCHW tensor on config.device) -> `(255 x + U[0,1)) / 256` -> scaler.
Here the records come from the datasets' published file formats instead of tfds:
  CIFAR-10   `cifar-10-batches-py/` (pickles) or `cifar-10-batches-bin/` (1 label byte + 3072 CHW bytes per record)
  CelebA     a folder of aligned 178x218 JPEG/PNG files (`img_align_celeba/`), read with PIL
and the stream semantics (repeat, streaming shuffle buffer, flip, drop_remainder, evaluation = one pass, no flip) are
restated on numpy.  One process per GPU: rank r of R reads records r, r+R, ... and batches config.training.batch_size / R.

PARITY NOTE: `tf.image.resize(antialias=True)` is third-party arithmetic that is absent from this image (TensorFlow is
not installed); `resize_antialias` restates its published algorithm (scale_and_translate with the triangle
kernel, half-pixel centres, kernel widened by the down-scale factor, weights normalised per output pixel).  It is pinned
only by properties (identity at scale 1, constants preserved, exact 2x box-of-triangle weights), not by TF outputs:
parity unpinned for CelebA's resize.  CIFAR-10's resize is the identity (scale 1), so that path is exact.
"""
import os
import pickle
import queue
import threading

import numpy as np
import torch

from . import ops
from .rng import mix64
from .utils import get_data_scaler, get_data_inverse_scaler  # noqa: F401  (reference keeps them in datasets.py:56-71)

SHUFFLE_BUFFER = 10000  # datasets.py:178


# ----------------------------------------------------------------------------- image ops (datasets.py:74-106)
def _resize_weights(in_size, out_size):
    """[out_size, in_size] row-stochastic matrix of tf.image.resize(method=bilinear, antialias=True)."""
    inv_scale = in_size / out_size
    kscale = max(inv_scale, 1.0)
    W = np.zeros((out_size, in_size), dtype=np.float64)
    for x in range(out_size):
        s = (x + 0.5) * inv_scale
        lo = max(int(np.ceil(s - kscale - 0.5)), 0)
        hi = min(int(np.floor(s + kscale - 0.5)), in_size - 1) + 1
        src = np.arange(lo, hi)
        w = np.maximum(0.0, 1.0 - np.abs((src + 0.5 - s) / kscale))
        if w.sum() > 0:
            W[x, lo:hi] = w / w.sum()
    return W.astype(np.float32)


def resize_antialias(img, out_h, out_w):
    """img float32 [H, W, C] -> [out_h, out_w, C] (separable: rows then columns)"""
    H, W, _ = img.shape
    if (H, W) == (out_h, out_w):
        return img
    t = np.tensordot(_resize_weights(H, out_h), img, axes=(1, 0))              # [out_h, W, C]
    return np.ascontiguousarray(np.tensordot(_resize_weights(W, out_w), t, axes=(1, 1)).transpose(1, 0, 2))


def central_crop(img, size):
    """datasets.py:101-105"""
    top, left = (img.shape[0] - size) // 2, (img.shape[1] - size) // 2
    return img[top:top + size, left:left + size]


def resize_small(img, resolution):
    """datasets.py:89-98: shorter side -> resolution (sizes truncated, as int())"""
    h, w = img.shape[0], img.shape[1]
    ratio = resolution / min(h, w)
    return resize_antialias(img, int(h * ratio), int(w * ratio))


# ----------------------------------------------------------------------------- record sources
def load_cifar10(root, train):
    """-> (uint8 [N, 32, 32, 3], int64 [N]) from the python or the binary distribution under `root`"""
    py, bn = os.path.join(root, "cifar-10-batches-py"), os.path.join(root, "cifar-10-batches-bin")
    imgs, labels = [], []
    if os.path.isdir(py):
        for name in ([f"data_batch_{i}" for i in range(1, 6)] if train else ["test_batch"]):
            with open(os.path.join(py, name), "rb") as f:
                d = pickle.load(f, encoding="bytes")
            imgs.append(np.asarray(d[b"data"], dtype=np.uint8).reshape(-1, 3, 32, 32))
            labels.append(np.asarray(d[b"labels"], dtype=np.int64))
    elif os.path.isdir(bn):
        for name in ([f"data_batch_{i}.bin" for i in range(1, 6)] if train else ["test_batch.bin"]):
            rec = np.fromfile(os.path.join(bn, name), dtype=np.uint8).reshape(-1, 3073)
            imgs.append(rec[:, 1:].reshape(-1, 3, 32, 32))
            labels.append(rec[:, 0].astype(np.int64))
    else:
        raise FileNotFoundError(f"no cifar-10-batches-py/ or cifar-10-batches-bin/ under {root}")
    return np.ascontiguousarray(np.concatenate(imgs).transpose(0, 2, 3, 1)), np.concatenate(labels)


class _ArraySource:
    """records held in memory as uint8 NHWC; float32 [0,1] on access (tf.image.convert_image_dtype)"""

    def __init__(self, images, labels, image_size):
        self.images, self.labels, self.image_size = images, labels, image_size

    def __len__(self):
        return len(self.images)

    def get(self, i):
        img = self.images[i].astype(np.float32) / 255.0
        return resize_antialias(img, self.image_size, self.image_size), self.labels[i]


class _CelebASource:
    """aligned CelebA files; resize_op of datasets.py:218-222 (central crop 140, shorter side -> image_size)"""

    def __init__(self, folder, files, image_size):
        self.folder, self.files, self.image_size = folder, files, image_size

    def __len__(self):
        return len(self.files)

    def get(self, i):
        from PIL import Image
        with Image.open(os.path.join(self.folder, self.files[i])) as im:
            img = np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0
        return resize_small(central_crop(img, 140), self.image_size), -1


def _celeba_split(root, evaluation):
    """tfds celeb_a: partition file 0 = train, 2 = test (datasets.py:213-216); without it, every file is `train`"""
    folder = os.path.join(root, "img_align_celeba")
    if not os.path.isdir(folder):
        raise FileNotFoundError(f"no img_align_celeba/ under {root}")
    part = os.path.join(root, "list_eval_partition.txt")
    want = "2" if evaluation else "0"
    if os.path.exists(part):
        files = [ln.split()[0] for ln in open(part) if ln.split() and ln.split()[1] == want]
    else:
        files = sorted(f for f in os.listdir(folder) if f.lower().endswith((".jpg", ".jpeg", ".png")))
    return folder, files


# ----------------------------------------------------------------------------- the stream
class Dataset:
    """Iterable of dict(image=float32 [B, H, W, C] in [0,1], label=int64 [B]) like the tf.data pipeline of
    datasets.py:262-288: (repeat ->) shuffle(10000) -> map(resize, flip) -> batch(drop_remainder) -> prefetch."""

    def __init__(self, source, batch_size, evaluation, random_flip, seed=0, rank=0, world=1, prefetch=2):
        if batch_size % world:
            raise ValueError(f"Batch sizes ({batch_size} must be divided by the number of devices")  # datasets.py:174
        self.source, self.evaluation, self.flip = source, evaluation, bool(random_flip) and not evaluation
        self.batch_size = batch_size // world
        self.ids = np.arange(rank, len(source), world)
        self.seed, self.prefetch = (seed, rank), prefetch

    def __len__(self):
        return len(self.ids)

    def _indices(self, rs):
        """record order: repeat (training) then a streaming shuffle buffer of SHUFFLE_BUFFER records"""
        def stream():
            while True:
                yield from self.ids
                if self.evaluation:
                    return
        buf = []
        for i in stream():
            if len(buf) < SHUFFLE_BUFFER:
                buf.append(i)
                continue
            j = rs.randint(len(buf))
            yield buf[j]
            buf[j] = i
        rs.shuffle(buf)
        yield from buf

    def _batches(self):
        rs = np.random.RandomState(mix64(*self.seed) % (2 ** 31))
        imgs, labels = [], []
        for i in self._indices(rs):
            img, lab = self.source.get(int(i))
            if self.flip and rs.rand() < 0.5:
                img = img[:, ::-1]
            imgs.append(img)
            labels.append(lab)
            if len(imgs) == self.batch_size:
                yield dict(image=np.ascontiguousarray(np.stack(imgs)), label=np.asarray(labels, dtype=np.int64))
                imgs, labels = [], []
        # drop_remainder=True: a short last batch is dropped

    def __iter__(self):
        if not self.prefetch:
            return self._batches()
        q, done = queue.Queue(maxsize=self.prefetch), object()

        def work():
            try:
                for b in self._batches():
                    q.put(b)
                q.put(done)
            except BaseException as e:  # surfaced in the consumer
                q.put(e)

        threading.Thread(target=work, daemon=True).start()

        def gen():
            while True:
                b = q.get()
                if b is done:
                    return
                if isinstance(b, BaseException):
                    raise b
                yield b

        return gen()


def _data_root(config):
    root = config.data.get("root", None) if hasattr(config.data, "get") else getattr(config.data, "root", None)
    return root or os.environ.get("INDM_DATA", "data")


def get_dataset_split(config, evaluation=False, rank=0, world=1):
    """datasets.py:161-288 for the datasets the INDM configs use (CIFAR10, CELEBA)"""
    root, size = _data_root(config), config.data.image_size
    name = config.data.dataset
    if name == "CIFAR10":
        images, labels = load_cifar10(root, train=not evaluation)
        source = _ArraySource(images, labels, size)
    elif name == "CELEBA":
        source = _CelebASource(*_celeba_split(root, evaluation), size)
    else:
        raise NotImplementedError(f"Dataset {name} not yet supported.")   # datasets.py:246-248
    bs = config.eval.batch_size if evaluation else config.training.batch_size
    return Dataset(source, bs, evaluation, config.data.random_flip, seed=config.get("seed", 0) if hasattr(config, "get") else 0,
                   rank=rank, world=world)


def get_dataset(config, rank=0, world=1):
    """datasets.py:131-138 -> (train_ds, eval_ds)"""
    return (get_dataset_split(config, False, rank, world), get_dataset_split(config, True, rank, world))


def get_batch_(config, batch):
    """datasets.py:117-129: NHWC [0,1] record batch -> NCHW float tensor on config.device (pinned, non-blocking copy)"""
    img = batch["image"] if isinstance(batch, dict) else batch
    img = img._numpy() if hasattr(img, "_numpy") else img
    t = torch.from_numpy(np.ascontiguousarray(img)).float() if not torch.is_tensor(img) else img.float()
    t = t.permute(0, 3, 1, 2)
    assert t.shape == (t.shape[0], config.data.num_channels, config.data.image_size, config.data.image_size)
    dev = torch.device(config.device)
    if dev.type == "cuda":
        return t.contiguous().pin_memory().to(dev, non_blocking=True)
    return t.contiguous().to(dev)


def get_batch(config, data_iter, data):
    """datasets.py:108-115: next batch, restarting the iterator at the end of an epoch -> (batch, data_iter)"""
    try:
        batch = get_batch_(config, next(data_iter))
    except StopIteration:
        data_iter = iter(data)
        batch = get_batch_(config, next(data_iter))
    return batch, data_iter


def dequantize_and_scale(config, batch, rng):
    """run_lib.py:86-87 in ONE launch: scaler((255 x + u) / 256), u ~ U[0,1) from the step's noise source."""
    u = rng.rand(*batch.shape)
    if config.data.centered:
        return ops.axpby(batch, u, 2.0 * 255.0 / 256.0, 2.0 / 256.0, -1.0)
    return ops.axpby(batch, u, 255.0 / 256.0, 1.0 / 256.0, 0.0)
