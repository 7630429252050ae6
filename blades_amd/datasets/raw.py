"""Raw dataset readers without torchvision, plus shape-identical synthetic data.

No download is attempted (this framework targets air-gapped nodes).  Looked-up
layouts under ``data_root``:

* CIFAR-10: ``cifar-10-batches-bin/{data_batch_[1-5],test_batch}.bin`` (binary
  version: 1 label byte + 3072 pixel bytes per record);
* MNIST / Fashion-MNIST: ``<Name>/raw/{train,t10k}-{images-idx3,labels-idx1}-ubyte[.gz]``
  (torchvision's layout) or the same four files directly under ``<name>/``.

When the files are absent and synthetic data is allowed
(``BLADES_SYNTHETIC_DATA=1`` or ``dataset_config.synthetic: true``) a
deterministic synthetic set of identical shape/dtype is generated instead:
every class has a fixed random low-frequency prototype image and samples are
prototype + noise, so models *can* learn it and attack/defence curves behave
qualitatively like on the real data.
"""
import gzip
import os
import struct
from pathlib import Path
from typing import Tuple

import numpy as np

SHAPES = {
    "cifar10": ((3, 32, 32), 50000, 10000, 10),
    "mnist": ((28, 28), 60000, 10000, 10),
    "fashionmnist": ((28, 28), 60000, 10000, 10),
}


def synthetic_allowed(flag=None) -> bool:
    if flag is not None:
        return bool(flag)
    return os.environ.get("BLADES_SYNTHETIC_DATA", "0") not in ("0", "", "false", "False")


def make_synthetic(name: str, n_train=None, n_test=None, seed: int = 0, noise: float = None
                   ) -> Tuple[Tuple[np.ndarray, np.ndarray], Tuple[np.ndarray, np.ndarray]]:
    """uint8 images in the dataset's native layout (CIFAR: NCHW after transpose).  ``noise`` (default 0.35,
    or the ``BLADES_SYNTHETIC_NOISE`` environment variable) is the per-pixel standard deviation added to
    the class prototype: 0.35 is learned to 100 % within a few rounds, ~3 gives a task hard enough for
    attacks and defences to separate."""
    if noise is None:
        noise = float(os.environ.get("BLADES_SYNTHETIC_NOISE", "0.35"))
    shape, ntr, nte, ncls = SHAPES[name]
    n_train = ntr if n_train is None else n_train
    n_test = nte if n_test is None else n_test
    rng = np.random.default_rng(1000 + seed)
    h, w = shape[-2], shape[-1]
    c = shape[0] if len(shape) == 3 else 1
    # low-frequency prototypes: upsampled 4x4 random fields
    low = rng.uniform(0.1, 0.9, size=(ncls, c, 4, 4)).astype(np.float32)
    proto = np.kron(low, np.ones((1, 1, -(-h // 4), -(-w // 4)), dtype=np.float32))[..., :h, :w]

    def gen(n, r):
        y = r.integers(0, ncls, size=n)
        x = proto[y] + noise * r.standard_normal((n, c, h, w), dtype=np.float32)
        x = np.clip(x * 255.0, 0, 255).astype(np.uint8)
        if len(shape) == 2:
            x = x[:, 0]
        return x, y.astype(np.int64)

    return gen(n_train, rng), gen(n_test, np.random.default_rng(2000 + seed))


def _read_idx(path: Path) -> np.ndarray:
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        data = f.read()
    magic, = struct.unpack(">I", data[:4])
    ndim = magic & 0xFF
    dims = struct.unpack(">" + "I" * ndim, data[4: 4 + 4 * ndim])
    return np.frombuffer(data, dtype=np.uint8, offset=4 + 4 * ndim).reshape(dims)


def _find(root: Path, candidates):
    for c in candidates:
        for suffix in ("", ".gz"):
            p = root / (c + suffix)
            if p.exists():
                return p
    return None


def load_mnist_like(data_root, name: str):
    root = Path(data_root)
    dirs = [root / d for d in ({"mnist": ["MNIST/raw", "mnist", "MNIST"],
                                "fashionmnist": ["FashionMNIST/raw", "fashionmnist", "FashionMNIST"]}[name])]
    for d in dirs:
        f = [_find(d, [n]) for n in ("train-images-idx3-ubyte", "train-labels-idx1-ubyte",
                                     "t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte")]
        if all(f):
            xtr, ytr, xte, yte = (_read_idx(p) for p in f)
            return (xtr, ytr.astype(np.int64)), (xte, yte.astype(np.int64))
    return None


def load_cifar10(data_root):
    root = Path(data_root) / "cifar-10-batches-bin"
    files = [root / f"data_batch_{i}.bin" for i in range(1, 6)]
    test = root / "test_batch.bin"
    if not (all(p.exists() for p in files) and test.exists()):
        return None

    def rd(p):
        a = np.fromfile(p, dtype=np.uint8).reshape(-1, 3073)
        return a[:, 1:].reshape(-1, 3, 32, 32), a[:, 0].astype(np.int64)

    parts = [rd(p) for p in files]
    xtr = np.concatenate([p[0] for p in parts])
    ytr = np.concatenate([p[1] for p in parts])
    xte, yte = rd(test)
    return (xtr, ytr), (xte, yte)


def load(name: str, data_root, synthetic=None, seed: int = 0, n_train=None, n_test=None):
    """Returns ((x_train, y_train), (x_test, y_test), is_synthetic)."""
    real = None
    if not (synthetic is True):
        real = load_cifar10(data_root) if name == "cifar10" else load_mnist_like(data_root, name)
    if real is not None:
        return real[0], real[1], False
    if synthetic_allowed(synthetic):
        tr, te = make_synthetic(name, n_train, n_test, seed)
        return tr, te, True
    raise FileNotFoundError(
        f"{name}: raw files not found under {data_root} and no download is possible here. "
        "Place the files there, or allow the shape-identical synthetic stand-in with "
        "BLADES_SYNTHETIC_DATA=1 / dataset_config.synthetic: true."
    )
