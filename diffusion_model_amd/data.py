"""STL10 input pipeline for ``main_stl10.train(train_loader=...)`` -- SURVEY.md section 8(f) item 1: the data format
on the input side of the hot path.

The reference reads ``torchvision.datasets.STL10(root='stl10_data', split='train')`` with
``ToTensor()`` + ``Normalize((0.5,)*3, (0.5,)*3)`` behind a shuffling ``DataLoader`` (utils.py:134-139,
main_stl10.py:421-432).  torchvision is not part of this package; what is restated here is the published binary
format of the dataset (``stl10_binary/{train,test,unlabeled}_X.bin``: uint8, one image = 3 x 96 x 96 values stored
channel-major with each channel COLUMN-major; ``*_y.bin``: uint8 labels 1..10) and the two transforms, as the same
fp32 operations in the same order (``/255``, ``-0.5``, ``/0.5``), so a batch equals the reference's bit for bit.

MI355X-first: 288 GB of HBM hold the whole split (5000 x 27,648 bytes = 138 MB as uint8), so the loader uploads it
once and builds every batch on the device with a gather + a 256-entry lookup of the transform -- no worker processes, no
per-batch PCIe traffic, nothing for the training step to wait on.  Shuffling draws one ``torch.randperm`` per epoch
from a seeded CPU generator (the reference's DataLoader order is unseeded; order is not part of parity)."""
from __future__ import annotations

import os
from typing import Iterator, Optional, Tuple

import numpy as np
import torch

IMG_BYTES = 3 * 96 * 96
_SPLITS = {"train": ("train_X.bin", "train_y.bin"), "test": ("test_X.bin", "test_y.bin"),
           "unlabeled": ("unlabeled_X.bin", None)}


def read_stl10_binary(root: str, split: str = "train") -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """-> (images uint8 [N,3,96,96] in the usual row-major (C,H,W) order, labels int64 [N] in 0..9 or None).
    ``root`` is the directory that holds ``stl10_binary/`` (the reference's ``stl10_data``) or that directory itself."""
    if split not in _SPLITS:
        raise ValueError(f"split must be one of {sorted(_SPLITS)}")
    xname, yname = _SPLITS[split]
    base = os.path.join(root, "stl10_binary") if os.path.isdir(os.path.join(root, "stl10_binary")) else root
    xpath = os.path.join(base, xname)
    if not os.path.exists(xpath):
        raise FileNotFoundError(f"{xpath}: STL10 binary files not found (this package never downloads data)")
    raw = np.fromfile(xpath, dtype=np.uint8)
    if raw.size == 0 or raw.size % IMG_BYTES:
        raise ValueError(f"{xpath}: size {raw.size} is not a multiple of {IMG_BYTES}")
    # stored as [N][C][W][H] (column-major channels): swap the last two axes to get [N][C][H][W]
    images = np.ascontiguousarray(raw.reshape(-1, 3, 96, 96).transpose(0, 1, 3, 2))
    labels = None
    if yname is not None:
        ypath = os.path.join(base, yname)
        y = np.fromfile(ypath, dtype=np.uint8)
        if y.size != images.shape[0]:
            raise ValueError(f"{ypath}: {y.size} labels for {images.shape[0]} images")
        if y.min() < 1 or y.max() > 10:
            raise ValueError(f"{ypath}: labels must be 1..10")
        labels = y.astype(np.int64) - 1
    return images, labels


def normalize_uint8(x: torch.Tensor) -> torch.Tensor:
    """ToTensor() then Normalize(0.5, 0.5) of the reference (utils.py:134-137), same fp32 ops in the same order
    (on the CPU, where the reference's DataLoader workers run them)."""
    return (x.to(torch.float32) / 255.0 - 0.5) / 0.5


def normalize_lut() -> torch.Tensor:
    """The 256 possible results of normalize_uint8, computed on the CPU: device batches are a table lookup, so they
    equal the reference's CPU-transformed batches bit for bit (a device-side ``/255`` may round differently)."""
    return normalize_uint8(torch.arange(256, dtype=torch.int32).to(torch.uint8))


def _rank_shard(order: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank ``rank``'s stride of one epoch order.  Every rank must run the SAME number of batches (a rank with one
    batch more would wait for ever in its gradient all-reduce), so the order is first padded to a multiple of
    ``world`` by wrapping around to its beginning -- torch's DistributedSampler convention; with world == 1 nothing
    is added."""
    n = order.numel()
    pad = (-n) % world
    if pad:
        order = torch.cat([order, order[:pad]])
    return order[rank::world]


def _n_batches(n: int, world: int, batch_size: int, drop_last: bool) -> int:
    n_local = (n + world - 1) // world          # identical on every rank (see _rank_shard)
    return n_local // batch_size if drop_last else (n_local + batch_size - 1) // batch_size


class STL10Loader:
    """Iterable of ``(x [B,3,96,96] fp32 in [-1,1], labels [B] int64)`` device batches with ``len()``, the interface
    ``main_stl10.train`` expects from ``train_loader``.  The split lives on ``device`` as uint8."""

    def __init__(self, root: str = "stl10_data", split: str = "train", batch_size: int = 64, shuffle: bool = True,
                 drop_last: bool = False, seed: int = 0, device: Optional[torch.device] = None,
                 rank: int = 0, world_size: int = 1):
        images, labels = read_stl10_binary(root, split)
        if labels is None:
            labels = np.full(images.shape[0], -1, dtype=np.int64)
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self.images = torch.from_numpy(images).to(dev)
        self.labels = torch.from_numpy(labels).to(dev)
        self.lut = normalize_lut().to(dev)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), bool(shuffle), bool(drop_last)
        self.gen = torch.Generator().manual_seed(int(seed))
        self.rank, self.world = int(rank), int(world_size)  # data-parallel: every rank takes its stride of the epoch order
        self.n = images.shape[0]

    def __len__(self) -> int:
        return _n_batches(self.n, self.world, self.batch_size, self.drop_last)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        order = torch.randperm(self.n, generator=self.gen) if self.shuffle else torch.arange(self.n)
        order = _rank_shard(order, self.rank, self.world).to(self.images.device)
        for i in range(len(self)):
            idx = order[i * self.batch_size:(i + 1) * self.batch_size]
            u8 = self.images.index_select(0, idx)
            x = self.lut.index_select(0, u8.reshape(-1).to(torch.int32)).view(u8.shape)
            yield x, self.labels.index_select(0, idx)


class LatentLoader:
    """Latent-diffusion input: the reference's ``LatentDataset`` (utils.py:70-91) + shuffled
    ``DataLoader`` as one device-resident loader.  The file is the reference's ``{'latents': [N,4,h,w], 'labels': [N]}``
    dict of pre-computed VAE latents (``vae_encoder.py``), read with a tensor-only ``torch.load``; every sample is
    ``latents[i] * scaling_factor`` (:90).  ``scaling_factor`` is the VAE's ``config.scaling_factor`` in the reference
    (VAE.py:42); the VAE itself cannot be fetched here, so the factor is an argument, defaulting to the published
    value of ``stabilityai/sd-vae-ft-ema`` (0.18215).  Same iteration / sharding interface as ``STL10Loader``."""

    def __init__(self, latent_path: str, scaling_factor: float = 0.18215, batch_size: int = 64, shuffle: bool = True,
                 drop_last: bool = False, seed: int = 0, device: Optional[torch.device] = None, rank: int = 0,
                 world_size: int = 1):
        d = torch.load(latent_path, map_location="cpu", weights_only=True)
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self.latents = d["latents"].float().to(dev)
        self.labels = torch.as_tensor(d["labels"]).long().to(dev)
        assert self.latents.ndim == 4 and self.latents.shape[0] == self.labels.shape[0]
        self.scaling_factor = float(scaling_factor)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), bool(shuffle), bool(drop_last)
        self.gen = torch.Generator().manual_seed(int(seed))
        self.rank, self.world = int(rank), int(world_size)
        self.n = self.latents.shape[0]

    def __len__(self) -> int:
        return _n_batches(self.n, self.world, self.batch_size, self.drop_last)

    def __getitem__(self, idx: int):
        return self.latents[idx] * self.scaling_factor, self.labels[idx]

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        order = torch.randperm(self.n, generator=self.gen) if self.shuffle else torch.arange(self.n)
        order = _rank_shard(order, self.rank, self.world).to(self.latents.device)
        for i in range(len(self)):
            idx = order[i * self.batch_size:(i + 1) * self.batch_size]
            yield self.latents.index_select(0, idx) * self.scaling_factor, self.labels.index_select(0, idx)
