"""DataModule with the reference's constructor contract (data/datamodule.py:14-76): yields
``(x float32 [B,C,S,S] in [-1,1], y int64)`` batches.  The real torchvision datasets are outside
the hot path and unavailable offline, so every dataset name is served by a seeded synthetic
source of the configured shape (the reference's post-Normalize range).  The per-process batch is
``batch_size / world_size`` exactly like the reference (datamodule.py:33)."""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Dataset


class SyntheticImages(Dataset):
    def __init__(self, n: int, channels: int, size: int, num_classes: int = 10, seed: int = 10):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.rand(n, channels, size, size, generator=g) * 2 - 1
        self.y = torch.randint(0, num_classes, (n,), generator=g)

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.y[i]


class DataModule:
    def __init__(self, name: str, img_size: int, img_channels: int, data_dir=None, batch_size: int = 32,
                 num_workers: int = 0, pin_memory: bool = True, persistent_workers: bool = True,
                 train_val_split: float = 0.8, download: bool = True, num_samples: int = 2048):
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if world > 1 else 0
        self.name, self.img_size, self.img_channels = str(name), img_size, img_channels
        self.batch_size = int(batch_size / world)
        self.num_workers, self.pin_memory = int(num_workers), pin_memory
        self.persistent_workers = bool(persistent_workers) and self.num_workers > 0
        self.train_val_split = train_val_split
        n_train = int(num_samples * train_val_split)
        self.train = SyntheticImages(n_train, img_channels, img_size, seed=10 + self.rank)
        self.val = SyntheticImages(num_samples - n_train, img_channels, img_size, seed=1000 + self.rank)

    # pinned host batches (reference: pin_memory=True, datamodule.py:24): the trainer's non_blocking copy is then a
    # real asynchronous H2D - from pageable memory it blocks the host until the previous step has drained
    def _pin(self):
        return bool(self.pin_memory) and torch.cuda.is_available()

    def train_dataloader(self):
        return DataLoader(self.train, batch_size=self.batch_size, shuffle=True, drop_last=True,
                          generator=torch.Generator().manual_seed(10), pin_memory=self._pin(),
                          num_workers=self.num_workers, persistent_workers=self.persistent_workers)

    def val_dataloader(self):
        return DataLoader(self.val, batch_size=self.batch_size, shuffle=False, drop_last=True, pin_memory=self._pin(),
                          num_workers=self.num_workers, persistent_workers=self.persistent_workers)
