"""Device-side input pipeline — the reference DataModule's transform stack (data/datamodule.py:41-76:
ToTensor, Normalize(0.5, 0.5), CenterCropMinXY, Resize(img_size, antialias=True) and, for the training
split, RandomHorizontalFlip(0.5)) as ONE HIP kernel over a batch of decoded uint8 images, so that a real
dataset run is not bound by a Python/CPU loader once a training step takes ~10 ms.
"""
from __future__ import annotations

from typing import Optional

import torch

from lgm_hip import ops


class DeviceTransforms:
    def __init__(self, img_size: int, train: bool = True, flip_p: float = 0.5, seed: Optional[int] = None):
        self.img_size, self.train, self.flip_p = img_size, train, flip_p
        self._gen = None
        self._seed = seed

    def __call__(self, images_u8: torch.Tensor, flip: Optional[torch.Tensor] = None) -> torch.Tensor:
        """images_u8: [B,H,W,C] uint8 on the GPU.  ``flip`` (uint8 [B]) may be injected (parity tests);
        by default the training split draws it with probability ``flip_p`` per image."""
        if flip is None and self.train and self.flip_p > 0:
            if self._gen is None:
                self._gen = torch.Generator(device=images_u8.device)
                if self._seed is not None:
                    self._gen.manual_seed(self._seed)
            flip = (torch.rand(images_u8.shape[0], device=images_u8.device, generator=self._gen) < self.flip_p).to(
                torch.uint8)
        return ops.image_transform(images_u8.contiguous(), self.img_size, flip)
