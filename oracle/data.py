"""Oracle: the reference DataModule's transform stack on CPU tensors.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference anchors: data/datamodule.py:41-53 (ToTensor, Normalize([0.5]*C, [0.5]*C), CenterCropMinXY,
Resize(img_size, antialias=True), RandomHorizontalFlip(0.5)), data/utils.py:7-35 (CenterCropMinXY).
torchvision is not installed here; its tensor code paths are thin wrappers over the torch ops used below
(``ToTensor`` = permute + /255, ``Resize`` on a tensor = ``F.interpolate(mode="bilinear",
align_corners=False, antialias=True)`` to the smaller-edge size, ``hflip`` = ``flip(-1)``), so the oracle
is pinned by torch itself; the torchvision wrappers themselves are **parity unpinned**.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def transform(img_u8_hwc: torch.Tensor, img_size: int, flip: bool) -> torch.Tensor:
    x = img_u8_hwc.permute(2, 0, 1).float() / 255.0                  # ToTensor
    x = (x - 0.5) / 0.5                                              # Normalize
    _, h, w = x.shape                                                # CenterCropMinXY (data/utils.py:23-33)
    d = min(h, w)
    top, left = (h - d) // 2, (w - d) // 2
    x = x[:, top:top + d, left:left + d]
    x = F.interpolate(x[None], size=(img_size, img_size), mode="bilinear", align_corners=False, antialias=True)[0]
    return x.flip(-1) if flip else x
