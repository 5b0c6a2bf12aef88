"""LSGAN on the MI355X HIP engine — drop-in for the reference's models/generative/gan/lsgan.py
(constructor arguments, loss names).  The generator / critic are DCGAN's HIP networks; the least
squares losses are [B]-sized arithmetic on the critic's logits (reference :53-97), so they stay as
torch ops whose gradient enters the hand-written critic backward through ``_DiscFn``.
"""
from __future__ import annotations

from typing import List

import torch

from models.generative.gan.dcgan import DCGAN


class LSGAN(DCGAN):
    def __init__(self, img_channels: int = 3, img_size: int = 64, latent_dim: int = 100, lr: float = 1e-4,
                 b1: float = 0.5, b2: float = 0.999, weight_decay: float = 1e-5, calculate_metrics: bool = False,
                 metrics: List[str] = [], summary: bool = True) -> None:
        super().__init__(img_channels=img_channels, img_size=img_size, latent_dim=latent_dim, lr=lr, b1=b1, b2=b2,
                         weight_decay=weight_decay, calculate_metrics=calculate_metrics, metrics=metrics,
                         summary=summary)

    def _calculate_d_loss(self, x, x_hat):
        """reference :53-79: 0.5*mean((D(x)-1)^2) + 0.5*mean(D(G(z))^2)"""
        logits_real = self.D(x)
        d_loss_real = 0.5 * torch.mean((logits_real - 1) ** 2)
        logits_fake = self.D(x_hat.detach())
        d_loss_fake = 0.5 * torch.mean(logits_fake ** 2)
        d_loss = d_loss_real + d_loss_fake
        return {"d_loss": d_loss, "d_loss_real": d_loss_real, "d_loss_fake": d_loss_fake,
                "logits_real": logits_real.mean(), "logits_fake": logits_fake.mean()}

    def _calculate_g_loss(self, x_hat):
        """reference :81-97: 0.5*mean((D(G(z))-1)^2)"""
        logits_fake = self.D(x_hat)
        g_loss = 0.5 * torch.mean((logits_fake - 1) ** 2)
        return {"g_loss": g_loss, "logits_fake": logits_fake.mean()}
