"""GAN base LightningModule — the part of the reference's models/generative/gan/gan.py that the
DCGAN / WGAN hot path inherits (manual optimisation, two Adam optimisers, the alternating
_common_step).  The reference's MLP generator/discriminator (gan.py:15-89) are outside the hot
path (they are constructed and immediately overwritten by DCGAN's conv nets), so
this base class leaves ``G`` / ``D`` to the subclass.
"""
from __future__ import annotations

from typing import List

import torch

from lgm_hip.lightning import LightningModule, multi_rank
from lgm_hip.optim import FusedAdam


class GAN(LightningModule):
    def __init__(self, img_channels: int = 1, img_size: int = 28, latent_dim: int = 100, lr: float = 1e-4,
                 b1: float = 0.5, b2: float = 0.999, weight_decay: float = 1e-5, loss_type: str = "non-saturating",
                 calculate_metrics: bool = False, metrics: List[str] = [], summary: bool = True):
        super().__init__()
        self.save_hyperparameters()
        self.automatic_optimization = False      # reference gan.py:118
        self.calculate_metrics = calculate_metrics
        self.metrics = metrics
        self.G = None
        self.D = None

    def prepare_hip(self, device):
        self.G.prepare_hip(device)
        self.D.prepare_hip(device)

    def forward(self, z):
        return self.G(z)

    def _common_step(self, batch, mode: str):
        """reference gan.py:144-174: one D update then one G update per batch."""
        x, _ = batch
        x_hat = self.G.random_sample(x.size(0))
        d_optim, g_optim = self.optimizers()
        loss_dict = self._calculate_d_loss(x, x_hat)
        if self.training:
            d_optim.zero_grad(set_to_none=True)
            self.manual_backward(loss_dict["d_loss"])
            d_optim.step()
        loss_dict.update(self._calculate_g_loss(x_hat))
        if self.training:
            g_optim.zero_grad(set_to_none=True)
            self.manual_backward(loss_dict["g_loss"])
            g_optim.step()
        loss_dict = {f"{mode}_{k}": v for k, v in loss_dict.items()}
        self.log_dict(loss_dict, prog_bar=True, logger=True, sync_dist=multi_rank())
        return x, x_hat, loss_dict

    def training_step(self, batch):
        _, _, loss_dict = self._common_step(batch, "train")
        return loss_dict

    def validation_step(self, batch):
        self._common_step(batch, "val")

    def configure_optimizers(self):
        """reference gan.py:243-256: ([d_optim, g_optim], [])"""
        kw = dict(lr=self.hparams.lr, betas=(self.hparams.b1, self.hparams.b2), weight_decay=self.hparams.weight_decay)
        return [FusedAdam(self.D.parameters(), **kw), FusedAdam(self.G.parameters(), **kw)], []
