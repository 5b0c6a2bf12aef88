"""MLP VAE — CPU plumbing model (BASELINE config 1: proves train.py / config / trainer wiring).
Same constructor arguments, losses (L1 reconstruction + kld_weight * KLD) and optimiser as the
reference's models/generative/vae/vae.py:100-215; not on the HIP hot path."""
from __future__ import annotations

import math

import torch
from torch import nn

from lgm_hip.lightning import LightningModule


def _mlp(sizes, last_act=None):
    layers = []
    for i, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
        layers.append(nn.Linear(a, b))
        if i < len(sizes) - 2:
            layers.append(nn.LeakyReLU(0.2))
    if last_act is not None:
        layers.append(last_act)
    return nn.Sequential(*layers)


class Encoder(nn.Module):
    def __init__(self, img_channels, img_size, latent_dim):
        super().__init__()
        n = img_channels * img_size * img_size
        self.layers = nn.Sequential(nn.Linear(n, 512), nn.LeakyReLU(0.2), nn.Linear(512, 256), nn.LeakyReLU(0.2),
                                    nn.Linear(256, 128), nn.LeakyReLU(0.2))
        self.mu = nn.Linear(128, latent_dim)
        self.log_var = nn.Linear(128, latent_dim)

    def forward(self, x):
        h = self.layers(x.flatten(1))
        return self.mu(h), self.log_var(h)


class Decoder(nn.Module):
    def __init__(self, img_channels, img_size, latent_dim):
        super().__init__()
        self.shape = (img_channels, img_size, img_size)
        self.layers = _mlp([latent_dim, 128, 256, 512, math.prod(self.shape)], nn.Tanh())

    def forward(self, z):
        return self.layers(z).view(-1, *self.shape)


class VAE(LightningModule):
    def __init__(self, img_channels: int, img_size: int, latent_dim: int = 20, lr: float = 1e-4, b1: float = 0.9,
                 b2: float = 0.999, weight_decay: float = 1e-5, kld_weight: float = 1e-2):
        super().__init__()
        self.save_hyperparameters()
        self.encoder = Encoder(img_channels, img_size, latent_dim)
        self.decoder = Decoder(img_channels, img_size, latent_dim)

    def reparameterize(self, mu, log_var):
        return mu + torch.randn_like(mu) * torch.exp(log_var / 2)

    def forward(self, x):
        mu, log_var = self.encoder(x)
        return self.decoder(self.reparameterize(mu, log_var)), mu, log_var

    def _common_step(self, batch, batch_idx, split):
        x, _ = batch
        x_hat, mu, log_var = self(x)
        recon = torch.nn.functional.l1_loss(x_hat, x)
        kld = -0.5 * torch.mean(1 + log_var - mu.pow(2) - log_var.exp())
        loss = recon + self.hparams.kld_weight * kld
        self.log_dict({f"{split}_loss": loss, f"{split}_recon_loss": recon, f"{split}_kld": kld})
        return loss

    def training_step(self, batch, batch_idx):
        return self._common_step(batch, batch_idx, "train")

    def validation_step(self, batch, batch_idx):
        return self._common_step(batch, batch_idx, "val")

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=self.hparams.lr, betas=(self.hparams.b1, self.hparams.b2),
                                weight_decay=self.hparams.weight_decay)
