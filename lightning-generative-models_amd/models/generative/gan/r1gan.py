"""R1GAN on the MI355X HIP engine — drop-in for the reference's models/generative/gan/r1gan.py.

d_loss = (BCE(D(x), 1) + BCE(D(G(z)), 0)) / 2 + r1_penalty * R1,  R1 = 0.5 * mean_b ||dD(x)/dx||^2
(reference :62-94; per-sample FLATTENED squared norm, :77).  The R1 term needs the gradient of an
input-gradient; it reuses the critic's hand-derived second-order sweep (``Discriminator.
gradient_penalty`` / ``gp_backward``: convolutions are linear, LeakyReLU'' = 0, train-mode BatchNorm
through its adjoint operators) with the R1 functional on top instead of WGAN-GP's channel norm.
"""
from __future__ import annotations

from typing import List

import torch

from lgm_hip.nn import GradCtx
from models.generative.gan.dcgan import DCGAN, _to_nhwc


class R1GAN(DCGAN):
    def __init__(self, img_channels: int, img_size: int, latent_dim: int, lr: float, b1: float, b2: float,
                 weight_decay: float, r1_penalty: float = 10.0, calculate_metrics: bool = False,
                 metrics: List[str] = []) -> None:
        super().__init__(img_channels=img_channels, img_size=img_size, latent_dim=latent_dim, lr=lr, b1=b1, b2=b2,
                         weight_decay=weight_decay, calculate_metrics=calculate_metrics, metrics=metrics)
        self.save_hyperparameters()

    def _calculate_d_loss(self, x, x_hat):
        bce = torch.nn.functional.binary_cross_entropy_with_logits
        logits_real = self.D(x)
        d_loss_real = bce(logits_real, torch.ones_like(logits_real))
        logits_fake = self.D(x_hat.detach())
        d_loss_fake = bce(logits_fake, torch.zeros_like(logits_fake))
        d_loss = (d_loss_real + d_loss_fake) / 2
        r1 = _R1PenaltyFn.apply(self.D._anchor(x.device), self.D, x)      # third critic forward, like :73-76
        d_loss = d_loss + self.hparams.r1_penalty * r1
        return {"d_loss": d_loss, "d_loss_real": d_loss_real, "d_loss_fake": d_loss_fake, "r1_penalty": r1,
                "logits_real": logits_real.mean(), "logits_fake": logits_fake.mean()}


class _R1PenaltyFn(torch.autograd.Function):
    """R1 = 0.5 * mean_b sum (dD(x)/dx)^2; backward = d(R1)/d(theta_D) (x is data: no input gradient)."""

    @staticmethod
    def forward(ctx, anchor, D, x):
        D.prepare_hip(x.device)
        pen, state = D.gradient_penalty(_to_nhwc(x), 1.0, kind="r1")
        ctx.stuff = (D, state)
        return pen.reshape(()).clone()

    @staticmethod
    def backward(ctx, gpen):
        D, state = ctx.stuff
        gl = gpen.detach().float().reshape(1).contiguous()
        gc = GradCtx(D._flat)
        D.gp_backward(gc, state, gl)
        D._flat.bind_grad_views()
        ctx.stuff = None
        return None, None, None
