"""Oracle: DCGAN generator / critic and the WGAN-GP losses (fp32 CPU restatement).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference anchors:
  models/generative/gan/dcgan.py  Generator :35-104, Discriminator :107-164,
                                   initialize_weights :23-32
  models/generative/gan/wgan.py   _calculate_d_loss :84-110, _calculate_g_loss :112-115,
                                   _calculate_gradient_penalty :117-156 (norm over dim=1 only),
                                   _weight_clipping :158-168
  models/generative/gan/dcgan.py  DCGAN._calculate_d_loss / _calculate_g_loss :223-245 (BCE with logits)
  models/generative/gan/lsgan.py  _calculate_d_loss :53-79, _calculate_g_loss :81-97
  models/generative/gan/r1gan.py  _calculate_d_loss :62-94 (R1 = 0.5 * mean_b ||dD(x)/dx||^2, flattened)

State-dict keys follow the reference modules: ``model.{i}.0.weight`` (conv / convT),
``model.{i}.1.{weight,bias,running_mean,running_var,num_batches_tracked}`` (BatchNorm2d).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

# (cin, cout, k, stride, pad, bn)
def generator_spec(img_size: int, img_channels: int, latent_dim: int) -> List[Tuple]:
    if img_size == 64:
        return [(latent_dim, 1024, 4, 1, 0, True), (1024, 512, 4, 2, 1, True),
                (512, 256, 4, 2, 1, True), (256, 128, 4, 2, 1, True),
                (128, img_channels, 4, 2, 1, False)]
    if img_size == 28:
        return [(latent_dim, 256, 7, 1, 0, True), (256, 128, 4, 2, 1, True),
                (128, img_channels, 4, 2, 1, False)]
    raise ValueError("img_size must be 64 or 28")


def discriminator_spec(img_size: int, img_channels: int) -> List[Tuple]:
    # last entry: final layer (no BN, no activation)
    if img_size == 64:
        return [(img_channels, 64, 4, 2, 1, False), (64, 128, 4, 2, 1, True),
                (128, 256, 4, 2, 1, True), (256, 512, 4, 2, 1, True),
                (512, 1, 4, 1, 0, False)]
    if img_size == 28:
        return [(img_channels, 64, 4, 2, 1, False), (64, 128, 4, 2, 1, True),
                (128, 256, 7, 1, 0, True), (256, 1, 1, 1, 0, False)]
    raise ValueError("img_size must be 64 or 28")


def gan_init(img_size=64, img_channels=3, latent_dim=100, seed=0) -> Tuple[Params, Params]:
    """N(0,0.02) conv weights, N(1,0.02) BN weights, zero BN bias (dcgan.py:23-32)."""
    g = torch.Generator().manual_seed(seed)
    G: Params = {}
    D: Params = {}
    for i, (ci, co, k, s, p, bn) in enumerate(generator_spec(img_size, img_channels, latent_dim)):
        G[f"model.{i}.0.weight"] = torch.randn(ci, co, k, k, generator=g) * 0.02
        if bn:
            G[f"model.{i}.1.weight"] = 1 + torch.randn(co, generator=g) * 0.02
            G[f"model.{i}.1.bias"] = torch.zeros(co)
    for i, (ci, co, k, s, p, bn) in enumerate(discriminator_spec(img_size, img_channels)):
        D[f"model.{i}.0.weight"] = torch.randn(co, ci, k, k, generator=g) * 0.02
        if bn:
            D[f"model.{i}.1.weight"] = 1 + torch.randn(co, generator=g) * 0.02
            D[f"model.{i}.1.bias"] = torch.zeros(co)
    return G, D


def _bn_train(x, w, b, eps=1e-5):
    # train-mode BatchNorm2d: batch statistics (biased variance); running stats are
    # side effects the losses do not depend on.
    return F.batch_norm(x, None, None, w, b, training=True, momentum=0.1, eps=eps)


def generator(G: Params, z, img_size=64, img_channels=3):
    spec = generator_spec(img_size, img_channels, z.shape[1])
    x = z
    for i, (ci, co, k, s, p, bn) in enumerate(spec):
        x = F.conv_transpose2d(x, G[f"model.{i}.0.weight"], None, s, p)
        if bn:
            x = F.relu(_bn_train(x, G[f"model.{i}.1.weight"], G[f"model.{i}.1.bias"]))
        else:
            x = torch.tanh(x)
    return x


def discriminator(D: Params, x, img_size=64):
    spec = discriminator_spec(img_size, x.shape[1])
    n = len(spec)
    for i, (ci, co, k, s, p, bn) in enumerate(spec):
        x = F.conv2d(x, D[f"model.{i}.0.weight"], None, s, p)
        if bn:
            x = _bn_train(x, D[f"model.{i}.1.weight"], D[f"model.{i}.1.bias"])
        if i != n - 1:
            x = F.leaky_relu(x, 0.2)
    return x.squeeze()


def gradient_penalty(D: Params, x, x_hat, alpha, lam=10.0, img_size=64):
    """wgan.py:134-156 with alpha injected: channel-dim norm, mean over [B,H,W]."""
    inter = (alpha * x + (1 - alpha) * x_hat).detach().requires_grad_(True)
    score = discriminator(D, inter, img_size)
    grads = torch.autograd.grad(score, inter, torch.ones_like(score),
                                create_graph=True, retain_graph=True)[0]
    nrm = grads.norm(2, dim=1)
    return ((nrm - 1) ** 2).mean() * lam


def wgan_d_loss(D: Params, x, x_hat, alpha, lam=10.0, img_size=64):
    """wgan.py:84-110 (training, gp).  x_hat must already be detached."""
    real = discriminator(D, x, img_size).mean()
    fake = discriminator(D, x_hat, img_size).mean()
    gp = gradient_penalty(D, x, x_hat, alpha, lam, img_size)
    return dict(d_loss=fake - real + gp, d_loss_real=real, d_loss_fake=fake, gradient_penalty=gp)


def wgan_g_loss(D: Params, x_hat, img_size=64):
    return -discriminator(D, x_hat, img_size).mean()


# ---- the other heads on the same generator / critic -------------------------------
def dcgan_d_loss(D: Params, x, x_hat, img_size=64):
    """dcgan.py:223-239"""
    lr_ = discriminator(D, x, img_size)
    lf = discriminator(D, x_hat, img_size)
    real = F.binary_cross_entropy_with_logits(lr_, torch.ones_like(lr_))
    fake = F.binary_cross_entropy_with_logits(lf, torch.zeros_like(lf))
    return dict(d_loss=(real + fake) / 2, d_loss_real=real, d_loss_fake=fake, logits_real=lr_.mean(),
                logits_fake=lf.mean())


def dcgan_g_loss(D: Params, x_hat, img_size=64):
    """dcgan.py:241-245"""
    lf = discriminator(D, x_hat, img_size)
    return F.binary_cross_entropy_with_logits(lf, torch.ones_like(lf))


def lsgan_d_loss(D: Params, x, x_hat, img_size=64):
    """lsgan.py:53-79"""
    lr_ = discriminator(D, x, img_size)
    lf = discriminator(D, x_hat, img_size)
    real = 0.5 * torch.mean((lr_ - 1) ** 2)
    fake = 0.5 * torch.mean(lf ** 2)
    return dict(d_loss=real + fake, d_loss_real=real, d_loss_fake=fake, logits_real=lr_.mean(), logits_fake=lf.mean())


def lsgan_g_loss(D: Params, x_hat, img_size=64):
    """lsgan.py:81-97"""
    lf = discriminator(D, x_hat, img_size)
    return 0.5 * torch.mean((lf - 1) ** 2)


def r1_penalty(D: Params, x, img_size=64):
    """r1gan.py:72-77: 0.5 * mean_b sum_{chw} (d sum(D(x)) / dx)^2"""
    xr = x.detach().clone().requires_grad_(True)
    score = discriminator(D, xr, img_size)
    grad = torch.autograd.grad(score.sum(), xr, create_graph=True)[0]
    return 0.5 * grad.pow(2).reshape(grad.shape[0], -1).sum(1).mean()


def r1gan_d_loss(D: Params, x, x_hat, lam=10.0, img_size=64):
    """r1gan.py:62-94"""
    out = dcgan_d_loss(D, x, x_hat, img_size)
    r1 = r1_penalty(D, x, img_size)
    out["d_loss"] = out["d_loss"] + lam * r1
    out["r1_penalty"] = r1
    return out


def wgan_clip_d_loss(D: Params, x, x_hat, img_size=64):
    """wgan.py:84-93 with constraint_method = "clip": no penalty term (the weights are clamped as a
    side effect, see weight_clip)."""
    real = discriminator(D, x, img_size).mean()
    fake = discriminator(D, x_hat, img_size).mean()
    return dict(d_loss=fake - real, d_loss_real=real, d_loss_fake=fake)


def weight_clip(D: Params, c: float) -> Params:
    """wgan.py:158-168"""
    return {k: v.clamp(-c, c) for k, v in D.items()}
