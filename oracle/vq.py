"""Oracle: VQ-VAE encoder / vector-quantiser / decoder (fp32 CPU restatement).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference anchors:
  models/modules/vector_quantizer.py  _quantize :45-69, losses :71-78,
      perplexity :80-88, STE :90-93, EMA update :128-147, EMA _quantize :149-179
  models/modules/residual.py          ResidualBlock :5-21 (in-place ReLU quirk), stack :24-43
  models/generative/vae/vqvae.py      Encoder :36-51, Decoder :74-85, _common_step :169-199

Keys follow ``VQVAE.state_dict()``: ``encoder.layers.N.*``, ``decoder.layers.N.*``,
``vector_quantizer.embedding.weight`` (+ ``_ema_cluster_size`` / ``_ema_embedding``).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


def vqvae_param_shapes(img_channels=3, embedding_dim=64, num_embeddings=512, hidden_dim=128,
                       num_residual_layers=2, num_residual_hiddens=32):
    s: Dict[str, Tuple[int, ...]] = {}
    h = hidden_dim

    def conv(name, cin, cout, k, bias=True):
        s[name + ".weight"] = (cout, cin, k, k)
        if bias:
            s[name + ".bias"] = (cout,)

    def convT(name, cin, cout, k):
        s[name + ".weight"] = (cin, cout, k, k)
        s[name + ".bias"] = (cout,)

    def stack(name):
        for i in range(num_residual_layers):
            conv(f"{name}.layers.{i}.block.1", h, num_residual_hiddens, 3, bias=False)
            conv(f"{name}.layers.{i}.block.3", num_residual_hiddens, h, 1, bias=False)

    conv("encoder.layers.0", img_channels, h // 4, 4)
    conv("encoder.layers.2", h // 4, h // 2, 4)
    conv("encoder.layers.4", h // 2, h, 4)
    conv("encoder.layers.6", h, h, 3)
    stack("encoder.layers.7")
    conv("encoder.layers.8", h, embedding_dim, 1)
    conv("decoder.layers.0", embedding_dim, h, 3)
    stack("decoder.layers.1")
    convT("decoder.layers.2", h, h // 2, 4)
    convT("decoder.layers.4", h // 2, h // 4, 4)
    convT("decoder.layers.6", h // 4, img_channels, 4)
    s["vector_quantizer.embedding.weight"] = (num_embeddings, embedding_dim)
    return s


def vqvae_init(seed=0, **kw) -> Params:
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    shapes = vqvae_param_shapes(**kw)
    for name, shp in shapes.items():
        if name == "vector_quantizer.embedding.weight":
            K = shp[0]
            P[name] = (torch.rand(shp, generator=g) * 2 - 1) / K   # vector_quantizer.py:39-43
        elif name.endswith(".weight"):
            is_T = name.startswith("decoder.layers.") and name.split(".")[2] in ("2", "4", "6")
            fan_in = (shp[0] if is_T else shp[1]) * shp[2] * shp[3]
            b = 1.0 / math.sqrt(fan_in)
            P[name] = (torch.rand(shp, generator=g) * 2 - 1) * b
        else:
            P[name] = (torch.rand(shp, generator=g) * 2 - 1) * 0.05
    return P


def residual_stack(P, pre, x, n_layers):
    # residual.py:20-21 : `x + self.block(x)` where block[0] is ReLU(inplace=True):
    # the in-place ReLU rewrites x before the add, so the sum is relu(x) + block(x).
    for i in range(n_layers):
        r = F.relu(x)
        y = F.conv2d(r, P[f"{pre}.layers.{i}.block.1.weight"], padding=1)
        y = F.conv2d(F.relu(y), P[f"{pre}.layers.{i}.block.3.weight"])
        x = r + y
    return F.relu(x)


def encoder(P, x, n_layers=2):
    x = F.relu(F.conv2d(x, P["encoder.layers.0.weight"], P["encoder.layers.0.bias"], 2, 1))
    x = F.relu(F.conv2d(x, P["encoder.layers.2.weight"], P["encoder.layers.2.bias"], 2, 1))
    x = F.relu(F.conv2d(x, P["encoder.layers.4.weight"], P["encoder.layers.4.bias"], 2, 1))
    x = F.conv2d(x, P["encoder.layers.6.weight"], P["encoder.layers.6.bias"], 1, 1)
    x = residual_stack(P, "encoder.layers.7", x, n_layers)
    return F.conv2d(x, P["encoder.layers.8.weight"], P["encoder.layers.8.bias"])


def decoder(P, q, n_layers=2):
    x = F.conv2d(q, P["decoder.layers.0.weight"], P["decoder.layers.0.bias"], 1, 1)
    x = residual_stack(P, "decoder.layers.1", x, n_layers)
    x = F.relu(F.conv_transpose2d(x, P["decoder.layers.2.weight"], P["decoder.layers.2.bias"], 2, 1))
    x = F.relu(F.conv_transpose2d(x, P["decoder.layers.4.weight"], P["decoder.layers.4.bias"], 2, 1))
    return torch.tanh(F.conv_transpose2d(x, P["decoder.layers.6.weight"], P["decoder.layers.6.bias"], 2, 1))


def vq_distances(flat: torch.Tensor, codebook: torch.Tensor) -> torch.Tensor:
    # vector_quantizer.py:53-57 — same algebraic form / op order as the reference
    return ((flat ** 2).sum(dim=1, keepdim=True) + (codebook ** 2).sum(dim=1)
            - 2 * flat @ codebook.T)


def vq_indices(latents: torch.Tensor, codebook: torch.Tensor) -> torch.Tensor:
    B, D, H, W = latents.shape
    flat = latents.permute(0, 2, 3, 1).reshape(B * H * W, D)
    return vq_distances(flat, codebook).argmin(dim=1)


def ema_update(cluster_size, ema_embedding, idx, flat, K, decay, eps):
    """vector_quantizer.py:128-147; returns (cluster_size', ema_embedding', codebook')."""
    enc = F.one_hot(idx, K).float()
    cs = cluster_size * decay + enc.sum(0) * (1 - decay)
    n = cs.sum()
    cw = (cs + eps) / (n + K * eps) * n
    dw = enc.T @ flat
    ee = ema_embedding * decay + dw * (1 - decay)
    return cs, ee, ee / cw.unsqueeze(1)


def vector_quantizer(latents, codebook, commitment_cost=0.25, ema_state=None, decay=0.99, eps=1e-5):
    """VectorQuantizer.forward :30-35 (ema_state=None) or the EMA variant :149-179
    in training mode (ema_state=(cluster_size, ema_embedding); the codebook is
    replaced by the EMA estimate BEFORE the lookup, and the new state is returned).
    Returns (quantized_ste, vq_loss, perplexity, indices, new_state)."""
    B, D, H, W = latents.shape
    K = codebook.shape[0]
    flat = latents.permute(0, 2, 3, 1).reshape(B * H * W, D)
    idx = vq_distances(flat, codebook).argmin(dim=1)
    new_state = None
    if ema_state is not None:
        with torch.no_grad():
            cs, ee, cb = ema_update(ema_state[0], ema_state[1], idx, flat.detach(), K, decay, eps)
        new_state = (cs, ee, cb)
        # reference does embedding.weight.data.copy_(...): same Parameter, new values,
        # autograd still routes e_latent_loss gradient to it.
        codebook = codebook + (cb - codebook).detach()
    q = F.embedding(idx, codebook).reshape(B, H, W, D).permute(0, 3, 1, 2).contiguous()
    e_loss = F.mse_loss(q, latents.detach())
    q_loss = F.mse_loss(q.detach(), latents)
    vq_loss = e_loss + commitment_cost * q_loss
    q_ste = latents + (q - latents).detach()
    probs = F.one_hot(idx, K).float().mean(dim=0)
    perplexity = torch.exp(-torch.sum(probs * torch.log(probs + 1e-10)))
    return q_ste, vq_loss, perplexity, idx, new_state


def vqvae_step(P, x, n_layers=2, commitment_cost=0.25, w_recon=1.0, w_vq=1.0,
               ema_state=None, decay=0.99, eps=1e-5):
    """VQVAE._common_step (vqvae.py:169-199): returns dict of losses etc."""
    lat = encoder(P, x, n_layers)
    q, vq_loss, ppl, idx, new_state = vector_quantizer(
        lat, P["vector_quantizer.embedding.weight"], commitment_cost, ema_state, decay, eps)
    x_hat = decoder(P, q, n_layers)
    recon = F.mse_loss(x_hat, x)
    loss = w_recon * recon + w_vq * vq_loss
    return dict(loss=loss, recon_loss=recon, vq_loss=vq_loss, perplexity=ppl, indices=idx,
                x_hat=x_hat, latents=lat, ema_state=new_state)
