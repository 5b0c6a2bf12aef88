"""Reverse-diffusion sampling loops on the HIP engine.

Reference: GaussianDiffusion.p_sample_loop ddpm.py:759-780, ddim_sample :782-834,
model_predictions :707-734, p_sample :748-757.  The reference copies the image to the host
at EVERY step (``img.detach().cpu()`` :775,829); here the whole chain stays on the device:
per step = one UNet forward (HIP engine, NHWC) + one fused update kernel (lgm_sample_step)
whose per-timestep scalars come from host copies of the schedule buffers (no device syncs).
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from . import ops
from .flat import _r4


def _host_schedule(gd):
    hs = getattr(gd, "_host_sched", None)
    if hs is None:
        names = ["alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                 "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
                 "posterior_mean_coef2", "posterior_log_variance_clipped"]
        hs = {n: getattr(gd, n).detach().cpu() for n in names}
        gd._host_sched = hs
    return hs


def _f32(x) -> float:
    return float(torch.as_tensor(x, dtype=torch.float32))


class _Chain:
    """Device-resident state of one sampling run (NHWC, padded channels)."""

    def __init__(self, gd, shape, init_noise: Optional[torch.Tensor]):
        self.gd = gd
        self.net = gd.model
        B, C, H, W = shape
        self.shape = shape
        dev = gd.betas.device
        self.net.prepare_hip(dev)
        self.Cp = _r4(C)
        if init_noise is None:
            init_noise = torch.randn(shape, device=dev)
        self.x = torch.empty((B, H, W, self.Cp), device=dev)
        ops.nchw_to_nhwc(init_noise.float().contiguous(), self.x)
        self.x_next = torch.empty_like(self.x)
        self.x0 = torch.empty_like(self.x)
        self.tbuf = {}

    def times(self, t: int) -> torch.Tensor:
        tb = self.tbuf.get(t)
        if tb is None:
            tb = torch.full((self.shape[0],), t, device=self.x.device, dtype=torch.long)
            if len(self.tbuf) < 4096:
                self.tbuf[t] = tb
        return tb

    def step(self, t: int, noise: Optional[torch.Tensor], clip: bool, C0, C1, C2, C3):
        hs = _host_schedule(self.gd)
        B, C, H, W = self.shape
        v, _ = self.net.forward_nhwc(self.x, self.times(t), False)
        ops.lib().lgm_sample_step(self.x.data_ptr(), v.data_ptr(), None if noise is None else noise.data_ptr(),
                                  self.x_next.data_ptr(), self.x0.data_ptr(), B, C, H * W, self.Cp,
                                  _f32(hs["sqrt_alphas_cumprod"][t]), -_f32(hs["sqrt_one_minus_alphas_cumprod"][t]),
                                  1 if clip else 0, _f32(hs["sqrt_recip_alphas_cumprod"][t]),
                                  _f32(hs["sqrt_recipm1_alphas_cumprod"][t]), C0, C1, C2, C3, ops.stream())
        self.x, self.x_next = self.x_next, self.x

    def image(self, unnormalize: bool) -> torch.Tensor:
        B, C, H, W = self.shape
        out = torch.empty(self.shape, device=self.x.device)
        ops.nhwc_to_nchw(self.x, out)
        if unnormalize:
            out.mul_(0.5).add_(0.5)     # unnormalize_to_zero_to_one, once per sampling run
        return out


def p_sample_step(chain: _Chain, t: int, noise: Optional[torch.Tensor]):
    """One ancestral step (p_sample :748-757): clip x0, posterior mean + sigma * noise (t > 0)."""
    hs = _host_schedule(chain.gd)
    sigma = _f32(torch.as_tensor(0.5 * hs["posterior_log_variance_clipped"][t]).exp()) if t > 0 else 0.0
    chain.step(t, noise if t > 0 else None, True, _f32(hs["posterior_mean_coef1"][t]),
               _f32(hs["posterior_mean_coef2"][t]), 0.0, sigma)


def ddim_step(chain: _Chain, t: int, t_next: int, noise: Optional[torch.Tensor], eta: float):
    """One DDIM step (loop body :805-829)."""
    hs = _host_schedule(chain.gd)
    if t_next < 0:
        chain.step(t, None, True, 1.0, 0.0, 0.0, 0.0)
        return
    a, an = hs["alphas_cumprod"][t], hs["alphas_cumprod"][t_next]
    sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
    c = (1 - an - sigma ** 2).sqrt()
    chain.step(t, noise if float(sigma) != 0.0 else None, True, _f32(an.sqrt()), 0.0, _f32(c), _f32(sigma))


@torch.no_grad()
def p_sample_loop(gd, shape, return_all_timesteps=False, init_noise=None, noises: Optional[List[torch.Tensor]] = None):
    chain = _Chain(gd, shape, init_noise)
    dev = chain.x.device
    frames = [chain.image(False)] if return_all_timesteps else None
    for i, t in enumerate(reversed(range(gd.num_timesteps))):
        nz = None
        if t > 0:
            nz = noises[i] if noises is not None else torch.randn(shape, device=dev)
        p_sample_step(chain, t, nz)
        if return_all_timesteps:
            frames.append(chain.image(False))
    if return_all_timesteps:
        ret = torch.stack(frames, dim=1)
        return (ret + 1) * 0.5 if gd.auto_normalize else ret
    return chain.image(gd.auto_normalize)


@torch.no_grad()
def ddim_sample(gd, shape, return_all_timesteps=False, init_noise=None, noises: Optional[List[torch.Tensor]] = None):
    chain = _Chain(gd, shape, init_noise)
    dev = chain.x.device
    eta = gd.ddim_sampling_eta
    frames = [chain.image(False)] if return_all_timesteps else None
    for i, (t, t_next) in enumerate(gd.ddim_time_pairs()):
        nz = None
        if t_next >= 0 and eta != 0.0:
            nz = noises[i] if noises is not None else torch.randn(shape, device=dev)
        ddim_step(chain, t, t_next, nz, eta)
        if return_all_timesteps:
            frames.append(chain.image(False))
    if return_all_timesteps:
        ret = torch.stack(frames, dim=1)
        return (ret + 1) * 0.5 if gd.auto_normalize else ret
    return chain.image(gd.auto_normalize)
