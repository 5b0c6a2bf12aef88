"""Reverse-diffusion sampling loops on the HIP engine.

Reference: GaussianDiffusion.p_sample_loop ddpm.py:759-780, ddim_sample :782-834,
model_predictions :707-734, p_sample :748-757.  The reference copies the image to the host
at EVERY step (``img.detach().cpu()`` :775,829); here the whole chain stays on the device:
per step = one UNet forward (HIP engine, NHWC) + one fused update kernel.

Two ways to run a chain, bit-identical in their results (tests/test_hip_unet.py):
  * graph replay (default on the GPU when only the final image is wanted): ONE HIP graph per (network, shape)
    holds a whole step — t[b] <- device table[step counter], UNet forward, noise draw, in-place update with the
    step's scalars read from a device table, counter += 1 — and is replayed once per step: no Python between
    the ~230 launches of a step, no host syncs, no per-step allocations (``lgm_sample_step_table``);
  * eager launches (``return_all_timesteps``, ``LGM_NO_SAMPLER_GRAPH=1``, or when capture fails): per-timestep
    scalars from host copies of the schedule buffers (``lgm_sample_step``).
"""
from __future__ import annotations

import math
import os
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


def _p_sample_coeffs(gd, t: int):
    """(A, Bv, R, Rm1, C0, C1, C2, C3) of one ancestral step — the scalars p_sample_step hands to the kernel"""
    hs = _host_schedule(gd)
    sigma = _f32(torch.as_tensor(0.5 * hs["posterior_log_variance_clipped"][t]).exp()) if t > 0 else 0.0
    return (_f32(hs["sqrt_alphas_cumprod"][t]), -_f32(hs["sqrt_one_minus_alphas_cumprod"][t]),
            _f32(hs["sqrt_recip_alphas_cumprod"][t]), _f32(hs["sqrt_recipm1_alphas_cumprod"][t]),
            _f32(hs["posterior_mean_coef1"][t]), _f32(hs["posterior_mean_coef2"][t]), 0.0, sigma)


def _ddim_coeffs(gd, t: int, t_next: int, eta: float):
    hs = _host_schedule(gd)
    head = (_f32(hs["sqrt_alphas_cumprod"][t]), -_f32(hs["sqrt_one_minus_alphas_cumprod"][t]),
            _f32(hs["sqrt_recip_alphas_cumprod"][t]), _f32(hs["sqrt_recipm1_alphas_cumprod"][t]))
    if t_next < 0:
        return head + (1.0, 0.0, 0.0, 0.0)
    a, an = hs["alphas_cumprod"][t], hs["alphas_cumprod"][t_next]
    sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
    c = (1 - an - sigma ** 2).sqrt()
    return head + (_f32(an.sqrt()), 0.0, _f32(c), _f32(sigma))


# net -> {(shape, with_noise): _GraphedChain}.  Weak on the network: a sampled model that goes away takes its graphs
# (and their memory pool) with it.  Every entry remembers which flat parameter storage its launches were captured
# against (see _GraphedChain.matches): a graph bakes buffer ADDRESSES in, so after prepare_hip() rebuilt the flat
# storage (model.to(), replaced parameter storage) the entry is dropped and the step recaptured.
import weakref

_GRAPHS = weakref.WeakKeyDictionary()
_CAPTURE_RETRY_AFTER = 8      # a failed capture is retried after this many eager chains, not cached for ever


class _GraphedChain:
    """One captured sampling step for a (network, batch shape); replayed once per step of any chain on it."""

    def __init__(self, gd, shape, with_noise: bool, max_steps: int = 4096):
        net = gd.model
        self._net = weakref.ref(net)                 # the cache is keyed weakly on the network: no strong reference here
        B, C, H, W = shape
        dev = gd.betas.device
        self.shape, self.with_noise = shape, with_noise
        fp = net._flat
        # identity of everything whose address the captured launches carry: the flat object and its buffers
        self._bound = (weakref.ref(fp), fp.data.data_ptr(),
                       None if fp.data_uf is None else fp.data_uf.data_ptr(),
                       None if fp.data_t is None else fp.data_t.data_ptr())
        Cp = _r4(C)
        self.x = torch.zeros((B, H, W, Cp), device=dev)
        self.t = torch.zeros(B, dtype=torch.long, device=dev)
        self.noise = torch.zeros(shape, device=dev) if with_noise else None     # injected noise goes here
        self.table = torch.zeros((max_steps, 8), device=dev)
        self.ttable = torch.zeros(max_steps, dtype=torch.long, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.inject = False
        self.max_steps = max_steps
        L = ops.lib()

        def one_step():
            st = ops.stream()
            L.lgm_sampler_time(self.ttable.data_ptr(), self.counter.data_ptr(), self.t.data_ptr(), B, st)
            v, _ = net.forward_nhwc(self.x, self.t, False, refresh_weights=False)
            nz = None
            if with_noise:
                nz = self.noise if self.inject else torch.randn(shape, device=dev)
            L.lgm_sample_step_table(self.x.data_ptr(), v.data_ptr(), None if nz is None else nz.data_ptr(), None, B, C,
                                    H * W, Cp, self.table.data_ptr(), self.counter.data_ptr(), 1, 1, ops.stream())

        net.refresh_derived_weights(False)
        rng_state = torch.cuda.get_rng_state(dev)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):            # eager warm-up (sizes workspaces, sets kernel attributes)
            for _ in range(2):
                one_step()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graphs = {}
        for inject in ((False, True) if with_noise else (False,)):
            self.inject = inject
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                one_step()
            self.graphs[inject] = g
        torch.cuda.set_rng_state(rng_state, dev)     # capture leaves the random stream where it was

    def matches(self, net) -> bool:
        """True while the network still owns the flat storage this step was captured against."""
        fp = net._flat
        ref, data, uf, dt = self._bound
        return (fp is not None and ref() is fp and fp.still_bound() and fp.data.data_ptr() == data
                and (None if fp.data_uf is None else fp.data_uf.data_ptr()) == uf
                and (None if fp.data_t is None else fp.data_t.data_ptr()) == dt)

    def run(self, x0_nhwc, times, coeffs, noises):
        """times[i], coeffs[i] (8 floats) per step; noises: None (draw on device) or a list with one NCHW tensor or
        None per step.  Returns the final NHWC image (a view of the static buffer)."""
        n = len(times)
        assert n <= self.max_steps
        self._net().refresh_derived_weights(False)   # the weights may have moved since the last chain (EMA updates)
        self.x.copy_(x0_nhwc)
        self.table[:n].copy_(torch.tensor(coeffs, dtype=torch.float32), non_blocking=False)
        self.ttable[:n].copy_(torch.tensor(times, dtype=torch.long))
        self.counter.zero_()
        for i in range(n):
            if noises is not None and self.with_noise and noises[i] is not None:
                self.noise.copy_(noises[i])
                self.graphs[True].replay()
            elif noises is not None and self.with_noise:
                self.noise.zero_()
                self.graphs[True].replay()
            else:
                self.graphs[False].replay()
        return self.x


def _graph_chain(gd, shape, with_noise: bool):
    """-> a _GraphedChain for (network, shape), or None (graph replay disabled / capture failed: eager launches)"""
    if os.environ.get("LGM_NO_SAMPLER_GRAPH", "0") == "1" or gd.betas.device.type != "cuda":
        return None
    net = gd.model
    net.prepare_hip(gd.betas.device)                 # may rebuild the flat storage (model.to(), new parameter storage)
    per_net = _GRAPHS.get(net)
    if per_net is None:
        per_net = {}
        _GRAPHS[net] = per_net
    key = (tuple(shape), bool(with_noise))
    ent = per_net.get(key)
    if isinstance(ent, _GraphedChain) and not ent.matches(net):
        ent = None                                   # captured against buffers the network no longer uses
        per_net.pop(key, None)
    if isinstance(ent, int):                         # a capture failed earlier: eager for a while, then try again
        if ent > 0:
            per_net[key] = ent - 1
            return None
        ent = None
    if ent is None:
        try:
            ent = _GraphedChain(gd, tuple(shape), with_noise)
        except Exception as e:  # capture is an optimisation
            import sys
            print(f"[lgm_hip] sampler graph capture unavailable ({type(e).__name__}: {e}); eager launches",
                  file=sys.stderr, flush=True)
            per_net[key] = _CAPTURE_RETRY_AFTER
            return None
        per_net[key] = ent
    return ent


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
def warm_chain(gd, shape, replays: int = 20) -> bool:
    """Capture the ancestral chain's per-step graph for (network, shape) and run ``replays`` steps of it on noise
    (benchmarks: warm-up without paying a whole 1000-step chain).  False when graph replay is unavailable."""
    gc = _graph_chain(gd, shape, True)
    if gc is None:
        return False
    ts = list(reversed(range(gd.num_timesteps)))[:replays]
    gc.run(_Chain(gd, shape, None).x, ts, [_p_sample_coeffs(gd, t) for t in ts], None)
    return True


@torch.no_grad()
def p_sample_loop(gd, shape, return_all_timesteps=False, init_noise=None, noises: Optional[List[torch.Tensor]] = None,
                  start: Optional[int] = None, unnormalize: Optional[bool] = None):
    """``start``: walk the chain from step start - 1 down to 0 (GaussianDiffusion.interpolate :861-865) instead of from
    T - 1; ``unnormalize``: default = the model's auto_normalize (p_sample_loop :779), False for interpolate."""
    chain = _Chain(gd, shape, init_noise)
    dev = chain.x.device
    ts = list(reversed(range(gd.num_timesteps if start is None else int(start))))
    unn = gd.auto_normalize if unnormalize is None else bool(unnormalize)
    gc = None if return_all_timesteps or not ts else _graph_chain(gd, shape, True)
    if gc is not None:
        x = gc.run(chain.x, ts, [_p_sample_coeffs(gd, t) for t in ts], noises)
        chain.x = x
        return chain.image(unn)
    frames = [chain.image(False)] if return_all_timesteps else None
    for i, t in enumerate(ts):
        nz = None
        if t > 0:
            nz = noises[i] if noises is not None else torch.randn(shape, device=dev)
        p_sample_step(chain, t, nz)
        if return_all_timesteps:
            frames.append(chain.image(False))
    if return_all_timesteps:
        ret = torch.stack(frames, dim=1)
        return (ret + 1) * 0.5 if unn else ret
    return chain.image(unn)


@torch.no_grad()
def ddim_sample(gd, shape, return_all_timesteps=False, init_noise=None, noises: Optional[List[torch.Tensor]] = None):
    chain = _Chain(gd, shape, init_noise)
    dev = chain.x.device
    eta = gd.ddim_sampling_eta
    pairs = gd.ddim_time_pairs()
    gc = None if return_all_timesteps else _graph_chain(gd, shape, eta != 0.0)
    if gc is not None:
        x = gc.run(chain.x, [a for a, _ in pairs], [_ddim_coeffs(gd, a, b, eta) for a, b in pairs],
                   noises if eta != 0.0 else None)
        chain.x = x
        return chain.image(gd.auto_normalize)
    frames = [chain.image(False)] if return_all_timesteps else None
    for i, (t, t_next) in enumerate(pairs):
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
