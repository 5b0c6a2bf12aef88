"""Oracle: DDPM / DDIM UNet + Gaussian diffusion, functional fp32 CPU restatement.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The network is expressed as pure functions over a flat ``{name: tensor}``
dictionary whose keys are exactly the reference ``Unet.state_dict()`` keys
(``models/generative/diffusion/ddpm.py:275-471``), so the same weights can be
loaded into the reference, the oracle and the HIP engine.

Reference anchors (all in models/generative/diffusion/ddpm.py unless noted):
  Block            :157-173      ResnetBlock       :176-200
  RMSNorm          :107-113      SinusoidalPosEmb  :119-132
  LinearAttention  :203-239      Attention         :242-271 + modules/attend.py:97-126
  Up/Downsample    :93-104       Unet.forward      :428-471
  schedules        :491-529      buffers           :577-662
  q_sample         :869-876      p_losses          :878-925
  model_predictions:707-734      p_sample          :748-757
  ddim_sample      :782-834
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# architecture description
# --------------------------------------------------------------------------

def unet_dims(dim: int, dim_mults: Sequence[int] = (1, 2, 4, 8)) -> List[Tuple[int, int]]:
    """(dim_in, dim_out) per stage — ddpm.py:306-307."""
    dims = [dim] + [dim * m for m in dim_mults]
    return list(zip(dims[:-1], dims[1:]))


def unet_param_shapes(dim: int = 64, channels: int = 3,
                      dim_mults: Sequence[int] = (1, 2, 4, 8),
                      heads: int = 4, dim_head: int = 32, num_mem_kv: int = 4
                      ) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of every Unet parameter, in reference registration order."""
    shapes: Dict[str, Tuple[int, ...]] = {}
    time_dim = dim * 4
    hidden = heads * dim_head

    def conv(name, cin, cout, k, bias=True):
        shapes[name + ".weight"] = (cout, cin, k, k)
        if bias:
            shapes[name + ".bias"] = (cout,)

    def linear(name, cin, cout):
        shapes[name + ".weight"] = (cout, cin)
        shapes[name + ".bias"] = (cout,)

    def resblock(name, cin, cout):
        linear(name + ".mlp.1", time_dim, cout * 2)
        for blk, ci in (("block1", cin), ("block2", cout)):
            conv(f"{name}.{blk}.proj", ci, cout, 3)
            shapes[f"{name}.{blk}.norm.weight"] = (cout,)
            shapes[f"{name}.{blk}.norm.bias"] = (cout,)
        if cin != cout:
            conv(name + ".res_conv", cin, cout, 1)

    def lin_attn(name, c):
        shapes[name + ".norm.g"] = (1, c, 1, 1)
        shapes[name + ".mem_kv"] = (2, heads, dim_head, num_mem_kv)
        conv(name + ".to_qkv", c, hidden * 3, 1, bias=False)
        conv(name + ".to_out.0", hidden, c, 1)
        shapes[name + ".to_out.1.g"] = (1, c, 1, 1)

    def full_attn(name, c):
        shapes[name + ".norm.g"] = (1, c, 1, 1)
        shapes[name + ".mem_kv"] = (2, heads, num_mem_kv, dim_head)
        conv(name + ".to_qkv", c, hidden * 3, 1, bias=False)
        conv(name + ".to_out", hidden, c, 1)

    conv("init_conv", channels, dim, 7)
    linear("time_mlp.1", dim, time_dim)
    linear("time_mlp.3", time_dim, time_dim)

    in_out = unet_dims(dim, dim_mults)
    n = len(in_out)
    for i, (ci, co) in enumerate(in_out):
        last = i == n - 1
        resblock(f"downs.{i}.0", ci, ci)
        resblock(f"downs.{i}.1", ci, ci)
        (full_attn if last else lin_attn)(f"downs.{i}.2", ci)
        if last:
            conv(f"downs.{i}.3", ci, co, 3)
        else:
            conv(f"downs.{i}.3.1", ci * 4, co, 1)
    mid = in_out[-1][1]
    resblock("mid_block1", mid, mid)
    full_attn("mid_attn", mid)
    resblock("mid_block2", mid, mid)
    for i, (ci, co) in enumerate(reversed(in_out)):
        last = i == n - 1
        is_full = i == 0
        resblock(f"ups.{i}.0", co + ci, co)
        resblock(f"ups.{i}.1", co + ci, co)
        (full_attn if is_full else lin_attn)(f"ups.{i}.2", co)
        if last:
            conv(f"ups.{i}.3", co, ci, 3)
        else:
            conv(f"ups.{i}.3.1", co, ci, 3)
    resblock("final_res_block", dim * 2, dim)
    conv("final_conv", dim, channels, 1)
    return shapes


def unet_init(dim: int = 64, channels: int = 3, seed: int = 0,
              dim_mults: Sequence[int] = (1, 2, 4, 8)) -> Params:
    """Random-init weights with torch-default-like statistics (kaiming-uniform
    fan-in bounds for conv/linear, U(0.5,1.5) / U(-0.3,0.3) for the norms' scale / shift, N(0,1) mem_kv).  The
    exact RNG stream of the reference constructor is NOT reproduced (parity tests
    always load identical weights into both sides)."""
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    for name, shp in unet_param_shapes(dim, channels, dim_mults).items():
        if name.endswith(".g") or name.endswith("norm.weight"):
            # NOT the constructor's ones/zeros: a mis-routed gamma / beta / g pointer must be visible in
            # every network-level comparison, so the affine parameters are random too
            P[name] = 0.5 + torch.rand(shp, generator=g)
        elif name.endswith("norm.bias"):
            P[name] = (torch.rand(shp, generator=g) * 2 - 1) * 0.3
        elif name.endswith("mem_kv"):
            P[name] = torch.randn(shp, generator=g)
        elif name.endswith(".weight"):
            fan_in = int(torch.tensor(shp[1:]).prod())
            b = 1.0 / math.sqrt(fan_in)
            P[name] = (torch.rand(shp, generator=g) * 2 - 1) * b
        else:  # conv / linear bias
            w = P[name[:-len("bias")] + "weight"]
            fan_in = int(torch.tensor(w.shape[1:]).prod())
            b = 1.0 / math.sqrt(fan_in)
            P[name] = (torch.rand(shp, generator=g) * 2 - 1) * b
    return P


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------

def rms_norm(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    # ddpm.py:112-113 : F.normalize(x, dim=1) * g * sqrt(C)   (eps 1e-12 on the norm)
    nrm = x.pow(2).sum(dim=1, keepdim=True).sqrt().clamp_min(1e-12)
    return x / nrm * g * (x.shape[1] ** 0.5)


def sinusoidal_pos_emb(t: torch.Tensor, dim: int, theta: float = 10000.0) -> torch.Tensor:
    # ddpm.py:125-132
    half = dim // 2
    step = math.log(theta) / (half - 1)
    freqs = torch.exp(torch.arange(half, device=t.device) * -step)
    arg = t[:, None] * freqs[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def time_mlp(P: Params, t: torch.Tensor, dim: int) -> torch.Tensor:
    # ddpm.py:328-333  Linear -> GELU(erf) -> Linear
    e = sinusoidal_pos_emb(t, dim)
    e = F.linear(e, P["time_mlp.1.weight"], P["time_mlp.1.bias"])
    e = F.gelu(e)
    return F.linear(e, P["time_mlp.3.weight"], P["time_mlp.3.bias"])


def block(P: Params, pre: str, x, scale_shift=None, groups: int = 8):
    # ddpm.py:164-173
    x = F.conv2d(x, P[pre + ".proj.weight"], P[pre + ".proj.bias"], padding=1)
    x = F.group_norm(x, groups, P[pre + ".norm.weight"], P[pre + ".norm.bias"], eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return F.silu(x)


def resnet_block(P: Params, pre: str, x, temb):
    # ddpm.py:189-200
    ss = F.linear(F.silu(temb), P[pre + ".mlp.1.weight"], P[pre + ".mlp.1.bias"])
    ss = ss[:, :, None, None]
    scale, shift = ss.chunk(2, dim=1)
    h = block(P, pre + ".block1", x, (scale, shift))
    h = block(P, pre + ".block2", h)
    if (pre + ".res_conv.weight") in P:
        x = F.conv2d(x, P[pre + ".res_conv.weight"], P[pre + ".res_conv.bias"])
    return h + x


def linear_attention(P: Params, pre: str, x, heads: int = 4, dim_head: int = 32):
    # ddpm.py:217-239
    b, c, hh, ww = x.shape
    n = hh * ww
    xn = rms_norm(x, P[pre + ".norm.g"])
    qkv = F.conv2d(xn, P[pre + ".to_qkv.weight"])
    q, k, v = (t.reshape(b, heads, dim_head, n) for t in qkv.chunk(3, dim=1))
    mk, mv = (m.unsqueeze(0).expand(b, -1, -1, -1) for m in P[pre + ".mem_kv"])
    k = torch.cat((mk, k), dim=-1)
    v = torch.cat((mv, v), dim=-1)
    q = q.softmax(dim=-2) * (dim_head ** -0.5)
    k = k.softmax(dim=-1)
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q)
    out = out.reshape(b, heads * dim_head, hh, ww)
    out = F.conv2d(out, P[pre + ".to_out.0.weight"], P[pre + ".to_out.0.bias"])
    return rms_norm(out, P[pre + ".to_out.1.g"])


def full_attention(P: Params, pre: str, x, heads: int = 4, dim_head: int = 32):
    # ddpm.py:255-271 + attend.py:106-126 (non-flash branch, dropout 0)
    b, c, hh, ww = x.shape
    n = hh * ww
    xn = rms_norm(x, P[pre + ".norm.g"])
    qkv = F.conv2d(xn, P[pre + ".to_qkv.weight"])
    q, k, v = (t.reshape(b, heads, dim_head, n).transpose(-1, -2) for t in qkv.chunk(3, dim=1))
    mk, mv = (m.unsqueeze(0).expand(b, -1, -1, -1) for m in P[pre + ".mem_kv"])
    k = torch.cat((mk, k), dim=-2)
    v = torch.cat((mv, v), dim=-2)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * (dim_head ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, v)
    out = out.transpose(-1, -2).reshape(b, heads * dim_head, hh, ww)
    return F.conv2d(out, P[pre + ".to_out.weight"], P[pre + ".to_out.bias"])


def pixel_unshuffle_cp1p2(x):
    # Rearrange("b c (h p1) (w p2) -> b (c p1 p2) h w", p1=2, p2=2)  ddpm.py:102
    b, c, H, W = x.shape
    x = x.reshape(b, c, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4)
    return x.reshape(b, c * 4, H // 2, W // 2)


def unet_forward(P: Params, x: torch.Tensor, time: torch.Tensor, dim: int = 64,
                 dim_mults: Sequence[int] = (1, 2, 4, 8),
                 taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """ddpm.py:428-471.  ``taps`` (optional) collects named intermediates."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    in_out = unet_dims(dim, dim_mults)
    n = len(in_out)
    x = F.conv2d(x, P["init_conv.weight"], P["init_conv.bias"], padding=3)
    tap("init_conv", x)
    r = x
    temb = tap("time_mlp", time_mlp(P, time, dim))
    hs = []
    for i in range(n):
        last = i == n - 1
        x = resnet_block(P, f"downs.{i}.0", x, temb)
        tap(f"downs.{i}.0", x)
        hs.append(x)
        x = resnet_block(P, f"downs.{i}.1", x, temb)
        att = full_attention if last else linear_attention
        x = att(P, f"downs.{i}.2", x) + x
        tap(f"downs.{i}.2", x)
        hs.append(x)
        if last:
            x = F.conv2d(x, P[f"downs.{i}.3.weight"], P[f"downs.{i}.3.bias"], padding=1)
        else:
            x = F.conv2d(pixel_unshuffle_cp1p2(x), P[f"downs.{i}.3.1.weight"], P[f"downs.{i}.3.1.bias"])
        tap(f"downs.{i}.3", x)
    x = resnet_block(P, "mid_block1", x, temb)
    x = full_attention(P, "mid_attn", x) + x
    x = resnet_block(P, "mid_block2", x, temb)
    tap("mid", x)
    for i in range(n):
        last = i == n - 1
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(P, f"ups.{i}.0", x, temb)
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(P, f"ups.{i}.1", x, temb)
        att = full_attention if i == 0 else linear_attention
        x = att(P, f"ups.{i}.2", x) + x
        if last:
            x = F.conv2d(x, P[f"ups.{i}.3.weight"], P[f"ups.{i}.3.bias"], padding=1)
        else:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = F.conv2d(x, P[f"ups.{i}.3.1.weight"], P[f"ups.{i}.3.1.bias"], padding=1)
        tap(f"ups.{i}.3", x)
    x = torch.cat((x, r), dim=1)
    x = resnet_block(P, "final_res_block", x, temb)
    return F.conv2d(x, P["final_conv.weight"], P["final_conv.bias"])


# --------------------------------------------------------------------------
# Gaussian diffusion
# --------------------------------------------------------------------------

def sigmoid_beta_schedule(timesteps: int, start=-3, end=3, tau=1) -> torch.Tensor:
    # ddpm.py:514-529 (float64)
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0, 0.999)


def linear_beta_schedule(timesteps: int) -> torch.Tensor:
    # ddpm.py:491-498
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def cosine_beta_schedule(timesteps: int, s: float = 0.008) -> torch.Tensor:
    # ddpm.py:501-511
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=torch.float64) / timesteps
    ac = torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0, 0.999)


def diffusion_buffers(timesteps: int = 1000, schedule: str = "sigmoid",
                      objective: str = "pred_v") -> Dict[str, torch.Tensor]:
    """The 13 fp32 buffers of GaussianDiffusion (ddpm.py:577-662), computed in
    float64 and cast to float32 exactly as the reference does."""
    fn = {"sigmoid": sigmoid_beta_schedule, "linear": linear_beta_schedule,
          "cosine": cosine_beta_schedule}[schedule]
    betas = fn(timesteps)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    snr = ac / (1 - ac)
    if objective == "pred_noise":
        lw = snr / snr
    elif objective == "pred_x0":
        lw = snr
    else:
        lw = snr / (snr + 1)
    bufs = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(post_var.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
        "loss_weight": lw,
    }
    return {k: v.to(torch.float32) for k, v in bufs.items()}


def _ext(a: torch.Tensor, t: torch.Tensor, ndim: int) -> torch.Tensor:
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))


def q_sample(bufs, x0, t, noise):
    # ddpm.py:869-876
    return (_ext(bufs["sqrt_alphas_cumprod"], t, x0.ndim) * x0
            + _ext(bufs["sqrt_one_minus_alphas_cumprod"], t, x0.ndim) * noise)


def predict_v(bufs, x0, t, noise):
    # ddpm.py:684-688
    return (_ext(bufs["sqrt_alphas_cumprod"], t, x0.ndim) * noise
            - _ext(bufs["sqrt_one_minus_alphas_cumprod"], t, x0.ndim) * x0)


def p_losses(P, bufs, x0, t, noise, dim=64, dim_mults=(1, 2, 4, 8), return_out=False):
    """ddpm.py:878-925 with objective pred_v, no self-cond, no offset noise.
    ``x0`` is already normalised (img*2-1)."""
    x = q_sample(bufs, x0, t, noise)
    out = unet_forward(P, x, t, dim, dim_mults)
    target = predict_v(bufs, x0, t, noise)
    loss = F.mse_loss(out, target, reduction="none")
    loss = loss.reshape(loss.shape[0], -1).mean(dim=1)
    loss = loss * bufs["loss_weight"].gather(-1, t)
    loss = loss.mean()
    return (loss, out) if return_out else loss


def diffusion_forward(P, bufs, img, t, noise, **kw):
    """GaussianDiffusion.forward (ddpm.py:927-946) with t and noise injected:
    auto_normalize img*2-1 then p_losses."""
    return p_losses(P, bufs, img * 2 - 1, t, noise, **kw)


def model_predictions(P, bufs, x, t, clip_x_start=False, dim=64, dim_mults=(1, 2, 4, 8)):
    # ddpm.py:707-734, pred_v branch
    v = unet_forward(P, x, t, dim, dim_mults)
    x_start = (_ext(bufs["sqrt_alphas_cumprod"], t, x.ndim) * x
               - _ext(bufs["sqrt_one_minus_alphas_cumprod"], t, x.ndim) * v)
    if clip_x_start:
        x_start = x_start.clamp(-1.0, 1.0)
    pred_noise = ((_ext(bufs["sqrt_recip_alphas_cumprod"], t, x.ndim) * x - x_start)
                  / _ext(bufs["sqrt_recipm1_alphas_cumprod"], t, x.ndim))
    return pred_noise, x_start, v


def p_sample(P, bufs, x, t_int: int, noise, dim=64, dim_mults=(1, 2, 4, 8)):
    """ddpm.py:748-757 (+ p_mean_variance :736-746, q_posterior :696-705).
    ``noise`` is injected (ignored when t == 0)."""
    b = x.shape[0]
    t = torch.full((b,), t_int, dtype=torch.long)
    _, x_start, _ = model_predictions(P, bufs, x, t, clip_x_start=False, dim=dim, dim_mults=dim_mults)
    x_start = x_start.clamp(-1.0, 1.0)
    mean = (_ext(bufs["posterior_mean_coef1"], t, x.ndim) * x_start
            + _ext(bufs["posterior_mean_coef2"], t, x.ndim) * x)
    logvar = _ext(bufs["posterior_log_variance_clipped"], t, x.ndim)
    if t_int > 0:
        return mean + (0.5 * logvar).exp() * noise, x_start
    return mean, x_start


def ddim_time_pairs(total_timesteps: int, sampling_timesteps: int) -> List[Tuple[int, int]]:
    # ddpm.py:792-798
    times = torch.linspace(-1, total_timesteps - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


def ddim_step(P, bufs, img, time: int, time_next: int, noise, eta: float = 0.0,
              dim=64, dim_mults=(1, 2, 4, 8)):
    """One iteration of the loop body ddpm.py:805-829."""
    b = img.shape[0]
    t = torch.full((b,), time, dtype=torch.long)
    pred_noise, x_start, _ = model_predictions(P, bufs, img, t, clip_x_start=True,
                                               dim=dim, dim_mults=dim_mults)
    if time_next < 0:
        return x_start, x_start
    alpha = bufs["alphas_cumprod"][time]
    alpha_next = bufs["alphas_cumprod"][time_next]
    sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
    c = (1 - alpha_next - sigma ** 2).sqrt()
    img = x_start * alpha_next.sqrt() + c * pred_noise + sigma * noise
    return img, x_start


def draw_loop_noise(seed: int, shape, n_steps: int) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """The draws the reference loops make from the global CPU generator after ``torch.manual_seed(seed)``:
    ``randn(shape)`` for the start image, then one ``randn_like`` per step that draws (ddpm.py:763,755 for the
    ancestral loop: every t > 0; ddpm.py:802,825 for DDIM: every pair with time_next >= 0)."""
    g = torch.Generator().manual_seed(seed)
    init = torch.randn(shape, generator=g)
    return init, [torch.randn(shape, generator=g) for _ in range(n_steps)]


def p_sample_loop(P, bufs, init, noises, dim=64, dim_mults=(1, 2, 4, 8)):
    """ddpm.py:759-780: T ancestral steps from ``init``; returns the unnormalised image ((x+1)/2)."""
    T = bufs["betas"].shape[0]
    img = init
    for i, t in enumerate(reversed(range(T))):
        img, _ = p_sample(P, bufs, img, t, noises[i] if t > 0 else None, dim=dim, dim_mults=dim_mults)
    return (img + 1) * 0.5


def ddim_sample_loop(P, bufs, init, noises, sampling_timesteps: int, eta: float = 0.0, dim=64,
                     dim_mults=(1, 2, 4, 8)):
    """ddpm.py:782-834; returns the unnormalised image."""
    T = bufs["betas"].shape[0]
    img = init
    for i, (a, b) in enumerate(ddim_time_pairs(T, sampling_timesteps)):
        img, _ = ddim_step(P, bufs, img, a, b, noises[i] if b >= 0 else None, eta=eta, dim=dim, dim_mults=dim_mults)
    return (img + 1) * 0.5
