"""Generate tests/golden/*.npz by running the REAL reference on CPU.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference,
which never travels to the GPU box).  Usage:  python oracle/make_golden.py

The reference's trainer / logging packages (pytorch_lightning, wandb, torchinfo,
torchvision.utils, ema_pytorch, torchmetrics) are not installed; they are replaced by
inert import-time stubs that touch no arithmetic.  Weights are created by
the oracle's seeded initialisers and loaded into the reference modules with
``load_state_dict(strict=True)`` — which also pins the oracle's parameter names/shapes.
Only inputs seeds + expected outputs are stored (no reference source, no weights).
"""
from __future__ import annotations

import os
import sys
import types
import typing

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def install_stubs():
    import torch.nn as nn

    class AttrDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self._hp = AttrDict()
            self.global_step = 0
            self.automatic_optimization = True

        def save_hyperparameters(self):
            import inspect
            frame = inspect.currentframe().f_back
            loc = frame.f_locals
            for k, v in loc.items():
                if k not in ("self", "__class__") and not k.startswith("_"):
                    self._hp[k] = v

        @property
        def hparams(self):
            return self._hp

        @property
        def device(self):
            return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    pl = mod("pytorch_lightning", LightningModule=LightningModule, Trainer=_Dummy,
             seed_everything=lambda *a, **k: None)
    pl.strategies = mod("pytorch_lightning.strategies", DDPStrategy=_Dummy,
                        SingleDeviceStrategy=_Dummy, Strategy=_Dummy)
    pl.callbacks = mod("pytorch_lightning.callbacks", ModelCheckpoint=_Dummy, Callback=_Dummy)
    pl.loggers = mod("pytorch_lightning.loggers", WandbLogger=_Dummy)
    mod("wandb", Image=_Dummy, Table=_Dummy, Artifact=_Dummy)
    mod("torchinfo", summary=lambda *a, **k: None)
    tv = mod("torchvision")
    tv.utils = mod("torchvision.utils", make_grid=lambda *a, **k: None)
    tm = mod("torchmetrics")
    tm.image = mod("torchmetrics.image")
    mod("torchmetrics.image.fid", FrechetInceptionDistance=_Dummy)
    mod("torchmetrics.image.inception", InceptionScore=_Dummy)
    mod("torchmetrics.image.kid", KernelInceptionDistance=_Dummy)

    class EMA(nn.Module):
        def __init__(self, model, beta=0.995, update_every=10, **k):
            super().__init__()
            import copy
            self.online_model = model
            self.ema_model = copy.deepcopy(model)

        @property
        def model(self):
            return self.online_model

        def update(self):
            pass

    mod("ema_pytorch", EMA=EMA)
    torch.List = typing.List  # dcgan.py:14 `from torch import List`
    sys.path.insert(0, REF)


def to_np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def gen_diffusion():
    from models.generative.diffusion import ddpm as R
    from oracle import diffusion as O

    out = {}
    # ---- schedules / buffers -------------------------------------------------
    torch.manual_seed(0)
    unet = R.Unet(dim=16, channels=3)
    gd = R.GaussianDiffusion(unet, img_size=16, timesteps=1000)
    idx = [0, 1, 2, 10, 100, 250, 500, 750, 900, 990, 998, 999]
    sched = {"idx": np.asarray(idx)}
    for name in O.diffusion_buffers(1000):
        sched[name] = getattr(gd, name)[idx].numpy()
        # full-array checksum in float64
        sched[name + "__sum"] = np.float64(getattr(gd, name).double().sum().item())
    for S in (50, 250, 1000):
        gd_s = R.GaussianDiffusion(unet, img_size=16, timesteps=1000, sampling_timesteps=S)
        times = torch.linspace(-1, 999, steps=S + 1)
        times = list(reversed(times.int().tolist()))
        sched[f"ddim_times_{S}"] = np.asarray(times, dtype=np.int64)
        sched[f"is_ddim_{S}"] = np.asarray(gd_s.is_ddim_sampling)
    # sinusoidal known answers
    pe = R.SinusoidalPosEmb(64)(torch.tensor([0, 1, 17, 999]))
    sched["posemb_t"] = np.asarray([0, 1, 17, 999])
    sched["posemb"] = pe.numpy()
    np.savez_compressed(os.path.join(OUT, "diffusion_schedule.npz"), **sched)

    # ---- UNet / p_losses, small (dim 16, 16x16) and full (dim 64, 32x32) ------
    for tag, dim, S, B, seed in (("small", 16, 16, 2, 1), ("full", 64, 32, 2, 2), ("full64", 64, 64, 2, 3)):
        P = O.unet_init(dim=dim, channels=3, seed=seed)
        unet = R.Unet(dim=dim, channels=3)
        missing = unet.load_state_dict(P, strict=True)
        gd = R.GaussianDiffusion(unet, img_size=S, timesteps=1000, sampling_timesteps=50)
        g = torch.Generator().manual_seed(100 + seed)
        img = torch.rand(B, 3, S, S, generator=g)            # data in [0,1]; forward() does *2-1
        noise = torch.randn(B, 3, S, S, generator=g)
        t = torch.tensor([37, 912][:B])
        x0 = img * 2 - 1
        for p in unet.parameters():
            p.grad = None
        x_t = gd.q_sample(x0, t, noise)
        v_out = unet(x_t, t)
        loss = gd.p_losses(x0, t, noise)
        loss.backward()
        fx = {"seed": seed, "dim": dim, "S": S, "B": B, "data_seed": 100 + seed,
              "t": t.numpy(), "x_t": x_t.detach().numpy(), "unet_out": v_out.detach().numpy(),
              "loss": loss.detach().numpy()}
        sd = dict(unet.named_parameters())
        gnames = ["init_conv.weight", "init_conv.bias", "time_mlp.1.weight", "time_mlp.3.bias",
                  "downs.0.0.mlp.1.weight", "downs.0.0.block1.proj.weight", "downs.0.0.block1.norm.weight",
                  "downs.0.0.block1.norm.bias", "downs.0.2.mem_kv", "downs.0.2.norm.g",
                  "downs.0.2.to_qkv.weight", "downs.0.2.to_out.0.bias", "downs.0.2.to_out.1.g",
                  "downs.0.3.1.weight", "downs.3.2.mem_kv", "downs.3.2.to_out.weight", "downs.3.3.weight",
                  "mid_attn.to_qkv.weight", "mid_block1.block2.proj.bias", "ups.0.0.res_conv.weight",
                  "ups.1.2.mem_kv", "ups.2.3.1.weight", "ups.3.3.bias", "final_res_block.res_conv.weight",
                  "final_conv.weight", "final_conv.bias"]
        for n in gnames:
            gten = sd[n].grad
            if tag == "small":
                fx["grad:" + n] = gten.numpy()
            else:  # full-size: store norm + a strided sample to keep the file small
                flat = gten.reshape(-1)
                fx["gradnorm:" + n] = np.float64(flat.double().norm().item())
                fx["gradsample:" + n] = flat[:: max(1, flat.numel() // 64)][:64].numpy()
        # all-parameter gradient norm (double) as a global checksum
        fx["gradnorm_all"] = np.float64(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in unet.parameters())).item())
        with torch.no_grad():
            pn, xs = gd.model_predictions(x_t, t, clip_x_start=True, rederive_pred_noise=True)
            fx["pred_noise_clip"] = pn.numpy()
            fx["x_start_clip"] = xs.numpy()
            # one ancestral step at t=500 with injected noise (p_sample draws randn_like internally:
            # reproduce through manual seed before the call)
            torch.manual_seed(4242)
            img_next, xs2 = gd.p_sample(x_t, 500)
            torch.manual_seed(4242)
            fx["p_sample_noise"] = torch.randn_like(x_t).numpy()
            fx["p_sample_500"] = img_next.numpy()
            img_next0, _ = gd.p_sample(x_t, 0)
            fx["p_sample_0"] = img_next0.numpy()
            # one DDIM step 999 -> 979 (eta = 0: the noise term is multiplied by sigma = 0)
            tt = torch.full((B,), 999, dtype=torch.long)
            pn, xs = gd.model_predictions(x_t, tt, clip_x_start=True, rederive_pred_noise=True)
            a, an = gd.alphas_cumprod[999], gd.alphas_cumprod[979]
            sigma = 0.0 * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
            c = (1 - an - sigma ** 2).sqrt()
            fx["ddim_999_979"] = (xs * an.sqrt() + c * pn).numpy()
            if tag == "small":
                # whole sampling LOOPS of the reference (ddpm.py:759-780, 782-834).  Both loops draw from the
                # global RNG (randn(shape), then one randn_like per step): a test replays the same draws by
                # seeding the CPU generator identically.  Stored: the returned (unnormalised) images only.
                torch.manual_seed(9001)
                fx["ddim_loop_50"] = gd.ddim_sample((B, 3, S, S)).numpy()
                fx["ddim_loop_seed"] = 9001
                gd_a = R.GaussianDiffusion(unet, img_size=S, timesteps=200)     # ancestral chain, T = 200
                torch.manual_seed(9002)
                fx["p_sample_loop_200"] = gd_a.p_sample_loop((B, 3, S, S)).numpy()
                fx["p_sample_loop_seed"] = 9002
        np.savez_compressed(os.path.join(OUT, f"diffusion_unet_{tag}.npz"), **to_np(fx))
        print("diffusion", tag, "loss", float(loss), "gradnorm", fx["gradnorm_all"])


def gen_vq():
    from models.modules.vector_quantizer import VectorQuantizer, VectorQuantizerEMA
    from models.generative.vae import vqvae as R
    from oracle import vq as O

    fx = {}
    g = torch.Generator().manual_seed(7)
    K, D = 512, 64
    lat = torch.randn(8, D, 4, 4, generator=g) * 0.05
    cb = (torch.rand(K, D, generator=g) * 2 - 1) / K
    vq = VectorQuantizer(K, D, 0.25)
    vq.embedding.weight.data.copy_(cb)
    lat_r = lat.clone().requires_grad_(True)
    q, loss, ppl = vq(lat_r)
    (q.sum() * 0.5 + loss).backward()
    flat = lat.permute(0, 2, 3, 1).reshape(-1, D)
    dist = ((flat ** 2).sum(1, keepdim=True) + (cb ** 2).sum(1) - 2 * flat @ cb.T)
    top2 = dist.topk(2, dim=1, largest=False).values
    fx.update(seed=7, indices=dist.argmin(1).numpy(), margin=(top2[:, 1] - top2[:, 0]).numpy(),
              vq_loss=loss.detach().numpy(), perplexity=ppl.detach().numpy(),
              quantized=q.detach().numpy(), grad_latents=lat_r.grad.numpy(),
              grad_codebook=vq.embedding.weight.grad.numpy())
    # EMA variant, 3 training steps on fresh latents
    vqe = VectorQuantizerEMA(K, D, 0.25, 0.99, 1e-5)
    vqe.embedding.weight.data.copy_(cb)
    vqe._ema_embedding.copy_(cb)
    vqe.train()
    for step in range(3):
        lat_s = torch.randn(8, D, 4, 4, generator=g) * 0.05
        q, loss, ppl = vqe(lat_s)
        fx[f"ema_loss_{step}"] = loss.detach().numpy()
        fx[f"ema_ppl_{step}"] = ppl.detach().numpy()
    fx["ema_cluster_size"] = vqe._ema_cluster_size.numpy()
    fx["ema_embedding_sum"] = np.float64(vqe._ema_embedding.double().sum().item())
    fx["ema_codebook_sample"] = vqe.embedding.weight.detach()[::37].numpy()
    # full VQVAE step (both variants) at vqvae.json sizes, B=4
    for use_ema in (False, True):
        kw = dict(img_channels=3, img_size=32, embedding_dim=64, num_embeddings=512, hidden_dim=128,
                  num_residual_layers=2, num_residual_hiddens=32, commitment_cost=0.25,
                  use_ema=use_ema, decay=0.99, epsilon=1e-5,
                  loss_weights={"recon_loss": 1, "vq_loss": 10 if use_ema else 1})
        m = R.VQVAE(**kw)
        P = O.vqvae_init(seed=11)
        sd = dict(P)
        if use_ema:
            sd["vector_quantizer._ema_cluster_size"] = torch.zeros(512)
            sd["vector_quantizer._ema_embedding"] = P["vector_quantizer.embedding.weight"].clone()
        m.load_state_dict(sd, strict=True)
        m.train()
        gx = torch.Generator().manual_seed(12)
        x = torch.rand(4, 3, 32, 32, generator=gx) * 2 - 1
        loss = m.training_step((x, None), 0)
        loss.backward()
        tag = "ema" if use_ema else "plain"
        fx[f"vqvae_{tag}_loss"] = loss.detach().numpy()
        gn = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in m.parameters() if p.grad is not None))
        fx[f"vqvae_{tag}_gradnorm"] = np.float64(gn.item())
        fx[f"vqvae_{tag}_grad_enc0"] = m.encoder.layers[0].weight.grad.numpy()
        with torch.no_grad():
            m.eval()
            lat = m.encoder(x)
            fx[f"vqvae_{tag}_latents"] = lat.numpy()
        print("vqvae", tag, float(loss), float(gn))
    np.savez_compressed(os.path.join(OUT, "vq.npz"), **to_np(fx))


def gen_gan():
    from models.generative.gan import wgan as R
    from oracle import gan as O

    fx = {}
    for img_size, ch, latent, B in ((64, 3, 100, 4), (28, 1, 128, 4)):
        m = R.WGAN(img_channels=ch, img_size=img_size, latent_dim=latent, lr=1e-4, b1=0.5, b2=0.9,
                   weight_decay=0, n_critic=5, grad_penalty=10, constraint_method="gp", summary=False)
        G, D = O.gan_init(img_size, ch, latent, seed=21)
        gsd = m.G.state_dict()
        gsd.update(G)
        m.G.load_state_dict(gsd, strict=True)
        dsd = m.D.state_dict()
        dsd.update(D)
        m.D.load_state_dict(dsd, strict=True)
        m.train()
        g = torch.Generator().manual_seed(22)
        x = torch.rand(B, ch, img_size, img_size, generator=g) * 2 - 1
        z = torch.randn(B, latent, 1, 1, generator=g)
        alpha = torch.rand(B, 1, 1, 1, generator=g)
        x_hat = m.G(z)
        # _calculate_gradient_penalty draws alpha with torch.rand: reproduce through the global seed
        torch.manual_seed(777)
        alpha_ref = torch.rand(B, 1, 1, 1)
        torch.manual_seed(777)
        ld = m._calculate_d_loss(x, x_hat)
        m.D.zero_grad()
        ld["d_loss"].backward()
        tag = f"{img_size}"
        fx[f"alpha_{tag}"] = alpha_ref.numpy()
        fx[f"x_hat_{tag}"] = x_hat.detach().numpy()
        for k in ("d_loss", "d_loss_real", "d_loss_fake", "gradient_penalty"):
            fx[f"{k}_{tag}"] = ld[k].detach().numpy()
        for n, p in m.D.named_parameters():
            fx[f"dgrad_{tag}:{n}"] = (p.grad.numpy() if p.numel() < 20000
                                      else p.grad.reshape(-1)[:: p.numel() // 256][:256].numpy())
            fx[f"dgradnorm_{tag}:{n}"] = np.float64(p.grad.double().norm().item())
        m.G.zero_grad()
        m.D.zero_grad()
        x_hat2 = m.G(z)
        gl = m._calculate_g_loss(x_hat2)["g_loss"]
        gl.backward()
        fx[f"g_loss_{tag}"] = gl.detach().numpy()
        for n, p in m.G.named_parameters():
            fx[f"ggradnorm_{tag}:{n}"] = np.float64(p.grad.double().norm().item())
        print("wgan", tag, {k: float(v) for k, v in ld.items()}, float(gl))
    np.savez_compressed(os.path.join(OUT, "wgan.npz"), **to_np(fx))


def gen_gan_heads():
    """DCGAN (BCE), LSGAN, R1GAN heads and the weight-clipping WGAN + RMSprop, from the
    reference classes on CPU, same injected weights / inputs as gen_gan."""
    from models.generative.gan import dcgan as RD, lsgan as RL, r1gan as RR, wgan as RW
    from oracle import gan as O

    fx = {}

    def grads(prefix, module):
        for n, p in module.named_parameters():
            fx[f"{prefix}:{n}"] = (p.grad.numpy().copy() if p.numel() < 20000
                                   else p.grad.reshape(-1)[:: p.numel() // 256][:256].numpy().copy())
            fx[f"{prefix}norm:{n}"] = np.float64(p.grad.double().norm().item())

    for img_size, ch, latent, B in ((64, 3, 100, 4), (28, 1, 128, 4)):
        tag = f"{img_size}"
        kw = dict(img_channels=ch, img_size=img_size, latent_dim=latent, lr=2e-4, b1=0.5, b2=0.999, weight_decay=1e-5)
        models = {"dcgan": RD.DCGAN(summary=False, **kw), "lsgan": RL.LSGAN(summary=False, **kw),
                  "r1gan": RR.R1GAN(r1_penalty=10.0, **kw)}
        G, D = O.gan_init(img_size, ch, latent, seed=21)
        g = torch.Generator().manual_seed(22)
        x = torch.rand(B, ch, img_size, img_size, generator=g) * 2 - 1
        z = torch.randn(B, latent, 1, 1, generator=g)
        for name, m in models.items():
            gsd = m.G.state_dict(); gsd.update(G); m.G.load_state_dict(gsd, strict=True)
            dsd = m.D.state_dict(); dsd.update(D); m.D.load_state_dict(dsd, strict=True)
            m.train()
            x_hat = m.G(z)
            ld = m._calculate_d_loss(x.clone(), x_hat)
            m.D.zero_grad()
            ld["d_loss"].backward()
            for k, v in ld.items():
                fx[f"{name}_{k}_{tag}"] = v.detach().numpy()
            grads(f"{name}_dgrad_{tag}", m.D)
            m.G.zero_grad(); m.D.zero_grad()
            gl = m._calculate_g_loss(m.G(z))["g_loss"]
            gl.backward()
            fx[f"{name}_g_loss_{tag}"] = gl.detach().numpy()
            for n, p in m.G.named_parameters():
                fx[f"{name}_ggradnorm_{tag}:{n}"] = np.float64(p.grad.double().norm().item())
            print(name, tag, {k: float(v) for k, v in ld.items()}, float(gl))
        # ---- weight-clipping WGAN: loss, clipped weights, one RMSprop step of the critic ----
        m = RW.WGAN(img_channels=ch, img_size=img_size, latent_dim=latent, lr=5e-5, n_critic=5, clip_value=0.01,
                    constraint_method="clip", summary=False)
        gsd = m.G.state_dict(); gsd.update(G); m.G.load_state_dict(gsd, strict=True)
        dsd = m.D.state_dict(); dsd.update(D); m.D.load_state_dict(dsd, strict=True)
        m.train()
        d_optim, g_optim = m.configure_optimizers()[0]
        x_hat = m.G(z)
        ld = m._calculate_d_loss(x, x_hat)          # clamps the critic's weights as a side effect
        d_optim.zero_grad()
        ld["d_loss"].backward()
        for k, v in ld.items():
            fx[f"wgancp_{k}_{tag}"] = v.detach().numpy()
        grads(f"wgancp_dgrad_{tag}", m.D)
        d_optim.step()
        for n, p in m.D.named_parameters():
            fx[f"wgancp_after_{tag}:{n}"] = (p.detach().numpy().copy() if p.numel() < 20000
                                             else p.detach().reshape(-1)[:: p.numel() // 256][:256].numpy().copy())
        print("wgan_cp", tag, {k: float(v) for k, v in ld.items()})
    np.savez_compressed(os.path.join(OUT, "gan_heads.npz"), **to_np(fx))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["diffusion", "vq", "gan", "gan_heads"]
    if "diffusion" in which:
        gen_diffusion()
    if "vq" in which:
        gen_vq()
    if "gan" in which:
        gen_gan()
    if "gan_heads" in which:
        gen_gan_heads()
