"""CPU: the oracle (oracle/*.py) against fixtures captured from the real reference
(tests/golden/*.npz, produced by oracle/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import diffusion as OD
from oracle import gan as OG
from oracle import optim as OO
from oracle import vq as OV

RTOL = 1e-4  # north_star: fp32 outputs / losses within 1e-4 relative


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_schedule_buffers_bit_exact(golden_dir):
    fx = load(golden_dir, "diffusion_schedule.npz")
    bufs = OD.diffusion_buffers(1000, "sigmoid", "pred_v")
    idx = torch.as_tensor(fx["idx"])
    for name, val in bufs.items():
        assert np.array_equal(val[idx].numpy(), fx[name]), name
        assert val.double().sum().item() == float(fx[name + "__sum"]), name
    # known answers quoted in SURVEY.md §8(a)
    assert abs(bufs["betas"][0].item() - 3.002792e-4) < 1e-9
    assert abs(bufs["alphas_cumprod"][500].item() - 0.49834281) < 1e-7
    assert torch.equal(bufs["loss_weight"], bufs["alphas_cumprod"])


def test_ddim_time_indices_bit_exact(golden_dir):
    fx = load(golden_dir, "diffusion_schedule.npz")
    for S in (50, 250, 1000):
        pairs = OD.ddim_time_pairs(1000, S)
        times = [pairs[0][0]] + [p[1] for p in pairs]
        assert times == fx[f"ddim_times_{S}"].tolist()
        assert bool(fx[f"is_ddim_{S}"]) == (S < 1000)
    assert OD.ddim_time_pairs(1000, 50)[0] == (999, 979)
    assert OD.ddim_time_pairs(1000, 50)[-1] == (19, -1)
    assert OD.ddim_time_pairs(1000, 250)[1] == (995, 991)


def test_posemb_known_answers(golden_dir):
    fx = load(golden_dir, "diffusion_schedule.npz")
    pe = OD.sinusoidal_pos_emb(torch.as_tensor(fx["posemb_t"]), 64)
    assert np.array_equal(pe.numpy(), fx["posemb"])
    assert abs(pe[1, 0].item() - 0.84147096) < 1e-7 and abs(pe[1, 32].item() - 0.54030234) < 1e-7


def _unet_case(golden_dir, tag):
    fx = load(golden_dir, f"diffusion_unet_{tag}.npz")
    dim, S, B = int(fx["dim"]), int(fx["S"]), int(fx["B"])
    P = OD.unet_init(dim=dim, channels=3, seed=int(fx["seed"]))
    g = torch.Generator().manual_seed(int(fx["data_seed"]))
    img = torch.rand(B, 3, S, S, generator=g)
    noise = torch.randn(B, 3, S, S, generator=g)
    t = torch.as_tensor(fx["t"])
    return fx, dim, P, img, noise, t


def test_unet_small_forward_backward(golden_dir):
    fx, dim, P, img, noise, t = _unet_case(golden_dir, "small")
    bufs = OD.diffusion_buffers(1000)
    for p in P.values():
        p.requires_grad_(True)
    x0 = img * 2 - 1
    x_t = OD.q_sample(bufs, x0, t, noise)
    assert rel_err(x_t, fx["x_t"]) < 1e-6
    loss, out = OD.p_losses(P, bufs, x0, t, noise, dim=dim, return_out=True)
    assert rel_err(out, fx["unet_out"]) < RTOL
    assert abs(loss.item() - float(fx["loss"])) / abs(float(fx["loss"])) < RTOL
    loss.backward()
    for k in fx:
        if k.startswith("grad:"):
            assert rel_err(P[k[5:]].grad, fx[k]) < RTOL, k
    gn = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in P.values())).item()
    assert abs(gn - float(fx["gradnorm_all"])) / float(fx["gradnorm_all"]) < RTOL


@pytest.mark.parametrize("tag", ["full", "full64"])
def test_unet_full_forward_backward(golden_dir, tag):
    fx, dim, P, img, noise, t = _unet_case(golden_dir, tag)
    bufs = OD.diffusion_buffers(1000)
    for p in P.values():
        p.requires_grad_(True)
    loss, out = OD.diffusion_forward(P, bufs, img, t, noise, dim=dim, return_out=True)
    assert rel_err(out, fx["unet_out"]) < RTOL
    assert abs(loss.item() - float(fx["loss"])) / abs(float(fx["loss"])) < RTOL
    loss.backward()
    for k in fx:
        if k.startswith("gradnorm:"):
            n = k[len("gradnorm:"):]
            gnorm = P[n].grad.double().norm().item()
            assert abs(gnorm - float(fx[k])) / max(float(fx[k]), 1e-12) < RTOL, k
            flat = P[n].grad.reshape(-1)
            samp = flat[:: max(1, flat.numel() // 64)][:64]
            assert rel_err(samp, fx["gradsample:" + n]) < 5 * RTOL, k


def test_sampling_steps(golden_dir):
    fx, dim, P, img, noise, t = _unet_case(golden_dir, "small")
    bufs = OD.diffusion_buffers(1000)
    x_t = torch.as_tensor(fx["x_t"])
    with torch.no_grad():
        pn, xs, _ = OD.model_predictions(P, bufs, x_t, t, clip_x_start=True, dim=dim)
        assert rel_err(pn, fx["pred_noise_clip"]) < RTOL
        assert rel_err(xs, fx["x_start_clip"]) < RTOL
        nz = torch.as_tensor(fx["p_sample_noise"])
        img500, _ = OD.p_sample(P, bufs, x_t, 500, nz, dim=dim)
        assert rel_err(img500, fx["p_sample_500"]) < RTOL
        img0, _ = OD.p_sample(P, bufs, x_t, 0, nz, dim=dim)
        assert rel_err(img0, fx["p_sample_0"]) < RTOL
        nxt, _ = OD.ddim_step(P, bufs, x_t, 999, 979, torch.zeros_like(x_t), dim=dim)
        assert rel_err(nxt, fx["ddim_999_979"]) < RTOL


def test_sampling_loops_match_reference(golden_dir):
    """Whole loops of the reference (50-pair DDIM chain, 200-step ancestral chain) replayed by the oracle from
    the same CPU-generator draws."""
    fx, dim, P, img, noise, t = _unet_case(golden_dir, "small")
    shape = tuple(fx["x_t"].shape)
    with torch.no_grad():
        init, nz = OD.draw_loop_noise(int(fx["ddim_loop_seed"]), shape, 49)
        out = OD.ddim_sample_loop(P, OD.diffusion_buffers(1000), init, nz, 50, dim=dim)
        e1 = rel_err(out, fx["ddim_loop_50"])
        init, nz = OD.draw_loop_noise(int(fx["p_sample_loop_seed"]), shape, 199)
        out = OD.p_sample_loop(P, OD.diffusion_buffers(200), init, nz, dim=dim)
        e2 = rel_err(out, fx["p_sample_loop_200"])
    print(f"oracle loops vs reference: ddim50 {e1:.2e} ancestral200 {e2:.2e}")
    assert e1 < RTOL and e2 < RTOL


def test_vq_quantizer(golden_dir):
    fx = load(golden_dir, "vq.npz")
    g = torch.Generator().manual_seed(int(fx["seed"]))
    K, D = 512, 64
    lat = (torch.randn(8, D, 4, 4, generator=g) * 0.05).requires_grad_(True)
    cb = ((torch.rand(K, D, generator=g) * 2 - 1) / K).requires_grad_(True)
    q, loss, ppl, idx, _ = OV.vector_quantizer(lat, cb, 0.25)
    assert np.array_equal(idx.numpy(), fx["indices"])          # bit-exact indices
    assert rel_err(loss, fx["vq_loss"]) < RTOL and rel_err(ppl, fx["perplexity"]) < RTOL
    assert rel_err(q, fx["quantized"]) < 1e-6
    (q.sum() * 0.5 + loss).backward()
    assert rel_err(lat.grad, fx["grad_latents"]) < RTOL
    assert rel_err(cb.grad, fx["grad_codebook"]) < RTOL
    # EMA variant: three training steps
    state = (torch.zeros(K), cb.detach().clone())
    codebook = cb.detach().clone()
    for step in range(3):
        lat_s = torch.randn(8, D, 4, 4, generator=g) * 0.05
        q, loss, ppl, idx, state3 = OV.vector_quantizer(lat_s, codebook, 0.25, ema_state=state)
        state, codebook = (state3[0], state3[1]), state3[2]
        assert rel_err(loss, fx[f"ema_loss_{step}"]) < RTOL
        assert rel_err(ppl, fx[f"ema_ppl_{step}"]) < RTOL
    assert rel_err(state[0], fx["ema_cluster_size"]) < 1e-6
    assert abs(state[1].double().sum().item() - float(fx["ema_embedding_sum"])) < 1e-4 * abs(float(fx["ema_embedding_sum"])) + 1e-9
    assert rel_err(codebook[::37], fx["ema_codebook_sample"]) < RTOL


def test_vqvae_step(golden_dir):
    fx = load(golden_dir, "vq.npz")
    for tag, w_vq in (("plain", 1.0), ("ema", 10.0)):
        P = OV.vqvae_init(seed=11)
        for p in P.values():
            p.requires_grad_(True)
        gx = torch.Generator().manual_seed(12)
        x = torch.rand(4, 3, 32, 32, generator=gx) * 2 - 1
        ema_state = None
        if tag == "ema":
            ema_state = (torch.zeros(512), P["vector_quantizer.embedding.weight"].detach().clone())
        r = OV.vqvae_step(P, x, w_recon=1.0, w_vq=w_vq, ema_state=ema_state)
        assert rel_err(r["latents"], fx[f"vqvae_{tag}_latents"]) < RTOL
        assert rel_err(r["loss"], fx[f"vqvae_{tag}_loss"]) < RTOL
        r["loss"].backward()
        gn = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in P.values() if p.grad is not None)).item()
        assert abs(gn - float(fx[f"vqvae_{tag}_gradnorm"])) / float(fx[f"vqvae_{tag}_gradnorm"]) < RTOL
        assert rel_err(P["encoder.layers.0.weight"].grad, fx[f"vqvae_{tag}_grad_enc0"]) < RTOL


def test_wgan_gp_losses(golden_dir):
    fx = load(golden_dir, "wgan.npz")
    for img_size, ch, latent, B in ((64, 3, 100, 4), (28, 1, 128, 4)):
        tag = str(img_size)
        G, D = OG.gan_init(img_size, ch, latent, seed=21)
        for p in list(G.values()) + list(D.values()):
            p.requires_grad_(True)
        g = torch.Generator().manual_seed(22)
        x = torch.rand(B, ch, img_size, img_size, generator=g) * 2 - 1
        z = torch.randn(B, latent, 1, 1, generator=g)
        x_hat = OG.generator(G, z, img_size, ch)
        assert rel_err(x_hat, fx[f"x_hat_{tag}"]) < RTOL
        alpha = torch.as_tensor(fx[f"alpha_{tag}"])
        ld = OG.wgan_d_loss(D, x, x_hat.detach(), alpha, 10.0, img_size)
        for k in ("d_loss", "d_loss_real", "d_loss_fake", "gradient_penalty"):
            assert rel_err(ld[k], fx[f"{k}_{tag}"]) < RTOL, k
        ld["d_loss"].backward()
        for n, p in D.items():
            gnorm = p.grad.double().norm().item()
            ref = float(fx[f"dgradnorm_{tag}:{n}"])
            assert abs(gnorm - ref) / max(ref, 1e-12) < 5 * RTOL, n
        gl = OG.wgan_g_loss(D, OG.generator(G, z, img_size, ch), img_size)
        assert rel_err(gl, fx[f"g_loss_{tag}"]) < RTOL
        for p in G.values():
            p.grad = None
        gl.backward()
        for n, p in G.items():
            ref = float(fx[f"ggradnorm_{tag}:{n}"])
            assert abs(p.grad.double().norm().item() - ref) / max(ref, 1e-12) < 5 * RTOL, n


def test_gan_heads_losses(golden_dir):
    """DCGAN (BCE), LSGAN, R1GAN, weight-clipping WGAN + RMSprop against the reference fixture."""
    from oracle import optim as OO
    fx = load(golden_dir, "gan_heads.npz")
    heads = {"dcgan": (OG.dcgan_d_loss, OG.dcgan_g_loss), "lsgan": (OG.lsgan_d_loss, OG.lsgan_g_loss),
             "r1gan": (lambda D, x, xh, s: OG.r1gan_d_loss(D, x, xh, 10.0, s), OG.dcgan_g_loss)}
    for img_size, ch, latent, B in ((64, 3, 100, 4), (28, 1, 128, 4)):
        tag = str(img_size)
        g = torch.Generator().manual_seed(22)
        x = torch.rand(B, ch, img_size, img_size, generator=g) * 2 - 1
        z = torch.randn(B, latent, 1, 1, generator=g)
        for name, (dl, gl_fn) in heads.items():
            G, D = OG.gan_init(img_size, ch, latent, seed=21)
            for p in list(G.values()) + list(D.values()):
                p.requires_grad_(True)
            x_hat = OG.generator(G, z, img_size, ch)
            ld = dl(D, x, x_hat.detach(), img_size)
            for k, v in ld.items():
                assert rel_err(v, fx[f"{name}_{k}_{tag}"]) < RTOL, (name, k)
            ld["d_loss"].backward()
            for n, p in D.items():
                ref = float(fx[f"{name}_dgrad_{tag}norm:{n}"])
                assert abs(p.grad.double().norm().item() - ref) / max(ref, 1e-12) < 5 * RTOL, (name, n)
            gl = gl_fn(D, OG.generator(G, z, img_size, ch), img_size)
            assert rel_err(gl, fx[f"{name}_g_loss_{tag}"]) < RTOL, name
            for p in G.values():
                p.grad = None
            gl.backward()
            for n, p in G.items():
                ref = float(fx[f"{name}_ggradnorm_{tag}:{n}"])
                assert abs(p.grad.double().norm().item() - ref) / max(ref, 1e-12) < 5 * RTOL, (name, n)
        # weight clipping + one RMSprop step of the critic (wgan.py:101-102,158-181)
        G, D = OG.gan_init(img_size, ch, latent, seed=21)
        x_hat = OG.generator(G, z, img_size, ch).detach()
        ld = OG.wgan_clip_d_loss(D, x, x_hat, img_size)          # loss on the UNclipped weights
        for k, v in ld.items():
            assert rel_err(v, fx[f"wgancp_{k}_{tag}"]) < RTOL, k
        Dc = {k: v.clone().requires_grad_(True) for k, v in OG.weight_clip(D, 0.01).items()}
        # the reference clamps .data in place after the forward: the backward sees the clipped weights
        # in every weight-dependent node but the activations of the unclipped forward.  Reproduce with
        # autograd on the unclipped graph is impossible; the fixture's gradients are checked on the
        # GPU path (which has the same "late weights" semantics); here only RMSprop is pinned.
        for n in D:
            got_key = f"wgancp_after_{tag}:{n}"
            gkey = f"wgancp_dgrad_{tag}:{n}"
            if D[n].numel() < 20000:
                gref = torch.as_tensor(fx[gkey]).reshape(D[n].shape)
                p1, _ = OO.rmsprop_step(Dc[n].detach(), gref, torch.zeros_like(gref), lr=5e-5)
                assert rel_err(p1, fx[got_key]) < 1e-6, n


def test_adam_matches_torch():
    torch.manual_seed(0)
    for wd in (0.0, 1e-5):
        p = torch.randn(257)
        ref = p.clone().requires_grad_(True)
        opt = torch.optim.Adam([ref], lr=2e-5, betas=(0.9, 0.99), weight_decay=wd)
        m = torch.zeros_like(p)
        v = torch.zeros_like(p)
        for step in range(1, 6):
            g = torch.randn(257)
            ref.grad = g.clone()
            opt.step()
            p, m, v = OO.adam_step(p, g, m, v, step, 2e-5, 0.9, 0.99, 1e-8, wd)
            assert torch.allclose(p, ref.detach(), rtol=1e-6, atol=1e-9)


def test_ema_schedule_closed_form():
    st = OO.EmaState(beta=0.995, update_every=10, update_after_step=100)
    acts = [st.next_action() for _ in range(200)]
    assert acts[0][0] == "copy" and acts[1][0] == "skip" and acts[100][0] == "copy"
    assert acts[110][0] == "copy"          # first post-warm-up call initialises the shadow
    kind, w = acts[120]
    assert kind == "lerp"
    expect = 1 - min(1 - (1 + (121 - 100 - 1)) ** (-2 / 3), 0.995)
    assert abs(w - expect) < 1e-12
