"""GPU: the HIP UNet / diffusion training step against (a) golden fixtures captured from the real
reference and (b) the CPU oracle on fresh seeded inputs.  fp32, 1e-4 relative (north_star)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    if a.shape != b.shape and a.numel() == b.numel():
        a = a.reshape(b.shape)      # Downsample's weight: held as [N, C, 2, 2], row-major = the reference's [N, 4 C, 1, 1]
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def build(dim, S, P, dev, sampling_timesteps=None):
    from models.generative.diffusion.ddpm import GaussianDiffusion, Unet
    net = Unet(dim=dim, channels=3)
    net.load_state_dict(P, strict=True)
    gd = GaussianDiffusion(net, img_size=S, timesteps=1000, sampling_timesteps=sampling_timesteps).to(dev)
    net.prepare_hip(dev)
    return net, gd


def case(golden_dir, tag):
    from oracle import diffusion as OD
    fx = dict(np.load(os.path.join(golden_dir, f"diffusion_unet_{tag}.npz")))
    dim, S, B = int(fx["dim"]), int(fx["S"]), int(fx["B"])
    P = OD.unet_init(dim=dim, channels=3, seed=int(fx["seed"]))
    g = torch.Generator().manual_seed(int(fx["data_seed"]))
    img = torch.rand(B, 3, S, S, generator=g)
    noise = torch.randn(B, 3, S, S, generator=g)
    return fx, dim, S, P, img, noise, torch.as_tensor(fx["t"])


@pytest.mark.parametrize("C,N,S,B", [(64, 64, 32, 4), (64, 128, 16, 3), (128, 256, 8, 8), (16, 16, 16, 2)])
def test_downsample_is_one_strided_convolution(dev, C, N, S, B, parity):
    """Downsample (reference ddpm.py:100-104, Rearrange 'b c (h p1) (w p2) -> b (c p1 p2) h w' + Conv2d(4C, N, 1))
    runs as ONE 2x2 / stride-2 convolution of the un-shuffled tensor.  Forward, input gradient (overwrite and accumulate),
    weight and bias gradient against the reference's two operations in float64, with the weight in the reference's shape."""
    import torch.nn.functional as F
    from lgm_hip import ops
    from lgm_hip.flat import FlatParams
    from lgm_hip.nn import GradCtx, param_kind
    from models.generative.diffusion.ddpm import _Down
    torch.manual_seed(C + N + S)
    d = _Down(C, N)
    w_ref = d.state_dict()["1.weight"].double().requires_grad_(True)          # [N, 4C, 1, 1]
    b_ref = d.state_dict()["1.bias"].double().requires_grad_(True)
    fp = FlatParams([(n, p, param_kind(n, p)) for n, p in d.named_parameters()], dev)
    x = torch.randn(B, C, S, S)
    gy = torch.randn(B, N, S // 2, S // 2)
    xr = x.double().requires_grad_(True)
    lo = xr.view(B, C, S // 2, 2, S // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(B, 4 * C, S // 2, S // 2)
    want = F.conv2d(lo, w_ref, b_ref)
    want.backward(gy.double())
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.empty(B, S // 2, S // 2, N, device=dev)
    saved = d.fwd(xh, out, True)
    parity("forward", rel(out.permute(0, 3, 1, 2), want), RTOL)
    gc = GradCtx(fp)
    gyh = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    gx = torch.full((B, S, S, C), 7.0, device=dev)
    d.bwd(gc, saved, gyh, gx, False)
    gc.flush()
    parity("input gradient", rel(gx.permute(0, 3, 1, 2), xr.grad), RTOL)
    fp.bind_grad_views()
    conv = d._modules["1"]
    parity("weight gradient", rel(conv.weight.grad, w_ref.grad), RTOL)
    parity("bias gradient", rel(conv.bias.grad, b_ref.grad), RTOL)
    base = torch.randn(B, S, S, C, device=dev)
    gx2 = base.clone()
    gc2 = GradCtx(fp)
    d.bwd(gc2, saved, gyh, gx2, True)
    gc2.flush()
    parity("accumulated input gradient", rel((gx2 - base).permute(0, 3, 1, 2), xr.grad), 10 * RTOL)
    parity("accumulated weight gradient (beta = 1)", rel(conv.weight.grad, 2 * w_ref.grad), RTOL)
    assert ops.lib() is not None


@pytest.mark.parametrize("tag", ["small", "full", "full64"])
def test_unet_matches_reference_fixture(dev, golden_dir, tag, parity):
    """small: dim 16 @16^2; full: dim 64 @32^2 (BASELINE config 2's network); full64: dim 64 @64^2 (config 5's).
    Weights incl. RANDOM GroupNorm / RMSNorm affine parameters (oracle.unet_init)."""
    fx, dim, S, P, img, noise, t = case(golden_dir, tag)
    net, gd = build(dim, S, P, dev)
    x_t = torch.as_tensor(fx["x_t"]).to(dev)
    with torch.no_grad():
        out = net(x_t, t.to(dev))
    parity("unet_out", rel(out, fx["unet_out"]), RTOL)
    loss = gd.p_losses((img * 2 - 1).to(dev), t.to(dev), noise.to(dev))
    parity("loss", abs(loss.item() - float(fx["loss"])) / float(fx["loss"]), RTOL)
    loss.backward()
    sd = dict(net.named_parameters())
    worst, worst_n, worst_s = 0.0, 0.0, 0.0
    for k in fx:
        if k.startswith("grad:"):
            worst = max(worst, rel(sd[k[5:]].grad, fx[k]))
        elif k.startswith("gradnorm:"):
            n = k[len("gradnorm:"):]
            gn = sd[n].grad.double().norm().item()
            worst_n = max(worst_n, abs(gn - float(fx[k])) / max(float(fx[k]), 1e-12))
            flat = sd[n].grad.reshape(-1)
            samp = flat[:: max(1, flat.numel() // 64)][:64]
            worst_s = max(worst_s, rel(samp, fx["gradsample:" + n]))
    if tag == "small":
        parity("worst parameter gradient (26 tensors)", worst, RTOL)
    else:
        parity("worst parameter gradient norm (26 tensors)", worst_n, RTOL)
        parity("worst 64-element gradient sample", worst_s, RTOL)
    gn = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in net.parameters())).item()
    parity("all-parameter gradient norm", abs(gn - float(fx["gradnorm_all"])) / float(fx["gradnorm_all"]), RTOL)


def test_product_schedule_buffers_and_posemb_bit_exact(dev, golden_dir):
    """The PRODUCT's 13 schedule buffers (not the oracle's) against the reference fixture, bit for bit, on the
    host and after .to(device); lgm_posemb against the reference's SinusoidalPosEmb rows."""
    from lgm_hip import ops
    from models.generative.diffusion.ddpm import GaussianDiffusion, Unet
    fx = dict(np.load(os.path.join(golden_dir, "diffusion_schedule.npz")))
    gd = GaussianDiffusion(Unet(dim=16, channels=3), img_size=16, timesteps=1000)
    idx = torch.as_tensor(fx["idx"])
    names = [k for k in fx if k + "__sum" in fx]
    assert len(names) == 13
    for where in ("cpu", dev):
        gd.to(where)
        for name in names:
            b = getattr(gd, name).cpu()
            assert np.array_equal(b[idx].numpy(), fx[name]), (name, where)
            assert b.double().sum().item() == float(fx[name + "__sum"]), (name, where)
    tt = torch.as_tensor(fx["posemb_t"]).to(dev)
    pe = torch.empty(tt.shape[0], 64, device=dev)
    ops.posemb(tt, 64, 10000.0, pe)
    err = float((pe.cpu() - torch.as_tensor(fx["posemb"])).abs().max())
    print(f"[parity] posemb vs reference rows: max abs err {err:.2e}")
    assert err < 1e-6


def test_unet_matches_oracle_fresh_inputs(dev, parity):
    """dim 32, 32x32, B=3: every parameter gradient against the oracle's autograd."""
    from oracle import diffusion as OD
    dim, S, B = 32, 32, 3
    P = OD.unet_init(dim=dim, channels=3, seed=5)
    bufs = OD.diffusion_buffers(1000)
    g = torch.Generator().manual_seed(77)
    img = torch.rand(B, 3, S, S, generator=g)
    noise = torch.randn(B, 3, S, S, generator=g)
    t = torch.tensor([3, 450, 999])
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss_ref, out_ref = OD.diffusion_forward(Pr, bufs, img, t, noise, dim=dim, return_out=True)
    loss_ref.backward()
    net, gd = build(dim, S, P, dev)
    loss = gd.p_losses(img.to(dev), t.to(dev), noise.to(dev), _normalize=True)
    parity("loss", abs(loss.item() - loss_ref.item()) / loss_ref.item(), RTOL)
    loss.backward()
    errs = {n: rel(p.grad, Pr[n].grad) for n, p in net.named_parameters()}
    wn = max(errs, key=errs.get)
    parity(f"worst of ALL {len(errs)} parameter gradients ({wn})", errs[wn], RTOL)
    # second backward without zero_grad accumulates (beta = 1 path)
    loss2 = gd.p_losses(img.to(dev), t.to(dev), noise.to(dev), _normalize=True)
    loss2.backward()
    n0, p0 = next(iter(net.named_parameters()))
    assert rel(p0.grad, 2 * Pr[n0].grad) < 2 * RTOL


@pytest.mark.parametrize("B", [1, 5])
def test_full_width_unet_ragged_batches_match_oracle(dev, B, parity):
    """dim 64 (the benchmark network), batches that do not fill the 3x3 kernels' image groups (8 images
    per tile at 4x4, 2 at 8x8): those layers must take the generic path and still match the oracle."""
    from oracle import diffusion as OD
    dim, S = 64, 32
    P = OD.unet_init(dim=dim, channels=3, seed=11)
    bufs = OD.diffusion_buffers(1000)
    g = torch.Generator().manual_seed(100 + B)
    img = torch.rand(B, 3, S, S, generator=g)
    noise = torch.randn(B, 3, S, S, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    loss_ref = OD.diffusion_forward(Pr, bufs, img, t, noise, dim=dim)
    loss_ref.backward()
    net, gd = build(dim, S, P, dev)
    loss = gd.p_losses(img.to(dev), t.to(dev), noise.to(dev), _normalize=True)
    parity("loss", abs(loss.item() - loss_ref.item()) / loss_ref.item(), RTOL)
    loss.backward()
    errs = {n: rel(p.grad, Pr[n].grad) for n, p in net.named_parameters()}
    wn = max(errs, key=errs.get)
    parity(f"worst of ALL {len(errs)} parameter gradients ({wn})", errs[wn], RTOL)


def test_sampling_steps_match_reference_fixture(dev, golden_dir, parity):
    from lgm_hip import sampler
    fx, dim, S, P, img, noise, t = case(golden_dir, "small")
    net, gd = build(dim, S, P, dev, sampling_timesteps=50)
    x_t = torch.as_tensor(fx["x_t"])
    nz = torch.as_tensor(fx["p_sample_noise"]).to(dev)
    ch = sampler._Chain(gd, tuple(x_t.shape), x_t.to(dev))
    sampler.p_sample_step(ch, 500, nz)
    parity("p_sample t=500", rel(ch.image(False), fx["p_sample_500"]), RTOL)
    ch = sampler._Chain(gd, tuple(x_t.shape), x_t.to(dev))
    sampler.p_sample_step(ch, 0, nz)
    parity("p_sample t=0", rel(ch.image(False), fx["p_sample_0"]), RTOL)
    ch = sampler._Chain(gd, tuple(x_t.shape), x_t.to(dev))
    sampler.ddim_step(ch, 999, 979, None, 0.0)
    parity("ddim 999->979", rel(ch.image(False), fx["ddim_999_979"]), RTOL)
    # DDIM index lists are bit-exact with the reference's (tests/golden/diffusion_schedule.npz)
    sch = dict(np.load(os.path.join(golden_dir, "diffusion_schedule.npz")))
    pairs = gd.ddim_time_pairs()
    assert [pairs[0][0]] + [p[1] for p in pairs] == sch["ddim_times_50"].tolist()


def test_whole_sampling_loops_match_reference_fixture(dev, golden_dir, parity):
    """The complete loops (reference ddpm.py:759-780 and :782-834) against images the REFERENCE returned:
    the 50-pair DDIM chain (eta = 0) and a 200-step ancestral chain, replaying the reference's CPU-generator
    draws (start image + one noise tensor per step)."""
    from lgm_hip import sampler
    from oracle import diffusion as OD
    fx, dim, S, P, img, noise, t = case(golden_dir, "small")
    shape = tuple(fx["x_t"].shape)
    net, gd = build(dim, S, P, dev, sampling_timesteps=50)
    init, nz = OD.draw_loop_noise(int(fx["ddim_loop_seed"]), shape, 49)
    out = sampler.ddim_sample(gd, shape, init_noise=init.to(dev), noises=[n.to(dev) for n in nz] + [None])
    parity("50-pair DDIM loop, final image", rel(out, fx["ddim_loop_50"]), RTOL)
    from models.generative.diffusion.ddpm import GaussianDiffusion
    gd_a = GaussianDiffusion(net, img_size=S, timesteps=200).to(dev)
    init, nz = OD.draw_loop_noise(int(fx["p_sample_loop_seed"]), shape, 199)
    out = sampler.p_sample_loop(gd_a, shape, init_noise=init.to(dev), noises=[n.to(dev) for n in nz] + [None])
    parity("200-step ancestral loop, final image", rel(out, fx["p_sample_loop_200"]), RTOL)


def test_graph_replayed_sampling_is_bit_identical_to_eager_launches(dev, golden_dir, monkeypatch):
    """The captured sampling step (device-side step counter + scalar table, in-place update) against the eager
    per-step launches: same DDIM chain (eta = 0) and same ancestral chain with injected noise, torch.equal."""
    from lgm_hip import sampler
    from models.generative.diffusion.ddpm import GaussianDiffusion
    from oracle import diffusion as OD
    fx, dim, S, P, img, noise, t = case(golden_dir, "small")
    shape = tuple(fx["x_t"].shape)
    net, gd = build(dim, S, P, dev, sampling_timesteps=50)
    gd_a = GaussianDiffusion(net, img_size=S, timesteps=60).to(dev)
    init, nz = OD.draw_loop_noise(7, shape, 59)
    nzd = [n.to(dev) for n in nz] + [None]
    outs = {}
    for mode in ("graph", "eager"):
        monkeypatch.setenv("LGM_NO_SAMPLER_GRAPH", "0" if mode == "graph" else "1")
        outs[mode] = (sampler.ddim_sample(gd, shape, init_noise=init.to(dev)),
                      sampler.p_sample_loop(gd_a, shape, init_noise=init.to(dev), noises=nzd))
    ents = [e for per in sampler._GRAPHS.values() for e in per.values()]
    assert ents and all(isinstance(e, sampler._GraphedChain) for e in ents), "graph capture did not happen"
    assert torch.equal(outs["graph"][0], outs["eager"][0])
    assert torch.equal(outs["graph"][1], outs["eager"][1])
    # the cache follows the network's flat storage: after the parameter storage was replaced (module.to(), a loaded
    # model) prepare_hip() rebuilds the flat buffers, the entry captured against the old ones is dropped and the step
    # recaptured -- sampling then uses the NEW weights (a stale graph would keep reading the old buffers)
    monkeypatch.setenv("LGM_NO_SAMPLER_GRAPH", "0")
    old = sampler._GRAPHS[net][(shape, False)]
    with torch.no_grad():
        for p_ in net.parameters():
            p_.data = (p_.data * 1.01).clone()          # new storage, different values
    assert not net._flat.still_bound()
    new_img = sampler.ddim_sample(gd, shape, init_noise=init.to(dev))
    assert sampler._GRAPHS[net][(shape, False)] is not old and sampler._GRAPHS[net][(shape, False)].matches(net)
    monkeypatch.setenv("LGM_NO_SAMPLER_GRAPH", "1")
    assert torch.equal(new_img, sampler.ddim_sample(gd, shape, init_noise=init.to(dev)))
    assert not torch.equal(new_img, outs["graph"][0])
    # device-drawn noise: runs, finite, and two chains differ (fresh draws per replay)
    monkeypatch.setenv("LGM_NO_SAMPLER_GRAPH", "0")
    a = sampler.p_sample_loop(gd_a, shape, init_noise=init.to(dev))
    b = sampler.p_sample_loop(gd_a, shape, init_noise=init.to(dev))
    assert torch.isfinite(a).all() and not torch.equal(a, b)


def test_config5_ancestral_chain_64x64_graph_equals_eager(dev, monkeypatch):
    """BASELINE config 5's sampling half on its own network size: the dim-64 UNet at 64x64 (ddpm_64.json), the
    ancestral chain ``sample()`` dispatches to when sampling_timesteps == timesteps (reference ddpm.py:759-780, 836-845).
    24 steps with injected noise: graph replay is torch.equal to eager launches, the result is finite, and a second
    replayed chain reproduces it bit for bit.  (The 1000-step chain itself is bench.py's ``ddpm64_sampling_1000`` leg.)"""
    from lgm_hip import sampler
    from models.generative.diffusion.ddpm import DDPM
    from oracle import diffusion as OD
    torch.manual_seed(3)
    m = DDPM(img_channels=3, img_size=64, dim=64, diffusion_timesteps=24).to(dev)
    m.sample_every = 0
    m.prepare_hip(dev)
    gd = m.ema.ema_model
    gd.eval()
    assert not gd.is_ddim_sampling and gd.num_timesteps == 24
    shape = (4, 3, 64, 64)
    init, nz = OD.draw_loop_noise(11, shape, 23)
    nzd = [n.to(dev) for n in nz] + [None]
    outs = {}
    for mode in ("graph", "eager", "graph2"):
        monkeypatch.setenv("LGM_NO_SAMPLER_GRAPH", "1" if mode == "eager" else "0")
        outs[mode] = sampler.p_sample_loop(gd, shape, init_noise=init.to(dev), noises=nzd).clone()
    ents = [e for e in sampler._GRAPHS[gd.model].values()]
    assert ents and all(isinstance(e, sampler._GraphedChain) for e in ents), "graph capture did not happen"
    assert torch.isfinite(outs["graph"]).all() and float(outs["graph"].std()) > 0
    assert torch.equal(outs["graph"], outs["eager"])
    assert torch.equal(outs["graph"], outs["graph2"])
    # the public entry takes the same route
    monkeypatch.setenv("LGM_NO_SAMPLER_GRAPH", "0")
    img = gd.sample(batch_size=4)
    assert img.shape == shape and torch.isfinite(img).all()


def test_ddpm_module_training_steps(dev, parity):
    """LightningModule surface: training_step -> backward -> FusedAdam.step -> EMA, 3 steps,
    against the oracle + torch.optim.Adam on CPU."""
    from models.generative.diffusion.ddpm import DDPM
    from oracle import diffusion as OD
    torch.manual_seed(0)
    m = DDPM(img_channels=3, img_size=16, dim=16, lr=1e-3, betas=(0.9, 0.99))
    m.sample_every = 0
    P0 = {k: v.detach().clone() for k, v in m.ema.online_model.model.state_dict().items()}
    m.to(dev)
    m.prepare_hip(dev)
    m.train()
    opt = m.configure_optimizers()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    ref_opt = torch.optim.Adam(list(Pr.values()), lr=1e-3, betas=(0.9, 0.99))
    bufs = OD.diffusion_buffers(1000)
    gd = m.ema.online_model
    g = torch.Generator().manual_seed(3)
    for step in range(3):
        img = torch.rand(4, 3, 16, 16, generator=g)
        noise = torch.randn(4, 3, 16, 16, generator=g)
        t = torch.randint(0, 1000, (4,), generator=g)
        loss = gd.p_losses(img.to(dev), t.to(dev), noise.to(dev), _normalize=True)
        opt.zero_grad()
        loss.backward()
        opt.step()
        m.on_train_batch_end(None, None, step)
        if step == 0:   # ema.update() #0 hard-copies online -> shadow; #1, #2 are skipped (update_every = 10)
            snap = m.ema.online_model.model.state_dict()["init_conv.weight"].detach().clone()
        ref_opt.zero_grad()
        lr = OD.diffusion_forward(Pr, bufs, img, t, noise, dim=16)
        lr.backward()
        ref_opt.step()
        parity(f"loss at step {step}", abs(loss.item() - lr.item()) / lr.item(), RTOL)
    sd = m.ema.online_model.model.state_dict()
    for k in ("init_conv.weight", "downs.0.0.block1.proj.weight", "mid_attn.mem_kv", "final_conv.bias",
              "ups.1.2.to_out.1.g", "time_mlp.3.weight", "downs.2.1.mlp.1.bias"):
        parity(f"parameter after 3 Adam steps: {k}", rel(sd[k], Pr[k]), RTOL)
    # EMA shadow == the online weights at the last executed update (hard copy during warm-up)
    assert rel(m.ema.ema_model.model.state_dict()["init_conv.weight"], snap) < 1e-6
    # the stock training_step path (random t / noise on device) runs and returns a finite scalar
    out = m.training_step((torch.rand(4, 3, 16, 16, device=dev) * 2 - 1, torch.zeros(4, dtype=torch.long, device=dev)))
    assert out.dim() == 0 and torch.isfinite(out)


def test_checkpoint_resume_and_torch_optimizer_interchange(dev, tmp_path):
    """a Lightning-layout checkpoint (reference key names, torch-format Adam state) written
    after 3 steps resumes to the same parameters as 5 uninterrupted steps, and its optimizer state
    loads into a real torch.optim.Adam with identical next-step results."""
    from lgm_hip.lightning import MiniTrainer, save_checkpoint
    from models.generative.diffusion.ddpm import DDPM

    def make():
        torch.manual_seed(3)
        m = DDPM(img_channels=3, img_size=16, dim=8, diffusion_timesteps=50, lr=1e-3, ema_update_every=2)
        m.sample_every = 0
        return m

    g = torch.Generator().manual_seed(5)
    batches = [(torch.rand(4, 3, 16, 16, generator=g) * 2 - 1, torch.zeros(4, dtype=torch.long)) for _ in range(5)]

    def feed(bs, reseed_at):
        for i, b in enumerate(bs):
            if i == reseed_at:
                torch.manual_seed(77)      # (t, noise) come from the device RNG: same draws in both runs
            yield b

    def fit(m, data, ckpt=None):
        MiniTrainer(max_epochs=1, default_root_dir=None, log_every=0, device=dev).fit(m, train_dataloader=data,
                                                                                     ckpt_path=ckpt)
        return m

    torch.manual_seed(11)
    a = fit(make(), feed(batches, 3))                       # five uninterrupted steps
    torch.manual_seed(11)
    b1 = fit(make(), feed(batches[:3], -1))                  # three steps, checkpoint ...
    path = str(tmp_path / "step3.ckpt")
    save_checkpoint(b1, list(b1._optimizers), path)
    ck = torch.load(path, weights_only=False)
    assert ck["global_step"] == 3 and "ema.online_model.model.init_conv.weight" in ck["state_dict"]
    assert "ema.ema_model.model.init_conv.weight" in ck["state_dict"] and "ema.step" in ck["state_dict"]
    osd = ck["optimizer_states"][0]
    n_params = len(list(b1.ema.online_model.parameters()))
    assert len(osd["state"]) == n_params and set(osd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert osd["state"][0]["exp_avg"].shape == b1.ema.online_model.model.init_conv.weight.shape
    b2 = fit(make(), feed(batches[3:], 0), ckpt=path)        # ... fresh process state, two more steps
    assert b2.global_step == 5 == a.global_step
    for (n, p), (_, q) in zip(a.state_dict().items(), b2.state_dict().items()):
        assert torch.allclose(p.float(), q.float(), rtol=1e-6, atol=1e-7), n
    # torch.optim.Adam accepts the exported state and reproduces the fused step
    c = make()
    c.load_state_dict(ck["state_dict"])
    # (the torch optimizer holds the parameters in the REFERENCE's shapes - what state_dict() hands out: Downsample's
    # weight is [N, 4 C, 1, 1] there and [N, C, 2, 2], same row-major order, in this package)
    csd = c.ema.online_model.state_dict()
    params = [csd[n].detach().clone().requires_grad_(True) for n, _ in c.ema.online_model.named_parameters()]
    assert any(p.shape != q.shape for p, q in zip(params, c.ema.online_model.parameters()))
    topt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.99))
    topt.load_state_dict(osd)
    gg = torch.Generator().manual_seed(9)
    grads = [torch.randn(p.shape, generator=gg) * 1e-2 for p in params]
    for p, gr in zip(params, grads):
        p.grad = gr.clone()
    topt.step()
    d = fit(make(), [], ckpt=path)                           # loads weights + optimizer state, no steps
    unet = d.ema.online_model.model
    unet._flat.bind_grad_views()
    for p, gr in zip(d.ema.online_model.parameters(), grads):
        p.grad.copy_(gr.reshape(p.shape).to(dev))
    d._optimizers[0].step()
    for p, q in zip(params, d.ema.online_model.parameters()):
        assert torch.allclose(p.detach(), q.detach().cpu().reshape(p.shape), rtol=2e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------------------------------
# GaussianDiffusion's public method surface (reference ddpm.py:673-757, 782-876) on the HIP path, and the north_star's
# headline parity quantity: eps = model_predictions(...).pred_noise within 1e-4
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["small", "full", "full64"])
def test_model_predictions_match_reference_fixture(dev, golden_dir, tag, parity):
    """``model_predictions(x_t, t, clip_x_start=True, rederive_pred_noise=True)`` of the PRODUCT class against what the
    reference's own method returned (fixture keys pred_noise_clip / x_start_clip; per-sample t = (37, 912))."""
    fx, dim, S, P, img, noise, t = case(golden_dir, tag)
    net, gd = build(dim, S, P, dev, sampling_timesteps=50)
    x_t = torch.as_tensor(fx["x_t"]).to(dev)
    pred = gd.model_predictions(x_t, t.to(dev), clip_x_start=True, rederive_pred_noise=True)
    assert type(pred).__name__ == "ModelPrediction" and pred._fields == ("pred_noise", "pred_x_start")
    parity("pred_noise (eps), clipped x_start", rel(pred.pred_noise, fx["pred_noise_clip"]), RTOL)
    parity("pred_x_start, clipped", rel(pred.pred_x_start, fx["x_start_clip"]), RTOL)
    pn, xs = pred                                            # unpacks like the reference's namedtuple (:808)
    assert pn is pred.pred_noise and xs is pred.pred_x_start
    # q_sample(x0, t, noise) reproduces the fixture's x_t (the reference's own q_sample output)
    parity("q_sample vs the reference's x_t", rel(gd.q_sample((img * 2 - 1).to(dev), t.to(dev), noise.to(dev)), fx["x_t"]),
           1e-6)


def test_eps_prediction_at_the_benchmark_batch_against_the_oracle(dev, parity):
    """UNCLIPPED eps and x_start at BASELINE config 2's shape (dim 64, 32x32, B = 128, random per-sample t) against
    oracle.diffusion.model_predictions; and the raw network output (check both)."""
    from oracle import diffusion as OD
    dim, S, B = 64, 32, 128
    P = OD.unet_init(dim=dim, channels=3, seed=21)
    bufs = OD.diffusion_buffers(1000)
    g = torch.Generator().manual_seed(2100)
    x = torch.randn(B, 3, S, S, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    with torch.no_grad():
        pn_ref, xs_ref, v_ref = OD.model_predictions(P, bufs, x, t, clip_x_start=False, dim=dim)
        pn_c_ref, xs_c_ref, _ = OD.model_predictions(P, bufs, x, t, clip_x_start=True, dim=dim)
    net, gd = build(dim, S, P, dev)
    pred = gd.model_predictions(x.to(dev), t.to(dev))
    parity("raw network output v, B=128", rel(net(x.to(dev), t.to(dev)), v_ref), RTOL)
    parity("pred_noise (eps) unclipped, B=128", rel(pred.pred_noise, pn_ref), RTOL)
    parity("pred_x_start unclipped, B=128", rel(pred.pred_x_start, xs_ref), RTOL)
    # worst single sample: small-t samples divide by sqrt_recipm1 -> the per-sample error is the interesting one
    per = ((pred.pred_noise.cpu().double() - pn_ref.double()).flatten(1).norm(dim=1)
           / pn_ref.double().flatten(1).norm(dim=1))
    parity("pred_noise (eps) unclipped, worst of the 128 samples", float(per.max()), RTOL)
    pc = gd.model_predictions(x.to(dev), t.to(dev), clip_x_start=True, rederive_pred_noise=True)
    parity("pred_noise (eps) clipped, B=128", rel(pc.pred_noise, pn_c_ref), RTOL)
    parity("pred_x_start clipped, B=128", rel(pc.pred_x_start, xs_c_ref), RTOL)


def test_gaussian_diffusion_algebra_methods_match_the_reference_expressions(dev, parity):
    """q_sample / predict_v / predict_start_from_v / predict_start_from_noise / predict_noise_from_start / q_posterior
    (reference :673-705, 869-876) with a per-sample t, against the reference's ATen expressions evaluated on the CPU
    (oracle buffers).  Elementwise fp32 with separately rounded products: bit-equal is expected, 1e-6 is asserted."""
    from oracle import diffusion as OD
    bufs = OD.diffusion_buffers(1000)
    _, gd = build(16, 16, OD.unet_init(dim=16, channels=3, seed=1), dev)
    g = torch.Generator().manual_seed(5)
    B = 7
    a, b = torch.randn(B, 3, 16, 16, generator=g), torch.randn(B, 3, 16, 16, generator=g)
    t = torch.tensor([0, 1, 37, 500, 912, 998, 999])
    e = lambda n: OD._ext(bufs[n], t, 4)  # noqa: E731
    ad, bd, td = a.to(dev), b.to(dev), t.to(dev)
    cases = {
        "q_sample": (gd.q_sample(ad, td, bd), e("sqrt_alphas_cumprod") * a + e("sqrt_one_minus_alphas_cumprod") * b),
        "predict_v": (gd.predict_v(ad, td, bd), e("sqrt_alphas_cumprod") * b - e("sqrt_one_minus_alphas_cumprod") * a),
        "predict_start_from_v": (gd.predict_start_from_v(ad, td, bd),
                                 e("sqrt_alphas_cumprod") * a - e("sqrt_one_minus_alphas_cumprod") * b),
        "predict_start_from_noise": (gd.predict_start_from_noise(ad, td, bd),
                                     e("sqrt_recip_alphas_cumprod") * a - e("sqrt_recipm1_alphas_cumprod") * b),
        "predict_noise_from_start": (gd.predict_noise_from_start(ad, td, bd),
                                     (e("sqrt_recip_alphas_cumprod") * a - b) / e("sqrt_recipm1_alphas_cumprod")),
        "q_posterior mean": (gd.q_posterior(ad, bd, td)[0], e("posterior_mean_coef1") * a + e("posterior_mean_coef2") * b),
    }
    for name, (got, want) in cases.items():
        assert got.shape == want.shape and got.device.type == "cuda"
        parity(name, float((got.cpu() - want).abs().max() / want.abs().max()), 1e-6)
    _, var, logvar = gd.q_posterior(ad, bd, td)
    assert var.shape == (B, 1, 1, 1) and torch.equal(var.cpu(), e("posterior_variance"))
    assert torch.equal(logvar.cpu(), e("posterior_log_variance_clipped"))
    # q_sample without noise draws it (reference default(noise, randn_like)): right shape, finite, not x itself
    qs = gd.q_sample(ad, td)
    assert qs.shape == ad.shape and torch.isfinite(qs).all() and not torch.equal(qs, ad)
    assert torch.equal(gd.normalize(ad), ad * 2 - 1) and torch.equal(gd.unnormalize(ad), (ad + 1) * 0.5)
    with pytest.raises(AssertionError):
        gd.q_sample(ad, td[:3], bd)                          # one timestep per sample


def test_p_sample_p_mean_variance_and_loops_through_the_class_methods(dev, golden_dir, parity, monkeypatch):
    """p_mean_variance / p_sample (reference signatures, :736-757) against the reference's fixture images, and
    p_sample_loop / ddim_sample / sample / interpolate as METHODS: interpolate against the oracle's p_sample chain on
    the very draws the method made (replayed from the device generator's seed)."""
    from oracle import diffusion as OD
    fx, dim, S, P, img, noise, t = case(golden_dir, "small")
    net, gd = build(dim, S, P, dev, sampling_timesteps=50)
    bufs = OD.diffusion_buffers(1000)
    x_t = torch.as_tensor(fx["x_t"]).to(dev)
    nz = torch.as_tensor(fx["p_sample_noise"]).to(dev)
    pred_img, x_start = gd.p_sample(x_t, 500, noise=nz)
    parity("p_sample(x, 500) image", rel(pred_img, fx["p_sample_500"]), RTOL)
    pred0, _ = gd.p_sample(x_t, 0)
    parity("p_sample(x, 0) image", rel(pred0, fx["p_sample_0"]), RTOL)
    tb = torch.full((x_t.shape[0],), 500, device=dev, dtype=torch.long)
    mean, var, logvar, xs = gd.p_mean_variance(x_t, tb, clip_denoised=True)
    with torch.no_grad():
        _, xs_ref = OD.p_sample(P, bufs, x_t.cpu(), 500, nz.cpu(), dim=dim)
    parity("p_mean_variance x_start", rel(xs, xs_ref), RTOL)
    parity("p_sample's x_start == p_mean_variance's", rel(x_start, xs), 1e-6)
    parity("p_mean_variance mean + sigma * noise == p_sample", rel(mean + (0.5 * logvar).exp() * nz, fx["p_sample_500"]), RTOL)
    # loops as methods: shapes, range, and the DDIM loop is the same chain lgm_hip.sampler runs
    shape = tuple(fx["x_t"].shape)
    assert gd.is_ddim_sampling
    torch.manual_seed(11)
    a = gd.sample(batch_size=shape[0])
    torch.manual_seed(11)
    b = gd.ddim_sample(shape)
    assert a.shape == shape and torch.equal(a, b) and 0.0 <= float(a.min()) and float(a.max()) <= 1.0
    allt = gd.ddim_sample(shape, return_all_timesteps=True)
    assert allt.shape == (shape[0], 51) + shape[1:]
    # interpolate (:847-867), eager launches so that the draws are randn(shape) in program order
    monkeypatch.setenv("LGM_NO_SAMPLER_GRAPH", "1")
    x1, x2 = (img * 2 - 1).to(dev), (img.flip(0) * 2 - 1).to(dev)
    T0, lam = 6, 0.3
    torch.manual_seed(77)
    out = gd.interpolate(x1, x2, t=T0, lam=lam)
    torch.manual_seed(77)
    n1, n2 = torch.randn_like(x1), torch.randn_like(x2)
    steps = [torch.randn(shape, device=dev) for _ in range(T0 - 1)]          # steps T0-1 .. 1 draw, step 0 does not
    tb = torch.full((shape[0],), T0, dtype=torch.long)
    ref = (1 - lam) * OD.q_sample(bufs, x1.cpu(), tb, n1.cpu()) + lam * OD.q_sample(bufs, x2.cpu(), tb, n2.cpu())
    with torch.no_grad():
        for k, i in enumerate(reversed(range(T0))):
            ref, _ = OD.p_sample(P, bufs, ref, i, steps[k].cpu() if i > 0 else None, dim=dim)
    assert out.shape == shape
    parity(f"interpolate(x1, x2, t={T0}, lam={lam}) vs the oracle's chain on the same draws", rel(out, ref), RTOL)


@pytest.mark.parametrize("graph", [True, False], ids=["graph_replay", "eager"])
def test_stochastic_ddim_chain_matches_the_oracle(dev, golden_dir, parity, monkeypatch, graph):
    """DDIM with eta > 0 (reference ddpm.py:820-827: sigma = eta sqrt((1 - a/a_next)(1 - a_next)/(1 - a)), c = sqrt(1 - a_next -
    sigma^2), + sigma * noise): a 20-pair chain with eta = 0.7 and injected noise against oracle.ddim_sample_loop - the fixture
    loops only cover eta = 0, where the noise term is multiplied by zero."""
    from lgm_hip import sampler
    from models.generative.diffusion.ddpm import GaussianDiffusion
    from oracle import diffusion as OD
    if not graph:
        monkeypatch.setenv("LGM_NO_SAMPLER_GRAPH", "1")
    fx, dim, S, P, img, noise, t = case(golden_dir, "small")
    shape = tuple(fx["x_t"].shape)
    net, _ = build(dim, S, P, dev)
    gd = GaussianDiffusion(net, img_size=S, timesteps=1000, sampling_timesteps=20, ddim_sampling_eta=0.7).to(dev)
    bufs = OD.diffusion_buffers(1000)
    init, nz = OD.draw_loop_noise(31337, shape, 20)
    with torch.no_grad():
        ref = OD.ddim_sample_loop(P, bufs, init, nz, 20, eta=0.7, dim=dim)
    out = sampler.ddim_sample(gd, shape, init_noise=init.to(dev), noises=[n.to(dev) for n in nz])
    parity(f"20-pair DDIM chain, eta = 0.7 ({'graph replay' if graph else 'eager launches'})", rel(out, ref), RTOL)
    # the method itself draws its own noise: right shape and range, and a different draw gives a different image
    torch.manual_seed(1)
    a = gd.ddim_sample(shape)
    torch.manual_seed(2)
    b = gd.ddim_sample(shape)
    assert a.shape == shape and not torch.equal(a, b) and 0.0 <= float(a.min()) and float(a.max()) <= 1.0


def test_opt_in_kernel_routes_match_the_oracle(dev, parity, monkeypatch):
    """The round-6 opt-in routes - the time embedding as one launch (LGM_TIME_MLP=1) and the mid-sized 1x1 convolutions through the
    engine's un-split GEMM (LGM_GEMM1X1=1) - through the whole network: dim 64, 32x32, B = 8 (2048 pixel rows at 16x16, so the 1x1
    rule takes layers), loss and ALL parameter gradients against the oracle's autograd."""
    from lgm_hip import ops
    from oracle import diffusion as OD
    monkeypatch.setattr(ops, "TIME_MLP", True)
    monkeypatch.setattr(ops, "GEMM1X1", True)
    ops._TIME_MLP_OK.clear()
    taken = []
    real = ops.lib().lgm_weng_gemm_epi
    monkeypatch.setattr(ops.lib(), "lgm_weng_gemm_epi", lambda *a: (taken.append(a[3:6]), real(*a))[1])
    try:
        dim, S, B = 64, 32, 8
        P = OD.unet_init(dim=dim, channels=3, seed=13)
        bufs = OD.diffusion_buffers(1000)
        g = torch.Generator().manual_seed(1300)
        img = torch.rand(B, 3, S, S, generator=g)
        noise = torch.randn(B, 3, S, S, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        loss_ref = OD.diffusion_forward(Pr, bufs, img, t, noise, dim=dim)
        loss_ref.backward()
        net, gd = build(dim, S, P, dev)
        loss = gd.p_losses(img.to(dev), t.to(dev), noise.to(dev), _normalize=True)
        parity("loss with the opt-in routes", abs(loss.item() - loss_ref.item()) / loss_ref.item(), RTOL)
        loss.backward()
        errs = {n: rel(p.grad, Pr[n].grad) for n, p in net.named_parameters()}
        wn = max(errs, key=errs.get)
        parity(f"worst of ALL {len(errs)} parameter gradients with the opt-in routes ({wn})", errs[wn], RTOL)
        assert len(taken) >= 2, taken                      # the 1x1 rule really routed layers through the engine's GEMM (16x16 maps)
    finally:
        ops._TIME_MLP_OK.clear()
