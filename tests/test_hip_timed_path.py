"""GPU: parity of the path bench.py TIMES (``GraphedDDPMStep``: two HIP graphs per step) and of the BASELINE
sizes themselves (B=128 DDPM against the CPU oracle, N=4096 x K=512 VQ with a non-collapsed codebook), plus
the 2-rank data-parallel DDPM step (gloo ranks sharing the one GPU of the box)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lightning-generative-models_amd")


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


def _ddpm(dev, dim, S, seed=10, **kw):
    from models.generative.diffusion.ddpm import DDPM
    torch.manual_seed(seed)
    m = DDPM(img_channels=3, img_size=S, dim=dim, **kw)
    m.sample_every = 0
    m.to(dev)
    m.prepare_hip(dev)
    m.train()
    return m


@pytest.mark.parametrize("pipeline", [False, True])
def test_graph_replay_is_bit_identical_to_eager_steps(dev, pipeline, monkeypatch):
    """(pipeline=True: the opt-in LGM_STEP_PIPELINE structure - four graphs, each bucket's slab reduction and Adam slice
    on a side stream - must give the same bits.)
    The timed path at the BASELINE configuration (dim 64, 32x32, B=128, configs/diffusion/ddpm.json's
    optimiser): after ONE replay the loss and the WHOLE flat gradient buffer are torch.equal to an eager
    p_losses + backward on a twin with identical weights fed the (t, noise) the graph drew; after 12 replays
    the parameters, Adam moments and the EMA shadow are torch.equal to 12 eager steps fed the same draws
    (covers static-buffer aliasing across the two captures, RNG under capture, zero_grad-in-graph ordering,
    Adam's step count and the EMA update at step 10)."""
    from lgm_hip import graph as lgm_graph
    from lgm_hip.graph import GraphedDDPMStep
    monkeypatch.setattr(lgm_graph, "_STEP_PIPELINE", pipeline)
    kw = dict(lr=2e-5, betas=(0.9, 0.99), ema_update_every=10, ema_decay=0.995)
    a, b = _ddpm(dev, 64, 32, **kw), _ddpm(dev, 64, 32, **kw)
    fa, fb = a.ema.online_model.model._flat, b.ema.online_model.model._flat
    assert torch.equal(fa.data, fb.data)
    oa, ob = a.configure_optimizers(), b.configure_optimizers()
    g = torch.Generator().manual_seed(10)
    x = (torch.rand(128, 3, 32, 32, generator=g) * 2 - 1).to(dev)
    step = GraphedDDPMStep(a, oa, x.clone())
    gd_b = b.ema.online_model
    for i in range(12):
        loss_a = step.step(i).clone()
        t, noise = step.t.clone(), step.noise.clone()
        assert int(t.min()) >= 0 and int(t.max()) < 1000 and abs(float(noise.mean())) < 0.02
        ob.zero_grad()
        loss_b = gd_b.p_losses(x, t, noise, _normalize=True)
        loss_b.backward()
        if i == 0:
            assert torch.equal(loss_a.reshape(()), loss_b.detach().reshape(())), (float(loss_a), float(loss_b))
            assert torch.equal(fa.grad, fb.grad)
            assert float(fa.grad.abs().max()) > 0
        ob.step()
        b.on_train_batch_end(None, None, i)
    assert torch.equal(fa.data, fb.data)
    sa, sb = oa._flat_state[id(fa)], ob._flat_state[id(fb)]
    assert sa["step"] == sb["step"] == 12 and torch.equal(sa["m"], sb["m"]) and torch.equal(sa["v"], sb["v"])
    ea, eb = a.ema.ema_model.model._flat, b.ema.ema_model.model._flat
    assert torch.equal(ea.data, eb.data) and not torch.equal(ea.data, fa.data)   # shadow = weights at step 10
    # consecutive replays draw fresh (t, noise)
    t0 = step.t.clone()
    step.step(12)
    assert not torch.equal(t0, step.t)


def _plan_caches():
    from lgm_hip import ops
    return [getattr(ops, n) for n in ("_WINO4_OK", "_WINO_WS", "_EPI_STATS", "_PAIR_OK", "_WG2_OK", "_WG2_WS", "_WINO_OK")
            if hasattr(ops, n)]


@pytest.mark.parametrize("B,light", [(128, 0), (64, 1), (16, 1)], ids=["b128", "rank_of_2_b64_light", "rank_of_8_b16_light"])
def test_ddpm_at_the_baseline_batch_matches_the_cpu_oracle(dev, parity, B, light):
    """dim 64, 32x32 (BASELINE config 2 as benched): HIP loss and every parameter gradient against
    oracle.diffusion_forward + autograd on the CPU (a few seconds) - at B = 128 with the kernel selection of one GPU, and at
    the per-rank batches of 2 and 8 GPUs with the selection a RANK gets (light F(4x4) workgroups, csrc/winograd4l.hip, and launch plans for 240 of the 256 CUs: the
    library switches both on when WORLD_SIZE > 1, here through lgm_wino4_set_light / lgm_set_cu_margin)."""
    from lgm_hip import ops
    for c in _plan_caches():
        c.clear()                 # plans are cached per geometry: none may survive a change of the kernel selection
    ops.lib().lgm_wino4_set_light(1 if light else -1)
    ops.lib().lgm_set_cu_margin(16 if light else -1)      # a rank's launch plans leave 16 CUs to the collective
    try:
        _ddpm_vs_oracle(dev, parity, B, light)
    finally:
        ops.lib().lgm_wino4_set_light(-1)
        ops.lib().lgm_set_cu_margin(-1)
        for c in _plan_caches():
            c.clear()


def _ddpm_vs_oracle(dev, parity, B, light):
    from models.generative.diffusion.ddpm import GaussianDiffusion, Unet
    from oracle import diffusion as OD
    from lgm_hip import ops
    dim, S = 64, 32
    P = OD.unet_init(dim=dim, channels=3, seed=128)
    g = torch.Generator().manual_seed(1280)
    img = torch.rand(B, 3, S, S, generator=g)
    noise = torch.randn(B, 3, S, S, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    loss_ref = OD.diffusion_forward(Pr, OD.diffusion_buffers(1000), img, t, noise, dim=dim)
    loss_ref.backward()
    net = Unet(dim=dim, channels=3)
    net.load_state_dict(P, strict=True)
    gd = GaussianDiffusion(net, img_size=S, timesteps=1000).to(dev)
    net.prepare_hip(dev)
    loss = gd.p_losses(img.to(dev), t.to(dev), noise.to(dev), _normalize=True)
    parity("loss", abs(loss.item() - loss_ref.item()) / loss_ref.item(), RTOL)
    loss.backward()
    if light:                     # the light kernel really ran (its name is the last F(4x4) launch the library noted somewhere
        g32 = ops.make_geom(B, S, S, dim, dim, 3, 3, 1, 1)      # in this pass; asked directly: the first level's 3x3 layers take it)
        assert ops.lib().lgm_conv3x3_wino4_preferred(__import__("ctypes").byref(g32), 0) == 1
        assert ops.lib().lgm_conv3x3_wino4l_supported(__import__("ctypes").byref(g32), 0) == 1
    errs = {n: rel(p.grad, Pr[n].grad) for n, p in net.named_parameters()}
    wn = max(errs, key=errs.get)
    parity(f"worst of ALL {len(errs)} parameter gradients ({wn})", errs[wn], RTOL)
    gn = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in net.parameters())).item()
    gr = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in Pr.values())).item()
    parity("all-parameter gradient norm", abs(gn - gr) / gr, RTOL)


def test_vq_full_size_non_collapsed_codebook(dev, parity):
    """N=4096 rows (B=256 x 4x4), K=512, D=64 — configs/vae/vqvae.json's quantiser problem — with latents
    scaled so that hundreds of codes are in use: indices torch.equal to the oracle's argmin (reference
    vector_quantizer.py:45-69) outside a 1e-6 top-2 margin band (count reported), segment sums exact."""
    from lgm_hip import ops
    from oracle import vq as OV
    N, K, D = 4096, 512, 64
    g = torch.Generator().manual_seed(4096)
    cb = (torch.rand(K, D, generator=g) * 2 - 1) / K
    x = torch.randn(N, D, generator=g) * (0.6 / K)          # comparable to the codebook's spread
    dist = OV.vq_distances(x, cb)
    ref = dist.argmin(1)
    top2 = dist.topk(2, dim=1, largest=False).values
    margin = (top2[:, 1] - top2[:, 0]) / top2[:, 0].abs().clamp_min(1e-30)
    used = int(torch.unique(ref).numel())
    p = torch.bincount(ref, minlength=K).float() / N
    ppl = float(torch.exp(-(p * (p + 1e-10).log()).sum()))
    print(f"[parity] vq full size: {used} of {K} codes used, perplexity {ppl:.1f}")
    assert used > 200 and ppl > 100
    L = ops.lib()
    xd, cbd = x.to(dev), cb.to(dev)
    idx = torch.empty(N, dtype=torch.long, device=dev)
    L.lgm_vq_assign(xd.data_ptr(), D, cbd.data_ptr(), N, K, D, idx.data_ptr(), None, ops.stream())
    mism = idx.cpu() != ref
    band = margin <= 1e-6
    # both counts go on record (profiles/rNN_parity_errors.json): whether the band was ever needed is auditable
    parity("rows inside the 1e-6 relative top-2 margin band (count)", int(band.sum()), N + 1)
    parity("index mismatches in total (count; allowed only inside the band)", int(mism.sum()), int(band.sum()) + 1)
    parity("index mismatches OUTSIDE the band (count)", int((mism & ~band).sum()), 1)
    # a mismatch inside the band must still pick one of the two (numerically tied) nearest codes
    if int(mism.sum()):
        second = dist.topk(2, dim=1, largest=False).indices[:, 1]
        assert torch.equal(idx.cpu()[mism], second[mism])
    dw, counts = torch.empty(K, D, device=dev), torch.empty(K, device=dev)
    L.lgm_vq_segment_sum(xd.data_ptr(), D, idx.data_ptr(), N, K, D, dw.data_ptr(), counts.data_ptr(), ops.stream())
    onehot = torch.nn.functional.one_hot(idx.cpu(), K).float()
    assert torch.equal(counts.cpu(), onehot.sum(0))
    parity("segment sums dw = onehot^T x", rel(dw, onehot.double().T @ x.double()), 1e-6)



@pytest.mark.parametrize("ema", [False, True])
def test_vqvae_graph_replay_is_bit_identical_to_eager_steps(dev, ema):
    """What bench.py --workload vqvae times and MiniTrainer.fit drives (``VQVAE.make_fast_step`` ->
    ``ModuleFastStep``: training_step + backward in ONE HIP graph, Adam eager): after 6 steps on 6 different
    batches the parameters, the Adam moments, the EMA codebook buffers and every logged value are torch.equal to
    the same steps issued from eager launches - i.e. the warm-up / capture left the training state (codebook EMA,
    random stream) untouched and the static buffers do not alias."""
    from models.generative.vae.vqvae import VQVAE

    def make():
        torch.manual_seed(3)
        m = VQVAE(img_channels=3, img_size=32, embedding_dim=64, num_embeddings=512, hidden_dim=128,
                  num_residual_layers=2, num_residual_hiddens=32, use_ema=ema, lr=1e-3, b1=0.9, b2=0.999,
                  loss_weights={"recon_loss": 1, "vq_loss": 10 if ema else 1}).to(dev)
        m.prepare_hip(dev)
        m.train()
        return m, m.configure_optimizers()

    (a, oa), (b, ob) = make(), make()
    assert torch.equal(a._flat.data, b._flat.data)
    fa, fb = a.make_fast_step(oa, 1, True), b.make_fast_step(ob, 1, False)
    g = torch.Generator().manual_seed(4)
    for i in range(6):
        x = (torch.rand(64, 3, 32, 32, generator=g) * 2 - 1).to(dev)
        la = fa.step((x, None), i).detach().clone()
        lb = fb.step((x.clone(), None), i).detach().clone()
        assert torch.equal(la, lb), (i, float(la), float(lb))
        for k in b.logged:
            assert torch.equal(a.logged[k].detach(), b.logged[k].detach()), (i, k)
    assert fa.mode.startswith("hipGraph") and fb.mode == "eager"
    assert torch.equal(a._flat.data, b._flat.data)
    sa, sb = a.state_dict(), b.state_dict()
    for k in sb:
        assert torch.equal(sa[k], sb[k]), k
    for pa, pb in zip(oa.state_dict()["state"].values(), ob.state_dict()["state"].values()):
        for k in pb:
            assert torch.equal(torch.as_tensor(pa[k]), torch.as_tensor(pb[k])), k


@pytest.mark.parametrize("ema", [False, True])
def test_vqvae_trainer_fast_path_checkpoint_resume(dev, tmp_path, ema):
    """MiniTrainer.fit drives a VQVAE through ``make_fast_step`` (HIP-graph replay): a Lightning-layout checkpoint
    written after 3 steps resumes - new process state, graph captured again at the first batch after the resume -
    to exactly the parameters, EMA codebook buffers and Adam moments of 6 uninterrupted steps."""
    from lgm_hip.graph import ModuleFastStep
    from lgm_hip.lightning import MiniTrainer, save_checkpoint
    from models.generative.vae.vqvae import VQVAE

    def make():
        torch.manual_seed(3)
        return VQVAE(img_channels=3, img_size=32, embedding_dim=64, num_embeddings=512, hidden_dim=64,
                     num_residual_layers=2, num_residual_hiddens=32, use_ema=ema, lr=1e-3, b1=0.9, b2=0.999,
                     loss_weights={"recon_loss": 1, "vq_loss": 10 if ema else 1})

    g = torch.Generator().manual_seed(6)
    batches = [(torch.rand(32, 3, 32, 32, generator=g) * 2 - 1, torch.zeros(32, dtype=torch.long)) for _ in range(6)]

    def fit(m, data, ckpt=None):
        tr = MiniTrainer(max_epochs=1, default_root_dir=None, log_every=0, device=dev)
        tr.fit(m, train_dataloader=data, ckpt_path=ckpt)
        return m, tr

    a, tra = fit(make(), batches)
    assert isinstance(tra.fast, ModuleFastStep) and tra.fast.mode.startswith("hipGraph")
    b1, _ = fit(make(), batches[:3])
    path = str(tmp_path / "step3.ckpt")
    save_checkpoint(b1, list(b1._optimizers), path)
    b2, trb = fit(make(), batches[3:], ckpt=path)
    assert b2.global_step == 6 == a.global_step and trb.fast.mode.startswith("hipGraph")
    sa, sb = a.state_dict(), b2.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    oa, ob = a._optimizers[0].state_dict()["state"], b2._optimizers[0].state_dict()["state"]
    for i in oa:
        for k in oa[i]:
            assert torch.equal(torch.as_tensor(oa[i][k]), torch.as_tensor(ob[i][k])), (i, k)


def test_module_fast_step_mixed_batch_shapes(dev):
    """A batch of another shape in the middle of a run (a ragged last batch, a larger evaluation batch) goes through
    eager launches - which may grow the shared workspace the captured graph has baked in - and the next full batch
    replays the graph again: parameters equal an all-eager twin after B = 64, 64, 24, 96, 64, 64."""
    from models.generative.vae.vqvae import VQVAE

    def make():
        torch.manual_seed(5)
        m = VQVAE(img_channels=3, img_size=32, embedding_dim=64, num_embeddings=512, hidden_dim=64,
                  num_residual_layers=2, num_residual_hiddens=32, use_ema=True, lr=1e-3, b1=0.9, b2=0.999,
                  loss_weights={"recon_loss": 1, "vq_loss": 10}).to(dev)
        m.prepare_hip(dev)
        m.train()
        return m, m.configure_optimizers()

    (a, oa), (b, ob) = make(), make()
    fa, fb = a.make_fast_step(oa, 1, True), b.make_fast_step(ob, 1, False)
    g = torch.Generator().manual_seed(8)
    for i, B in enumerate((64, 64, 24, 96, 64, 64)):
        x = (torch.rand(B, 3, 32, 32, generator=g) * 2 - 1).to(dev)
        la, lb = fa.step((x, None), i), fb.step((x.clone(), None), i)
        assert torch.equal(la.detach(), lb.detach()), (i, B)
    assert fa.mode.startswith("hipGraph") and fa.static[0].shape[0] == 64
    assert torch.equal(a._flat.data, b._flat.data)
    for k, v in b.state_dict().items():
        assert torch.equal(a.state_dict()[k], v), k

def test_gp_penalty_zero_gradient_pixel(dev):
    """A pixel whose channel gradient is exactly zero: penalty (0-1)^2 and a ZERO subgradient (torch's
    backward of norm(2, dim=1)), not NaN."""
    from lgm_hip import ops
    L = ops.lib()
    npix, C = 512, 3
    g = torch.randn(npix, 4)
    g[:, 3] = 0
    g[7] = 0
    g[300] = 0
    gt = g[:, :3].clone().requires_grad_(True)
    r = gt.norm(2, dim=1)
    pen_ref = 10.0 * ((r - 1) ** 2).mean()
    pen_ref.backward()
    gd_, pen, gbar = g.to(dev), torch.empty(1, device=dev), torch.empty(npix, 4, device=dev)
    one = torch.ones(1, device=dev)
    ws = ops.workspace(L.lgm_gp_penalty_workspace(npix), dev)
    L.lgm_gp_penalty(gd_.data_ptr(), npix, C, 10.0, one.data_ptr(), pen.data_ptr(), gbar.data_ptr(), ws.data_ptr(),
                     ops.stream())
    assert torch.isfinite(gbar).all()
    assert float(gbar[7].abs().max()) == 0 and float(gbar[300].abs().max()) == 0
    assert rel(pen, pen_ref) < 1e-6 and rel(gbar[:, :3], gt.grad) < 1e-6


_DDP_WORKER = r'''
import json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.dirname(sys.argv[1]))
mode = sys.argv[2]                                   # "graph" | "eager"
from lgm_hip.graph import DDPMFastStep
from models.generative.diffusion.ddpm import DDPM
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
def make():
    torch.manual_seed(10)
    m = DDPM(img_channels=3, img_size=16, dim=16, lr=1e-3, betas=(0.9, 0.99), ema_update_every=2)
    m.sample_every = 0
    m.to(dev); m.prepare_hip(dev); m.train()
    return m
B = 8
n = B // world
g = torch.Generator().manual_seed(3)
xs = [(torch.rand(B, 3, 16, 16, generator=g) * 2 - 1) for _ in range(3)]
m = make()
opt = m.configure_optimizers()
fast = DDPMFastStep(m, opt, world, use_graph=(mode == "graph"))
torch.manual_seed(10 + 0)                            # every rank seeds identically (reference train.py:20)
ref = make() if rank == 0 else None
ropt = ref.configure_optimizers() if rank == 0 else None
worst = 0.0
for i in range(3):
    xr = xs[i][rank * n:(rank + 1) * n].to(dev)
    if mode == "graph":
        fast.step((xr, None), i)
        t, noise = fast.graphed.t.clone(), fast.graphed.noise.clone()
    else:
        # eager path of the same object: inject the draws through p_losses so the twin can replay them
        t = torch.randint(0, 1000, (n,), device=dev)
        noise = torch.randn(n, 3, 16, 16, device=dev)
        net = m.ema.online_model.model
        net.grad_sync = fast.sync
        loss = m.ema.online_model.p_losses(xr, t, noise, _normalize=True)
        loss.backward()
        fast.sync.finish()
        opt.step(); opt.zero_grad()
        m.on_train_batch_end(None, None, i)
    tl = [torch.empty_like(t.cpu()) for _ in range(world)]
    nl = [torch.empty_like(noise.cpu()) for _ in range(world)]
    dist.all_gather(tl, t.cpu()); dist.all_gather(nl, noise.cpu())
    if rank == 0:                                    # one process, concatenated batch, same draws
        ropt.zero_grad()
        l = ref.ema.online_model.p_losses(xs[i].to(dev), torch.cat(tl).to(dev), torch.cat(nl).to(dev), _normalize=True)
        l.backward()
        ropt.step()
        ref.on_train_batch_end(None, None, i)
mine = m.ema.online_model.model._flat.data.cpu()
shadow = m.ema.ema_model.model._flat.data.cpu()
gl = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(gl, mine)
same = all(torch.equal(gl[0], t) for t in gl)
if rank == 0:
    r = ref.ema.online_model.model._flat.data.cpu()
    rs = ref.ema.ema_model.model._flat.data.cpu()
    e = float((mine.double() - r.double()).norm() / r.double().norm())
    es = float((shadow.double() - rs.double()).norm() / rs.double().norm())
    moved = float((mine - make().ema.online_model.model._flat.data.cpu()).abs().max())
    print("DDP_RESULT " + json.dumps({"mode": fast.mode, "ranks_identical": same, "rel_err_vs_1rank": e,
                                      "ema_rel_err": es, "moved": moved}), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode", ["graph", "eager", "graph-late"])
def test_two_rank_ddpm_step_equals_one_rank_on_concatenated_batch(dev, tmp_path, mode, parity):
    """2 ranks (gloo collectives, both on the box's one GPU), per-rank batch 4, three
    optimizer steps through DDPMFastStep — bucketed all-reduce overlapped with the hand-written backward, 1/N
    folded into Adam, graph replay or eager launches — equal one process on the concatenated batch of 8 fed the
    same (t, noise) draws, within fp32 reduction-order tolerance; all ranks hold bit-identical parameters."""
    script = tmp_path / "ddp_worker.py"
    script.write_text(_DDP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    port = {"graph": "29551", "eager": "29553", "graph-late": "29555"}[mode]
    if mode == "graph-late":        # LGM_DDP_OVERLAP=0: ONE all-reduce of the whole gradient buffer after the backward
        env["LGM_DDP_OVERLAP"] = "0"
        mode = "graph"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), PKG, mode],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("DDP_RESULT ")]
    assert line, r.stdout + r.stderr[-2000:]
    res = json.loads(line[0][len("DDP_RESULT "):])
    assert res["ranks_identical"] and res["moved"] > 0
    if mode == "graph":
        assert res["mode"].startswith("hipGraph"), res
    parity(f"2-rank parameters vs 1-rank after 3 steps ({res['mode']})", res["rel_err_vs_1rank"], 1e-5)
    parity("2-rank EMA shadow vs 1-rank", res["ema_rel_err"], 1e-5)


def _wgan(dev, seed=11, size=64, ch=3, latent=100):
    from lgm_hip.lightning import _CountingOptimizer
    from models.generative.gan.wgan import WGAN
    torch.manual_seed(seed)
    m = WGAN(img_channels=ch, img_size=size, latent_dim=latent, lr=1e-4, b1=0.5, b2=0.999, weight_decay=1e-5,
             n_critic=5, grad_penalty=10, constraint_method="gp").to(dev)
    m.prepare_hip(dev)
    m.train()
    m._optimizers = [_CountingOptimizer(o, m) for o in m.configure_optimizers()[0]]
    return m


def test_wgan_graph_replay_is_bit_identical_to_eager_steps(dev):
    """What bench.py --workload wgan_gp64 times and MiniTrainer.fit drives (``WGAN.make_fast_step`` -> ``WGANFastStep``:
    the critic update and the generator update as one HIP graph each, 5 : 1 schedule + Adam on the host): after 13
    steps (two whole cycles + one critic step) on different batches the critic's and the generator's parameters, both
    Adam states, every BatchNorm running statistic AND batch counter and every logged loss are torch.equal to the same
    steps through the module's own ``training_step`` - i.e. warm-up / capture left running statistics, counters and the
    random stream untouched, and the replayed schedule is the reference's (wgan.py:58-82)."""
    a, b = _wgan(dev), _wgan(dev)
    assert torch.equal(a.D._flat.data, b.D._flat.data) and torch.equal(a.G._flat.data, b.G._flat.data)
    fa = a.make_fast_step(a._optimizers, 1, True)
    g = torch.Generator().manual_seed(12)
    torch.manual_seed(99)
    sa = torch.cuda.get_rng_state(dev)
    kinds = []
    for i in range(13):
        x = (torch.rand(16, 3, 64, 64, generator=g) * 2 - 1).to(dev)
        torch.cuda.set_rng_state(sa, dev)
        la = fa.step((x, None), i)
        sa_next = torch.cuda.get_rng_state(dev)
        torch.cuda.set_rng_state(sa, dev)                  # the twin draws the same z / alpha
        b.training_step((x.clone(), None))
        assert torch.equal(torch.cuda.get_rng_state(dev), sa_next), f"step {i}: graph and eager consume the RNG differently"
        sa = sa_next
        kinds.append("g" if "g_loss" in la else "d")
        for k, v in la.items():
            assert torch.equal(v.detach(), b.logged[k].detach()), (i, k, float(v), float(b.logged[k]))
        assert a.global_step == b.global_step == i + 1
    assert kinds == ["d"] * 5 + ["g"] + ["d"] * 5 + ["g"] + ["d"]
    assert fa.mode.startswith("hipGraph") and set(fa.graphs) == {"d", "g"}
    assert torch.equal(a.D._flat.data, b.D._flat.data) and torch.equal(a.G._flat.data, b.G._flat.data)
    sda, sdb = a.state_dict(), b.state_dict()
    assert any(k.endswith("num_batches_tracked") and int(v) > 0 for k, v in sdb.items())
    for k in sdb:
        assert torch.equal(sda[k], sdb[k]), k
    for oa, ob in zip(a._optimizers, b._optimizers):
        for pa, pb in zip(oa.state_dict()["state"].values(), ob.state_dict()["state"].values()):
            for k in pb:
                assert torch.equal(torch.as_tensor(pa[k]), torch.as_tensor(pb[k])), k


def test_wgan_trainer_drives_the_fast_step(dev):
    """MiniTrainer.fit builds the WGAN's fast step (manual optimisation module) and ends where the module's own
    training_step loop ends, bit for bit."""
    from lgm_hip.graph import WGANFastStep
    from lgm_hip.lightning import MiniTrainer
    from models.generative.gan.wgan import WGAN

    def make():
        torch.manual_seed(21)
        return WGAN(img_channels=1, img_size=28, latent_dim=128, lr=1e-4, b1=0.5, b2=0.999, weight_decay=1e-5,
                    n_critic=5, grad_penalty=10, constraint_method="gp")
    g = torch.Generator().manual_seed(22)
    data = [(torch.rand(8, 1, 28, 28, generator=g) * 2 - 1, torch.zeros(8, dtype=torch.long)) for _ in range(8)]
    out = []
    for fast_path in (True, False):
        torch.manual_seed(5)
        m = make()
        tr = MiniTrainer(max_epochs=1, log_every=0, device=dev, fast_path=fast_path)
        tr.fit(m, train_dataloader=data)
        assert isinstance(tr.fast, WGANFastStep) == fast_path
        out.append(m)
    a, b = out
    assert a.global_step == b.global_step == 8
    for k, v in b.state_dict().items():
        assert torch.equal(a.state_dict()[k], v), k


_WGAN_DDP_WORKER = r'''
import json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.dirname(sys.argv[1]))
mode = sys.argv[2]                                   # "graph" | "eager"
from lgm_hip.lightning import BufferSync, _CountingOptimizer
from models.generative.gan.wgan import WGAN
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
def make():
    torch.manual_seed(10)
    m = WGAN(img_channels=1, img_size=28, latent_dim=128, lr=1e-4, b1=0.5, b2=0.999, weight_decay=1e-5, n_critic=2,
             grad_penalty=10, constraint_method="gp").to(dev)
    m.prepare_hip(dev); m.train()
    m._optimizers = [_CountingOptimizer(o, m) for o in m.configure_optimizers()[0]]
    return m
B, n = 8, 8 // world
g = torch.Generator().manual_seed(3)
xs = [(torch.rand(B, 1, 28, 28, generator=g) * 2 - 1) for _ in range(6)]
m = make()
fast = m.make_fast_step(m._optimizers, world, use_graph=(mode == "graph"))
bs = BufferSync(m)
ref = make() if rank == 0 else None                  # one process: per-shard passes with the same draws, averaged by hand
rfast = ref.make_fast_step(ref._optimizers, 1, use_graph=False) if rank == 0 else None
torch.manual_seed(77)                                # every rank seeds identically (reference train.py:20)
state = torch.cuda.get_rng_state(dev)
exchanged = []
orig = dist.all_reduce
def counting(t, *a, **k):
    exchanged.append(t.numel())
    return orig(t, *a, **k)
dist.all_reduce = counting
for i in range(6):
    bs.broadcast()                                   # DDP: rank 0's running statistics before every forward
    torch.cuda.set_rng_state(state, dev)
    fast.step((xs[i][rank * n:(rank + 1) * n].to(dev), None), i)
    nxt = torch.cuda.get_rng_state(dev)
    if rank == 0:
        critic = (ref.global_step + 1) % 3 != 0
        net = ref.D if critic else ref.G
        fn = rfast._critic if critic else rfast._generator
        grads, keep = [], None
        bufs = [b_ for b_ in ref.buffers() if b_.dtype.is_floating_point]
        before = [b_.clone() for b_ in bufs]
        for r in range(world):
            for b_, s_ in zip(bufs, before):
                b_.copy_(s_)                         # every rank starts from rank 0's buffers
            torch.cuda.set_rng_state(state, dev)
            fn(xs[i][r * n:(r + 1) * n].to(dev))
            grads.append(net._flat.grad.clone())
            if r == 0:
                keep = [b_.clone() for b_ in bufs]   # rank 0's statistics are the ones that survive
        for b_, s_ in zip(bufs, keep):
            b_.copy_(s_)
        net._flat.grad.copy_(sum(grads) / world)
        opt = ref._optimizers[0 if critic else 1]
        opt.step(); opt.zero_grad()
    state = nxt
dist.all_reduce = orig
bs.broadcast()
def flat_of(mm):
    return torch.cat([mm.D._flat.data, mm.G._flat.data] + [b_.reshape(-1) for b_ in mm.buffers() if b_.dtype.is_floating_point]).cpu()
mine = flat_of(m)
gl = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(gl, mine)
same = all(torch.equal(gl[0], t) for t in gl)
if rank == 0:
    r = flat_of(ref)
    e = float((mine.double() - r.double()).norm() / r.double().norm())
    moved = float((mine - flat_of(make())).abs().max())
    # one exchange per update, of exactly the flat buffer that update wrote (4 critic + 2 generator steps)
    sizes = sorted(set(exchanged))
    print("DDP_RESULT " + json.dumps({"mode": fast.mode, "ranks_identical": same, "rel_err_vs_emulation": e, "moved": moved,
                                      "exchanges": len(exchanged), "sizes_ok": sizes == sorted({m.D._flat.total, m.G._flat.total})}),
          flush=True)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode", ["graph", "eager"])
def test_two_rank_wgan_step(dev, tmp_path, mode, parity):
    """2 ranks (gloo collectives on the box's one GPU) through ``WGANFastStep``: BatchNorm statistics stay per rank
    (no SyncBN, like the reference under DDP), each update exchanges exactly ONE flat gradient buffer (the critic's or
    the generator's) with 1/N folded into Adam, rank 0's running statistics are re-broadcast before every forward.
    Six steps (n_critic = 2) equal a one-process emulation - per-shard passes with the same random draws, gradients
    averaged by hand - and all ranks end bit-identical."""
    script = tmp_path / "wgan_ddp_worker.py"
    script.write_text(_WGAN_DDP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    port = "29557" if mode == "graph" else "29559"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), PKG, mode],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("DDP_RESULT ")]
    assert line, r.stdout + r.stderr[-2000:]
    res = json.loads(line[0][len("DDP_RESULT "):])
    assert res["ranks_identical"] and res["moved"] > 0 and res["exchanges"] == 6 and res["sizes_ok"], res
    if mode == "graph":
        assert res["mode"].startswith("hipGraph"), res
    parity(f"2-rank WGAN parameters + running statistics vs one-process emulation ({res['mode']})",
           res["rel_err_vs_emulation"], 1e-5)


def test_bench_starts_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` with NO launcher in the environment: the process becomes the parent of two ranks
    (lgm_hip/launch.py; here gloo ranks sharing the box's one GPU) and relays rank 0's line, whose n_gpus is 2.  The
    reference's ``python train.py`` does the same through Lightning's DDPStrategy (train.py:38, lightning_utils.py:37-43)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LGM_DIST_BACKEND="gloo", LGM_DIST_TIMEOUT="240")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--only", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and "[launch] starting 2 ranks" in r.stderr
    # the N > 1 line explains itself: exchange mode, the step's collectives, how much of the exchange
    # the backward did not hide, and the rank count the collectives really ran on
    cfg = rec["config"]
    assert cfg["rccl_ranks"] == 2 and cfg["backend"] == "gloo" and "all-reduce" in cfg["grad_exchange"]
    assert cfg["collectives_per_step"] == len(cfg["bucket_bytes"]) == 5
    # round 6: the rank selection (CU margin, light workgroups) follows the EXCHANGE, not WORLD_SIZE - gloo reduces on the
    # host, nothing is resident beside the backward, so these two ranks run the one-GPU rules and the line says why
    # (the RCCL side of the rule: tests/test_hip_rccl.py)
    assert cfg["kernel_selection"]["cu_margin"] == 0 and "gloo" in cfg["kernel_selection"]["rule"], cfg["kernel_selection"]
    assert sum(cfg["bucket_bytes"]) >= 4 * 35_719_555                      # padded channel lanes make the flat buffer larger
    assert rec["comm_exposed_ms"] >= 0 and rec["comm_host_wait_ms"] >= 0 and 0 <= rec["comm_exposed_frac_of_step"] < 1.5
    # under a launcher that disagrees with --gpus the run is refused; it never prints a line with another rank count
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--only", "--steps", "1", "--warmup", "0",
                         "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env2)
    assert r2.returncode != 0 and not [ln for ln in r2.stdout.splitlines() if ln.startswith("{")]
