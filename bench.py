"""bench.py — BASELINE.json metric: training images/sec, DDPM UNet 32x32, global batch 128.

    python bench.py --gpus N --steps K --warmup W [--workload ddpm32|ddpm64|wgan_gp64|vqvae] [--only]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Without a launcher in the environment ``--gpus N`` (N > 1) makes this process the parent of N ranks (lgm_hip/launch.py:
one torch.distributed.run child, started before any GPU call; rank 0's line is relayed, the exit code is the ranks').
Under a launcher WORLD_SIZE must equal --gpus; the line's n_gpus is always the number of ranks that ran.

One "step" = the full optimiser step of the reference loop on a synthetic batch already
resident in HBM.  Default workload (the headline, BASELINE config 2): DDPM.training_step (t ~ randint,
noise ~ randn on device, q_sample, UNet fwd, weighted MSE) -> hand-written HIP backward -> gradient all-reduce
over RCCL when N > 1 -> fused Adam -> EMA update (every 10th step), replayed as HIP graphs.  Strong
scaling: the global batch stays 128 (reference DataModule divides the config batch by the GPU count).

Rank 0 prints ONE JSON line.  Its top level is the headline; on one GPU (and without --only) the same line carries
  "secondary"       BASELINE's other configs measured the same way right after the headline leg, each with its own
                    "roofline" and "cpu_baseline":
                      ddpm64     config 5's network: DDPM UNet 64x64, B = 64 per GPU
                      wgan_gp64  config 3: WGAN-GP DCGAN G/D 64x64, B = 128, n_critic = 5 (one training_step = one D or G update)
                      vqvae / vqvae_ema   config 4: VQ-VAE 32x32, K = 512, B = 256 (plain / EMA codebook)
                      ddpm64_sampling     config 5's sampling half: DDIM, 64 images 64x64, 250 steps, one graph replay per step
                      ddpm64_sampling_1000  the same at its stated length: sample() -> 1000-step ancestral chain (p_sample_loop)
  "per_rank_proxy"  ms per step of the headline workload at the per-rank batches of 2 / 4 / 8 GPUs (64 / 32 / 16
                    images) on this one GPU: what a rank computes between its gradient exchanges under strong scaling.
Objects of every workload:
  "roofline"      the dominant SINGLE kernel of the convolution family (name with template arguments, as rocprofv3
                  prints it), timed live with HIP events on the launch stream in one extra instrumented step after the
                  timed region.  frac = EXECUTED MFMA FLOPs / event time / 157.3 TFLOP/s (<= 1 by construction): a
                  Winograd F(2x2,3x3) kernel executes algorithmic / 2.25 FLOPs; algorithmic_tflops is the
                  2 B Ho Wo Cout KH KW Cin count over the same time.  ceiling_img_per_s = the fp32 MFMA roof of the
                  algorithm actually run (Winograd on the 3x3 layers), next to SURVEY.md's direct-convolution ceiling.
  "cpu_baseline"  the CPU oracle (kind "port"), same workload at its own batch for ~10 s of CPU work, on this host's cores (count + model).
"""
import argparse
import json
import os
import re
import sys
import time

# Multi-process GPU work on this pool's driver stack needs dmabuf IPC: the image exports HSA_ENABLE_IPC_MODE_LEGACY=0
# (build-environment notes: "the host driver only supports dmabuf IPC, and without it RCCL / CUDA-tensor sharing
# across processes fails with hipIpcGetMemHandle: invalid argument").  The environment's value always wins; the
# default below only covers a shell that lost it, and main() prints the effective value and its source on stderr.
_IPC_ENV_SOURCE = "environment" if "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ else "bench.py default"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix == fp32 vector peak
WINO_FACTOR = 2.25              # direct 3x3 multiplies per Winograd F(2x2,3x3) multiply (36 / 16)
WINO4_FACTOR = 4.0              # ... per Winograd F(4x4,3x3) multiply (144 / 36)


def mfma_factor(kernel_name: str) -> float:
    """algorithmic (direct-convolution) FLOPs per FLOP the MFMA pipe executes, by kernel family"""
    if "weng" in kernel_name:        # the engine's GEMM: a plain 1x1 GEMM by default (factor 1); with LGM_WENG=1 it also runs the
        # 4x4 / stride-2 layers as F(4x4,2x2) on pixel phases (100 products per 16 outputs instead of 256)
        return 2.56 if os.environ.get("LGM_WENG", "0") == "1" else 1.0
    if "wino4" in kernel_name:
        return WINO4_FACTOR
    if "wino_" in kernel_name:
        return WINO_FACTOR
    return 1.0
HBM_PEAK = 8.0e12               # bytes / s (MI355X_MICROARCH.md)


def _conv_work(module, batch):
    """MAC counter over the convolution modules of a network AFTER it ran at ``batch`` (their geometries are cached):
    per layer, in registration order, (forward multiply-accumulates per image with the LOGICAL channel counts — no
    padding lanes —, input + output bytes per image in fp32).  Returns (layers, n_params)."""
    from lgm_hip.nn import Conv2d, ConvTranspose2d
    layers = []
    for m in module.modules():
        if not isinstance(m, (Conv2d, ConvTranspose2d)):
            continue
        for g in m._geoms.values():
            if g.B != batch:
                continue
            # the geometry's X side (H, W) is the convolution's input, and the LARGE map of a transposed convolution
            mm = float(g.Ho) * g.Wo * m.cin * m.cout * g.KH * g.KW
            xin, yout = ((g.H * g.W * m.cin, g.Ho * g.Wo * m.cout) if isinstance(m, Conv2d)
                         else (g.Ho * g.Wo * m.cin, g.H * g.W * m.cout))
            layers.append((mm, 4.0 * (xin + yout)))
            break
    return layers, sum(p.numel() for p in module.parameters())


def _threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))   # a 1-GPU box owns a 16-core share of the host


def _timed_cpu(one, batch, warmup, min_seconds, max_steps, what):
    nthreads = _threads()
    torch.set_num_threads(nthreads)
    times, i = [], 0
    while i < warmup or (sum(times) < min_seconds and len(times) < max_steps):
        if i < warmup or len(times) % 10 == 0:
            print(f"[bench] cpu_baseline step {i} ({nthreads} threads, {sum(times):.1f}s timed)", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        one()
        if i >= warmup:
            times.append(time.perf_counter() - t0)
        i += 1
    total = sum(times)
    return {"value": round(batch * len(times) / total, 2), "unit": "images/s", "cores": nthreads, "cpu_model": _cpu_model(),
            "kind": "port",
            "sample": f"{what}: {len(times)} steps = {total:.1f} s of CPU work after {warmup} warm-up"}


def _cpu_model() -> str:
    """'report the core count and CPU model with the number'."""
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline_ddpm(img, dim=64, batch=128, min_seconds=12.0):
    """The CPU oracle (validated against the reference by tests/golden): whole training steps
    (forward + backward + Adam) until ``min_seconds`` of timed work have accumulated."""
    from oracle import diffusion as OD
    torch.manual_seed(10)
    P = {k: v.requires_grad_(True) for k, v in OD.unet_init(dim=dim, channels=3, seed=0).items()}
    bufs = OD.diffusion_buffers(1000)
    opt = torch.optim.Adam(list(P.values()), lr=2e-5, betas=(0.9, 0.99))
    x = torch.rand(batch, 3, img, img) * 2 - 1

    def one():
        t = torch.randint(0, 1000, (batch,))
        noise = torch.randn_like(x)
        loss = OD.diffusion_forward(P, bufs, x, t, noise, dim=dim)
        opt.zero_grad()
        loss.backward()
        opt.step()
    return _timed_cpu(one, batch, 1, min_seconds, 200, f"oracle fwd+bwd+Adam at the workload's own batch, B={batch}, 3x{img}x{img}")


def cpu_baseline_wgan(batch=128, min_seconds=12.0):
    from oracle import gan as OG
    torch.manual_seed(10)
    G, D = OG.gan_init(64, 3, 100, seed=0)
    G = {k: v.requires_grad_(True) for k, v in G.items()}
    D = {k: v.requires_grad_(True) for k, v in D.items()}
    dopt = torch.optim.Adam(list(D.values()), lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-5)
    gopt = torch.optim.Adam(list(G.values()), lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-5)
    x = torch.rand(batch, 3, 64, 64) * 2 - 1
    step = [0]

    def one():      # reference schedule: 5 critic updates, then 1 generator update (wgan.py:58-82)
        z = torch.randn(batch, 100, 1, 1)
        x_hat = OG.generator(G, z, 64, 3)
        if (step[0] + 1) % 6 != 0:
            alpha = torch.rand(batch, 1, 1, 1)
            ld = OG.wgan_d_loss(D, x, x_hat.detach(), alpha, 10.0, 64)
            dopt.zero_grad()
            ld["d_loss"].backward()
            dopt.step()
        else:
            gl = OG.wgan_g_loss(D, x_hat, 64)
            gopt.zero_grad()
            gl.backward()
            gopt.step()
        step[0] += 1
    return _timed_cpu(one, batch, 1, min_seconds, 300, f"oracle WGAN-GP training_step (5 D : 1 G) + Adam, B={batch}, 3x64x64")


def cpu_baseline_vqvae(ema, batch=256, min_seconds=10.0):
    from oracle import vq as OV
    torch.manual_seed(10)
    P = {k: v.requires_grad_(True) for k, v in OV.vqvae_init(seed=0).items()}
    opt = torch.optim.Adam(list(P.values()), lr=1e-3, betas=(0.9, 0.999))
    x = torch.rand(batch, 3, 32, 32) * 2 - 1
    state = [(torch.zeros(512), P["vector_quantizer.embedding.weight"].detach().clone())] if ema else [None]

    def one():
        r = OV.vqvae_step(P, x, w_recon=1.0, w_vq=10.0 if ema else 1.0, ema_state=state[0])
        opt.zero_grad()
        r["loss"].backward()
        opt.step()
    return _timed_cpu(one, batch, 1, min_seconds, 400, f"oracle VQ-VAE fwd+bwd+Adam (use_ema={ema}), B={batch}, 3x32x32")


def torch_gpu_baseline(dev, img, dim, batch, warmup=5, steps=10):
    """Optional comparison leg (--torch-gpu-baseline, DDPM workloads): the same oracle (the reference's arithmetic as
    plain torch ops) with its tensors on the MI355X — PyTorch-ROCm eager kernels (MIOpen / rocBLAS / ATen) +
    torch.optim.Adam.  A reported comparison point, not the target."""
    from oracle import diffusion as OD
    torch.manual_seed(10)
    P = {k: v.to(dev).requires_grad_(True) for k, v in OD.unet_init(dim=dim, channels=3, seed=0).items()}
    bufs = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in OD.diffusion_buffers(1000).items()}
    opt = torch.optim.Adam(list(P.values()), lr=2e-5, betas=(0.9, 0.99))
    x = (torch.rand(batch, 3, img, img) * 2 - 1).to(dev)

    def one():
        t = torch.randint(0, 1000, (batch,), device=dev)
        noise = torch.randn_like(x)
        loss = OD.diffusion_forward(P, bufs, x, t, noise, dim=dim)
        opt.zero_grad()
        loss.backward()
        opt.step()

    for i in range(warmup):
        print(f"[bench] torch_gpu_baseline warm-up {i}", file=sys.stderr, flush=True)
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(batch / dt, 2), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3), "kind": "port",
            "sample": f"oracle (torch eager ops on the GPU) fwd+bwd+Adam, B={batch}, mean of {steps} steps after {warmup}"}


# ---- workloads: each returns (step(i) -> loss-like tensor, eager_step for the instrumented pass, info) ------------
def setup_ddpm(args, dev, world, rank, img, global_batch):
    from lgm_hip.graph import DDPMFastStep
    from models.generative.diffusion.ddpm import DDPM
    assert global_batch % world == 0
    per_gpu = global_batch // world
    model = DDPM(img_channels=3, img_size=img, dim=64, diffusion_timesteps=1000, sampling_timesteps=None,
                 lr=2e-5, betas=(0.9, 0.99), ema_update_every=10, ema_decay=0.995)   # configs/diffusion/ddpm[_64].json
    model.sample_every = 0
    model.to(dev)
    model.prepare_hip(dev)
    model.train()
    opt = model.configure_optimizers()
    g = torch.Generator(device="cpu").manual_seed(10 + rank)
    x = (torch.rand(per_gpu, 3, img, img, generator=g) * 2 - 1).to(dev)
    y = torch.zeros(per_gpu, dtype=torch.long, device=dev)
    # the object MiniTrainer.fit drives for a DDPM module: overlapped bucketed all-reduce + graph replay
    fast = DDPMFastStep(model, opt, world, use_graph=not args.no_graph)
    eager = DDPMFastStep.__new__(DDPMFastStep)
    eager.__dict__.update(fast.__dict__)
    eager.use_graph, eager.graphed = False, None

    def step(i):
        return fast.step((x, y), i)

    def eager_step(i):
        return eager.step((x, y), i)
    def work():     # forward x 3; bytes per image + (weights x 3, Adam 28 B/param, EMA / 10) per step
        return {32: 10.95e9, 64: 43.8e9}[img], (69.6e6 if img == 32 else 278e6), 1.472e9, {}
    info = dict(per_gpu=per_gpu, fast=fast, work=work, n_instr=1, model=model,
                workload=f"configs/diffusion/{'ddpm' if img == 32 else 'ddpm_64'}.json UNet dim=64, 3x{img}x{img} synthetic "
                         "NCHW fp32, training_step+backward+Adam+EMA")
    return step, eager_step, info


def setup_wgan(args, dev):
    from lgm_hip.lightning import _CountingOptimizer
    from models.generative.gan.wgan import WGAN
    B = 128
    m = WGAN(img_channels=3, img_size=64, latent_dim=100, lr=1e-4, b1=0.5, b2=0.999, weight_decay=1e-5, n_critic=5,
             grad_penalty=10, constraint_method="gp").to(dev)        # configs/gan/wgan_gp_celeba.json
    m.prepare_hip(dev)
    m.train()
    m._optimizers = [_CountingOptimizer(o, m) for o in m.configure_optimizers()[0]]
    x = torch.rand(B, 3, 64, 64, device=dev) * 2 - 1
    # what MiniTrainer.fit drives for a WGAN: critic graph / generator graph, schedule + Adam on the host
    fast = m.make_fast_step(m._optimizers, 1, use_graph=not args.no_graph)
    eager = m.make_fast_step(m._optimizers, 1, use_graph=False)
    if not args.no_graph:
        # both graphs are captured HERE (capture leaves the training state untouched): the first generator update is step 5,
        # which `--warmup 5` would put - with its capture - inside the timed region
        for key in ("d", "g"):
            fast._capture(key, x)

    def step(i):
        ld = fast.step((x, None), i)
        return ld.get("d_loss", ld.get("g_loss"))

    def eager_step(i):
        ld = eager.step((x, None), i)
        return ld.get("d_loss", ld.get("g_loss"))

    def work():
        """Exact multiply-accumulate count of one training_step (reference wgan.py:58-156), from the layers' cached
        geometries.  G, D = forward MACs per image; g_i / d_i per layer.
          critic update    G fwd + 3 D fwd (real, fake, interpolates) + 2 x [D wgrad + D dgrad without layer 1]
                           + first-order input gradient on the interpolates (D) + its double backward:
                           sweep 1 = wgrad of every layer (D) + W * u of layers 1..4 (D - d_5);
                           sweep 2 = dgrad + wgrad of layers 2..4, wgrad of layer 1
          generator update G fwd + D fwd + D dgrad (all layers) + G wgrad + G dgrad without layer 1
          one training_step = (5 critic + 1 generator) / 6"""
        gl, gp_ = _conv_work(m.G, B)
        dl, dp_ = _conv_work(m.D, B)
        G, D = sum(a for a, _ in gl), sum(a for a, _ in dl)
        d = [a for a, _ in dl]
        critic = G + 3 * D + 2 * (2 * D - d[0]) + D + D + (D - d[-1]) + 2 * sum(d[1:-1]) + d[0]
        gen = G + D + D + G + (G - gl[0][0])
        flop = 2.0 * (5 * critic + gen) / 6
        # bytes: every convolution-shaped pass reads its input and writes its output once; + weights per pass, Adam 28 B/param
        gb, db = sum(b for _, b in gl), sum(b for _, b in dl)
        critic_b = gb + 3 * db + 2 * 2 * db + db + 2 * db + 2 * db
        gen_b = gb + db + db + 2 * gb
        byts = (5 * critic_b + gen_b) / 6
        per_step = (5 * (dp_ * (4 * 9 + 28) + gp_ * 4) + (gp_ * (4 * 3 + 28) + dp_ * 4 * 2)) / 6
        return flop, byts, per_step, dict(G_fwd_gflop=round(2 * G / 1e9, 4), D_fwd_gflop=round(2 * D / 1e9, 4),
                                          critic_update_gflop=round(2 * critic / 1e9, 4),
                                          generator_update_gflop=round(2 * gen / 1e9, 4))
    info = dict(per_gpu=B, fast=fast, work=work, n_instr=6,
                workload="configs/gan/wgan_gp_celeba.json WGAN-GP DCGAN G/D, 3x64x64 synthetic, training_step "
                         "(n_critic = 5: one critic OR generator update incl. GP double backward) + fused Adam")
    return step, eager_step, info


def setup_vqvae(args, dev, ema):
    from models.generative.vae.vqvae import VQVAE
    B = 256
    v = VQVAE(img_channels=3, img_size=32, embedding_dim=64, num_embeddings=512, hidden_dim=128, num_residual_layers=2,
              num_residual_hiddens=32, use_ema=ema, lr=1e-3, b1=0.9, b2=0.999,
              loss_weights={"recon_loss": 1, "vq_loss": 10 if ema else 1}).to(dev)      # configs/vae/vqvae[_ema].json
    v.prepare_hip(dev)
    v.train()
    opt = v.configure_optimizers()
    xv = torch.rand(B, 3, 32, 32, device=dev) * 2 - 1
    fast = v.make_fast_step(opt, 1, use_graph=not args.no_graph)    # what MiniTrainer.fit drives for a VQVAE
    eager = v.make_fast_step(opt, 1, use_graph=False)               # the roofline leg times individual launches

    def step(i):
        return fast.step((xv, None), i)

    def eager_step(i):
        return eager.step((xv, None), i)

    def work():
        """Convolutions: forward + weight gradient + input gradient (not for the image-end layer of the encoder);
        quantiser: 2 N K D for the nearest-code search (N = B * 16 latent vectors) and as much for the one-hot^T x
        statistics (EMA) / codebook gradient."""
        el, _ = _conv_work(v.encoder, B)
        dl, _ = _conv_work(v.decoder, B)
        macs = 3 * (sum(a for a, _ in el) + sum(a for a, _ in dl)) - el[0][0]
        n_lat, K, D = 16, 512, 64
        macs += 2 * n_lat * K * D
        n_params = sum(p.numel() for p in v.parameters())
        byts = 3 * (sum(b for _, b in el) + sum(b for _, b in dl)) + 2 * 4 * n_lat * D * 2
        return 2.0 * macs, byts, n_params * (4 * 3 + 28.0), dict(
            encoder_fwd_mflop=round(2 * sum(a for a, _ in el) / 1e6, 3), decoder_fwd_mflop=round(2 * sum(a for a, _ in dl) / 1e6, 3),
            quantiser_mflop_per_step=round(2 * 2 * B * n_lat * K * D / 1e6, 1))
    info = dict(per_gpu=B, fast=fast, work=work, n_instr=1,
                workload=f"configs/vae/vqvae{'_ema' if ema else ''}.json VQ-VAE 3x32x32, K=512 D=64, "
                         "training_step+backward+Adam")
    return step, eager_step, info


def run_workload(wl, args, dev, world, rank, steps, warmup, batch=None, roofline=True, cpu=True):
    """Warm-up, the timed region (barrier + synchronize on both sides, max over ranks), the instrumented roofline
    step and the CPU baseline of ONE workload.  Returns the workload's JSON object (rank 0) or None."""
    from lgm_hip import ops
    torch.manual_seed(10)                         # reference train.py:20 seeds every rank identically
    if wl in ("ddpm32", "ddpm64"):
        img = 32 if wl == "ddpm32" else 64
        # ddpm32: strong scaling of the global batch 128 (the metric); ddpm64: 64 images per GPU (config 5, weak)
        gb = batch if batch is not None else (128 if wl == "ddpm32" else 64 * world)
        step, eager_step, info = setup_ddpm(args, dev, world, rank, img, gb)
        scaling = "strong" if wl == "ddpm32" else "weak"
        metric = f"training images/sec (DDPM UNet {img}x{img}, bs={gb if wl == 'ddpm32' else '64/GPU'})"
    else:
        assert world == 1, f"--workload {wl} is a single-GPU measurement"
        step, eager_step, info = (setup_wgan(args, dev) if wl == "wgan_gp64" else setup_vqvae(args, dev, wl == "vqvae_ema"))
        gb, scaling = info["per_gpu"], "weak"
        metric = ("training images/sec (WGAN-GP 64x64, bs=128)" if wl == "wgan_gp64"
                  else f"training images/sec (VQ-VAE 32x32 K=512{' EMA' if wl == 'vqvae_ema' else ''}, bs=256)")
    per_gpu = info["per_gpu"]

    tw = time.perf_counter()
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if rank == 0:
        print(f"[bench] {wl}: warm-up done: {warmup} steps in {time.perf_counter() - tw:.2f}s", file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    final_loss = float(loss.item()) if loss is not None else float("nan")
    ms = elapsed / steps * 1e3
    value = gb * steps / elapsed
    # N > 1: a few more steps (outside the timed region) with HIP events around the waits for the gradient exchange - how
    # much of the communication the backward did NOT hide - and the collectives one step issues, for the line's `config`
    comm = None
    sync = getattr(info.get("fast"), "sync", None)
    if sync is not None and not isinstance(sync, dict) and sync.active:
        sync.measure = True
        for i in range(min(5, steps)):
            step(warmup + steps + i)
        ev_ms, host_ms, n_meas = sync.exposed_ms()
        sync.measure = False
        cm = torch.tensor([ev_ms or 0.0, host_ms or 0.0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(cm, op=dist.ReduceOp.MAX)
        comm = {"comm_exposed_ms": round(float(cm[0]), 4), "comm_host_wait_ms": round(float(cm[1]), 4),
                "comm_measured_steps": n_meas,
                "bucket_bytes": [4 * int(n) for n in sync.last_buckets], "collectives_per_step": len(sync.last_buckets),
                "exchange": ("bucketed all-reduce overlapped with the backward" if sync.overlap
                             else "one all-reduce after the backward (LGM_DDP_OVERLAP=0)"),
                "backend": dist.get_backend(), "rccl_ranks": dist.get_world_size()}
    if rank == 0:
        print(f"[bench] {wl}: timed region: {steps} steps in {elapsed:.3f}s ({value:.1f} img/s)", file=sys.stderr, flush=True)
    launch = info["fast"].mode if info["fast"] is not None else "eager"
    if not roofline or args.no_roofline:
        return {"ms_per_step": round(ms, 3), "value": round(value, 2), "launch": launch, "steps": steps, "warmup": warmup}

    # ---- roofline leg: one extra instrumented step (a whole 5 D : 1 G cycle for the WGAN), per-launch HIP events on
    # the launch stream.  One un-instrumented eager pass first: the timed region replayed graphs, and the first launches
    # issued from Python afterwards pay one-off costs (allocator growth, lazily sized workspaces) that are not the kernels'
    n_instr = info["n_instr"]
    for k in range(n_instr):
        eager_step(warmup + steps + k)
    torch.cuda.synchronize()
    # The instrumented step is issued from Python, ~15 us of host work per launch: on an idle stream the event in front of
    # a launch completes at once and the span then contains the HOST's gap up to the launch (the F(4x4) dispatch does two
    # more dictionary look-ups than the others and read 132 instead of 47 us per launch).  So the stream is given ~0.1 s of
    # unrelated work first (plain torch GEMMs, measurement harness only): every launch of the step queues up behind it,
    # and the spans are GPU time.
    _busy = torch.randn(6144, 6144, device=dev)
    for _ in range(16):
        _busy = (_busy @ _busy) * 1e-4
    ops.TIMER = ops.KernelTimer()
    for k in range(n_instr):
        eager_step(warmup + steps + n_instr + k)
    del _busy
    fam = ops.TIMER.summary(False)
    kern = ops.TIMER.summary(True)
    ops.TIMER = None
    if rank != 0:
        return None

    def executed(name, flops):       # FLOPs the MFMA pipe executes for `flops` algorithmic ones
        return flops / mfma_factor(name)
    name, d = max(((k, v) for k, v in kern.items() if v["flops"] > 0), key=lambda kv: kv[1]["ms"])
    alg_tflops = d["flops"] / (d["ms"] * 1e-3) / 1e12
    exe_tflops = executed(name, d["flops"]) / (d["ms"] * 1e-3) / 1e12
    conv_ms = sum(v["ms"] for v in fam.values())
    conv_fl = sum(v["flops"] for v in fam.values())
    conv_exe = sum(executed(k, v["flops"]) for k, v in kern.items())
    wino_saved = sum(v["flops"] * (1.0 - 1.0 / mfma_factor(k)) for k, v in kern.items())   # FLOPs Winograd does not execute
    # HBM-side bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process; the
    # figure comes from the committed rocprofv3 --pmc summary of this same command (tools/pmc_kernels.py), if any
    # (newest round first; a file whose rows do not hold this kernel's exact name is skipped, never half-used).  The same
    # file's step totals give the counter-side HBM fraction of the whole step.
    traffic, traffic_src, step_counter_bytes, traffic_stale = None, None, None, None
    from lgm_hip._lib import source_fingerprint
    now = source_fingerprint()
    for cand in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):
        pmc = os.path.join(ROOT, "profiles", cand)
        if traffic is None and os.path.exists(pmc) and wl == "ddpm32" and gb == 128 and world == 1:
            with open(pmc) as fh:
                doc = json.load(fh)
            t = doc.get("kernels", {}).get(name)
            if not t:
                continue
            # counters measured on other kernel code are not this run's traffic: a file without a fingerprint, or whose
            # fingerprint differs from the tree in the file that defines this kernel (or in a shared header), is STALE
            then = doc.get("sources")
            base = re.sub(r"<.*$", "", name).split("::")[-1]
            own = [f for f in now if f.endswith(".h") or base in open(os.path.join(
                ROOT, "lightning-generative-models_amd", "csrc", f)).read()] if then else []
            if not then or any(then.get(f) != now[f] for f in own):
                traffic_stale = f"profiles/{cand}: measured on other sources" + (
                    "" if not then else " (" + ", ".join(f for f in own if then.get(f) != now[f]) + ")")
                break
            traffic, traffic_src = t["traffic_bytes_per_launch"], f"profiles/{cand}"
            st = doc.get("step_total") or {}
            if "fetch_bytes" in st and "write_bytes" in st and then == now:     # the step total depends on every kernel
                step_counter_bytes = st["fetch_bytes"] + st["write_bytes"]
    flop_per_img, bytes_per_img, bytes_per_step, detail = info["work"]()
    roof = {"bound": "mfma", "kernel": name, "achieved": round(exe_tflops, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(exe_tflops / FP32_MFMA_PEAK_TFLOPS, 4),
            "algorithmic_tflops": round(alg_tflops, 2),
            "executed_flops_per_launch": round(executed(name, d["flops"]) / d["launches"]),
            "algorithmic_flops_per_launch": round(d["flops"] / d["launches"]),
            "launches_per_step": round(d["launches"] / n_instr, 2), "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2),
            "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": bool(traffic_stale),
            "traffic_stale_why": traffic_stale,
            "algorithmic_bytes_per_launch": round(d.get("bytes", 0) / max(d["launches"], 1)) or None,
            "traffic_over_algorithmic": (round(traffic / (d["bytes"] / d["launches"]), 3)
                                         if traffic and d.get("bytes") else None),
            "kernel_hbm_counter_frac_of_8TBps": (round(traffic / (d["ms"] * 1e-3 / d["launches"]) / HBM_PEAK, 4)
                                                 if traffic else None),
            "note": ("achieved / frac count the FLOPs the MFMA pipe EXECUTES: a Winograd F(2x2,3x3) kernel executes "
                     "algorithmic / 2.25, an F(4x4,3x3) kernel (wino4) algorithmic / 4; algorithmic_tflops is the "
                     "direct-convolution count over the same time"),
            "kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                            "executed_tflops": round(executed(k, v["flops"]) / (v["ms"] * 1e-3) / 1e12, 2),
                            "frac": round(executed(k, v["flops"]) / (v["ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}
                        for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:10]},
            "family": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                           "algorithmic_tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in fam.items()},
            "conv_family_ms_per_step": round(conv_ms / n_instr, 3),
            "conv_family_executed_frac": round(conv_exe / (conv_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "conv_family_algorithmic_tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2)}
    # step level: algorithmic work (SURVEY.md §8(d) / the MAC counter) against both roofs, and the MFMA ceiling of the
    # algorithm actually run (Winograd on the layers that took it in the instrumented step)
    exe_per_img = flop_per_img - wino_saved / n_instr / per_gpu
    roof["flop_per_img"] = round(flop_per_img)
    roof["executed_flop_per_img"] = round(exe_per_img)
    roof["step_algorithmic_frac_of_peak"] = round(value * flop_per_img / (FP32_MFMA_PEAK_TFLOPS * 1e12 * world), 4)
    roof["step_executed_frac_of_peak"] = round(value * exe_per_img / (FP32_MFMA_PEAK_TFLOPS * 1e12 * world), 4)
    roof["ceiling_img_per_s"] = round(FP32_MFMA_PEAK_TFLOPS * 1e12 * world / exe_per_img, 1)
    roof["ceiling_img_per_s_direct_conv"] = round(FP32_MFMA_PEAK_TFLOPS * 1e12 * world / flop_per_img, 1)
    roof["hbm_frac_of_8TBps"] = round(value / world * (bytes_per_img + bytes_per_step / per_gpu) / HBM_PEAK, 4)
    # the same fraction from the counters (FETCH_SIZE + WRITE_SIZE of one step, committed PMC passes) over THIS run's step time
    roof["hbm_counter_bytes_per_step"] = step_counter_bytes
    roof["hbm_counter_frac_of_8TBps"] = (round(step_counter_bytes / (ms * 1e-3) / HBM_PEAK, 4)
                                         if step_counter_bytes else None)
    roof["hbm_counter_over_algorithmic"] = (round(step_counter_bytes / (per_gpu * bytes_per_img + bytes_per_step), 3)
                                            if step_counter_bytes else None)
    if detail:
        roof["work"] = detail
    conv_mode = "bf16x3 (LGM_CONV_MODE)" if ops.B3 else (
        ("fp32 MFMA, Winograd for the 3x3 layers: F(4x4,3x3) forward / input gradient on the large maps, F(2x2,3x3) elsewhere"
         if ops.WINO4 else "fp32 MFMA, Winograd F(2x2,3x3) for the 3x3 layers") if ops.WINO else "fp32 MFMA, direct")
    line = {"metric": metric, "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32" if not ops.B3 else "f32 (3x3 conv fwd/dgrad: opt-in bf16x3 split MFMA, fp32-level error)",
            "data": "synthetic",
            "config": {"workload": info["workload"], "global_batch": gb, "per_gpu_batch": per_gpu,
                       "parallelism": f"dp{world}", "final_loss": round(final_loss, 5), "launch": launch,
                       "conv_mode": conv_mode},
            "roofline": roof}
    if world > 1 or comm:
        line["config"]["grad_exchange"] = ("bucketed all-reduce overlapped with the backward (RCCL)"
                                           if os.environ.get("LGM_DDP_OVERLAP", "1") != "0"
                                           else "one all-reduce after the backward (LGM_DDP_OVERLAP=0)")
        from lgm_hip import ops as _ops
        # which rule chose the kernels (lgm_hip.lightning.FlatGradSync._select_kernels): margin + light workgroups only
        # while the exchange really overlaps the backward on RCCL; the environment's LGM_CU_MARGIN / LGM_WINO4_LIGHT win
        sel = dict(getattr(sync, "selection", None) or {}) if sync is not None and not isinstance(sync, dict) else {}
        sel.setdefault("cu_margin", int(_ops.lib().lgm_cu_margin()))
        sel["note"] = ("launch plans sized for 256 - cu_margin CUs (a collective's workgroups hold the rest); "
                       "LGM_CU_MARGIN / LGM_WINO4_LIGHT override the rule")
        line["config"]["kernel_selection"] = sel
        if comm:
            line["config"].update({k: comm[k] for k in ("bucket_bytes", "collectives_per_step", "backend", "rccl_ranks")})
            line["comm_exposed_ms"] = comm["comm_exposed_ms"]
            line["comm_host_wait_ms"] = comm["comm_host_wait_ms"]
            line["comm_exposed_frac_of_step"] = round(comm["comm_exposed_ms"] / ms, 4) if ms else None
    if world == 1 and cpu and not args.no_cpu_baseline:
        secs = args.cpu_seconds
        if wl == "ddpm32":
            line["cpu_baseline"] = cpu_baseline_ddpm(32, min_seconds=secs)
        elif wl == "ddpm64":
            line["cpu_baseline"] = cpu_baseline_ddpm(64, batch=64, min_seconds=secs)
        elif wl == "wgan_gp64":
            line["cpu_baseline"] = cpu_baseline_wgan(min_seconds=secs)
        else:
            line["cpu_baseline"] = cpu_baseline_vqvae(wl == "vqvae_ema", min_seconds=secs)
    if world == 1 and args.torch_gpu_baseline and wl in ("ddpm32", "ddpm64"):
        line["torch_gpu_baseline"] = torch_gpu_baseline(dev, 32 if wl == "ddpm32" else 64, 64, gb)
    return line


def _sampling_leg(dev, kind, steps, batch, img):
    from lgm_hip import sampler
    from models.generative.diffusion.ddpm import DDPM
    torch.manual_seed(10)
    m = DDPM(img_channels=3, img_size=img, dim=64, diffusion_timesteps=1000, sampling_timesteps=steps).to(dev)
    m.sample_every = 0
    m.prepare_hip(dev)
    gd = m.ema.ema_model
    gd.eval()
    shape = (batch, 3, img, img)
    if kind == "ddim":
        assert gd.is_ddim_sampling
        chain = lambda: sampler.ddim_sample(gd, shape)
        chain()                                      # capture + one whole chain as warm-up
    else:
        # T = 1000 sampling steps == T diffusion steps: ``sample()`` dispatches to the ancestral p_sample_loop
        # (reference ddpm.py:836-845).  Warm-up: the capture plus a short chain on the same graph
        # would need another schedule object, so the one warm-up here is the capture itself + 20 replays.
        assert not gd.is_ddim_sampling and gd.num_timesteps == steps
        chain = lambda: sampler.p_sample_loop(gd, shape)
        sampler.warm_chain(gd, shape, replays=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = chain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fwd_flop = {32: 3.651e9, 64: 14.594e9}[img] * batch
    # forward FLOPs by kernel family at 64 x 64 input (SURVEY section 8a layer table x 4): 3x3 layers on maps >= 16 x 16 run
    # F(4x4,3x3) (0.75 of the forward FLOPs), the 8 x 8 maps F(2x2,3x3) (0.23), the rest (7x7, 1x1, attention) direct
    share4, share2 = (0.75, 0.23) if img == 64 else (0.57, 0.41)
    from lgm_hip import ops as _ops
    if not _ops.WINO4:
        share4, share2 = 0.0, share4 + share2
    exe = fwd_flop * (1.0 - share4 * (1.0 - 1.0 / WINO4_FACTOR) - share2 * (1.0 - 1.0 / WINO_FACTOR))
    per = dt / steps
    name = "DDIM" if kind == "ddim" else "ancestral (p_sample_loop)"
    return {"metric": f"{name} sampling, {batch} images {img}x{img}, {steps} steps (UNet forward + update per step, one HIP "
                      "graph replay each)", "value": round(batch * steps / dt, 1), "unit": "image-steps/s",
            "ms_per_step": round(per * 1e3, 3), "seconds_per_chain": round(dt, 3), "finite": bool(torch.isfinite(out).all()),
            "dtype": "f32", "config": {"workload": "configs/diffusion/ddpm_64.json EMA network, "
                                       + ("ddim_sample, eta = 0" if kind == "ddim" else "sample() -> p_sample_loop, T = 1000"),
                                       "batch": batch, "steps": steps},
            "roofline": {"bound": "mfma", "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "step_algorithmic_frac_of_peak": round(fwd_flop / per / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                         "step_executed_frac_of_peak": round(exe / per / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                         "flop_per_step": round(fwd_flop)}}


def run_sampling(dev, steps=250, batch=64, img=64):
    """Config 5's second half, DDIM form: sampling with the EMA network of a 64x64 DDPM (reference ddim_sample,
    ddpm.py:782-834), ``batch`` images, ``steps`` sampling steps, each = one UNet forward + the fused update, replayed from ONE
    HIP graph per step.  Random-init weights, device-side noise.  Step-level roofline: 14.594 GFLOP per image and UNet
    forward at 64x64, against the fp32 MFMA peak (direct-convolution count) and against the
    Winograd-executed count."""
    return _sampling_leg(dev, "ddim", steps, batch, img)


def run_sampling_ancestral(dev, steps=1000, batch=64, img=64):
    """Config 5's sampling half AT ITS STATED LENGTH: the 1000-step chain ``sample()`` runs when sampling_timesteps ==
    timesteps (reference ddpm.py:759-780, 836-845): 64 images 64x64, 1000 x (UNet forward + posterior step with fresh
    noise), one graph replay per step, no host synchronisation inside the chain."""
    return _sampling_leg(dev, "ancestral", steps, batch, img)


def _release():
    """between workloads: drop what the finished one allocated (models, graph pools, caches keyed by its buffers)"""
    import gc
    from lgm_hip import ops
    gc.collect()
    ops.forget_dead_flats()
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=10)     # ... after >= 10 warm-up steps
    ap.add_argument("--workload", default="ddpm32", choices=["ddpm32", "ddpm64", "wgan_gp64", "vqvae", "vqvae_ema"])
    ap.add_argument("--vq-ema", action="store_true", help="vqvae workload: EMA codebook (= --workload vqvae_ema)")
    ap.add_argument("--only", action="store_true", help="only the named workload: no secondary configs, no per-rank proxy")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true",
                    help="profiling runs (rocprofv3 --kernel-trace / --pmc around this command): only warm-up + the timed "
                         "steps, no instrumented step and none of the stream filler it queues behind")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="timed CPU work per cpu_baseline leg")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch from Python (no HIP-graph replay)")
    ap.add_argument("--batch", type=int, default=None, help="global batch (default: the workload's config value)")
    ap.add_argument("--torch-gpu-baseline", action="store_true",
                    help="also time the oracle's plain torch ops on the GPU (PyTorch-ROCm eager) as a comparison point")
    args = ap.parse_args()
    if args.vq_ema and args.workload == "vqvae":
        args.workload = "vqvae_ema"

    from lgm_hip import launch
    if not launch.launched():
        if args.gpus > 1:
            # no launcher in the environment: this process becomes the PARENT of --gpus ranks (what the reference's
            # Trainer does for ``python train.py`` on a multi-GPU node).  Nothing above touched the GPU; the children
            # are started as a child process, rank 0's JSON line goes straight to our stdout, their exit code is ours.
            backend = os.environ.get("LGM_DIST_BACKEND", "nccl")
            ndev = launch.visible_gpu_count()
            if backend == "nccl" and ndev < args.gpus:
                print(f"[bench] --gpus {args.gpus} but {ndev} GPU(s) are visible (LGM_DIST_BACKEND=gloo rehearses the "
                      "N>1 path with several ranks on one GPU)", file=sys.stderr, flush=True)
                sys.exit(2)
            sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # --gpus N must never report another rank count: the driver computes scaling from n_gpus
        print(f"[bench] --gpus {args.gpus} but the launcher set WORLD_SIZE={world}", file=sys.stderr, flush=True)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("LGM_DIST_BACKEND", "nccl")     # "gloo": rehearse the N>1 path on one GPU
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if rank == 0:
        print(f"[bench] world={world} backend={backend} HSA_ENABLE_IPC_MODE_LEGACY="
              f"{os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '<unset>')} ({_IPC_ENV_SOURCE})", file=sys.stderr, flush=True)
    # LGM_DDP_FORCE=1 under a launcher with ONE rank: the process group comes up and FlatGradSync issues its collectives
    # although world == 1 - the one-rank RCCL rehearsal (tests/test_hip_rccl.py)
    forced = world == 1 and os.environ.get("LGM_DDP_FORCE", "0") == "1" and launch.launched()
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rendezvous or a first collective that cannot complete must END the run with the RCCL error text and a
        # non-zero exit code, not hang it: bounded process-group timeout + asynchronous error handling, and a probe
        # all-reduce right here, where a communicator set-up error has a clear message
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("LGM_DIST_TIMEOUT", "240")))
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=tmo)   # RCCL over xGMI
            else:
                dist.init_process_group(backend, timeout=tmo)
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)                   # first collective: communicator set-up errors surface HERE
            torch.cuda.synchronize()
            assert int(probe.item()) == world, f"all-reduce probe returned {probe.item()} on {world} ranks"
        except BaseException as e:
            print(f"[bench] rank {rank}: process group / first collective failed: {type(e).__name__}: {e}",
                  file=sys.stderr, flush=True)
            os._exit(3)

    single = world == 1 and not args.only and args.workload == "ddpm32" and args.batch is None
    line = run_workload(args.workload, args, dev, world, rank, args.steps, args.warmup, batch=args.batch)
    rc = 0
    if single and rank == 0:
        # The headline exists now: put it on record (stderr + gpurun_out/) BEFORE any secondary leg runs, so that a hard
        # fault, abort or hang in one of them cannot lose it; stdout still gets exactly ONE line, at the end.
        print("[bench] headline: " + json.dumps(line), file=sys.stderr, flush=True)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_headline.json"), "w") as f:
                f.write(json.dumps(line) + "\n")
        except OSError:
            pass
        from lgm_hip.lightning import is_device_error
        faulted = None          # name of the leg after which the GPU must not be used any more

        def leg(name, fn):
            """A secondary leg never takes the headline down with it.  A Python-level error is recorded and the next leg
            runs; a DEVICE error (hipError from the library, torch.AcceleratorError, RCCL) ends all GPU work of this
            process: the remaining legs are recorded as skipped and the run exits non-zero after printing its line."""
            nonlocal faulted
            if faulted is not None:
                return {"skipped": f"device error in leg '{faulted}'"}
            try:
                _release()
                return fn()
            except Exception as e:
                print(f"[bench] leg {name} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                if is_device_error(e):
                    faulted = name
                return {"error": f"{type(e).__name__}: {e}", "device_error": bool(is_device_error(e))}

        # ---- BASELINE's secondary configs, each measured the same way, then the per-rank batches of the
        # strong-scaled headline on this one GPU
        sec = {}
        for wl, st, wu in (("ddpm64", 15, 3), ("wgan_gp64", 60, 12), ("vqvae", 100, 10), ("vqvae_ema", 100, 10)):
            def run(wl=wl, st=st, wu=wu):
                r = run_workload(wl, args, dev, 1, 0, st, wu)
                return {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config",
                                          "roofline", "cpu_baseline") if k in r}
            sec[wl] = leg(wl, run)
        sec["ddpm64_sampling"] = leg("ddpm64_sampling", lambda: run_sampling(dev))
        sec["ddpm64_sampling_1000"] = leg("ddpm64_sampling_1000", lambda: run_sampling_ancestral(dev))
        line["secondary"] = sec
        proxy = {}
        # the proxies run what a RANK of an N > 1 job runs when its exchange overlaps the backward on RCCL: light F(4x4)
        # workgroups (csrc/winograd4.hip: wino4_use_light) and launch plans for 240 CUs (csrc/elementwise.hip: lgm_cu_budget) -
        # FlatGradSync sets both for such a rank (lgm_hip/lightning.py); the proxy legs set them the same way (and back)
        from lgm_hip import ops as _ops
        _ops.lib().lgm_wino4_set_light(1)
        _ops.lib().lgm_set_cu_margin(16)
        try:
            for b in (64, 32, 16):
                r = leg(f"proxy_b{b}", lambda b=b: run_workload("ddpm32", args, dev, 1, 0, 20, 5, batch=b, roofline=False,
                                                                 cpu=False))
                proxy[f"b{b}"] = r.get("ms_per_step") if isinstance(r, dict) else None
        finally:
            _ops.lib().lgm_wino4_set_light(-1)
            _ops.lib().lgm_set_cu_margin(-1)
        proxy["note"] = ("ms per step of the headline workload on ONE GPU at the per-rank batch of 2 / 4 / 8 GPUs "
                         "(global batch 128): compute between the gradient exchanges under strong scaling; kernel "
                         "selection of a rank whose exchange overlaps its backward on RCCL (light F(4x4) workgroups, launch plans "
                         "sized for 240 of the 256 CUs: lgm_set_cu_margin)")
        line["per_rank_proxy"] = proxy
        if faulted is not None:
            line["device_error_in"] = faulted
            rc = 4
    if rank == 0:
        print(json.dumps(line), flush=True)
    if rc != 0:
        sys.stdout.flush()
        os._exit(rc)             # after a device error: no destructors, no further GPU calls
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
