"""bench.py — BASELINE.json metric: training images/sec, DDPM UNet 32x32, global batch 128.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = the full optimiser step of the reference loop on a synthetic
batch already resident in HBM: DDPM.training_step (t ~ randint, noise ~ randn on device,
q_sample, UNet fwd, weighted MSE) -> loss.backward() (hand-written HIP backward) -> gradient
all-reduce over RCCL when N > 1 -> fused Adam -> EMA update (every 10th step).  The periodic
in-training sampling (reference F11, every 1000 steps) is outside the window.  Strong scaling:
the global batch stays 128 (reference DataModule divides the config batch by the GPU count).

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (convolution kernel family, fp32 MFMA
peak 157.3 TFLOP/s; per-launch durations from HIP events on the launch stream in one extra
instrumented step right after the timed region) and "cpu_baseline" (the CPU oracle, same
workload at a bounded batch, on this host's cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix == fp32 vector peak
GLOBAL_BATCH = 128
IMG = 32
DIM = 64


def cpu_baseline(batch=32, warmup=1, min_seconds=12.0, max_steps=200):
    """The CPU oracle (validated against the reference by tests/golden) on this host's cores: whole
    training steps (forward + backward + Adam) until ``min_seconds`` of timed work have accumulated."""
    from oracle import diffusion as OD
    torch.manual_seed(10)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        avail = os.cpu_count() or 1
    nthreads = max(1, min(avail, 16))   # a 1-GPU box owns a 16-core share of the host
    torch.set_num_threads(nthreads)
    P = {k: v.requires_grad_(True) for k, v in OD.unet_init(dim=DIM, channels=3, seed=0).items()}
    bufs = OD.diffusion_buffers(1000)
    opt = torch.optim.Adam(list(P.values()), lr=2e-5, betas=(0.9, 0.99))
    x = torch.rand(batch, 3, IMG, IMG) * 2 - 1
    times = []
    i = 0
    while i < warmup or (sum(times) < min_seconds and len(times) < max_steps):
        if i < warmup or len(times) % 10 == 0:
            print(f"[bench] cpu_baseline step {i} ({nthreads} threads, {sum(times):.1f}s timed)", file=sys.stderr,
                  flush=True)
        t0 = time.perf_counter()
        t = torch.randint(0, 1000, (batch,))
        noise = torch.randn_like(x)
        loss = OD.diffusion_forward(P, bufs, x, t, noise, dim=DIM)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if i >= warmup:
            times.append(time.perf_counter() - t0)
        i += 1
    total = sum(times)
    return {"value": round(batch * len(times) / total, 2), "unit": "images/s", "cores": nthreads, "kind": "port",
            "sample": f"oracle fwd+bwd+Adam, B={batch}, 3x{IMG}x{IMG}: {len(times)} steps = {total:.1f} s of CPU work "
                      f"after {warmup} warm-up"}


def torch_gpu_baseline(dev, batch=GLOBAL_BATCH, warmup=5, steps=10):
    """Baseline leg, optional (--torch-gpu-baseline): the same oracle (the reference's arithmetic as plain
    torch ops, i.e. what the reference's modules execute) with its tensors on the MI355X — PyTorch-ROCm eager
    kernels (MIOpen / rocBLAS / ATen) + torch.optim.Adam.  A reported comparison point, not the target."""
    from oracle import diffusion as OD
    torch.manual_seed(10)
    P = {k: v.to(dev).requires_grad_(True) for k, v in OD.unet_init(dim=DIM, channels=3, seed=0).items()}
    bufs = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in OD.diffusion_buffers(1000).items()}
    opt = torch.optim.Adam(list(P.values()), lr=2e-5, betas=(0.9, 0.99))
    x = (torch.rand(batch, 3, IMG, IMG) * 2 - 1).to(dev)

    def one():
        t = torch.randint(0, 1000, (batch,), device=dev)
        noise = torch.randn_like(x)
        loss = OD.diffusion_forward(P, bufs, x, t, noise, dim=DIM)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch from Python (no HIP-graph replay)")
    ap.add_argument("--batch", type=int, default=GLOBAL_BATCH, help="global batch (default 128 = the metric)")
    ap.add_argument("--torch-gpu-baseline", action="store_true",
                    help="also time the oracle's plain torch ops on the GPU (PyTorch-ROCm eager) as a comparison point")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, "--gpus must equal WORLD_SIZE under torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("LGM_DIST_BACKEND", "nccl")     # "gloo": rehearse the N>1 path on one GPU
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from lgm_hip import ops
    from lgm_hip.lightning import MiniTrainer
    from models.generative.diffusion.ddpm import DDPM

    torch.manual_seed(10)                         # reference train.py:20 seeds every rank identically
    assert args.batch % world == 0
    per_gpu = args.batch // world
    model = DDPM(img_channels=3, img_size=IMG, dim=DIM, diffusion_timesteps=1000, sampling_timesteps=None,
                 lr=2e-5, betas=(0.9, 0.99), ema_update_every=10, ema_decay=0.995)   # configs/diffusion/ddpm.json
    model.sample_every = 0
    model.to(dev)
    model.prepare_hip(dev)
    model.train()
    opt = model.configure_optimizers()
    trainer = MiniTrainer()
    from lgm_hip.lightning import FlatGradSync
    unet = model.ema.online_model.model
    sync = FlatGradSync(unet._flat) if world > 1 else None
    unet.grad_sync = sync          # buckets are all-reduced (async) while the backward is still running
    if sync is not None:
        opt.grad_scale = sync.grad_scale
    g = torch.Generator(device="cpu").manual_seed(10 + rank)
    x = (torch.rand(per_gpu, 3, IMG, IMG, generator=g) * 2 - 1).to(dev)
    y = torch.zeros(per_gpu, dtype=torch.long, device=dev)
    batch = (x, y)

    def step(i):
        loss = model.training_step(batch)
        loss.backward()
        if sync is not None:
            sync.finish()
        opt.step()
        opt.zero_grad()
        model.on_train_batch_end(None, batch, i)
        return loss

    eager_step = step
    graphed = None
    if not args.no_graph:
        try:
            from lgm_hip.graph import GraphedDDPMStep
            graphed = GraphedDDPMStep(model, opt, x, sync)
            step = graphed.step
        except Exception as e:  # capture is an optimisation: fall back to eager launches
            unet.grad_sync = sync
            print(f"[bench] rank {rank}: HIP-graph capture unavailable ({type(e).__name__}: {e}); eager launches",
                  file=sys.stderr, flush=True)
    tw = time.perf_counter()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if rank == 0:
        print(f"[bench] warm-up done: {args.warmup} steps in {time.perf_counter() - tw:.2f}s", file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    final_loss = float(loss.item())
    if rank == 0:
        print(f"[bench] timed region: {args.steps} steps in {elapsed:.3f}s "
              f"({args.batch * args.steps / elapsed:.1f} img/s)", file=sys.stderr, flush=True)

    # ---- roofline leg: one extra instrumented step, per-launch HIP events on the launch stream
    unet.grad_sync = sync
    ops.TIMER = ops.KernelTimer()
    eager_step(args.warmup + args.steps)
    summ = ops.TIMER.summary()
    ops.TIMER = None

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * args.steps / elapsed
        dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
        name, d = dom
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        conv_ms = sum(v["ms"] for v in summ.values())
        conv_fl = sum(v["flops"] for v in summ.values())
        # HBM-side bytes per launch of the dominant family: PMC counters cannot be read from inside this
        # process, so the figure comes from the committed rocprofv3 --pmc summary of this same command
        # (profiles/r01_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 correction)
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc) and args.batch == GLOBAL_BATCH and world == 1:
            with open(pmc) as fh:
                t = json.load(fh).get(name)
            if t:
                traffic, traffic_src = t["traffic_bytes_per_launch"], "profiles/r01_pmc_traffic.json"
        bytes_per_img = 69.6e6 + 1.472e9 / per_gpu        # algorithmic HBM bytes per image
        roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(d.get("bytes", 0) / max(d["launches"], 1)) or None,
                "hbm_frac_of_8TBps": round(value / world * bytes_per_img / 8.0e12, 4),
                "launches_per_step": d["launches"], "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2),
                "family": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                               "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in summ.items()},
                "conv_family_ms_per_step": round(conv_ms, 3),
                "conv_family_tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                "step_flop_frac_of_peak": round(value * 10.95e9 / (FP32_MFMA_PEAK_TFLOPS * 1e12 * world), 4)}
        line = {"metric": "training images/sec (DDPM UNet 32x32, bs=128)", "value": round(value, 2),
                "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32" if not ops.B3 else "f32 (3x3 conv fwd/dgrad: opt-in bf16x3 split MFMA, fp32-level error)",
                "data": "synthetic",
                "config": {"workload": "configs/diffusion/ddpm.json UNet dim=64, 3x32x32 synthetic NCHW fp32, "
                                       "training_step+backward+Adam+EMA", "global_batch": args.batch,
                           "per_gpu_batch": per_gpu, "parallelism": f"dp{world}", "final_loss": round(final_loss, 5),
                           "launch": "hipGraph replay (2 graphs/step)" if graphed is not None else "eager",
                           "conv_mode": "bf16x3 (LGM_CONV_MODE)" if ops.B3 else "fp32 MFMA"},
                "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        if world == 1 and args.torch_gpu_baseline:
            line["torch_gpu_baseline"] = torch_gpu_baseline(dev, args.batch)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
