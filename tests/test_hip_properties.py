"""GPU: size-independent properties at BASELINE.json's FULL sizes (B=128, 32x32 / dim 64): linearity and
adjoint identities of the convolution family, and bit-reproducibility (the reference trainer runs
deterministic=True) of the whole training step.  (The direct B=128 comparison with the CPU oracle — a few
seconds of CPU work — lives in tests/test_hip_timed_path.py.)"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("case", [(128, 32, 64, 64, 3, 1), (128, 16, 192, 128, 3, 1), (128, 4, 512, 512, 3, 1),
                                  (128, 32, 64, 384, 1, 0), (128, 8, 256, 128, 1, 0)])
def test_conv_linearity_and_adjoints_full_size(dev, case):
    from lgm_hip import ops
    B, S, ci, co, k, pad = case
    g = torch.Generator(device="cpu").manual_seed(sum(case))
    geom = ops.make_geom(B, S, S, ci, co, k, k, 1, pad)
    x1 = torch.randn(B, S, S, ci, generator=g).to(dev)
    x2 = torch.randn(B, S, S, ci, generator=g).to(dev)
    y1 = torch.randn(B, S, S, co, generator=g).to(dev)
    w = (torch.randn(co, k * k, ci, generator=g) * 0.05).to(dev)
    out = lambda: torch.empty(B, S, S, co, device=dev)  # noqa: E731
    c1, c2, c12 = out(), out(), out()
    ops.conv_xy(geom, x1, w.data_ptr(), None, None, c1)
    ops.conv_xy(geom, x2, w.data_ptr(), None, None, c2)
    ops.conv_xy(geom, (0.7 * x1 - 1.3 * x2).contiguous(), w.data_ptr(), None, None, c12)
    ref = 0.7 * c1 - 1.3 * c2
    assert float((c12 - ref).norm() / ref.norm()) < 1e-5                       # linearity in x
    gx = torch.empty_like(x1)
    ops.conv_yx(geom, y1, w.data_ptr(), None, None, gx)
    lhs, rhs = dot(c1, y1), dot(x1, gx)                                          # <W x, y> == <x, W^T y>
    assert abs(lhs - rhs) / max(abs(lhs), 1.0) < 1e-4
    gw = torch.empty_like(w)
    ops.conv_wgrad(geom, y1, x1, gw.data_ptr(), 0.0)
    lhs2, rhs2 = dot(gw, w), dot(c1, y1)                                         # <dW(y, x), W> == <W x, y>
    assert abs(lhs2 - rhs2) / max(abs(rhs2), 1.0) < 1e-4
    gw2 = torch.empty_like(w)
    ops.conv_wgrad(geom, y1, x1, gw2.data_ptr(), 0.0)
    assert torch.equal(gw, gw2)                                                  # deterministic split-K


@pytest.mark.parametrize("case", [(128, 64, 4, 64, 4, 2, 1), (128, 32, 64, 128, 4, 2, 1), (64, 8, 256, 512, 4, 2, 1),
                                  (128, 64, 4, 128, 4, 2, 1), (16, 28, 4, 64, 4, 2, 1)])
def test_strided_conv_adjoints_full_size(dev, case):
    """4x4 / stride-2 family of the DCGAN critic and generator at full size: the residue-class decomposed
    input gradient (and its small-N form at the image end) is the exact adjoint of the forward, and the
    weight gradient is the adjoint in W; deferred + batched slab reduction is bit-identical to the immediate one."""
    from lgm_hip import ops
    B, S, ci, co, k, st, pad = case
    g = torch.Generator(device="cpu").manual_seed(sum(case))
    geom = ops.make_geom(B, S, S, ci, co, k, k, st, pad)
    So = geom.Ho
    x1 = torch.randn(B, S, S, ci, generator=g).to(dev)
    y1 = torch.randn(B, So, So, co, generator=g).to(dev)
    w = (torch.randn(co, k * k, ci, generator=g) * 0.05).to(dev)
    c1 = torch.empty(B, So, So, co, device=dev)
    ops.conv_xy(geom, x1, w.data_ptr(), None, None, c1)
    gx = torch.empty_like(x1)
    ops.conv_yx(geom, y1, w.data_ptr(), None, None, gx)
    lhs, rhs = dot(c1, y1), dot(x1, gx)
    assert abs(lhs - rhs) / max(abs(lhs), 1.0) < 1e-4
    gw = torch.empty_like(w)
    ops.conv_wgrad(geom, y1, x1, gw.data_ptr(), 0.0)
    lhs2 = dot(gw, w)
    assert abs(lhs2 - lhs) / max(abs(lhs), 1.0) < 1e-4
    rows = []
    gw2 = torch.full_like(w, 7.0)
    ops.conv_wgrad(geom, y1, x1, gw2.data_ptr(), 0.0, None, defer=rows)
    ops.wgrad_reduce_batch(rows, dev)
    assert torch.equal(gw, gw2)


def test_full_size_training_step_is_bit_reproducible(dev):
    """Two runs of the BASELINE workload (dim 64, 32x32, B=128) from the same state and inputs give
    bit-identical loss and gradients (fixed-order reductions, no float atomics)."""
    from models.generative.diffusion.ddpm import GaussianDiffusion, Unet
    torch.manual_seed(0)
    net = Unet(dim=64, channels=3)
    gd = GaussianDiffusion(net, img_size=32).to(dev)
    net.prepare_hip(dev)
    g = torch.Generator().manual_seed(1)
    img = torch.rand(128, 3, 32, 32, generator=g).to(dev)
    noise = torch.randn(128, 3, 32, 32, generator=g).to(dev)
    t = torch.randint(0, 1000, (128,), generator=g).to(dev)
    runs = []
    for _ in range(2):
        net._flat.zero_grad()
        loss = gd.p_losses(img, t, noise, _normalize=True)
        loss.backward()
        runs.append((loss.detach().clone(), net._flat.grad.clone()))
    assert torch.equal(runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1])
    assert torch.isfinite(runs[0][1]).all() and float(runs[0][1].abs().max()) > 0


@pytest.mark.parametrize("cfg", ["diffusion/ddpm.json", "diffusion/ddim.json", "diffusion/ddpm_64.json",
                                 "gan/wgan_gp.json", "gan/wgan_gp_celeba.json", "gan/wgan_cp.json",
                                 "gan/dcgan_mnist.json", "gan/lsgan.json", "gan/r1gan.json", "vae/vqvae.json",
                                 "vae/vqvae_ema.json"])
def test_train_entry_runs_every_hot_path_config_on_the_gpu(cfg, tmp_path):
    """The reference's CLI surface: python train.py --config_path <cfg> --max_steps N on the HIP engine, then
    resume from the written Lightning-layout checkpoint for a few more steps."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "lightning-generative-models_amd")
    exp = f"pytest_gpu_{cfg.replace('/', '_').replace('.json', '')}"
    cmd = [sys.executable, os.path.join(pkg, "train.py"), "--config_path", os.path.join(pkg, "configs", cfg),
           "--max_steps", "6", "--experiment_name", exp]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    name = __import__("json").load(open(os.path.join(pkg, "configs", cfg)))["model"]["name"]
    ck = os.path.join(pkg, "experiments", name, exp, "last.ckpt")
    sd = torch.load(ck, map_location="cpu", weights_only=False)
    assert sd["global_step"] == 6 and len(sd["optimizer_states"]) >= 1      # 6 is a multiple of the 2-steps-per-batch GANs
    for v in sd["state_dict"].values():
        if v.is_floating_point():
            assert torch.isfinite(v).all()
    r = subprocess.run(cmd[:-4] + ["--max_steps", "9", "--experiment_name", exp, "--ckpt_path", ck], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # global_step counts optimizer steps (two per batch for the D/G-alternating GANs): first value >= max_steps
    assert torch.load(ck, map_location="cpu", weights_only=False)["global_step"] in (9, 10)
