"""Diagnostic: WGAN-GP 64x64 at B = 128 - per-parameter gradient distances (HIP vs fp32 oracle vs fp64 oracle), norms and
per-channel detail for the worst parameter.  usage (GPU box): python tools/wgan_b128_diag.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from oracle import gan as OG  # noqa: E402
import test_hip_gan as T  # noqa: E402

dev = torch.device("cuda", 0)
img_size, ch, latent = 64, 3, 100
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = T._load_wgan(img_size, ch, latent, dev)
g = torch.Generator().manual_seed(128)
x = torch.rand(B, ch, img_size, img_size, generator=g) * 2 - 1
z = torch.randn(B, latent, 1, 1, generator=g)
alpha = torch.rand(B, 1, 1, 1, generator=g)
G, D = OG.gan_init(img_size, ch, latent, seed=21)


def oracle_run(dt):
    Gp = {k: v.to(dt).requires_grad_(True) for k, v in G.items()}
    Dp = {k: v.to(dt).requires_grad_(True) for k, v in D.items()}
    xh = OG.generator(Gp, z.to(dt), img_size, ch)
    ld = OG.wgan_d_loss(Dp, x.to(dt), xh.detach(), alpha.to(dt), 10.0, img_size)
    dg = torch.autograd.grad(ld["d_loss"], list(Dp.values()))
    gl = OG.wgan_g_loss({k: v.detach() for k, v in Dp.items()}, xh, img_size)
    gg = torch.autograd.grad(gl, list(Gp.values()))
    return dict(zip(Dp, dg)), dict(zip(Gp, gg))


d32, g32 = oracle_run(torch.float32)
d64, g64 = oracle_run(torch.float64)
x_hat = m.G(z.to(dev))
ld = m._calculate_d_loss(x.to(dev), x_hat, alpha=alpha.to(dev))
d_opt, g_opt = m.configure_optimizers()[0]
d_opt.zero_grad()
ld["d_loss"].backward()


def rel(a, b):
    return float((a.double().cpu() - b.double()).norm() / b.double().norm())


def nrm(a, b):
    return abs(float(a.double().norm()) - float(b.double().norm())) / float(b.double().norm())


def report(net, r32, r64, what):
    for n, p in net.named_parameters():
        gr = p.grad
        print(f"{what} {n:18s} tensor: hip-64 {rel(gr, r64[n]):.2e} ref32-64 {rel(r32[n], r64[n]):.2e} | norm: hip-64 "
              f"{nrm(gr.cpu(), r64[n]):.2e} ref32-64 {nrm(r32[n], r64[n]):.2e} hip-ref32 {nrm(gr.cpu(), r32[n]):.2e}")


report(m.D, d32, d64, "D")
g_opt.zero_grad()
gl = m._calculate_g_loss(m.G(z.to(dev)))["g_loss"]
gl.backward()
report(m.G, g32, g64, "G")
p = dict(m.G.named_parameters())["model.1.1.weight"].grad.double().cpu()
r = g64["model.1.1.weight"]
e = (p - r)
print("G model.1.1.weight: signed mean rel err", float((e / r.abs().clamp_min(1e-12)).mean()), "corr", float((e * r).sum() / (r * r).sum()))
print("largest |g|:", r.abs().topk(5).values.tolist(), "their errs", e[r.abs().topk(5).indices].tolist())
