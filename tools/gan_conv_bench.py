"""Per-shape timing of the generic implicit-GEMM kernels at the DCGAN generator / critic layer shapes
(4x4 stride-2 convolutions and transposed convolutions).  usage (GPU box): python tools/gan_conv_bench.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from lgm_hip import ops  # noqa: E402

# critic convs: (name, Cin, Cout, H_in, k, stride, pad)
SHAPES = [("D 3->64 @64", 4, 64, 64, 4, 2, 1), ("D 64->128 @32", 64, 128, 32, 4, 2, 1),
          ("D 128->256 @16", 128, 256, 16, 4, 2, 1), ("D 256->512 @8", 256, 512, 8, 4, 2, 1),
          ("G 1024<-512 @4->8 (as conv 512->1024 @8)", 512, 1024, 8, 4, 2, 1),
          ("G 512<-256 @8->16", 256, 512, 16, 4, 2, 1), ("G 256<-128 @16->32", 128, 256, 32, 4, 2, 1),
          ("G 128<-3 @32->64", 4, 128, 64, 4, 2, 1)]


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda", 0)
    print(f"{'shape':44s} {'GFLOP':>7s} | {'xy us':>8s} {'TF':>6s} | {'yx us':>8s} {'TF':>6s} | {'wg us':>8s} {'TF':>6s}")
    for name, ci, co, h, k, st, pad in SHAPES:
        g = ops.make_geom(B, h, h, ci, co, k, k, st, pad)
        x = torch.randn(B, h, h, ci, device=dev)
        y = torch.randn(B, g.Ho, g.Wo, co, device=dev)
        w = torch.randn(co, k * k, ci, device=dev) * 0.05
        gw = torch.zeros_like(w)
        gx = torch.empty_like(x)
        fl = 2.0 * B * g.Ho * g.Wo * ci * co * k * k
        t_xy = timeit(lambda: ops.conv_xy(g, x, w.data_ptr(), None, None, y))
        t_yx = timeit(lambda: ops.conv_yx(g, y, w.data_ptr(), None, None, gx))
        t_wg = timeit(lambda: ops.conv_wgrad(g, y, x, gw.data_ptr(), 0.0, None))
        tf = lambda t: fl / (t * 1e-3) / 1e12  # noqa: E731
        print(f"{name:44s} {fl / 1e9:7.2f} | {t_xy * 1e3:8.1f} {tf(t_xy):6.1f} | {t_yx * 1e3:8.1f} {tf(t_yx):6.1f} | "
              f"{t_wg * 1e3:8.1f} {tf(t_wg):6.1f}", flush=True)


if __name__ == "__main__":
    main()
