"""Per-shape timing of the implicit-GEMM convolution kernels at the DDPM UNet's layer shapes.
usage (GPU box): python tools/conv_bench.py [B] [filter-substring]
Prints, per distinct conv shape: count per UNet forward, FLOPs, and achieved TFLOP/s of
xy (fwd) / yx (dgrad) / wgrad, plus the FLOP-weighted totals for one training step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from lgm_hip import ops  # noqa: E402

# (name, Cin, Cout, HW, k, pad, count)
UNET32 = [
    ("c3 64->64 @32", 64, 64, 32, 3, 1, 8),
    ("c3 128->64 @32", 128, 64, 32, 3, 1, 4),
    ("c3 64->64 @16", 64, 64, 16, 3, 1, 4),
    ("c3 192->128 @16", 192, 128, 16, 3, 1, 2),
    ("c3 128->128 @16", 128, 128, 16, 3, 1, 2),
    ("c3 256->128 @16", 256, 128, 16, 3, 1, 1),
    ("c3 128->128 @8", 128, 128, 8, 3, 1, 4),
    ("c3 384->256 @8", 384, 256, 8, 3, 1, 2),
    ("c3 256->256 @8", 256, 256, 8, 3, 1, 2),
    ("c3 512->256 @8", 512, 256, 8, 3, 1, 1),
    ("c3 256->256 @4", 256, 256, 4, 3, 1, 4),
    ("c3 512->512 @4", 512, 512, 4, 3, 1, 6),
    ("c3 768->512 @4", 768, 512, 4, 3, 1, 2),
    ("c3 256->512 @4", 256, 512, 4, 3, 1, 1),
    ("c1 128->64 @32", 128, 64, 32, 1, 0, 3),
    ("c1 192->128 @16", 192, 128, 16, 1, 0, 2),
    ("c1 384->256 @8", 384, 256, 8, 1, 0, 2),
    ("c1 768->512 @4", 768, 512, 4, 1, 0, 2),
    ("c1 64->384 @32", 64, 384, 32, 1, 0, 2),
    ("c1 128->384 @16", 128, 384, 16, 1, 0, 2),
    ("c1 256->384 @8", 256, 384, 8, 1, 0, 2),
    ("c1 512->384 @4", 512, 384, 4, 1, 0, 3),
    ("c1 128->64 @32 (to_out)", 128, 64, 32, 1, 0, 2),
    ("c1 128->128 @16 (to_out)", 128, 128, 16, 1, 0, 2),
    ("c1 128->256 @8 (to_out)", 128, 256, 8, 1, 0, 2),
    ("c1 128->512 @4 (to_out)", 128, 512, 4, 1, 0, 3),
    ("c1 256->64 @16 (down)", 256, 64, 16, 1, 0, 1),
    ("c1 256->128 @8 (down)", 256, 128, 8, 1, 0, 1),
    ("c1 512->256 @4 (down)", 512, 256, 4, 1, 0, 1),
    ("c7 4->64 @32", 4, 64, 32, 7, 3, 1),
    ("c1 64->4 @32", 64, 4, 32, 1, 0, 1),
]


def timeit(fn, iters=100):
    # long enough (and warmed) for the GPU to reach its sustained clock: 10-iteration bursts run at
    # ~2.07 GHz instead of 2.39 GHz and under-report by ~14 %
    for _ in range(30):
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
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    dev = torch.device("cuda", 0)
    tot = {"xy": [0.0, 0.0], "yx": [0.0, 0.0], "wg": [0.0, 0.0]}
    print(f"{'shape':28s} {'cnt':>3s} {'GFLOP':>8s} | {'xy us':>8s} {'TF':>6s} | {'yx us':>8s} {'TF':>6s} | {'wg us':>8s} {'TF':>6s}")
    for name, ci, co, hw, k, pad, cnt in UNET32:
        if flt and flt not in name:
            continue
        g = ops.make_geom(B, hw, hw, ci, co, k, k, 1, pad)
        x = torch.randn(B, hw, hw, ci, device=dev)
        y = torch.randn(B, hw, hw, co, device=dev)
        w = torch.randn(co, k * k, ci, device=dev) * 0.05
        gw = torch.zeros_like(w)
        wt = torch.zeros_like(w)       # [Cw][T][Nw] copy, as FlatParams.refresh_transposed() maintains it
        tbl = torch.tensor([[0, co, k * k, ci, 0]], dtype=torch.int32, device=dev)
        ops.lib().lgm_transpose_weights(w.data_ptr(), wt.data_ptr(), tbl.data_ptr(), 1,
                                        ((co + 31) // 32) * ((ci + 31) // 32) * k * k, ops.stream())
        gb = torch.zeros(co, device=dev)
        gx = torch.empty_like(x)
        fl = 2.0 * B * hw * hw * ci * co * k * k
        t_xy = timeit(lambda: ops.conv_xy(g, x, w.data_ptr(), gb.data_ptr(), None, y))
        t_yx = timeit(lambda: ops.conv_yx(g, y, w.data_ptr(), None, None, gx, wt.data_ptr()))
        t_wg = timeit(lambda: ops.conv_wgrad(g, y, x, gw.data_ptr(), 0.0, gb.data_ptr()))
        for key, t in (("xy", t_xy), ("yx", t_yx), ("wg", t_wg)):
            tot[key][0] += cnt * t
            tot[key][1] += cnt * fl
        tf = lambda t: fl / (t * 1e-3) / 1e12  # noqa: E731
        print(f"{name:28s} {cnt:3d} {fl / 1e9:8.2f} | {t_xy * 1e3:8.1f} {tf(t_xy):6.1f} | {t_yx * 1e3:8.1f} {tf(t_yx):6.1f} | "
              f"{t_wg * 1e3:8.1f} {tf(t_wg):6.1f}", flush=True)
    for key in tot:
        ms, fl = tot[key]
        if ms > 0:
            print(f"TOTAL {key}: {ms:.3f} ms/step  {fl / (ms * 1e-3) / 1e12:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
