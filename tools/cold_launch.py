"""What a COLD launch costs: the same convolution timed back to back (operands and code warm in L2) and behind a pass that
evicts the L2s and the Infinity Cache (a 512 MB fill), as every launch inside the training step finds the chip.
usage (GPU box): python tools/cold_launch.py [B]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from lgm_hip import ops  # noqa: E402
from wino_bench import timeit, wino, wino_weights  # noqa: E402
from wino4_bench import wino4, wino4_weights  # noqa: E402
from wino4l_bench import wino4l  # noqa: E402

SHAPES = [("64->64 @32", 64, 64, 32), ("128->64 @32", 128, 64, 32), ("192->128 @16", 192, 128, 16),
          ("256->256 @8", 256, 256, 8), ("512->512 @4", 512, 512, 4)]


def cold(fn, flush, n=12):
    tot = 0.0
    for _ in range(n):
        flush()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda", 0)
    big = torch.empty(128 << 20, device=dev)          # 512 MB

    def flush():
        big.fill_(1.0)

    def nothing():
        pass
    print(f"B = {B}: us per launch, warm (back to back) / cold (behind a 512 MB fill); event overhead alone: "
          f"{cold(nothing, flush):.1f} us")
    for name, ci, co, hw in SHAPES:
        g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
        x = torch.randn(B, hw, hw, ci, device=dev)
        y = torch.randn(B, hw, hw, co, device=dev)
        w = torch.randn(co, 9, ci, device=dev) * (1.0 / (3 * ci ** 0.5))
        bd = torch.randn(co, device=dev)
        uf, ub = wino_weights(w)
        uf4, ub4 = wino4_weights(w)
        L = ops.lib()
        rows = [("F(2x2)", lambda: wino(0, g, x, uf, bd, None, y))]
        L.lgm_wino4_set_light(0)
        if L.lgm_conv3x3_wino4_supported(ctypes.byref(g), 0) == 1:
            rows.append(("F(4x4) 32-tile", lambda: wino4(0, g, x, uf4, bd, None, y)))
        if L.lgm_conv3x3_wino4l_supported(ctypes.byref(g), 0) == 1:
            rows.append(("F(4x4) light", lambda: wino4l(0, g, x, uf4, bd, None, y)))
        for tag, fn in rows:
            print(f"  {name:14s} {tag:15s} warm {timeit(fn) * 1e3:7.1f}   cold {cold(fn, flush):7.1f}", flush=True)


if __name__ == "__main__":
    main()
