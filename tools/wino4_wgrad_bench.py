"""Winograd F(4x4,3x3) weight gradient (csrc/winograd4_wgrad.hip) vs the F(2x2) kernel: error against float64 autograd at a
small batch, and kernel + slab reduction timed at the UNet's large-map layer shapes.
usage (GPU box): python tools/wino4_wgrad_bench.py [B] [filter]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from lgm_hip import ops  # noqa: E402
from wino_bench import timeit  # noqa: E402

SHAPES = [("64->64 @32", 64, 64, 32, 9), ("128->64 @32", 128, 64, 32, 4), ("64->64 @16", 64, 64, 16, 4),
          ("192->128 @16", 192, 128, 16, 2), ("128->128 @16", 128, 128, 16, 2), ("256->128 @16", 256, 128, 16, 1),
          ("64->64 @64", 64, 64, 64, 0)]
_WS = {}


def wgrad4(g, y, x, gw, gb, beta=0.0):
    L = ops.lib()
    n = L.lgm_conv3x3_wino4_wgrad_workspace(ctypes.byref(g))
    key = (gw.data_ptr(), n)
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = torch.empty(n // 4 + 16, device=y.device)
    desc = (ctypes.c_int64 * 8)()
    L.lgm_conv3x3_wino4_wgrad(ctypes.byref(g), y.data_ptr(), ops.pitch(y), x.data_ptr(), ops.pitch(x), gw.data_ptr(),
                              None if gb is None else gb.data_ptr(), beta, ws.data_ptr(), ws.numel() * 4,
                              ctypes.addressof(desc), ops.stream())
    ops.wgrad_reduce_batch([tuple(desc)], y.device)


def wgrad2(g, y, x, gw, gb, beta=0.0):
    rows = []
    ops.conv_wgrad(g, y, x, gw.data_ptr(), beta, None if gb is None else gb.data_ptr(), defer=rows)
    ops.wgrad_reduce_batch(rows, y.device)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    dev = torch.device("cuda", 0)
    tot = {"F2": 0.0, "F4": 0.0}
    print(f"{'shape':16s} | err64 gw: F2  F4   gb: F2  F4 | us (kernel + reduce): F(2x2)  F(4x4)")
    for name, ci, co, hw, cnt in SHAPES:
        if flt and flt not in name:
            continue
        gen = torch.Generator().manual_seed(ci * 1000 + co + hw)
        Bs = 4
        x = torch.randn(Bs, hw, hw, ci, generator=gen)
        y = torch.randn(Bs, hw, hw, co, generator=gen)
        w0 = torch.zeros(co, ci, 3, 3, dtype=torch.double, requires_grad=True)
        out = F.conv2d(x.permute(0, 3, 1, 2).double(), w0, None, padding=1)
        gw_ref, = torch.autograd.grad(out, w0, y.permute(0, 3, 1, 2).double())
        gw_ref = gw_ref.permute(0, 2, 3, 1).reshape(co, 9, ci)
        gb_ref = y.double().sum((0, 1, 2))
        xd, yd = x.to(dev), y.to(dev)
        g = ops.make_geom(Bs, hw, hw, ci, co, 3, 3, 1, 1)
        assert ops.lib().lgm_conv3x3_wino4_wgrad_supported(ctypes.byref(g)) == 1
        res = []
        for fn in (wgrad2, wgrad4):
            gw = torch.full((co, 9, ci), float("nan"), device=dev)
            gb = torch.full((co,), float("nan"), device=dev)
            fn(g, yd, xd, gw, gb)
            res.append((gw, gb))
        err = lambda a, r: float((a.double().cpu() - r).abs().max() / r.abs().max())  # noqa: E731
        e = (err(res[0][0], gw_ref), err(res[1][0], gw_ref), err(res[0][1], gb_ref), err(res[1][1], gb_ref))
        t = (0.0, 0.0)
        if B > 0:
            x = torch.randn(B, hw, hw, ci, device=dev)
            y = torch.randn(B, hw, hw, co, device=dev)
            g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
            gw = torch.zeros(co, 9, ci, device=dev)
            gb = torch.zeros(co, device=dev)
            t = (timeit(lambda: wgrad2(g, y, x, gw, gb), 50), timeit(lambda: wgrad4(g, y, x, gw, gb), 50))
            tot["F2"] += cnt * t[0]
            tot["F4"] += cnt * t[1]
        print(f"{name:16s} | {e[0]:.1e} {e[1]:.1e}   {e[2]:.1e} {e[3]:.1e} | {t[0] * 1e3:8.1f} {t[1] * 1e3:8.1f}", flush=True)
    print("TOTAL ms/step: " + "  ".join(f"{k} {tot[k]:.3f}" for k in tot))


if __name__ == "__main__":
    main()
