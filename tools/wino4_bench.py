"""Winograd F(4x4,3x3) (csrc/winograd4.hip) vs F(2x2,3x3): error of both against an fp64 reference at a small batch and
sustained timing at the UNet's large-map 3x3 layer shapes.
usage (GPU box): python tools/wino4_bench.py [B] [filter]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from lgm_hip import ops  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tools"))
from wino_bench import timeit, wino, wino_weights  # noqa: E402

SHAPES = [("64->64 @32", 64, 64, 32, 8), ("128->64 @32", 128, 64, 32, 4), ("64->64 @16", 64, 64, 16, 4),
          ("192->128 @16", 192, 128, 16, 2), ("128->128 @16", 128, 128, 16, 2), ("256->128 @16", 256, 128, 16, 1),
          ("64->64 @64", 64, 64, 64, 0), ("128->128 @8", 128, 128, 8, 4), ("384->256 @8", 384, 256, 8, 2),
          ("256->256 @8", 256, 256, 8, 2), ("512->256 @8", 512, 256, 8, 1)]


def wino4_weights(w):
    Np, _, Cp = w.shape
    uf = torch.empty(Np * Cp * 36, device=w.device)
    ub = torch.empty(Np * Cp * 36, device=w.device)
    tab = torch.tensor([[0, Np, Cp, 0, 0, 0]], dtype=torch.int64, device=w.device)
    ops.lib().lgm_wino4_weights(w.data_ptr(), uf.data_ptr(), ub.data_ptr(), tab.data_ptr(), 1, (Np // 32) * (Cp // 32),
                                ops.stream())
    return uf, ub


def wino4(yx, g, a, u, bias, res, out):
    L = ops.lib()
    n = L.lgm_conv3x3_wino4_workspace(ctypes.byref(g), yx)
    ws = ops.workspace(n, a.device) if n > 0 else None
    L.lgm_conv3x3_wino4(yx, ctypes.byref(g), a.data_ptr(), ops.pitch(a), u.data_ptr(), None if bias is None else bias.data_ptr(),
                        None if res is None else res.data_ptr(), 0 if res is None else ops.pitch(res), out.data_ptr(),
                        ops.pitch(out), None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4, ops.stream())


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    dev = torch.device("cuda", 0)
    tot = {"w2_xy": 0.0, "w4_xy": 0.0, "w2_yx": 0.0, "w4_yx": 0.0}
    print(f"{'shape':16s} | err64 xy: F2  F4   yx: F2  F4 | xy us: F(2x2)  F(4x4) | yx us: F(2x2)  F(4x4) | F4 TF(alg) xy yx")
    for name, ci, co, hw, cnt in SHAPES:
        if flt and flt not in name:
            continue
        gen = torch.Generator().manual_seed(ci * 1000 + co + hw)
        Bs = 8
        x = torch.randn(Bs, hw, hw, ci, generator=gen)
        y = torch.randn(Bs, hw, hw, co, generator=gen)
        w = torch.randn(co, 9, ci, generator=gen) * (1.0 / (3 * ci ** 0.5))
        bias = torch.randn(co, generator=gen)
        res = torch.randn(Bs, hw, hw, co, generator=gen)
        w4 = w.reshape(co, 3, 3, ci).permute(0, 3, 1, 2).double()
        ref_xy = F.conv2d(x.permute(0, 3, 1, 2).double(), w4, bias.double(), padding=1).permute(0, 2, 3, 1) + res.double()
        ref_yx = F.conv_transpose2d(y.permute(0, 3, 1, 2).double(), w4, None, padding=1).permute(0, 2, 3, 1)
        xd, yd, wd, bd, rd = (t.to(dev) for t in (x, y, w, bias, res))
        uf, ub = wino_weights(wd)
        uf4, ub4 = wino4_weights(wd)
        g = ops.make_geom(Bs, hw, hw, ci, co, 3, 3, 1, 1)
        o2, o4 = torch.empty_like(yd), torch.full_like(yd, float("nan"))
        wino(0, g, xd, uf, bd, rd, o2)
        wino4(0, g, xd, uf4, bd, rd, o4)
        g2, g4 = torch.empty_like(xd), torch.full_like(xd, float("nan"))
        wino(1, g, yd, ub, None, None, g2)
        wino4(1, g, yd, ub4, None, None, g4)
        err = lambda a, r: float((a.double().cpu() - r).abs().max() / r.abs().max())  # noqa: E731
        e = (err(o2, ref_xy), err(o4, ref_xy), err(g2, ref_yx), err(g4, ref_yx))
        if B > 0:
            x = torch.randn(B, hw, hw, ci, device=dev)
            y = torch.randn(B, hw, hw, co, device=dev)
            gx = torch.empty_like(x)
            g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
            fl = 2.0 * B * hw * hw * ci * co * 9
            t = (timeit(lambda: wino(0, g, x, uf, bd, None, y)), timeit(lambda: wino4(0, g, x, uf4, bd, None, y)),
                 timeit(lambda: wino(1, g, y, ub, None, None, gx)), timeit(lambda: wino4(1, g, y, ub4, None, None, gx)))
            for k, v in zip(("w2_xy", "w4_xy", "w2_yx", "w4_yx"), t):
                tot[k] += cnt * v
        else:
            t, fl = (1, 1, 1, 1), 0.0
        print(f"{name:16s} | {e[0]:.1e} {e[1]:.1e}   {e[2]:.1e} {e[3]:.1e} | {t[0] * 1e3:8.1f} {t[1] * 1e3:8.1f} | "
              f"{t[2] * 1e3:8.1f} {t[3] * 1e3:8.1f} | {fl / t[1] / 1e9:6.1f} {fl / t[3] / 1e9:6.1f}", flush=True)
    print("TOTAL ms/step: " + "  ".join(f"{k} {tot[k]:.3f}" for k in tot))


if __name__ == "__main__":
    main()
