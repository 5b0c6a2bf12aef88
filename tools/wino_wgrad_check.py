"""Winograd weight gradient vs fp64 and vs the direct kernel (LGM_NO_WINO=1 in a second process for timing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from lgm_hip import ops
from wino_bench import SHAPES, timeit
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tot = 0.0
for name, ci, co, hw, cnt in SHAPES:
    if flt and flt not in name:
        continue
    gen = torch.Generator().manual_seed(ci + co + hw)
    Bs = 16
    x = torch.randn(Bs, hw, hw, ci, generator=gen)
    y = torch.randn(Bs, hw, hw, co, generator=gen)
    xr = x.permute(0, 3, 1, 2).double().requires_grad_(False)
    w0 = torch.zeros(co, ci, 3, 3, dtype=torch.double, requires_grad=True)
    out = F.conv2d(xr, w0, torch.zeros(co, dtype=torch.double), padding=1)
    gw_ref, = torch.autograd.grad(out, w0, y.permute(0, 3, 1, 2).double())
    gw_ref = gw_ref.permute(0, 2, 3, 1).reshape(co, 9, ci)
    gb_ref = y.double().sum((0, 1, 2))
    g = ops.make_geom(Bs, hw, hw, ci, co, 3, 3, 1, 1)
    gw = torch.full((co, 9, ci), float("nan"), device=dev)
    gb = torch.full((co,), float("nan"), device=dev)
    ops.conv_wgrad(g, y.to(dev), x.to(dev), gw.data_ptr(), 0.0, gb.data_ptr())
    e1 = float((gw.double().cpu() - gw_ref).abs().max() / gw_ref.abs().max())
    e2 = float((gb.double().cpu() - gb_ref).abs().max() / gb_ref.abs().max())
    # deferred + batched reduce, accumulate (beta = 1) on top of ones
    gw2 = torch.ones((co, 9, ci), device=dev)
    rows = []
    ops.conv_wgrad(g, y.to(dev), x.to(dev), gw2.data_ptr(), 1.0, None, defer=rows)
    ops.wgrad_reduce_batch(rows, dev)
    e3 = float((gw2.double().cpu() - 1.0 - gw_ref).abs().max() / gw_ref.abs().max())
    xb = torch.randn(B, hw, hw, ci, device=dev)
    yb = torch.randn(B, hw, hw, co, device=dev)
    gb2 = torch.zeros(co, device=dev)
    gB = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
    t = timeit(lambda: ops.conv_wgrad(gB, yb, xb, gw.data_ptr(), 0.0, gb2.data_ptr()))
    fl = 2.0 * B * hw * hw * ci * co * 9
    tot += cnt * t
    print(f"{name:16s} err gw {e1:.1e} gb {e2:.1e} deferred+beta {e3:.1e} | {t * 1e3:8.1f} us  {fl / t / 1e9:6.1f} TF(alg)", flush=True)
print(f"TOTAL wgrad ms/step {tot:.3f}")
