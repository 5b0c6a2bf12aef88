import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lightning-generative-models_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
from lgm_hip import ops
dev = torch.device("cuda", 0)
def run(B, hw, ci, co, xfn, yfn, tag):
    x = xfn(B, hw, ci); y = yfn(B, hw, co)
    w0 = torch.zeros(co, ci, 3, 3, dtype=torch.double, requires_grad=True)
    out = F.conv2d(x.permute(0, 3, 1, 2).double(), w0, None, padding=1)
    gw_ref, = torch.autograd.grad(out, w0, y.permute(0, 3, 1, 2).double())
    gw_ref = gw_ref.permute(0, 2, 3, 1).reshape(co, 9, ci)
    gb_ref = y.double().sum((0, 1, 2))
    g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
    gw = torch.full((co, 9, ci), float("nan"), device=dev)
    gb = torch.full((co,), float("nan"), device=dev)
    ops.conv_wgrad(g, y.to(dev), x.to(dev), gw.data_ptr(), 0.0, gb.data_ptr())
    gwc, gbc = gw.double().cpu(), gb.double().cpu()
    print(f"== {tag}: B{B} {hw}x{hw} {ci}->{co}")
    print("  gb got", gbc[:4].tolist(), "ref", gb_ref[:4].tolist())
    print("  gw[0,:,0] got", [round(v, 3) for v in gwc[0, :, 0].tolist()])
    print("  gw[0,:,0] ref", [round(v, 3) for v in gw_ref[0, :, 0].tolist()])
    print("  gw err", float((gwc - gw_ref).abs().max()), "nan", int(torch.isnan(gwc).sum()))
ones = lambda B, hw, c: torch.ones(B, hw, hw, c)
def ramp(B, hw, c):
    t = torch.zeros(B, hw, hw, c)
    t += torch.arange(hw).float()[None, :, None, None] * 10 + torch.arange(hw).float()[None, None, :, None]
    return t
def chan(B, hw, c):
    return torch.ones(B, hw, hw, c) * torch.arange(c).float()
def delta(py, px):
    def f(B, hw, c):
        t = torch.zeros(B, hw, hw, c); t[0, py, px, :] = 1; return t
    return f
run(1, 16, 64, 64, ones, ones, "ones/ones")
run(1, 16, 64, 64, ones, delta(0, 0), "x ones, y delta(0,0)")
run(1, 16, 64, 64, ramp, delta(5, 6), "x ramp, y delta(5,6)")
run(1, 16, 64, 64, chan, ones, "x chan, y ones")
def run2(B, hw, ci, co, xfn, yfn, tag):
    x = xfn(B, hw, ci); y = yfn(B, hw, co)
    g = ops.make_geom(B, hw, hw, ci, co, 3, 3, 1, 1)
    gw = torch.full((co, 9, ci), float("nan"), device=dev)
    gb = torch.full((co,), float("nan"), device=dev)
    ops.conv_wgrad(g, y.to(dev), x.to(dev), gw.data_ptr(), 0.0, gb.data_ptr())
    print(f"== {tag}")
    print("  gw[0,4,:] /256:", [round(v / 256, 2) for v in gw[0, 4, :].cpu().tolist()])
    print("  gw[:,4,1] /256:", [round(v / 256, 2) for v in gw[:, 4, 1].cpu().tolist()])
    print("  gb /256:", [round(v / 256, 2) for v in gb.cpu().tolist()])
run2(1, 16, 64, 64, chan, ones, "x chan (expect gw[0,4,c] = c)")
run2(1, 16, 64, 64, ones, chan, "y chan (expect gw[n,4,1] = n, gb[n] = n)")
